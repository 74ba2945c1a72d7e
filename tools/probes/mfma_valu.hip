// Probe: does VALU work issued between fp32 MFMAs (v_mfma_f32_32x32x2_f32) hide behind them, or add to them?
// One workgroup per CU, WAVES waves; each wave repeats [MFMA x4 (independent accumulators), V VALU ops after each MFMA].
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_valu tools/probes/mfma_valu.hip && tools/probes/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V, int KIND, bool MFMA>
__global__ void __launch_bounds__(512) probe(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (MFMA) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float& x = v[(m * V + k) & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                else asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(a));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int V, int KIND, bool MFMA>
void run(int waves, const char* label) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<V, KIND, MFMA><<<blocks, waves * 64>>>(out, 100, 1.f);
    hipEventRecord(e0);
    probe<V, KIND, MFMA><<<blocks, waves * 64>>>(out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = MFMA ? 2.0 * 32 * 32 * 2 * 16.0 * iters * waves * blocks : 0.0;
    // cycles per (MFMA + V VALU) group per SIMD at 2.4 GHz, waves/4 waves per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / (16.0 * iters * (waves / 4.0));
    printf("%-34s waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s  %6.1f cycles per [mfma + %d valu] per SIMD-wave\n", label, waves / 4, ms, mf / ms / 1e9, cyc, V);
    hipFree(out);
}

int main() {
    for (int waves : {4, 8}) {
        run<0, 0, true>(waves, "mfma only");
        run<1, 0, true>(waves, "mfma + 1 v_fma");
        run<2, 0, true>(waves, "mfma + 2 v_fma");
        run<4, 0, true>(waves, "mfma + 4 v_fma");
        run<8, 0, true>(waves, "mfma + 8 v_fma");
        run<16, 0, true>(waves, "mfma + 16 v_fma");
        run<4, 0, false>(waves, "4 v_fma only");
        run<16, 0, false>(waves, "16 v_fma only");
        run<2, 1, true>(waves, "mfma + 2 v_exp");
        run<4, 1, true>(waves, "mfma + 4 v_exp");
        run<4, 1, false>(waves, "4 v_exp only");
        run<4, 2, true>(waves, "mfma + 4 v_cvt_pk_f16");
        run<4, 2, false>(waves, "4 v_cvt_pk_f16 only");
    }
    return 0;
}
