// Probe: does VALU work issued between fp32 MFMAs (v_mfma_f32_32x32x2_f32) hide behind them, or add to them?
// One workgroup per CU, WAVES waves; each wave repeats [MFMA x4 (independent accumulators), V VALU ops after each MFMA].
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_valu tools/probes/mfma_valu.hip && tools/probes/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// the same with the fp16 matrix instruction (v_mfma_f32_32x32x16_f16, 8 passes = 32 cycles per SIMD) and optional LDS fragment reads
template <int V, int KIND, int NDS>
__global__ void __launch_bounds__(512) probe16(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 0.5f); }
    float fa = seed + threadIdx.x, fb = seed * 0.5f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x * 1e-3f;
    const unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 fr[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float& x = v[(m * V + k) & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(fa), "v"(fb));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
                else asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(fa));
            }
            if (NDS && (m % (16 / NDS)) == 0) {          // NDS ds_read_b128 per 16 MFMAs
                f4& d = fr[(m / (16 / NDS)) & 3];
                asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(la));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += fr[i][0];
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int V, int KIND, int NDS>
void run16(int waves, const char* label) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe16<V, KIND, NDS><<<blocks, waves * 64>>>(out, 100, 1.f);
    hipEventRecord(e0);
    probe16<V, KIND, NDS><<<blocks, waves * 64>>>(out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = 2.0 * 32 * 32 * 16 * 16.0 * iters * waves * blocks;
    const double cyc = ms * 1e-3 * 2.4e9 / (16.0 * iters * (waves / 4.0));
    printf("f16 %-30s waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s  %6.1f cycles per [mfma16 + %d valu + %d/16 ds_read_b128] per SIMD-wave\n", label, waves / 4, ms,
           mf / ms / 1e9, cyc, V, NDS);
    hipFree(out);
}

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
    for (int waves : {4, 8}) {
        run16<0, 0, 0>(waves, "mfma16 only");
        run16<1, 0, 0>(waves, "mfma16 + 1 v_fma");
        run16<2, 0, 0>(waves, "mfma16 + 2 v_fma");
        run16<4, 0, 0>(waves, "mfma16 + 4 v_fma");
        run16<8, 0, 0>(waves, "mfma16 + 8 v_fma");
        run16<2, 1, 0>(waves, "mfma16 + 2 v_exp");
        run16<2, 2, 0>(waves, "mfma16 + 2 v_cvt_pk_f16");
        run16<0, 0, 4>(waves, "mfma16 + 4/16 ds_read_b128");
        run16<0, 0, 8>(waves, "mfma16 + 8/16 ds_read_b128");
        run16<0, 0, 16>(waves, "mfma16 + 16/16 ds_read_b128");
        run16<2, 0, 16>(waves, "mfma16 + 2 v_fma + 16/16 ds_read");
    }
    return 0;
}
