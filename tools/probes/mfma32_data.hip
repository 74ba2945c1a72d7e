// Probe (round 3): is the fp32 matrix rate (v_mfma_f32_32x32x2_f32) data dependent like the fp16 one (mfma16_pattern.hip)?
// 256 workgroups x 8 waves, per step 16 MFMAs on 4 accumulators (the 64 x 64 wave tile of the fp32 convolution), operands constant vs random.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma32_data tools/probes/mfma32_data.hip && /tmp/mfma32_data
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int RANDOM, int NSETS>
__global__ void __launch_bounds__(512, 2) pat(float* out, int iters, unsigned seed) {
    unsigned s = seed + threadIdx.x * 7919u + blockIdx.x * 104729u;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = RANDOM ? ((int)(rnd(s) >> 12) - 524288) * (1.0f / 524288.f) : 0.5f;
        b[i] = RANDOM ? ((int)(rnd(s) >> 12) - 524288) * (1.0f / 524288.f) : 0.25f;
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ia = NSETS > 1 ? 2 * k : 0, ib = NSETS > 1 ? 2 * k : 0;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ia], b[ib], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ia], b[ib + (NSETS > 1)], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ia + (NSETS > 1)], b[ib], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ia + (NSETS > 1)], b[ib + (NSETS > 1)], acc[3], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 123.456f) out[threadIdx.x] = t;
}

template <int RANDOM, int NSETS>
void run(const char* label) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    pat<RANDOM, NSETS><<<blocks, 512>>>(out, 100, 1u);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        pat<RANDOM, NSETS><<<blocks, 512>>>(out, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fl = 2.0 * 32 * 32 * 2 * 16.0 * iters * 8 * blocks;
    printf("%-56s %8.3f ms  %7.1f TFLOP/s\n", label, best, fl / best / 1e9);
    hipFree(out);
}

int main() {
    run<0, 1>("fp32 MFMA, constant operands, one register set");
    run<1, 1>("fp32 MFMA, random operands, one register set");
    run<0, 2>("fp32 MFMA, constant operands, 2 x 2 register sets");
    run<1, 2>("fp32 MFMA, random operands, 2 x 2 register sets");
    return 0;
}
