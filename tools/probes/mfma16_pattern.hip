// Probe (round 3): what MFMA rate does the fp16-activation convolution's inner pattern allow?  v_mfma_f32_32x32x16_f16, eight waves
// per CU, per K step 2 A x NB B fragments -> 2 NB MFMAs on 2 NB accumulators.  Variants: operand data constant vs random (power),
// operands in one vs several register sets, with / without the LDS fragment reads and the per-tap barrier.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma16_pattern tools/probes/mfma16_pattern.hip && tools/probes/mfma16_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

// MODE bit 0: random operand data; bit 1: distinct operand registers (2 A, 3 B) instead of one A, one B; bit 2: LDS fragment reads
// (5 ds_read_b128 per K step, double buffered); bit 3: s_barrier every 4 K steps
template <int MODE>
__global__ void __launch_bounds__(512, 2) pat(float* out, int iters, unsigned seed) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    unsigned s = seed + threadIdx.x * 7919u + blockIdx.x * 104729u;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
        const _Float16 h0 = (MODE & 1) ? (_Float16)(((int)(rnd(s) >> 20) - 2048) * (1.0f / 2048.f)) : (_Float16)0.5f;
        const _Float16 h1 = (MODE & 1) ? (_Float16)(((int)(rnd(s) >> 20) - 2048) * (1.0f / 2048.f)) : (_Float16)0.25f;
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 pk = {h0, h1};
        lds[i] = __builtin_bit_cast(float, pk);
    }
    __syncthreads();
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned la = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
    f4 P[5], Q[5];
    for (int i = 0; i < 5; ++i) { P[i] = *(f4*)((char*)lds + ((threadIdx.x * 16 + i * 8192) & 65535)); Q[i] = *(f4*)((char*)lds + ((threadIdx.x * 16 + i * 8192 + 4096) & 65535)); }
#define MM(k, a, b) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), acc[k], 0, 0, 0)
#define GROUP(F)                                                                                   \
    if (MODE & 2) { MM(0, F[0], F[2]); MM(1, F[1], F[2]); MM(2, F[0], F[3]); MM(3, F[1], F[3]); MM(4, F[0], F[4]); MM(5, F[1], F[4]); } \
    else { MM(0, F[0], F[2]); MM(1, F[0], F[2]); MM(2, F[0], F[2]); MM(3, F[0], F[2]); MM(4, F[0], F[2]); MM(5, F[0], F[2]); }
#define READ(F, off)                                                                               \
    if (MODE & 4) { for (int i = 0; i < 5; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(F[i]) : "v"(la + (unsigned)i * 8192u), "n"(off)); }
#define WAIT(F, n)                                                                                 \
    if (MODE & 4) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]));
    for (int it = 0; it < iters; ++it) {
        READ(Q, 16); WAIT(P, 5); __builtin_amdgcn_sched_barrier(0); GROUP(P); __builtin_amdgcn_sched_barrier(0);
        READ(P, 32); WAIT(Q, 5); __builtin_amdgcn_sched_barrier(0); GROUP(Q); __builtin_amdgcn_sched_barrier(0);
        READ(Q, 48); WAIT(P, 5); __builtin_amdgcn_sched_barrier(0); GROUP(P); __builtin_amdgcn_sched_barrier(0);
        WAIT(Q, 0);
        if (MODE & 8) __builtin_amdgcn_s_barrier();
        READ(P, 0); __builtin_amdgcn_sched_barrier(0); GROUP(Q); __builtin_amdgcn_sched_barrier(0);
    }
    float t = 0.f;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 123.456f) out[threadIdx.x] = t;
}

template <int MODE>
void run(const char* label) {
    float* out; hipMalloc(&out, 4096);
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    pat<MODE><<<blocks, 512>>>(out, 200, 1u);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        pat<MODE><<<blocks, 512>>>(out, iters, 1u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double fl = 2.0 * 32 * 32 * 16 * 24.0 * iters * 8 * blocks;
    printf("%-64s %8.3f ms  %7.1f TFLOP/s  (%.1f ms of MFMA stream)\n", label, best, fl / best / 1e9, best);
    hipFree(out);
}

int main() {
    run<0>("const data, one A / one B register set");
    run<1>("random data, one register set");
    run<2>("const data, 2 A x 3 B register sets");
    run<3>("random data, 2 A x 3 B register sets");
    run<7>("random data, 2 x 3 sets, LDS fragment reads (5 per 6 MFMA)");
    run<15>("random data, 2 x 3 sets, LDS reads, barrier per 24 MFMA");
    run<11>("random data, 2 x 3 sets, barrier per 24 MFMA, no reads");
    run<6>("const data, 2 x 3 sets, LDS fragment reads");
    return 0;
}
