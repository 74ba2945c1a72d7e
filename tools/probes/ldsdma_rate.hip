// Probe (round 4): how fast can a CU pull operand tiles into LDS, and does the path matter?
//   path 0: LDS-DMA (global_load_lds_dwordx4: global -> LDS, no VGPR)           -- what conv3x3_f16dma / gemm_f16dma stage with
//   path 1: global_load_dwordx4 -> VGPR -> ds_write_b128                        -- the register path
// Sources: `l2`  = every workgroup re-reads the same 2 MB (weights-like: L2 / MALL resident after the first pass),
//          `hbm` = every workgroup streams its own slice of a 2 GB buffer (activations-like).
// Workgroups of 512 threads, one per CU (256) or two of 256 threads per CU (512), DEPTH 16-byte requests per lane in flight before a wait.
// Prints GB/s per CU and chip-wide.  Every fp16-activation kernel of this engine moves 15 - 24 GB/s per CU through path 0
// (docs/HISTORY.md section E.12); this probe says whether that is the path's ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/ldsdma_rate tools/probes/ldsdma_rate.hip && tools/probes/ldsdma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int PATH, int DEPTH, int THREADS>
__global__ void __launch_bounds__(THREADS) pull(const char* __restrict__ src, size_t slice_bytes, size_t wrap_bytes, int rounds, float* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // this workgroup's slice: [blockIdx.x * slice_bytes, +slice_bytes) modulo wrap_bytes (wrap = 2 MB: every workgroup reads the same bytes)
    const size_t base = ((size_t)blockIdx.x * slice_bytes) % wrap_bytes;
    const size_t round_bytes = (size_t)THREADS * 16 * DEPTH;          // bytes one round of requests moves
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rounds; ++r) {
        const size_t off = (base + ((size_t)r * round_bytes) % slice_bytes) % wrap_bytes;
        if (PATH == 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + off + ((size_t)d * THREADS + tid) * 16),
                                                 (lptr_t)(lds + ((d & 7) * THREADS + wave * 64) * 16), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f4 v[DEPTH];                                            // volatile asm: the compiler must not drop "dead" rounds
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[d]) : "v"(src + off + ((size_t)d * THREADS + tid) * 16) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
                asm volatile("ds_write_b128 %0, %1" ::"v"((unsigned)(size_t)(lptr_t)(lds + ((d & 7) * THREADS + tid) * 16)), "v"(v[d]) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        (void)lane;
    }
    __syncthreads();
    acc += *reinterpret_cast<const f4*>(lds + tid * 16);
    if (acc[0] == 123.456f) out[tid] = acc[1];
}

template <int PATH, int DEPTH, int THREADS>
void run(const char* label, const char* src, size_t total, bool l2, int blocks) {
    float* out; (void)hipMalloc(&out, 4096);
    const size_t wrap = l2 ? (2u << 20) : total;
    const size_t slice = l2 ? (2u << 20) : total / blocks;
    const size_t round_bytes = (size_t)THREADS * 16 * DEPTH;
    const int rounds = (int)((l2 ? (64u << 20) : slice) / round_bytes);          // l2: 64 MB per workgroup, hbm: its whole slice once
    const int smem = 8 * THREADS * 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pull<PATH, DEPTH, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    pull<PATH, DEPTH, THREADS><<<blocks, THREADS, smem>>>(src, slice, wrap, rounds > 64 ? 64 : rounds, out);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        pull<PATH, DEPTH, THREADS><<<blocks, THREADS, smem>>>(src, slice, wrap, rounds, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = (double)rounds * round_bytes * blocks;
    const double cus = blocks > 256 ? 256.0 : (double)blocks;
    printf("%-58s %8.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip\n", label, best, bytes / best / 1e6 / cus, bytes / best / 1e9);
    (void)hipFree(out);
}

int main() {
    const size_t total = 2ull << 30;
    char* src; (void)hipMalloc(&src, total); (void)hipMemset(src, 1, total);
    printf("# LDS fill rate by path (MI355X); 'l2' = all workgroups re-read the same 2 MB, 'hbm' = disjoint slices of 2 GB\n");
    run<0, 8, 512>("LDS-DMA   l2   512 thr x 1 WG/CU, 8 in flight / lane", src, total, true, 256);
    run<0, 16, 512>("LDS-DMA   l2   512 thr x 1 WG/CU, 16 in flight / lane", src, total, true, 256);
    run<0, 8, 256>("LDS-DMA   l2   256 thr x 2 WG/CU, 8 in flight / lane", src, total, true, 512);
    run<1, 8, 512>("registers l2   512 thr x 1 WG/CU, 8 in flight / lane", src, total, true, 256);
    run<1, 8, 256>("registers l2   256 thr x 2 WG/CU, 8 in flight / lane", src, total, true, 512);
    run<0, 8, 512>("LDS-DMA   hbm  512 thr x 1 WG/CU, 8 in flight / lane", src, total, false, 256);
    run<0, 16, 512>("LDS-DMA   hbm  512 thr x 1 WG/CU, 16 in flight / lane", src, total, false, 256);
    run<0, 8, 256>("LDS-DMA   hbm  256 thr x 2 WG/CU, 8 in flight / lane", src, total, false, 512);
    run<1, 8, 512>("registers hbm  512 thr x 1 WG/CU, 8 in flight / lane", src, total, false, 256);
    run<1, 8, 256>("registers hbm  256 thr x 2 WG/CU, 8 in flight / lane", src, total, false, 512);
    (void)hipFree(src);
    return 0;
}
