// 256x128-tile NT GEMM with both operands staged by LDS-DMA: the 1x1 / Linear kernel for many-row layers (every projection
// of SD-1.5's SpatialTransformer at batch >= 8, the qkv / proj 1x1s of the EDM nets at batch >= 64).  Eight waves in a 4x2
// grid, 64x64 outputs each (2x2 v_mfma_f32_32x32x2_f32 tiles), exactly the wave layout of the 8-wave convolution; a tile
// step moves 48 KB for 2 MFLOP (generic gather kernel: 32 KB per MFLOP) by global_load_lds_dwordx4 -- no VGPR round trip,
// no ds_write -- into the unpadded [row][32 floats] LDS image with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7
// (same involution on the DMA source address and on the fragment read), like the weight tiles of conv3x3_halo.hip.
// A 256x256-tile variant with ONE wave per SIMD (128x128 accumulators per wave, 4 MFLOP per 64 KB) was built first and
// rejected: 7.8 us per K tile in steady state (6.8 us = MFMA peak) but ~50 us of prologue / epilogue per tile that nothing
// overlaps (one workgroup per CU, all CUs in phase, epilogues at HBM speed): 116 TFLOP/s at K = 1280, 100 at K = 640, 80 at
// K = 320 -- this kernel does 125 / 115 / 91.
// Measured against the generic kernel (tools/bench_linear.py, bit-identical results): +8 ... +18 % (qkv 64x64 K = 320: 82 ->
// 91 TFLOP/s; ff.proj 16x16 K = 1280: 106 -> 125).
#include "igemm_common.h"

namespace igemm {
namespace {

constexpr int TM8 = 256, TN8 = 128;
constexpr int STAGE8 = (TM8 + TN8) * 32;
constexpr int SMEM8 = 2 * STAGE8 * (int)sizeof(float);      // 96 KB

__device__ float g_zero_page_dma8[64];

__global__ void __launch_bounds__(512, 2) gemm_dma8_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt)) return;
    const int m0 = mt * TM8, n0 = nt * TN8;
    const float* zero = g_zero_page_dma8 + (lane & 7) * 4;
    // DMA: wave w stages A rows w*32 .. w*32+31 (4 instructions of 8 rows) and W rows w*16 .. w*16+15 (2 instructions);
    // (row >> 1) & 7 of row = base + i*8 + lrow with base % 16 == 0 is 4 (i & 1) + (lrow >> 1)
    const int lrow = lane >> 3, lchunk = lane & 7;
    const int sw0 = (lchunk ^ (lrow >> 1)) * 4, sw1 = (lchunk ^ (4 + (lrow >> 1))) * 4;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define DS_DMA8_A(i_)                                                                                                       \
    do {                                                                                                                    \
        const int r_ = wave * 32 + (i_) * 8 + lrow;                                                                         \
        const float* g_ = (m0 + r_ < p.M) ? src_ + (size_t)(m0 + r_) * ld_ + (((i_) & 1) ? sw1 : sw0) : zero;               \
        __builtin_amdgcn_global_load_lds((gptr_t)g_, (lptr_t)(As_ + (wave * 32 + (i_) * 8) * 32), 16, 0, 0);                \
    } while (0)
#define DS_DMA8_B(i_)                                                                                                       \
    do {                                                                                                                    \
        const int r_ = wave * 16 + (i_) * 8 + lrow;                                                                         \
        const float* g_ = (n0 + r_ < p.nrows_b) ? p.b + (size_t)(n0 + r_) * p.ldb + k_ + (((i_) & 1) ? sw1 : sw0) : zero;  \
        __builtin_amdgcn_global_load_lds((gptr_t)g_, (lptr_t)(Bs_ + (wave * 16 + (i_) * 8) * 32), 16, 0, 0);                \
    } while (0)
#define DS_DMA8(kt_, buf_)                                                                                                  \
    do {                                                                                                                    \
        const int k_ = (kt_) * BK;                                                                                          \
        const bool first_ = k_ < p.c0;                                                                                      \
        const float* src_ = first_ ? p.a0 + k_ : p.a1 + (k_ - p.c0);                                                        \
        const int ld_ = first_ ? p.lda0 : p.lda1;                                                                           \
        float* As_ = smem + (buf_) * STAGE8;                                                                                \
        float* Bs_ = As_ + TM8 * 32;                                                                                        \
        DS_RACE_SKEW(wave);                                                                                                 \
        DS_DMA8_A(0); DS_DMA8_A(1); DS_DMA8_A(2); DS_DMA8_A(3); DS_DMA8_B(0); DS_DMA8_B(1);                                 \
    } while (0)

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = p.K / BK;
    DS_DMA8(0, 0);
    __syncthreads();
    const int fswz = ((lane & 31) >> 1) & 7;
    const int a_row = (wr * 64 + (lane & 31)) * 32;
    const int b_row = (wc * 64 + (lane & 31)) * 32;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) DS_DMA8(kt + 1, cur ^ 1);
        const float* As = smem + cur * STAGE8;
        const float* Bs = As + TM8 * 32;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int co = ((ks * 2 + (lane >> 5)) ^ fswz) * 4;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + a_row + co);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + a_row + 32 * 32 + co);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + b_row + co);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + b_row + 32 * 32 + co);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b0[r], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b1[r], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b0[r], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b1[r], acc[1][1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    epilogue<0, true>(p, acc, smem + wave * 32 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 64, p.out);
}

}  // namespace

bool gemm_dma8_applicable(const KParams& p) {
    if (p.taps != 1 || p.stride != 1 || p.ec0 || p.norm) return false;
    if (p.K % BK || p.c0 % BK) return false;
    const long long tiles = (long long)((p.M + TM8 - 1) / TM8) * ((p.N + TN8 - 1) / TN8);
    return tiles >= 512;           // the 256 CUs covered at least twice (320 tiles: 84 vs 88 TFLOP/s for the generic kernel)
}

int launch_gemm_dma8(KParams& p, hipStream_t stream) {
    DS_ENSURE_DYN_LDS((&gemm_dma8_kernel), SMEM8);
    p.mtiles = (p.M + TM8 - 1) / TM8;
    p.ntiles = (p.N + TN8 - 1) / TN8;
    p.splits = 1;
    hipLaunchKernelGGL(gemm_dma8_kernel, dim3(grid_1d(p.mtiles, p.ntiles)), dim3(512), SMEM8, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace igemm
