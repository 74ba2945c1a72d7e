// Large-tile NT GEMM for the 1x1 convolutions and Linear layers with many rows (SD-1.5's SpatialTransformer: 33 % of its
// time): C[m, n] = sum_k A(m, k) W[n, k], both operands k-contiguous, through the same fused epilogue as the other
// contraction kernels.
//
// Why a second shape: the generic gather kernel (gemm_conv.hip, 128x128 tiles, 4 waves x 64x64, 2 workgroups per CU) is
// bound by the per-CU global->LDS staging path: every 128x128x32 tile step moves 32 KB for 1 MFLOP.
// Here one workgroup owns a 256x256 tile with FOUR waves, one per SIMD, each holding a 128x128 accumulator block
// (4x4 v_mfma_f32_32x32x2_f32 tiles = 256 accumulator registers, which is why there is exactly one wave per SIMD): a tile
// step moves 64 KB for 4 MFLOP, half the staging per FLOP, and each LDS fragment feeds 4 MFMAs instead of 2.
// Measured (tools/bench_linear.py): 7.8 us per K tile (6.8 us would be the MFMA peak) + ~50 us of exposed prologue /
// epilogue per tile => 118 TFLOP/s at K = 1280 (generic kernel: 107), 104 at K = 640 (100), 80 at K = 320 (89): taken
// for K >= 640 only.
// Both operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), double-buffered, in the
// unpadded [row][32 floats] image with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7 (same involution on the DMA
// source address and on the fragment read), exactly like the weight tiles of conv3x3_halo.hip.
#include "igemm_common.h"

namespace igemm {
namespace {

constexpr int TM = 256, TN = 256;
constexpr int STAGE_FLOATS = (TM + TN) * 32;                 // one K tile of A and of W
constexpr int G_SMEM = 2 * STAGE_FLOATS * (int)sizeof(float);   // 128 KB

__device__ float g_zero_page_g256[64];

__global__ void __launch_bounds__(256, 1) gemm256_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt)) return;
    const int m0 = mt * TM, n0 = nt * TN;
    const float* zero = g_zero_page_g256 + (lane & 7) * 4;

    // DMA bookkeeping: wave w stages rows w*64 .. w*64+63 of both operand tiles, 8 rows x 128 B per instruction; lane l
    // writes LDS chunk (l & 7) of row (l >> 3), which must hold source chunk (l & 7) ^ ((row >> 1) & 7); within a wave's 64
    // rows (row >> 1) & 7 = 4 (i & 1) + (lrow >> 1), so two swizzled column offsets per lane cover all 8 instructions.
    const int lrow = lane >> 3, lchunk = lane & 7;
    const int row0 = wave * 64 + lrow;
    const int sw0 = (lchunk ^ (lrow >> 1)) * 4, sw1 = (lchunk ^ (4 + (lrow >> 1))) * 4;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
#define DS_G256_DMA(kt_, buf_)                                                                                              \
    do {                                                                                                                    \
        const int k_ = (kt_) * BK;                                                                                          \
        const bool first_ = k_ < p.c0;                                                                                      \
        const float* src_ = first_ ? p.a0 + k_ : p.a1 + (k_ - p.c0);                                                        \
        const int ld_ = first_ ? p.lda0 : p.lda1;                                                                           \
        float* As_ = smem + (buf_) * STAGE_FLOATS + wave * 64 * 32;                                                         \
        float* Bs_ = As_ + TM * 32;                                                                                         \
        const float* pa_ = src_ + (size_t)(m0 + row0) * ld_;                                                                \
        const float* pb_ = p.b + (size_t)(n0 + row0) * p.ldb + k_;                                                          \
        DS_G256_ONE(0); DS_G256_ONE(1); DS_G256_ONE(2); DS_G256_ONE(3);                                                     \
        DS_G256_ONE(4); DS_G256_ONE(5); DS_G256_ONE(6); DS_G256_ONE(7);                                                     \
    } while (0)
#define DS_G256_ONE(i_)                                                                                                     \
    do {                                                                                                                    \
        const int sw_ = ((i_) & 1) ? sw1 : sw0;                                                                             \
        const float* ga_ = (m0 + row0 + (i_) * 8 < p.M) ? pa_ + (size_t)((i_) * 8) * ld_ + sw_ : zero;                      \
        const float* gb_ = (n0 + row0 + (i_) * 8 < p.nrows_b) ? pb_ + (size_t)((i_) * 8) * p.ldb + sw_ : zero;              \
        __builtin_amdgcn_global_load_lds((gptr_t)ga_, (lptr_t)(As_ + (i_) * 8 * 32), 16, 0, 0);                             \
        __builtin_amdgcn_global_load_lds((gptr_t)gb_, (lptr_t)(Bs_ + (i_) * 8 * 32), 16, 0, 0);                             \
    } while (0)

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = p.K / BK;
    DS_G256_DMA(0, 0);
    __syncthreads();

    // fragment offsets: lane l reads row (l & 31) of a 32-row MFMA tile, 16-B chunk (2 ks + (l >> 5)) ^ swizzle(row)
    const int fswz = ((lane & 31) >> 1) & 7;
    const int a_row = (wr * 128 + (lane & 31)) * 32;
    const int b_row = (wc * 128 + (lane & 31)) * 32;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) DS_G256_DMA(kt + 1, cur ^ 1);       // buffer cur^1 was last read before the previous barrier
        const float* As = smem + cur * STAGE_FLOATS;
        const float* Bs = As + TM * 32;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int co = ((ks * 2 + (lane >> 5)) ^ fswz) * 4;
            f32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(As + a_row + i * 32 * 32 + co);
                b[i] = *reinterpret_cast<const f32x4*>(Bs + b_row + i * 32 * 32 + co);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][r], b[j][r], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                                      // drains the DMA (vmcnt) and frees buffer cur
    }

    // epilogue: four 64x64 sub-blocks per wave through the shared staged float4 epilogue (written out: the accumulator
    // array must only ever be indexed by constants, or it is demoted to scratch memory)
    float* stage = smem + wave * 64 * EPI_LD;
#define DS_G256_EPI(bi_, bj_)                                                                                               \
    do {                                                                                                                    \
        f32x16 sub_[2][2] = {{acc[(bi_) * 2][(bj_) * 2], acc[(bi_) * 2][(bj_) * 2 + 1]},                                    \
                             {acc[(bi_) * 2 + 1][(bj_) * 2], acc[(bi_) * 2 + 1][(bj_) * 2 + 1]}};                          \
        epilogue<0, false>(p, sub_, stage, lane, m0 + wr * 128 + (bi_) * 64, n0 + wc * 128 + (bj_) * 64, p.out);          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                              \
        __builtin_amdgcn_wave_barrier();                                                                                    \
    } while (0)
    DS_G256_EPI(0, 0); DS_G256_EPI(0, 1); DS_G256_EPI(1, 0); DS_G256_EPI(1, 1);
}

}  // namespace

// Does the large-tile kernel take this layer?  Plain 1x1 / Linear contraction with enough 256x256 tiles to cover the chip
// and no more padding waste along N than the 128-wide tiles would have.
bool gemm256_applicable(const KParams& p) {
    if (p.taps != 1 || p.stride != 1 || p.ec0 || p.norm) return false;
    if (p.K % BK || p.c0 % BK) return false;
    // one workgroup per CU: nothing overlaps its prologue / epilogue (which all CUs run at the same time, at HBM speed), so
    // the K loop must be long enough to amortise them -- measured break-even against the 128x128 kernel near K = 512
    if (p.K < 640) return false;
    const long long tiles = (long long)((p.M + TM - 1) / TM) * ((p.N + TN - 1) / TN);
    if (tiles < 256) return false;
    const int pad256 = ((p.N + TN - 1) / TN) * TN, pad128 = ((p.N + BN - 1) / BN) * BN;
    return pad256 * 100 <= pad128 * 106;
}

int launch_gemm256(KParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G_SMEM);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    p.mtiles = (p.M + TM - 1) / TM;
    p.ntiles = (p.N + TN - 1) / TN;
    p.splits = 1;
    hipLaunchKernelGGL(gemm256_kernel, dim3(grid_1d(p.mtiles, p.ntiles)), dim3(256), G_SMEM, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace igemm
