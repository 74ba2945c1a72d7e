// 3x3 convolution as an implicit GEMM with an LDS-resident input halo (gfx950, fp32 MFMA).
//
// Round-1 ablations of the generic gather kernel (gemm_conv.hip) showed it is bound by the per-CU global->LDS staging
// path, not by the matrix pipe: with the staging loads removed it runs at 147-152 TFLOP/s (93-96 % of the 157.3 peak),
// with them at 112-122 -- even when the loads hit L1 and even with prefetch distance 2 (so it is issue/throughput of
// the vector-memory path, ~32 KB per 128x128x32 tile, not latency).  The 9 taps of a 3x3 conv re-read the SAME
// activations shifted by one pixel, so this kernel stages each 32-channel slab of the input ONCE per M tile as a
// halo tile in LDS and derives all 9 tap operands from it:
//
//   M tile  = 128 output pixels = nimg image slots x TH rows x W columns (TH*W*nimg = 128)
//   halo    = nimg x (TH+2) x (W+2) pixels x 32 channels, rows padded to 36 floats (same conflict-free ds_read_b128
//             pattern as the generic kernel); out-of-image halo pixels are zero (loaded from a zero page)
//   tap (dy,dx) operand of output pixel (s,r,c) = halo[(s*(TH+2) + r+1+dy)*(W+2) + c+1+dx]  -> one uniform LDS
//             offset per tap added to a per-lane base: no per-tap address arithmetic, no per-tap global loads for A
//   weights = [Cout_pad][K], K = (chunk*9 + tap)*32 + cc, double-buffered per tap exactly like the generic kernel
//
// Global bytes per 32-channel slab and tile: halo 26 KB (W=32) + weights 9 x 16 KB = 170 KB instead of 288 KB, and
// the number of vector-memory instructions per MFMA drops by the same factor.  The halo for slab c+1 is loaded into
// registers while slab c is multiplied and written to LDS at the slab boundary (one extra barrier per 576 MFMAs).
#include "igemm_common.h"

namespace igemm {
namespace {

constexpr int NS_MAX = 10;               // halo float4 slots per thread (256 threads x 10 x 16 B = 320 pixels x 128 B)
constexpr int HALO_MAX = NS_MAX * 32;    // 320 halo pixels
constexpr int B_BYTES = 2 * BN * LDSK * (int)sizeof(float);

__device__ float g_zero_page_halo[64];   // zero-initialised

__global__ void __launch_bounds__(256, 2) conv3x3_halo_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                               // [2][BN][LDSK]
    float* Ah = smem + 2 * BN * LDSK;               // [NP][LDSK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int ld_row = tid >> 3, ld_col = (tid & 7) * 4;
    const float* zero = g_zero_page_halo;

    // ---- tile geometry -----------------------------------------------------------------------------------------
    const int img0 = m0 / p.HW;
    const int r0 = (p.nimg == 1) ? (m0 - img0 * p.HW) / p.W : 0;
    const int n_images = p.M / p.HW;
    const int ns = (p.NP * 8 + 255) >> 8;           // halo slots per thread actually used (uniform)

    // per-thread halo slots: pixel index in the image tensor (or -1: zero) -- fixed for the whole K loop
    int h_pix[NS_MAX];
#pragma unroll
    for (int j = 0; j < NS_MAX; ++j) {
        const int q = tid + j * 256;
        const int hp = q >> 3;
        const int s = hp / (p.HP * p.WP);
        const int rem = hp - s * p.HP * p.WP;
        const int hr = rem / p.WP, hc = rem - hr * p.WP;
        const int img = img0 + s, y = r0 + hr - 1, x = hc - 1;
        const bool ok = hp < p.NP && img < n_images && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        h_pix[j] = ok ? (img * p.H + y) * p.W + x : -1;
    }
    // per-lane A fragment bases inside the halo (two 32-row MFMA tiles per wave)
    int a_foff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wr * 64 + i * 32 + (lane & 31);
        const int s = m / (p.TH * p.W);
        const int rem = m - s * p.TH * p.W;
        const int r = rem / p.W, c = rem - r * p.W;
        a_foff[i] = ((s * p.HP + r + 1) * p.WP + c + 1) * LDSK + (lane >> 5) * 4;
    }
    // weight rows staged by this thread
    size_t b_off[4];
    bool b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ld_row + 32 * i;
        b_ok[i] = n < p.nrows_b;
        b_off[i] = (size_t)n * p.ldb + ld_col;
    }

    f32x4 hreg[NS_MAX];
    f32x4 rb[4];
    auto halo_load = [&](int chunk) {
        const int c = chunk * BK;
        const bool first = c < p.c0;
        const float* src = first ? p.a0 + c + ld_col : p.a1 + (c - p.c0) + ld_col;
        const int ld = first ? p.lda0 : p.lda1;
#pragma unroll
        for (int j = 0; j < NS_MAX; ++j) {
            if (j < ns) {
                const float* ptr = h_pix[j] >= 0 ? src + (size_t)h_pix[j] * ld : zero;
                hreg[j] = *reinterpret_cast<const f32x4*>(ptr);
            }
        }
    };
    auto halo_store = [&]() {
#pragma unroll
        for (int j = 0; j < NS_MAX; ++j) {
            if (j < ns) {
                const int q = tid + j * 256;
                if ((q >> 3) < p.NP) *reinterpret_cast<f32x4*>(Ah + (q >> 3) * LDSK + (q & 7) * 4) = hreg[j];
            }
        }
    };
    auto b_addr = [&](int kt, int i) -> const float* { return b_ok[i] ? p.b + b_off[i] + kt * BK : zero; };
    auto b_store = [&](int buf) {
        float* bs = Bs + buf * BN * LDSK + ld_row * LDSK + ld_col;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(bs + 32 * i * LDSK) = rb[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (p.c0 + p.c1) / BK;
    const int KT = nchunks * 9;

    // ---- prologue ----------------------------------------------------------------------------------------------
    halo_load(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_addr(0, i));
    halo_store();
    b_store(0);
    if (nchunks > 1) halo_load(1);
    __syncthreads();

    const int b_foff = (wc * 64 + (lane & 31)) * LDSK + (lane >> 5) * 4;
    int kt = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        for (int tap = 0; tap < 9; ++tap, ++kt) {
            const int ty = tap / 3;
            const int toff = ((ty - 1) * p.WP + (tap - ty * 3 - 1)) * LDSK;
            const int cur = kt & 1;
            const int nxt = min(kt + 1, KT - 1);            // past the end: re-stage the last tile (branch-free body)
            const float* as0 = Ah + a_foff[0] + toff;
            const float* as1 = Ah + a_foff[1] + toff;
            const float* bs = Bs + cur * BN * LDSK + b_foff;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(as0 + ks * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(as1 + ks * 8);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs + ks * 8);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bs + 32 * LDSK + ks * 8);
                rb[ks] = *reinterpret_cast<const f32x4*>(b_addr(nxt, ks));     // one weight-staging load per 16 MFMAs
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b0[r], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b1[r], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b0[r], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b1[r], acc[1][1], 0, 0, 0);
                }
            }
            b_store(cur ^ 1);
            __syncthreads();
        }
        if (chunk + 1 < nchunks) {
            // every wave has passed the barrier of tap 8: the halo of this slab is dead, publish the next one
            halo_store();
            if (chunk + 2 < nchunks) halo_load(chunk + 2);
            __syncthreads();
        }
    }

    // epilogue staging overlays the whole LDS allocation (>= 4 * 64 * EPI_LD floats, see the launcher)
    epilogue<0>(p, acc, smem + wave * 64 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 64, p.out);
}

}  // namespace

bool conv3x3_halo_supported(const KParams& p) {
    if (p.taps != 9) return false;
    if (p.W < 4 || p.W > 64) return false;
    if (p.HW >= BM) { if (BM % p.W || p.HW % BM) return false; }
    else if (BM % p.HW) return false;
    const int TH = (p.HW >= BM) ? BM / p.W : p.H;
    const int nimg = (p.HW >= BM) ? 1 : BM / p.HW;
    return nimg * (TH + 2) * (p.W + 2) <= HALO_MAX;
}

int launch_conv3x3_halo(KParams& p, hipStream_t stream) {
    p.TH = (p.HW >= BM) ? BM / p.W : p.H;
    p.nimg = (p.HW >= BM) ? 1 : BM / p.HW;
    p.HP = p.TH + 2; p.WP = p.W + 2; p.NP = p.nimg * p.HP * p.WP;
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.N + BN - 1) / BN;
    int smem = B_BYTES + p.NP * LDSK * (int)sizeof(float);
    const int epi = 4 * 64 * EPI_LD * (int)sizeof(float);
    if (smem < epi) smem = epi;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, B_BYTES + HALO_MAX * LDSK * (int)sizeof(float));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL(conv3x3_halo_kernel, dim3(grid_1d(p.mtiles, p.ntiles)), dim3(256), smem, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace igemm
