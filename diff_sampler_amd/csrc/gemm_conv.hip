// Generic implicit-GEMM / batched NT GEMM on the gfx950 fp32 matrix pipe, and the C-ABI entry points of both
// contraction kernels (3x3 convolutions that fit the halo geometry are routed to conv3x3_halo.hip).
//
// One kernel core serves 1x1 convolutions, Linear layers, QK^T, PV and any 3x3 convolution the halo kernel rejects:
//     C[m, n] = sum_k A(m, k) * B(n, k)          (both operands k-contiguous in memory: "NT")
//   conv mode : A(m, k) is gathered on the fly from the NHWC activation(s).  K is ordered CHUNK-MAJOR, TAP-MINOR:
//               k = (chunk * taps + tap) * 32 + cc, where chunk indexes 32-channel slabs of the concatenation
//               [x0 | x1] (the decoder's torch.cat is never materialised) and tap the zero-padded 3x3 neighbour.
//               B = packed weights [Cout_pad][K] in the same K order.
//   gemm mode : A, B plain strided row-major matrices, batched over blockIdx.z (attention).
//
// Tiling (CDNA4, wave64): block tile 128(M) x 128(N) x 32(K), 256 threads = 4 waves in a 2x2 grid, each wave
// owns a 64x64 sub-tile = 2x2 MFMA tiles of v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 cycles each,
// 157 TFLOP/s chip peak).  Global -> register -> LDS staging with double-buffered LDS (one barrier per K tile).
// LDS rows are padded to 36 floats so that the ds_read_b128 fragment reads (16 distinct rows per lane group,
// stride 144 B) hit 16 distinct 16-B bank slots: conflict free.
// Fragment trick: lane l of an MFMA holds A[row = l&31][kslot = l>>5]; one ds_read_b128 fetches 4 consecutive k
// for that lane, and register r of the read feeds MFMA number r, i.e. MFMA r contracts k = {8*ks + r, 8*ks+4+r}.
// A and B use the same k permutation, so the sum over k is unchanged.
// Staging is branch-free: out-of-image taps, rows past M and B rows past N read from a zero page instead of being
// predicated, so the K-tile body is one basic block and the staging loads ride in the shadow of the MFMAs.
#include "igemm_common.h"

namespace igemm {
namespace {

__device__ float g_zero_page[64];     // zero-initialised; target of the predicated-off staging loads

template <int MODE>   // 0 = conv gather, 1 = batched gemm
__global__ void __launch_bounds__(256, 2) igemm_f32_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDSK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (MODE == 0) {
        if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt, blockIdx.y)) return;
    } else {
        mt = blockIdx.x; nt = blockIdx.y;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    const int ld_row = tid >> 3, ld_col = (tid & 7) * 4;

    const float* a_base = p.a0;
    const float* b_base = p.b;
    float* o_base = p.out;
    if (MODE == 1) {
        const int zb = blockIdx.z / p.heads, zh = blockIdx.z - zb * p.heads;
        a_base += zb * p.a_bs + zh * p.a_hs;
        b_base += zb * p.b_bs + zh * p.b_hs;
        o_base += zb * p.o_bs + zh * p.o_hs;
    }

    // Per-thread row bookkeeping for the 4 A rows and 4 B rows this thread stages.
    bool a_ok[4];
    int a_oh[4], a_ow[4];
    size_t a_off[4];
    size_t b_off[4];
    bool b_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ld_row + 32 * i;
        a_ok[i] = m < p.M;
        if (MODE == 0) {
            const int img = m / p.HW;
            const int rem = m - img * p.HW;
            a_oh[i] = rem / p.W;
            a_ow[i] = rem - a_oh[i] * p.W;
            a_off[i] = (size_t)img * p.IH * p.IW;
        } else {
            a_oh[i] = a_ow[i] = 0;
            a_off[i] = (size_t)m * p.lda0 + ld_col;
        }
        const int n = n0 + ld_row + 32 * i;
        b_ok[i] = n < p.nrows_b;
        b_off[i] = (size_t)n * p.ldb + ld_col;
    }

    f32x4 ra[4], rb[4];
    const float* zero = g_zero_page;
    auto a_addr = [&](int kt, int i) -> const float* {
        if (MODE == 0) {
            const int nmain = p.taps * ((p.c0 + p.c1) / BK);
            const bool extra = kt >= nmain;                       // appended 1x1 sources (centre tap)
            const int chunk = extra ? kt - nmain : kt / p.taps;
            const int tap = extra ? 4 : kt - chunk * p.taps;
            const int c = chunk * BK;
            const int ty = (p.taps == 9) ? tap / 3 : 1;
            const int dy = ty - 1, dx = (p.taps == 9) ? tap - ty * 3 - 1 : 0;
            const int cc0 = extra ? p.ec0 : p.c0;
            const bool first = c < cc0;
            const float* s0 = extra ? p.e0 : p.a0;
            const float* s1 = extra ? p.e1 : p.a1;
            const float* src = first ? s0 + c + ld_col : s1 + (c - cc0) + ld_col;
            const int ld = first ? (extra ? p.elda0 : p.lda0) : (extra ? p.elda1 : p.lda1);
            const int ih = a_oh[i] * p.stride + dy, iw = a_ow[i] * p.stride + dx;
            const bool ok = a_ok[i] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
            return ok ? src + (a_off[i] + (size_t)(ih * p.IW + iw)) * ld : zero;
        } else {
            return a_ok[i] ? a_base + a_off[i] + kt * BK : zero;
        }
    };
    auto b_addr = [&](int kt, int i) -> const float* { return b_ok[i] ? b_base + b_off[i] + kt * BK : zero; };
    auto store_tiles = [&](int buf) {
        DS_RACE_SKEW(wave);
        float* as = As + buf * BM * LDSK + ld_row * LDSK + ld_col;
        float* bs = Bs + buf * BN * LDSK + ld_row * LDSK + ld_col;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(as + 32 * i * LDSK) = ra[i];
            *reinterpret_cast<f32x4*>(bs + 32 * i * LDSK) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // split-K (conv mode only): this block contracts K tiles [kt0, KT)
    const int KT_all = p.K / BK;
    const int kt0 = (MODE == 0) ? (int)((long long)blockIdx.y * KT_all / p.splits) : 0;
    const int KT = (MODE == 0) ? (int)((long long)(blockIdx.y + 1) * KT_all / p.splits) : KT_all;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const f32x4*>(a_addr(kt0, i));
        rb[i] = *reinterpret_cast<const f32x4*>(b_addr(kt0, i));
    }
    store_tiles(kt0 & 1);
    __syncthreads();

    const int frag_off = (lane & 31) * LDSK + (lane >> 5) * 4;
    for (int kt = kt0; kt < KT; ++kt) {
        const int cur = kt & 1;
        const int nxt = min(kt + 1, KT - 1);      // the last iteration re-stages its own tile (harmless, keeps the body branch-free)
        const float* as = As + cur * BM * LDSK + wr * 64 * LDSK + frag_off;
        const float* bs = Bs + cur * BN * LDSK + wc * 64 * LDSK + frag_off;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(as + ks * 8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(as + 32 * LDSK + ks * 8);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs + ks * 8);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bs + 32 * LDSK + ks * 8);
            // two of the eight staging loads ride in the shadow of each 16-MFMA group
            if (ks < 2) {
                ra[2 * ks] = *reinterpret_cast<const f32x4*>(a_addr(nxt, 2 * ks));
                ra[2 * ks + 1] = *reinterpret_cast<const f32x4*>(a_addr(nxt, 2 * ks + 1));
            } else {
                rb[2 * (ks - 2)] = *reinterpret_cast<const f32x4*>(b_addr(nxt, 2 * (ks - 2)));
                rb[2 * (ks - 2) + 1] = *reinterpret_cast<const f32x4*>(b_addr(nxt, 2 * (ks - 2) + 1));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b0[r], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b1[r], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b0[r], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b1[r], acc[1][1], 0, 0, 0);
            }
        }
        store_tiles(cur ^ 1);
        __syncthreads();
    }

    if (MODE == 0 && p.splits > 1) {
        const KParams q = split_params(p, blockIdx.y);
        epilogue<MODE>(q, acc, smem + wave * 64 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 64, q.out);
        return;
    }
    epilogue<MODE>(p, acc, smem + wave * 64 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 64, o_base);
}

// Sum of the split-K partial tiles + the fused epilogue: out = act((sum + colbias + cbias[img] + res) * scale).
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const KParams p) {
    const int n4 = (p.N + 3) >> 2;
    const long long total = (long long)p.M * n4;
    const size_t plane = (size_t)p.M * p.N;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / n4);
        const int col = (int)(idx - (long long)row * n4) * 4;
        const int img = p.cbias_bcast ? 0 : row / p.HW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = col + j;
            if (c >= p.N) break;
            float v = 0.f;
            for (int s = 0; s < p.splits; ++s) v += p.part[s * plane + (size_t)row * p.N + c];
            if (p.colbias) v += p.colbias[c];
            if (p.cbias) v += p.cbias[(size_t)img * p.cbias_ld + c];
            if (p.res) v += p.res[(size_t)row * p.res_ld + c];
            v *= p.scale;
            if (p.act == DS_ACT_SILU) v = ds_silu(v);
            if (p.out_planar) { const int im = row / p.HW; p.out[((size_t)im * p.N + c) * p.HW + (row - im * p.HW)] = v; }
            else p.out[(size_t)row * p.ldo + c] = v;
        }
    }
}

template <int MODE>
int launch(KParams& p, int batch, hipStream_t stream) {
    DS_ENSURE_DYN_LDS((&igemm_f32_kernel<MODE>), SMEM_BYTES);
    p.mtiles = (p.M + BM - 1) / BM;
    p.ntiles = (p.N + BN - 1) / BN;
    dim3 grid;
    if (MODE == 0) {
        p.splits = choose_splits((long long)p.mtiles * p.ntiles, false, p.K / BK, 1, p.part ? p.part_cap : 0, (long long)p.M * p.N, nullptr, p.t_splits);
        grid = dim3(grid_1d(p.mtiles, p.ntiles), p.splits, 1);
    } else {
        p.splits = 1;
        grid = dim3(p.mtiles, p.ntiles, batch);
    }
    hipLaunchKernelGGL(igemm_f32_kernel<MODE>, grid, dim3(256), SMEM_BYTES, stream, p);
    DS_CHECK_LAUNCH();
    if (p.splits > 1) return launch_splitk_reduce(p, stream);
    return DS_OK;
}

bool vec_epilogue_ok(const KParams& p) {
    auto a16 = [](const void* q) { return q == nullptr || ds_aligned16(q); };
    if ((p.ldo & 3) || !a16(p.out) || (p.o_bs & 3) || (p.o_hs & 3)) return false;
    if (p.res && ((p.res_ld & 3) || !a16(p.res))) return false;
    if (p.cbias && ((p.cbias_ld & 3) || !a16(p.cbias))) return false;
    if (!a16(p.colbias)) return false;
    return true;
}

}  // namespace

namespace {
// Column statistics of a finished output tensor in the epilogue's partial format (split-K layers: their epilogue runs in
// splitk_reduce_kernel, element-wise).  grid = (row blocks of 64, column chunks of 64); thread = (column, 16-row group).
__global__ void __launch_bounds__(256) colstats_kernel(const KParams p) {
    __shared__ float sh[2][4][64];
    const int rb = blockIdx.x, cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = blockIdx.y * 64 + cl;
    float s = 0.f, q = 0.f;
    if (col < p.N) {
        const int r0 = rb * 64 + rg * 16, r1 = min(r0 + 16, p.M);
#pragma unroll 4
        for (int r = r0; r < r1; ++r) { const float v = p.out[(size_t)r * p.ldo + col]; s += v; q += v * v; }
    }
    sh[0][rg][cl] = s; sh[1][rg][cl] = q;
    __syncthreads();
    if (rg == 0 && col < p.N) {
        p.stats[((size_t)rb * 2) * p.N + col] = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
        p.stats[((size_t)rb * 2 + 1) * p.N + col] = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
    }
}
}  // namespace

namespace {
// Split-K reduce of a layer whose output feeds a GroupNorm (p.stats), in ONE launch: a block owns a 64-row x 64-column block of the output
// -- exactly one (row block, column chunk) cell of the statistics -- sums the partial planes in split order (float4 rows), applies the
// epilogue, stores the rows and leaves the cell's column sums / sums of squares behind.  Replaces splitk_reduce_kernel + colstats_kernel
// (one dependent launch less per split layer: at 8 images per call two thirds of the CIFAR-10 net's convolutions are split).
// thread = (column quad: tid & 15, row lane: tid >> 4; rows lane, lane + 16, lane + 32, lane + 48).  Requires N % 64 == 0, float4-aligned
// output / residual / bias rows (p.vec_ok) and float4 partial planes (p.vec_part).
__global__ void __launch_bounds__(256) splitk_reduce_stats_kernel(const KParams p) {
    __shared__ f32x4 sh[2][16][16];
    const int rb = blockIdx.x, cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.y * 64 + cq * 4;
    const size_t plane = (size_t)p.M * p.N;
    f32x4 cb = {0.f, 0.f, 0.f, 0.f}, s4 = cb, q4 = cb;
    if (p.colbias) cb = *reinterpret_cast<const f32x4*>(p.colbias + col);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = rb * 64 + rl + 16 * k;
        if (row >= p.M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(p.part + (size_t)row * p.N + col);
        for (int sp = 1; sp < p.splits; ++sp) v += *reinterpret_cast<const f32x4*>(p.part + sp * plane + (size_t)row * p.N + col);
        v += cb;
        if (p.cbias) v += *reinterpret_cast<const f32x4*>(p.cbias + (size_t)(p.cbias_bcast ? 0 : row / p.HW) * p.cbias_ld + col);
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.res_ld + col);
        v *= p.scale;
        if (p.act == DS_ACT_SILU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ds_silu(v[j]);
        }
        *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col) = v;
        s4 += v; q4 += v * v;
    }
    sh[0][rl][cq] = s4; sh[1][rl][cq] = q4;
    __syncthreads();
    if (rl < 2) {                                   // rl = 0: sums, 1: sums of squares; fixed order over the 16 row lanes
        f32x4 t = sh[rl][0][cq];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += sh[rl][k][cq];
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)rb * 2 + rl) * p.N + col) = t;
    }
}
}  // namespace

namespace {
// Split-K reduce of the fp16-activation convolution (conv3x3_f16dma.hip): splitk_reduce_stats_kernel with the fp16 residual stream and fp16
// output rows of that family -- out = act((sum over splits, in split order + colbias + cbias[img] + res) * scale), rounded to nearest even
// when the rows are fp16; the column sums left for the consumer's GroupNorm are those of the values as STORED, like the fused epilogues'.
// Block = 64 rows x 64 columns = one cell of the statistics; thread = (column quad, row lane; rows lane + 16 k).  N % 64 == 0.
__global__ void __launch_bounds__(256) splitk_reduce_f16_kernel(const KParams p) {
    __shared__ f32x4 sh[2][16][16];
    typedef _Float16 rh4 __attribute__((ext_vector_type(4)));
    const int rb = blockIdx.x, cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int col = blockIdx.y * 64 + cq * 4;
    const size_t plane = (size_t)p.M * p.N;
    f32x4 cb = {0.f, 0.f, 0.f, 0.f}, s4 = cb, q4 = cb;
    if (p.colbias) cb = *reinterpret_cast<const f32x4*>(p.colbias + col);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = rb * 64 + rl + 16 * k;
        if (row >= p.M) break;
        f32x4 v = *reinterpret_cast<const f32x4*>(p.part + (size_t)row * p.N + col);
        for (int sp = 1; sp < p.splits; ++sp) v += *reinterpret_cast<const f32x4*>(p.part + sp * plane + (size_t)row * p.N + col);
        v += cb;
        if (p.cbias) v += *reinterpret_cast<const f32x4*>(p.cbias + (size_t)(p.cbias_bcast ? 0 : row / p.HW) * p.cbias_ld + col);
        if (p.res) {
            if (p.res_f16) {
                const rh4 r = *reinterpret_cast<const rh4*>(reinterpret_cast<const _Float16*>(p.res) + (size_t)row * p.res_ld + col);
                v += f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
            } else {
                v += *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.res_ld + col);
            }
        }
        v *= p.scale;
        if (p.act == DS_ACT_SILU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ds_silu(v[j]);
        }
        if (p.out_f16) {
            const rh4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *reinterpret_cast<rh4*>(reinterpret_cast<_Float16*>(p.out) + (size_t)row * p.ldo + col) = h;
            v = f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        } else {
            *reinterpret_cast<f32x4*>(p.out + (size_t)row * p.ldo + col) = v;
        }
        s4 += v; q4 += v * v;
    }
    if (!p.stats) return;
    sh[0][rl][cq] = s4; sh[1][rl][cq] = q4;
    __syncthreads();
    if (rl < 2) {                                   // rl = 0: sums, 1: sums of squares; fixed order over the 16 row lanes
        f32x4 t = sh[rl][0][cq];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += sh[rl][k][cq];
        *reinterpret_cast<f32x4*>(p.stats + ((size_t)rb * 2 + rl) * p.N + col) = t;
    }
}
}  // namespace

int launch_splitk_reduce_f16(const KParams& p, hipStream_t stream) {
    if ((p.N & 63) || !p.vec_ok || !p.vec_part || p.out_planar || p.act == DS_ACT_GEGLU || p.splits < 2) return DS_E_SHAPE;
    hipLaunchKernelGGL(splitk_reduce_f16_kernel, dim3((p.M + 63) / 64, p.N / 64), dim3(256), 0, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int launch_splitk_reduce(const KParams& p, hipStream_t stream) {
    if (p.stats && p.vec_ok && p.vec_part && !p.out_planar && (p.N & 63) == 0 && p.act != DS_ACT_GEGLU) {
        hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((p.M + 63) / 64, p.N / 64), dim3(256), 0, stream, p);
        DS_CHECK_LAUNCH();
        return DS_OK;
    }
    long long blocks = ((long long)p.M * ((p.N + 3) / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    DS_CHECK_LAUNCH();
    if (p.stats) {
        hipLaunchKernelGGL(colstats_kernel, dim3((p.M + 63) / 64, (p.N + 63) / 64), dim3(256), 0, stream, p);
        DS_CHECK_LAUNCH();
    }
    return DS_OK;
}

}  // namespace igemm

using namespace igemm;

// ds_conv_args.tune -> KParams: per-call kernel selection overrides (include/ds_engine.h); the library keeps no selection state.
static void set_tune(KParams& p, const ds_conv_args* a) {
    p.t_mode = a->tune.mode; p.t_variant = a->tune.variant; p.t_splits = a->tune.splits > 0 ? a->tune.splits : 0;
    p.t_nb = a->tune.f16dma_nb; p.t_nw = a->tune.f16dma_nw; p.t_ablate = a->tune.ablate;
}

// ---- 1x1 / Linear on AT MOST FOUR rows (round 6): the embedding path.  The noise / label embedding MLP and the per-block affine projections
// batched into one wide Linear see ONE row per sampler call when every image shares sigma (512 -> 8 448 on the CIFAR-10 net, 1 280 -> 20 160 on
// SD-1.5): on the matrix kernel that is one useful row of a 128-row tile and a K loop of 16 - 40 serial tiles -- 24 - 94 us for 17 - 103 MB of
// weights (tools/time_plan_ops.py, sessions r10a / r10j).  Here a wave owns an output column: 64 lanes x 16 bytes of its weight row per step,
// the <= 4 input rows from cache, a cross-lane sum at the end -- the weights are read once at streaming rate.  fp32 products and sums (another
// order than the matrix kernel's: fp32 rounding apart).  Kernel id 2573; ds_conv_tune.mode != 0 keeps the matrix kernels.
namespace igemm {
namespace {
__global__ void __launch_bounds__(256) gemv_rows_kernel(const KParams p) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= p.N) return;
    const float* w = p.b + (size_t)n * p.ldb;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane * 4; k < p.K; k += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < p.M) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(p.a0 + (size_t)m * p.lda0 + k);
                acc[m] = __builtin_fmaf(xv[0], wv[0], __builtin_fmaf(xv[1], wv[1], __builtin_fmaf(xv[2], wv[2], __builtin_fmaf(xv[3], wv[3], acc[m]))));
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[m] += __shfl_xor(acc[m], off);
    }
    if (lane == 0) {
        const float bias = p.colbias ? p.colbias[n] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < p.M) {
                float v = (acc[m] + bias) * p.scale;
                if (p.act == DS_ACT_SILU) v = ds_silu(v);
                p.out[(size_t)m * p.ldo + n] = v;
            }
        }
    }
}
}  // namespace

bool gemv_rows_applicable(const KParams& p) {
    if (p.taps != 1 || p.stride != 1 || p.M < 1 || p.M > 4 || p.c1 || p.ec0 || p.ec1 || p.norm) return false;
    if (p.res || p.cbias || p.rowbias || p.stats || p.out_planar || p.out_f16 || p.splits > 1) return false;
    if (p.act != DS_ACT_NONE && p.act != DS_ACT_SILU) return false;
    if ((p.K & 3) || (p.lda0 & 3) || (p.ldb & 3) || !ds_aligned16(p.a0) || !ds_aligned16(p.b) || p.nrows_b < p.N) return false;
    return p.N >= 64;                                          // (narrow outputs are latency either way)
}

int launch_gemv_rows(const KParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(gemv_rows_kernel, dim3((unsigned)((p.N + 3) / 4)), dim3(256), 0, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}
}  // namespace igemm

extern "C" int ds_conv2d_nhwc(const ds_conv_args* a, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!a || !a->x0 || !a->wgt || !a->out) return DS_E_ARG;
    if (a->taps != 1 && a->taps != 9) return DS_E_ARG;
    if (a->c0 <= 0 || a->c0 % 32 || a->c1 < 0 || a->c1 % 32) return DS_E_SHAPE;
    if (a->c1 && !a->x1) return DS_E_ARG;
    if (a->out_f16 && (!a->in_f16 || (a->cout & 63) || (a->out_ld & 7) || !ds_aligned16(a->out))) return DS_E_ARG;   // fp16 output rows: the fp16-activation kernels only
    if (a->res_f16 && (!a->in_f16 || !a->res || (a->res_ld & 7) || !ds_aligned16(a->res))) return DS_E_ARG;          // fp16 residual rows: the same kernels
    if (a->in_f16) {          // fp16 activations: pure matrix kernels (conv3x3_f16dma.hip / gemm_f16dma.hip); ld in halfs, 16-byte chunks
        if (a->wgt_f16 != 1 || a->out_nchw) return DS_E_ARG;
        // fused input normalisation on RAW fp16 sources (round 5, conv3x3_f16dma NORM instantiations): 3x3 stride 1 only; with it a second
        // source per operand is allowed (the concatenation is not materialised); without it the operand is ONE activated fp16 tensor
        if (a->norm_coefs && (a->taps != 9 || a->stride > 1)) return DS_E_ARG;
        if (!a->norm_coefs && (a->c1 || a->ec1)) return DS_E_ARG;
        if (a->stride > 1 && (a->stride != 2 || a->taps != 9 || a->ec0 || a->res)) return DS_E_ARG;      // stride 2: the Downsample convolution, csrc/gemm_f16dma.hip (gather)
        if (a->taps == 1 && a->ec0) return DS_E_ARG;
        if ((a->ld0 & 7) || (a->ec0 && ((a->eld0 & 7) || !a->e0 || !ds_aligned16(a->e0)))) return DS_E_ALIGN;
        if ((a->c1 && (a->ld1 & 7)) || (a->ec1 && ((a->eld1 & 7) || !a->e1 || !ds_aligned16(a->e1)))) return DS_E_ALIGN;
    }
    if ((a->ld0 & 3) || (a->c1 && (a->ld1 & 3))) return DS_E_ALIGN;
    if (!ds_aligned16(a->x0) || (a->c1 && !ds_aligned16(a->x1)) || !ds_aligned16(a->wgt)) return DS_E_ALIGN;
    if (a->n <= 0 || a->h <= 0 || a->w <= 0 || a->cout <= 0) return DS_E_ARG;
    const long long M = (long long)a->n * a->h * a->w;
    if (M > 0x7fffffffLL - BM) return DS_E_SHAPE;
    KParams p{};
    set_tune(p, a);
    p.a0 = a->x0; p.a1 = a->x1; p.c0 = a->c0; p.c1 = a->c1; p.lda0 = a->ld0; p.lda1 = a->ld1;
    p.H = a->h; p.W = a->w; p.HW = a->h * a->w; p.taps = a->taps;
    const int stride = a->stride ? a->stride : 1;
    if (stride != 1 && (stride != 2 || a->taps != 9 || a->norm_coefs || a->ec0)) return DS_E_ARG;
    p.stride = stride; p.IH = a->h * stride; p.IW = a->w * stride;
    if (a->ec0 < 0 || a->ec1 < 0 || a->ec0 % 32 || a->ec1 % 32 || (a->ec1 && !a->ec0)) return DS_E_SHAPE;
    if (a->ec0) {
        if (a->taps != 9 || !a->e0 || (a->ec1 && !a->e1)) return DS_E_ARG;
        if ((a->eld0 & 3) || (a->ec1 && (a->eld1 & 3)) || !ds_aligned16(a->e0) || (a->ec1 && !ds_aligned16(a->e1))) return DS_E_ALIGN;
    }
    if (a->norm_coefs && (a->taps != 9 || !ds_aligned16(a->norm_coefs))) return DS_E_ARG;
    p.M = (int)M; p.N = a->cout; p.K = a->taps * (a->c0 + a->c1) + a->ec0 + a->ec1;
    p.b = a->wgt; p.ldb = p.K; p.nrows_b = ((a->cout + BN - 1) / BN) * BN;   // weights are row-padded
    p.norm = a->norm_coefs; p.norm_act = a->norm_act;
    p.e0 = a->e0; p.e1 = a->e1; p.ec0 = a->ec0; p.ec1 = a->ec1; p.elda0 = a->eld0; p.elda1 = a->eld1;
    p.out = a->out; p.ldo = a->out_ld;
    p.colbias = a->bias; p.rowbias = nullptr;
    p.cbias = a->cbias; p.cbias_ld = a->cbias_ld; p.cbias_bcast = (a->cbias_rows == 1);
    p.res = a->res; p.res_ld = a->res_ld;
    p.scale = a->out_scale; p.act = a->act; p.heads = 1; p.acc_scale = 1.0f;
    p.vec_ok = vec_epilogue_ok(p) ? 1 : 0;
    p.stats = nullptr;
    if (a->stats_out) {
        if ((a->cout & 63) || !p.vec_ok || a->out_nchw || !ds_aligned16(a->stats_out)) return DS_E_ARG;
        p.stats = a->stats_out;
    }
    p.out_planar = 0;
    if (a->out_nchw) {
        if (a->cout >= 64) return DS_E_ARG;
        p.out_planar = 1; p.vec_ok = 0;
    }
    p.splits = 1; p.part = nullptr; p.part_cap = 0; p.vec_part = 0;
    if (a->act == DS_ACT_GEGLU) {
        // gate fused into the epilogue: needs the staged float4 path on whole 64-column wave tiles, and no split-K
        if (a->taps != 1 || (a->cout & 63) || !p.vec_ok || a->res || a->cbias || a->out_nchw || a->stats_out) return DS_E_ARG;
    } else
    if (a->workspace && a->workspace_floats > 0 && ds_aligned16(a->workspace)) {
        p.part = a->workspace; p.part_cap = a->workspace_floats; p.vec_part = (p.N & 3) ? 0 : 1;
    }
    if (a->wgt_f16) {
        if (a->update && (a->update->x_out || a->update->m_out)) return DS_E_ARG;      // the fused solver update exists in the fp32 head kernel only
        // fp16 operands (1) or split fp16 hi/lo operands (2): second-generation halo kernel only, every 128-column tile (the ragged
        // last one included)
        if (a->wgt_f16 == 1 && a->taps == 1 && stride == 1) {
            // 1x1 / Linear with fp16 operands: weights [cout_pad][K] halfs in plain K order
            if (a->wgt_shift) return DS_E_ARG;
            p.ldb = p.K / 2;
            p.part = nullptr; p.part_cap = 0; p.splits = 1;
#ifdef DS_TIMELINE
            if ((p.t_ablate & 0x8000) && a->workspace) p.part = a->workspace;         // diagnostics build only: phase stamps (ds_common.h)
#endif
            if (a->in_f16) {            // fp16 activations: both operands by LDS-DMA
                p.out_f16 = a->out_f16 ? 1 : 0; p.res_f16 = a->res_f16 ? 1 : 0;
                if (!gemm_f16dma_applicable(p)) return DS_E_SHAPE;
                return launch_gemm_f16dma(p, (hipStream_t)stream);
            }
            if (!gemm_f16_applicable(p)) return DS_E_SHAPE;
            return launch_gemm_f16(p, (hipStream_t)stream);
        }
        if (a->wgt_f16 == 1 && a->in_f16 && a->taps == 9 && stride == 2) {
            // the latent-diffusion Downsample on fp16 rows: the fp16-activation GEMM with a gathered A tile (weights packed like the stride-1 kernel's)
            if (a->wgt_shift) return DS_E_ARG;
            p.ldb = p.K / 2;
            p.out_f16 = a->out_f16 ? 1 : 0; p.res_f16 = 0;
            p.part = nullptr; p.part_cap = 0; p.splits = 1;
            if (!gemm_f16dma_gather_applicable(p)) return DS_E_SHAPE;
            return launch_gemm_f16dma(p, (hipStream_t)stream, true);
        }
        if (a->taps != 9 || stride != 1 || (a->wgt_f16 != 1 && a->wgt_f16 != 2)) return DS_E_ARG;
        if (a->wgt_shift < 0 || a->wgt_shift > 24 || (a->wgt_f16 == 1 && a->wgt_shift)) return DS_E_ARG;
        if (a->in_f16) {
            p.ldb = p.K / 2;
            p.out_f16 = a->out_f16 ? 1 : 0; p.res_f16 = a->res_f16 ? 1 : 0;
            p.splits = 1;                                   // p.part stays: under-filled layers split K (conv3x3_f16dma_splits)
            if (!conv3x3_f16dma_applicable(p)) return DS_E_SHAPE;
            return launch_conv3x3_f16dma(p, (hipStream_t)stream);
        }
        const int wide = (p.N + BN - 1) / BN;
        // row pitch of the fp16 weight matrix in float units: fp16 = K halfs per row, split = 2 K halfs (hi and lo)
        p.ldb = a->wgt_f16 == 1 ? p.K / 2 : p.K;
        p.acc_scale = 1.0f / (float)(1 << a->wgt_shift);
        p.part = nullptr; p.part_cap = 0; p.splits = 1;
        if (!conv3x3_halo2_applicable(p, wide, a->wgt_f16)) return DS_E_SHAPE;
        return launch_conv3x3_halo2(p, wide, a->wgt_f16, (hipStream_t)stream);
    }
    const bool generic = p.t_mode == 1;                    // tune.mode 1: the generic gather kernel (A/B runs, cross-checks)
    // network heads (cout <= 4): VALU kernel instead of a 64- / 128-column matrix tile (tune.mode != 0 keeps the matrix kernels: 8 = just that)
    p.upd = a->update;
    const bool want_update = a->update && (a->update->x_out || a->update->m_out);
    if (p.t_mode == 0 && p.t_variant == 0 && conv3x3_thin_applicable(p)) return launch_conv3x3_thin(p, (hipStream_t)stream);
    if (want_update) return DS_E_ARG;                      // the fused solver update exists in the head kernel only: fail loudly, never skip it
    if (!generic && stride == 1 && conv3x3_halo_supported(p)) return launch_conv3x3_halo(p, (hipStream_t)stream);
    if (p.norm) return DS_E_SHAPE;           // fused input normalisation exists only in the halo kernel
    if (!generic && p.t_mode != 6 && gemm_dma8_applicable(p)) return launch_gemm_dma8(p, (hipStream_t)stream);     // mode 6: no 8-wave DMA kernel
    if (p.t_mode == 0 && p.t_variant == 0 && gemv_rows_applicable(p)) return launch_gemv_rows(p, (hipStream_t)stream);
    return launch<0>(p, 1, (hipStream_t)stream);
}

extern "C" int ds_conv_kernel_id(const ds_conv_args* a) {
    if (!a) return DS_E_ARG;
    KParams p{};
    set_tune(p, a);
    p.taps = a->taps; p.H = a->h; p.W = a->w; p.HW = a->h * a->w; p.M = a->n * a->h * a->w; p.N = a->cout;
    p.c0 = a->c0; p.c1 = a->c1; p.ec0 = a->ec0; p.ec1 = a->ec1;
    if (a->workspace && a->workspace_floats > 0 && ds_aligned16(a->workspace) && a->act != DS_ACT_GEGLU) {
        p.part = a->workspace; p.part_cap = a->workspace_floats; p.vec_part = (p.N & 3) ? 0 : 1;
    }
    // the epilogue-related fields the tile choice looks at, as ds_conv2d_nhwc sets them
    p.out = a->out; p.ldo = a->out_ld; p.colbias = a->bias; p.cbias = a->cbias; p.cbias_ld = a->cbias_ld; p.res = a->res; p.res_ld = a->res_ld;
    p.act = a->act; p.out_planar = a->out_nchw ? 1 : 0;
    p.vec_ok = (vec_epilogue_ok(p) && !a->out_nchw) ? 1 : 0;
    p.stride = a->stride ? a->stride : 1; p.K = a->taps * (a->c0 + a->c1) + a->ec0 + a->ec1; p.norm = a->norm_coefs;
    p.nrows_b = ((a->cout + BN - 1) / BN) * BN;                                     // weights are row-padded, as in ds_conv2d_nhwc
    if (a->wgt_f16 == 1 && a->in_f16 && a->taps == 9 && p.stride == 2) return 2571;
    if (a->wgt_f16 == 1 && a->in_f16) return a->taps == 1 ? 2567 : (conv3x3_f16dma_use_half(p) ? 2569 : (a->norm_coefs ? 2572 : 2566));
    if (a->wgt_f16) return a->wgt_f16 == 2 ? 2563 : (a->taps == 1 ? 2564 : 2562);
    if (p.t_mode == 1) return 0;
    if (p.t_mode == 0 && p.t_variant == 0 && !a->res && !a->cbias && !a->stats_out) {
        KParams q = p; q.HW = p.HW; q.res = nullptr; q.cbias = nullptr; q.stats = nullptr; q.splits = 1; q.norm_act = a->norm_act;
        if (conv3x3_thin_applicable(q)) return 2570;
    }
    if (a->taps != 9 || a->stride > 1) {
        if (p.t_mode != 6 && gemm_dma8_applicable(p)) return 2561;
        KParams q = p; q.a0 = a->x0; q.lda0 = a->ld0; q.b = a->wgt; q.ldb = p.K; q.scale = a->out_scale; q.out_f16 = 0; q.splits = 1;
        q.rowbias = nullptr; q.stats = a->stats_out;
        return (p.t_mode == 0 && p.t_variant == 0 && gemv_rows_applicable(q)) ? 2573 : 0;
    }
    return conv3x3_halo_choice(p);
}

static int reduced_supported(int mode, int n, int h, int w, int c0, int c1, int ec0, int ec1) {
    KParams p{};
    p.taps = 9; p.H = h; p.W = w; p.HW = h * w; p.M = n * h * w; p.N = 128; p.c0 = c0; p.c1 = c1; p.ec0 = ec0; p.ec1 = ec1;
    p.nrows_b = 128;
    if (!conv3x3_halo2_applicable(p, 1, mode)) return 0;
    return w >= 16 ? 2 : 1;                      // 8x8: four images per tile, the per-image normalisation planes are not fused
}
extern "C" int ds_conv_f16_supported(int n, int h, int w, int c0, int c1, int ec0, int ec1) { return reduced_supported(1, n, h, w, c0, c1, ec0, ec1); }
extern "C" int ds_conv_f16dma_supported(int n, int h, int w, int c0, int ec0, int cout) {
    KParams p{};
    p.taps = 9; p.stride = 1; p.H = h; p.W = w; p.HW = h * w; p.M = n * h * w; p.N = cout; p.c0 = c0; p.ec0 = ec0;
    p.vec_ok = 1; p.nrows_b = ((cout + BN - 1) / BN) * BN;
    return conv3x3_f16dma_applicable(p) ? 1 : 0;
}
extern "C" int ds_gemm_f16dma_supported(long long rows, int k, int cout) {
    KParams p{};
    if (rows > 0x7fffffffLL || rows < 1) return 0;
    p.taps = 1; p.stride = 1; p.M = (int)rows; p.N = cout; p.K = k; p.c0 = k; p.vec_ok = 1; p.nrows_b = ((cout + BN - 1) / BN) * BN;
    return gemm_f16dma_applicable(p) ? 1 : 0;
}
extern "C" int ds_conv_f16dma_stride2_supported(int n, int h, int w, int c0, int cout) {
    KParams p{};
    const long long M = (long long)n * h * w;
    if (n < 1 || h < 1 || w < 1 || M > 0x7fffffffLL / 4) return 0;
    p.taps = 9; p.stride = 2; p.H = h; p.W = w; p.HW = h * w; p.IH = 2 * h; p.IW = 2 * w; p.M = (int)M; p.N = cout; p.c0 = c0; p.K = 9 * c0;
    p.vec_ok = 1; p.nrows_b = ((cout + BN - 1) / BN) * BN;
    return gemm_f16dma_gather_applicable(p) ? 1 : 0;
}
extern "C" int ds_gemm_f16_supported(long long rows, int c0, int c1) {
    KParams p{};
    if (rows > 0x7fffffffLL) return 0;
    p.taps = 1; p.stride = 1; p.M = (int)rows; p.N = 128; p.K = c0 + c1; p.c0 = c0; p.c1 = c1; p.nrows_b = 128;
    return gemm_f16_applicable(p) ? 1 : 0;
}
extern "C" int ds_conv_split_supported(int n, int h, int w, int c0, int c1, int ec0, int ec1) { return reduced_supported(2, n, h, w, c0, c1, ec0, ec1); }

extern "C" int ds_conv3x3_halo_supported(int h, int w) {
    KParams p{};
    p.taps = 9; p.H = h; p.W = w; p.HW = h * w;
    return conv3x3_halo_supported(p) ? 1 : 0;
}

extern "C" int ds_gemm_nt_batched(const ds_gemm_args* a, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!a || !a->a || !a->b || !a->c) return DS_E_ARG;
    if (a->m <= 0 || a->n <= 0 || a->k <= 0 || a->batch <= 0 || a->heads <= 0) return DS_E_ARG;
    if (a->k % 32) return DS_E_SHAPE;
    if ((a->lda & 3) || (a->ldb & 3) || (a->a_bstride & 3) || (a->a_hstride & 3) || (a->b_bstride & 3) || (a->b_hstride & 3))
        return DS_E_ALIGN;
    if (!ds_aligned16(a->a) || !ds_aligned16(a->b)) return DS_E_ALIGN;
    KParams p{};
    p.a0 = a->a; p.lda0 = a->lda; p.a_bs = a->a_bstride; p.a_hs = a->a_hstride;
    p.b = a->b; p.ldb = a->ldb; p.b_bs = a->b_bstride; p.b_hs = a->b_hstride; p.nrows_b = a->n;
    p.out = a->c; p.ldo = a->ldc; p.o_bs = a->c_bstride; p.o_hs = a->c_hstride;
    p.M = a->m; p.N = a->n; p.K = a->k; p.HW = 1; p.H = p.W = 1; p.taps = 1; p.c0 = a->k; p.c1 = 0;
    p.stride = 1; p.IH = p.IW = 1;
    p.colbias = a->colbias; p.rowbias = a->rowbias; p.cbias = nullptr; p.res = nullptr;
    p.scale = a->alpha; p.act = a->act; p.heads = a->heads; p.acc_scale = 1.0f;
    p.vec_ok = vec_epilogue_ok(p) ? 1 : 0; p.out_planar = 0; p.stats = nullptr;
    p.splits = 1; p.part = nullptr; p.part_cap = 0; p.vec_part = 0;
    return launch<1>(p, a->batch * a->heads, (hipStream_t)stream);
}
