// Shared pieces of the fp32-MFMA contraction kernels (gemm_conv.hip, conv3x3_halo.hip): tile constants, kernel
// parameter block, fused epilogue.
#pragma once
#include "ds_common.h"

namespace igemm {

constexpr int BM = 128, BN = 128, BK = 32, LDSK = 36;
constexpr int EPI_LD = 68;                                              // 64 + 4 floats: epilogue staging row
constexpr int SMEM_BYTES = 2 * (BM + BN) * LDSK * (int)sizeof(float);   // 73,728 B -> 2 blocks / CU
static_assert(4 * 64 * EPI_LD * (int)sizeof(float) <= SMEM_BYTES, "epilogue staging must fit in the tile buffers");

struct KParams {
    // A side (conv gather)
    const float* a0; const float* a1; int c0, c1, lda0, lda1; int H, W, HW, taps;
    int stride, IH, IW;            // generic kernel only: output stride and INPUT size (= H, W when stride == 1)
    // A side (gemm) uses a0/lda0 plus batch strides
    long long a_bs, a_hs;
    // B side
    const float* b; int ldb; long long b_bs, b_hs; int nrows_b;   // rows of B that may be read
    int M, N, K;
    int mtiles, ntiles;
    int n_begin;                   // first output column of this launch (halo kernel: the 64-column tail launch)
    // halo kernel geometry: a 128-pixel M tile = nimg image slots x TH rows x W columns
    int TH, nimg, HP, WP, NP;      // HP = TH + 2, WP = W + 2, NP = nimg * HP * WP halo pixels
    // fused input normalisation of the 3x3 sources: planes [n][3][c0+c1] = {mu, A, B}; in = act((x - mu) * A + B)
    const float* norm; int norm_act;
    int coef_lds;                  // halo kernel, tiles of several images: the planes of the tile's images are staged in LDS (set by the launcher)
    // extra 1x1 sources appended along K after the 9*(c0+c1) columns (skip projection fused into the conv)
    const float* e0; const float* e1; int ec0, ec1, elda0, elda1;
    // epilogue
    float* out; int ldo; long long o_bs, o_hs;
    const float* colbias; const float* rowbias;
    const float* cbias; int cbias_ld; int cbias_bcast;
    const float* res; int res_ld;
    float scale; int act; int heads;
    float acc_scale;                                               // conv epilogue: accumulators are multiplied by this first (1; 2**-shift for pre-scaled split-fp16 weights)
    int vec_ok;                                                    // float4 epilogue allowed (alignment, ld % 4)
    int out_planar;                                                // scalar epilogue writes out[(img * N + col) * HW + pixel]
    int out_f16;                                                   // vector epilogue stores fp16 rows [M][ldo halfs] (ds_conv_args.out_f16)
    int res_f16;                                                   // `res` is an fp16 tensor [M][res_ld halfs] (ds_conv_args.res_f16): the fp16 residual stream
    // optional per-(64-row block, column) sums of the OUTPUT for the consumer's GroupNorm: stats[(rb * 2 + {0: sum, 1: sum of
    // squares}) * N + col], rb = row / 64 (vector epilogue only: N % 64 == 0)
    float* stats;
    // split-K (small-M layers): blockIdx.y = split; each split contracts a contiguous range of K slabs / tiles and writes
    // its raw partial tile to part[split][M][N]; splitk_reduce_kernel sums them and applies the epilogue
    int splits; float* part; int vec_part; long long part_cap;     // part_cap: workspace capacity in floats (host side only)
    // per-call kernel selection overrides (ds_conv_args.tune, all 0 = the library's own choice; host side only).  There is no process-wide
    // selection state: two threads / streams may run layers with different overrides at the same time.
    int t_mode, t_variant, t_splits, t_nb, t_nw, t_ablate;
    // the solver update fused into the network head (ds_conv_args.update; host pointer, read by launch_conv3x3_thin only)
    const ds_update_args* upd;
};

// Fused epilogue of one wave's 64x64 accumulator tile (2x2 MFMA 32x32 tiles).
//   conv (MODE 0): out = act((acc + colbias + cbias[img] + res) * scale)
//   gemm (MODE 1): out = act(acc * scale + colbias + rowbias)
// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// Vector path: the tile is transposed through LDS (`stage`, 64 x EPI_LD floats owned by this wave, free once every
// wave of the block has passed the K loop's last barrier) so that bias / residual / output are accessed as float4
// rows (16 B per lane, 256 B contiguous per 16 lanes) instead of 64 dword accesses per lane.
// NTS: residual loads and output stores carry the non-temporal hint (read-once / write-once streams of a tile's epilogue): +0.6 % on the
// 256 x 256 conv tile; as a run-time option for every other kernel of the family (fp32 / fp16 GEMMs, fp16 convolutions; outputs of at
// least 32 MiB) it changed nothing on the CIFAR-10, ImageNet-64 fp16 and SD-1.5 fp16 benches (profiles/r2_conv_tile_options.txt), so
// only that kernel instantiates it.
template <int MODE, bool HALF = false, bool NTS = false>
__device__ __forceinline__ void epilogue(const KParams& p, const f32x16 (&acc)[2][2], float* stage, int lane, int wm0, int wn0,
                                         float* o_base) {
    // HALF: the staging area holds 32 x EPI_LD floats per wave (8-wave blocks) and the two 32-row halves go one after
    // the other; otherwise 64 x EPI_LD and the whole tile is staged at once.
    const bool full_cols = (wn0 + 64 <= p.N);
    if (p.vec_ok && full_cols) {
        const int c4 = (lane & 15) * 4;
        const int col = wn0 + c4;
        f32x4 cb = {0.f, 0.f, 0.f, 0.f};
        if (p.colbias) cb = *reinterpret_cast<const f32x4*>(p.colbias + col);
        f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};
        const bool geglu = (MODE == 0) && p.act == DS_ACT_GEGLU;      // columns [0,32) of the wave tile: values, [32,64): their gates
        f32x4 cbg = {0.f, 0.f, 0.f, 0.f};
        if (geglu && p.colbias && c4 < 32) cbg = *reinterpret_cast<const f32x4*>(p.colbias + col + 32);
        // Residual rows and the per-image bias of a whole group of 8 passes (32 rows) are requested BEFORE the tile goes through
        // LDS: one memory latency per group instead of one per unrolled quartet of passes, and no per-row integer division where
        // the rows of a group belong to one image (H*W a multiple of 32: every convolution; Linear layers have H*W = 1).
        constexpr int NP = 8;
        const bool cb_uniform = p.cbias && (p.cbias_bcast || p.HW % 32 == 0);
        f32x4 rv[NP], cvu = {0.f, 0.f, 0.f, 0.f};
        auto prefetch = [&](int rbase) {
            if (p.res) {
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) {
                    const int row = min(rbase + pass * 4 + (lane >> 4), p.M - 1);
                    const f32x4* rp = reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.res_ld + col);
                    rv[pass] = NTS ? __builtin_nontemporal_load(rp) : *rp;
                }
            }
            if (cb_uniform) {
                const int img = p.cbias_bcast ? 0 : __builtin_amdgcn_readfirstlane(min(rbase, p.M - 1)) / p.HW;
                cvu = *reinterpret_cast<const f32x4*>(p.cbias + (size_t)img * p.cbias_ld + col);
            }
        };
        auto process = [&](int rr0, int rbase) {                        // rows rbase .. rbase + 31 = staging rows rr0 .. rr0 + 31
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int rr = rr0 + pass * 4 + (lane >> 4);
                const int row = rbase + pass * 4 + (lane >> 4);
                if (row >= p.M) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + rr * EPI_LD + c4);
                if (MODE == 0) v *= p.acc_scale;
                if (MODE == 1) v *= p.scale;
                v += cb;
                if (p.rowbias) v += p.rowbias[row];
                if (cb_uniform) v += cvu;
                else if (p.cbias) v += *reinterpret_cast<const f32x4*>(p.cbias + (size_t)(row / p.HW) * p.cbias_ld + col);
                if (p.res) v += rv[pass];
                if (MODE == 0) v *= p.scale;
                if (geglu) {
                    if (c4 < 32) {
                        const f32x4 gt = *reinterpret_cast<const f32x4*>(stage + rr * EPI_LD + c4 + 32) + cbg;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] *= 0.5f * gt[q] * (1.0f + erff(gt[q] * 0.70710678118654752440f));
                        *reinterpret_cast<f32x4*>(o_base + (size_t)row * p.ldo + (wn0 >> 1) + c4) = v;
                    }
                    continue;
                }
                if (p.act == DS_ACT_SILU) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = ds_silu(v[q]);
                }
                if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(o_base + (size_t)row * p.ldo + col));
                else *reinterpret_cast<f32x4*>(o_base + (size_t)row * p.ldo + col) = v;
                st_s += v; st_q += v * v;
            }
        };
#pragma unroll
        for (int half = 0; half < (HALF ? 2 : 1); ++half) {
            prefetch(wm0 + half * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (HALF && i != half) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[((HALF ? 0 : i * 32) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + j * 32 + (lane & 31)] = acc[i][j][r];
            }
            process(0, wm0 + half * 32);
            if (!HALF) {                                                // 4-wave tiles stage all 64 rows at once: second group
                prefetch(wm0 + 32);
                process(32, wm0 + 32);
            }
        }
        if (p.stats && wm0 < p.M) {
            // column sums of this wave's 64 rows: the 4 lane groups (lane >> 4) hold disjoint rows of the same 4 columns
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                st_s[q] += __shfl_xor(st_s[q], 16); st_s[q] += __shfl_xor(st_s[q], 32);
                st_q[q] += __shfl_xor(st_q[q], 16); st_q[q] += __shfl_xor(st_q[q], 32);
            }
            if (lane < 16) {
                float* sp = p.stats + (size_t)(wm0 >> 6) * 2 * p.N + col;
                *reinterpret_cast<f32x4*>(sp) = st_s;
                *reinterpret_cast<f32x4*>(sp + p.N) = st_q;
            }
        }
        return;
    }
    // scalar fallback (ragged N such as the 3-channel output conv, or unaligned leading dimensions)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn0 + j * 32 + (lane & 31);
        if (col >= p.N) continue;
        const float cb = p.colbias ? p.colbias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                float v = acc[i][j][r];
                if (MODE == 0) v *= p.acc_scale;
                if (MODE == 1) v *= p.scale;
                v += cb;
                if (p.rowbias) v += p.rowbias[row];
                if (p.cbias) {
                    const int img = p.cbias_bcast ? 0 : row / p.HW;
                    v += p.cbias[(size_t)img * p.cbias_ld + col];
                }
                if (p.res) v += p.res[(size_t)row * p.res_ld + col];
                if (MODE == 0) v *= p.scale;
                if (p.act == DS_ACT_SILU) v = ds_silu(v);
                if (p.out_planar) { const int im = row / p.HW; o_base[((size_t)im * p.N + col) * p.HW + (row - im * p.HW)] = v; }
                else o_base[(size_t)row * p.ldo + col] = v;
            }
        }
    }
}

// Fused conv epilogue of a 64 x 32 wave tile (acc[i][0], i = the two 32-row MFMA blocks): the vector path of igemm_common.h's epilogue
// re-laid for 32 columns -- 8 lanes x float4 cover a row segment, lane >> 3 picks one of 8 rows per pass, 4 passes per 32-row group.
// The launcher only takes this kernel where the vector path is legal (p.vec_ok, whole 32-column tiles) and there is no GEGLU gate.
//   out = act((acc * acc_scale + colbias + cbias[img] + res) * scale), column sums / sums of squares of the wave's 64 rows to p.stats
template <bool HALF>
__device__ __forceinline__ void epilogue32(const KParams& p, const f32x16 (&acc)[2][2], float* stage, int lane, int wm0, int wn0, float* o_base) {
    static_assert(HALF, "8-wave tiles: 32 staging rows per wave, the two 32-row groups one after the other");
    const int c4 = (lane & 7) * 4;
    const int col = wn0 + c4;
    f32x4 cb = {0.f, 0.f, 0.f, 0.f};
    if (p.colbias) cb = *reinterpret_cast<const f32x4*>(p.colbias + col);
    f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};
    constexpr int NP = 4;
    const bool cb_uniform = p.cbias && (p.cbias_bcast || p.HW % 32 == 0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int rbase = wm0 + half * 32;
        f32x4 rv[NP], cvu = {0.f, 0.f, 0.f, 0.f};
        if (p.res) {
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int row = min(rbase + pass * 8 + (lane >> 3), p.M - 1);
                rv[pass] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.res_ld + col);
            }
        }
        if (cb_uniform) {
            const int img = p.cbias_bcast ? 0 : __builtin_amdgcn_readfirstlane(min(rbase, p.M - 1)) / p.HW;
            cvu = *reinterpret_cast<const f32x4*>(p.cbias + (size_t)img * p.cbias_ld + col);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + (lane & 31)] = acc[half][0][r];
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int rr = pass * 8 + (lane >> 3);
            const int row = rbase + rr;
            if (row >= p.M) continue;
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + rr * EPI_LD + c4);
            v *= p.acc_scale;
            v += cb;
            if (cb_uniform) v += cvu;
            else if (p.cbias) v += *reinterpret_cast<const f32x4*>(p.cbias + (size_t)(row / p.HW) * p.cbias_ld + col);
            if (p.res) v += rv[pass];
            v *= p.scale;
            if (p.act == DS_ACT_SILU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ds_silu(v[q]);
            }
            *reinterpret_cast<f32x4*>(o_base + (size_t)row * p.ldo + col) = v;
            st_s += v; st_q += v * v;
        }
    }
    if (p.stats && wm0 < p.M) {
        // column sums of this wave's 64 rows: the 8 lane groups (lane >> 3) hold disjoint rows of the same 4 columns
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            st_s[q] += __shfl_xor(st_s[q], 8); st_s[q] += __shfl_xor(st_s[q], 16); st_s[q] += __shfl_xor(st_s[q], 32);
            st_q[q] += __shfl_xor(st_q[q], 8); st_q[q] += __shfl_xor(st_q[q], 16); st_q[q] += __shfl_xor(st_q[q], 32);
        }
        if (lane < 8) {
            float* sp = p.stats + (size_t)(wm0 >> 6) * 2 * p.N + col;
            *reinterpret_cast<f32x4*>(sp) = st_s;
            *reinterpret_cast<f32x4*>(sp + p.N) = st_q;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Epilogue of the fp16-activation kernels (conv3x3_f16dma.hip, gemm_f16dma.hip): a 64-row wave tile made of one or two column blocks
// (WA = 32 / 64 columns at wn0, WB = 0 / 32 / 64 more at wn0 + WA), the same arithmetic as epilogue / epilogue32 above, with
//   * fp16 or fp32 residual rows and output rows (p.res_f16 / p.out_f16: the fp16 residual stream of the reference's fp16 mode);
//   * EIGHT columns per lane: a lane's residual / output access is one 16-byte vector of fp16 (two of fp32), a wave instruction covers
//     eight (W = 64) or sixteen (W = 32) whole row segments.  These layers are output-heavy (a K = 320 projection moves as many bytes
//     in its epilogue as in its main loop) and the epilogue is bound by the bytes a wave keeps in flight: with four columns per lane
//     (512 B per instruction) it was 5 - 10 % slower on the fp16 tensors (profiles/r3_gemm_f16dma_epilogue.txt);
//   * group by group (a group = 32 rows of one block) with the residual rows and per-image bias of the NEXT group requested before the
//     current group is staged, widened only where they are used (a conversion at the request would wait for the data: measured
//     +0.05 ms on a 0.25 ms convolution);
//   * the column sums left for the consumer's GroupNorm are those of the values as STORED (rounded when the output is fp16).
// GELU gate of the fp16-mode GEGLU epilogue: 0.5 g (1 + erf(g / sqrt 2)) with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute,
// evaluated as erfc on the negative side so the tail does not cancel) on the hardware reciprocal and exp2 -- about a third of erff's
// instructions.  The product is rounded to fp16 (2**-11 relative) right after, and the fp32 mode keeps erff (igemm_common.h:epilogue,
// norm_act.hip:geglu_kernel).  On SD-1.5 the 320 -> 2 560 GEGLU projection evaluates 168 M gates per call: with erff the epilogue's
// VALU work was as long as the whole main loop.
__device__ __forceinline__ float ds_gelu_gate_fast(float g) {
    const float x = g * 0.70710678118654752440f, ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
    const float poly = t * __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, __builtin_fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float pe = poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);          // erfc(|x|)
    return 0.5f * g * (x < 0.f ? pe : 2.0f - pe);
}

template <int W> struct EpiGeo { static constexpr int LPR = W / 8, RPP = 64 / LPR, NP = 32 / RPP; };
typedef _Float16 epi_h8 __attribute__((ext_vector_type(8)));
typedef unsigned epi_u4 __attribute__((ext_vector_type(4)));
// The requested rows of one group, RAW: an fp16 row segment (8 halfs) in raw[pass][0], an fp32 one (8 floats) in raw[pass][0 .. 1].
struct EpiRows { epi_u4 raw[4][2]; f32x4 cvu[2]; };
// fp16 residual rows of one group (16 VGPRs): small enough to request the rows of ALL groups of the wave tile when the epilogue starts --
// one memory latency for the tile instead of one per group (the request one group ahead is ~1 us ahead; HBM under load answers in 2 - 4).
struct EpiRes16 { epi_u4 r[4]; };

template <int W>
__device__ __forceinline__ void epi_request_res16(const KParams& p, int rbase, int col, int lane, EpiRes16& e) {
    using G = EpiGeo<W>;
#pragma unroll
    for (int pass = 0; pass < G::NP; ++pass) {
        const int row = min(rbase + pass * G::RPP + lane / G::LPR, p.M - 1);
        e.r[pass] = __builtin_nontemporal_load(reinterpret_cast<const epi_u4*>(reinterpret_cast<const _Float16*>(p.res) + (size_t)row * p.res_ld + col));
    }
}

// R16: the residual rows are fp16 and were requested up front (EpiRes16): only the per-image bias row is fetched here
template <int W, bool NTS, bool R16 = false>
__device__ __forceinline__ void epi_request(const KParams& p, int rbase, int col, int lane, EpiRows& e) {
    using G = EpiGeo<W>;
    if (!R16 && p.res) {
#pragma unroll
        for (int pass = 0; pass < G::NP; ++pass) {
            const int row = min(rbase + pass * G::RPP + lane / G::LPR, p.M - 1);
            if (p.res_f16) {
                e.raw[pass][0] = __builtin_nontemporal_load(reinterpret_cast<const epi_u4*>(reinterpret_cast<const _Float16*>(p.res) + (size_t)row * p.res_ld + col));
            } else {
                const epi_u4* rp = reinterpret_cast<const epi_u4*>(p.res + (size_t)row * p.res_ld + col);
                e.raw[pass][0] = NTS ? __builtin_nontemporal_load(rp) : rp[0];
                e.raw[pass][1] = NTS ? __builtin_nontemporal_load(rp + 1) : rp[1];
            }
        }
    }
    e.cvu[0] = e.cvu[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.cbias && (p.cbias_bcast || p.HW % 32 == 0)) {
        const int img = p.cbias_bcast ? 0 : __builtin_amdgcn_readfirstlane(min(rbase, p.M - 1)) / p.HW;
        const f32x4* cp = reinterpret_cast<const f32x4*>(p.cbias + (size_t)img * p.cbias_ld + col);
        e.cvu[0] = cp[0]; e.cvu[1] = cp[1];
    }
}

// a0 / a1: the group's accumulators of columns [0, 32) / [32, 64) of the block (a1 unused for W = 32) -> the wave's 32 staging rows
// TR: the accumulators of a SWAPPED product (weights as the MFMA's first operand: lane = pixel lane & 31, register r = channel
// 8 (r >> 2) + 4 (lane >> 5) + (r & 3); see epilogue_direct) -- four consecutive channels per 16-byte write.
template <int W, bool TR = false>
__device__ __forceinline__ void epi_stage(const f32x16& a0, const f32x16& a1, float* stage, int lane) {
    if constexpr (TR) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* sp = stage + (lane & 31) * EPI_LD + 8 * q + 4 * (lane >> 5);
            *reinterpret_cast<f32x4*>(sp) = f32x4{a0[4 * q], a0[4 * q + 1], a0[4 * q + 2], a0[4 * q + 3]};
            if (W == 64) *reinterpret_cast<f32x4*>(sp + 32) = f32x4{a1[4 * q], a1[4 * q + 1], a1[4 * q + 2], a1[4 * q + 3]};
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int sr = ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + (lane & 31);
        stage[sr] = a0[r];
        if (W == 64) stage[sr + 32] = a1[r];
    }
}

// bn0 = first column of the block; cb: the lane's eight column biases; st: column sums, sums of squares
template <int MODE, int W, bool NTS, bool R16 = false>
__device__ __forceinline__ void epi_process(const KParams& p, const float* stage, int rbase, const EpiRows& e, const EpiRes16& r16, int lane, int bn0,
                                            float* o_base, const f32x4 (&cb)[2], f32x4 (&st_s)[2], f32x4 (&st_q)[2]) {
    using G = EpiGeo<W>;
    const int c8 = (lane & (G::LPR - 1)) * 8, col = bn0 + c8;
    const bool cb_uniform = p.cbias && (p.cbias_bcast || p.HW % 32 == 0);
    const bool geglu = (MODE == 0) && W == 64 && p.act == DS_ACT_GEGLU;
    // The two scale factors as SCALARS read once (round 4).  `v *= p.acc_scale` on an f32x4 made the compiler fetch the factor with a
    // <4 x float> load straddling the neighbouring KParams fields, which kept a 28-byte slice of the kernel argument in PRIVATE memory:
    // every pass of every group then re-read it from scratch (two scratch_load_dwordx4 + s_waitcnt vmcnt(0): a memory round trip per
    // pass, 16 per wave and tile).  Found with the epilogue ablations of profiles/r4_gemm_f16dma_ablate.txt: the epilogue cost 0.24 ms of
    // the plain 320 -> 2 560 projection even with its stores AND its staging writes removed.
    const float acc_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.acc_scale)));
    const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.scale)));
    if (geglu) {
        // GEGLU gate (round 4): every lane takes FOUR value columns and their four gate columns (lane & 7 -> columns 4 (lane & 7) .. + 3 and
        // 32 + the same) instead of lanes 0..3 of a row taking eight values + eight gates while lanes 4..7 idle: the gate is ~25 VALU per
        // element and the epilogue of the 320 -> 2 560 projection (168 M gates per call) was bound by it at half-empty waves.  Same
        // arithmetic per element, 8-byte stores (a row's 64 output bytes per 8 lanes, as before).
        const int c4 = (lane & 7) * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, bg = bv;
        if (p.colbias) { bv = *reinterpret_cast<const f32x4*>(p.colbias + bn0 + c4); bg = *reinterpret_cast<const f32x4*>(p.colbias + bn0 + 32 + c4); }
#pragma unroll
        for (int pass = 0; pass < G::NP; ++pass) {
            const int rr = pass * G::RPP + lane / G::LPR;
            const int row = rbase + rr;
            if (row >= p.M) continue;
            f32x4 val = *reinterpret_cast<const f32x4*>(stage + rr * EPI_LD + c4);
            f32x4 gt = *reinterpret_cast<const f32x4*>(stage + rr * EPI_LD + 32 + c4);
            val = (val * acc_scale + bv) * scale;           // (no row / per-image bias, no residual: the launcher rejects them with GEGLU)
            gt += bg;
#pragma unroll
            for (int q = 0; q < 4; ++q) val[q] *= ds_gelu_gate_fast(gt[q]);
            const size_t ocol = (size_t)row * p.ldo + (bn0 >> 1) + c4;
            if (p.out_f16) {
                typedef _Float16 epi_h4 __attribute__((ext_vector_type(4)));
                const epi_h4 hv = {(_Float16)val[0], (_Float16)val[1], (_Float16)val[2], (_Float16)val[3]};
                epi_h4* op = reinterpret_cast<epi_h4*>(reinterpret_cast<_Float16*>(o_base) + ocol);
                if (NTS) __builtin_nontemporal_store(hv, op); else *op = hv;
            } else {
                f32x4* op = reinterpret_cast<f32x4*>(o_base + ocol);
                if (NTS) __builtin_nontemporal_store(val, op); else *op = val;
            }
        }
        return;
    }
#pragma unroll
    for (int pass = 0; pass < G::NP; ++pass) {
        const int rr = pass * G::RPP + lane / G::LPR;
        const int row = rbase + rr;
        if (row >= p.M) continue;
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v[h] = *reinterpret_cast<const f32x4*>(stage + rr * EPI_LD + c8 + 4 * h);
            if (MODE == 0) v[h] *= acc_scale;
            if (MODE == 1) v[h] *= scale;
            v[h] += cb[h];
            if (p.rowbias) v[h] += p.rowbias[row];
            if (cb_uniform) v[h] += e.cvu[h];
            else if (p.cbias) v[h] += *reinterpret_cast<const f32x4*>(p.cbias + (size_t)(row / p.HW) * p.cbias_ld + col + 4 * h);
            if (R16 || p.res) {
                if (R16 || p.res_f16) {
                    const epi_h8 hr = __builtin_bit_cast(epi_h8, R16 ? r16.r[pass] : e.raw[pass][0]);
                    v[h] += f32x4{(float)hr[4 * h], (float)hr[4 * h + 1], (float)hr[4 * h + 2], (float)hr[4 * h + 3]};
                } else v[h] += __builtin_bit_cast(f32x4, e.raw[pass][h]);
            }
            if (MODE == 0) v[h] *= scale;
        }
        size_t ocol = (size_t)row * p.ldo + col;
        if (p.act == DS_ACT_SILU) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[h][q] = ds_silu(v[h][q]);
        }
        if (p.out_f16) {
            const epi_h8 hv = {(_Float16)v[0][0], (_Float16)v[0][1], (_Float16)v[0][2], (_Float16)v[0][3],
                               (_Float16)v[1][0], (_Float16)v[1][1], (_Float16)v[1][2], (_Float16)v[1][3]};
            epi_h8* op = reinterpret_cast<epi_h8*>(reinterpret_cast<_Float16*>(o_base) + ocol);
            if (NTS) __builtin_nontemporal_store(hv, op); else *op = hv;
            if (p.stats) {
#pragma unroll
                for (int h = 0; h < 2; ++h) v[h] = f32x4{(float)hv[4 * h], (float)hv[4 * h + 1], (float)hv[4 * h + 2], (float)hv[4 * h + 3]};
            }
        } else {
            f32x4* op = reinterpret_cast<f32x4*>(o_base + ocol);
            if (NTS) { __builtin_nontemporal_store(v[0], op); __builtin_nontemporal_store(v[1], op + 1); }
            else { op[0] = v[0]; op[1] = v[1]; }
        }
        if (p.stats) {                                              // (never with GEGLU: the launcher rejects that combination)
#pragma unroll
            for (int h = 0; h < 2; ++h) { st_s[h] += v[h]; st_q[h] += v[h] * v[h]; }
        }
    }
}

template <int W>
__device__ __forceinline__ void epi_stats(const KParams& p, int lane, int wm0, int col, f32x4 (&st_s)[2], f32x4 (&st_q)[2]) {
    if (!p.stats || wm0 >= p.M) return;
    using G = EpiGeo<W>;
    // column sums of this wave's 64 rows: the lane groups (lane / LPR) hold disjoint rows of the same 8 columns
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = G::LPR; o < 64; o <<= 1) { st_s[h][q] += __shfl_xor(st_s[h][q], o); st_q[h][q] += __shfl_xor(st_q[h][q], o); }
        }
    if (lane < G::LPR) {
        float* sp = p.stats + (size_t)(wm0 >> 6) * 2 * p.N + col;
        reinterpret_cast<f32x4*>(sp)[0] = st_s[0]; reinterpret_cast<f32x4*>(sp)[1] = st_s[1];
        reinterpret_cast<f32x4*>(sp + p.N)[0] = st_q[0]; reinterpret_cast<f32x4*>(sp + p.N)[1] = st_q[1];
    }
}

// One column block (W columns at bn0) of the wave tile: its two 32-row groups.  `cur` holds the requested rows of the first group on
// entry; the rows of the second group are requested once the first is staged (its accumulators are dead by then: the two row sets
// never coexist with the whole accumulator tile), and -- if NEXT_W != 0 -- those of the next block's first group once the second is,
// so every request is issued BEFORE the stores of the group in front of it.  On return `cur` holds the next block's first group.
// R16: fp16 residual rows requested up front (r0 / r1 = the block's two groups); the requests here then fetch the bias rows only.
// R16: fp16 residual rows held raw in EpiRes16 (r0 / r1 = this block's two groups, requested by the caller); the rows of the NEXT block's
// groups (n0 / n1) are requested here as soon as a group's accumulators are staged and dead -- two groups ahead of their use, and with
// the first block's rows requested before anything else every residual row of the tile is in flight early, without ever holding more
// than three groups' rows next to the live accumulators.  The EpiRows requests then fetch the per-image bias rows only.
template <int MODE, bool NTS, int W, int NEXT_W, bool R16, bool TR = false>
__device__ __forceinline__ void epi_block(const KParams& p, const f32x16 (&acc)[2][2], float* stage, int lane, int wm0, int bn0, int next_col,
                                          float* o_base, EpiRows& cur, EpiRows& oth, const EpiRes16& r0, const EpiRes16& r1, EpiRes16& n0,
                                          EpiRes16& n1) {
    constexpr int NW_ = NEXT_W ? NEXT_W : 32;
    const int col = bn0 + (lane & (W / 8 - 1)) * 8;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 cb[2] = {zero, zero}, st_s[2] = {zero, zero}, st_q[2] = {zero, zero};
    if (p.colbias) { const f32x4* c = reinterpret_cast<const f32x4*>(p.colbias + col); cb[0] = c[0]; cb[1] = c[1]; }
    epi_stage<W, TR>(acc[0][0], acc[0][1], stage, lane);
    if constexpr (R16 && NEXT_W != 0) epi_request_res16<NW_>(p, wm0, next_col, lane, n0);
    epi_request<W, NTS, R16>(p, wm0 + 32, col, lane, oth);
    epi_process<MODE, W, NTS, R16>(p, stage, wm0, cur, r0, lane, bn0, o_base, cb, st_s, st_q);
    epi_stage<W, TR>(acc[1][0], acc[1][1], stage, lane);
    if constexpr (R16 && NEXT_W != 0) epi_request_res16<NW_>(p, wm0 + 32, next_col, lane, n1);
    if constexpr (NEXT_W != 0) epi_request<NW_, NTS, R16>(p, wm0, next_col, lane, cur);
    epi_process<MODE, W, NTS, R16>(p, stage, wm0 + 32, oth, r1, lane, bn0, o_base, cb, st_s, st_q);
    epi_stats<W>(p, lane, wm0, col, st_s, st_q);
}

// stage: 32 x EPI_LD floats owned by the wave.  The caller guarantees the vector path (p.vec_ok, whole blocks inside N, no split).
template <int MODE, bool NTS, int WA, int WB, bool TR = false>
__device__ __forceinline__ void epilogue_pipe(const KParams& p, const f32x16 (&accA)[2][2], const f32x16 (&accB)[2][2], float* stage, int lane,
                                              int wm0, int wn0, float* o_base) {
    constexpr int WBB = WB ? WB : 32;
    const int colA = wn0 + (lane & (WA / 8 - 1)) * 8;
    const int colB = wn0 + WA + (lane & (WBB / 8 - 1)) * 8;
    EpiRows e0, e1;
    if (p.res && p.res_f16 && !(p.coef_lds & 1024)) {             // the fp16 residual stream: raw rows, requested early (see epi_block); bit 10 of ds_conv_args.tune.ablate = A/B switch
        EpiRes16 a0, a1, b0, b1;
        epi_request_res16<WA>(p, wm0, colA, lane, a0);
        epi_request_res16<WA>(p, wm0 + 32, colA, lane, a1);
        epi_request<WA, NTS, true>(p, wm0, colA, lane, e0);
        epi_block<MODE, NTS, WA, WB, true, TR>(p, accA, stage, lane, wm0, wn0, colB, o_base, e0, e1, a0, a1, b0, b1);
        if constexpr (WB != 0) epi_block<MODE, NTS, WBB, 0, true, TR>(p, accB, stage, lane, wm0, wn0 + WA, 0, o_base, e0, e1, b0, b1, a0, a1);
        return;
    }
    EpiRes16 none;
    epi_request<WA, NTS>(p, wm0, colA, lane, e0);
    epi_block<MODE, NTS, WA, WB, false, TR>(p, accA, stage, lane, wm0, wn0, colB, o_base, e0, e1, none, none, none, none);
    if constexpr (WB != 0) epi_block<MODE, NTS, WBB, 0, false, TR>(p, accB, stage, lane, wm0, wn0 + WA, 0, o_base, e0, e1, none, none, none, none);
}

// XCD-aware decode of a 1-D workgroup id into (m tile, n tile): the dispatcher places workgroup b on XCD b % 8, so
// within each group of 8*NT ids the NT column tiles that share an A slab get the SAME b % 8 (same L2) and are
// dispatched back to back.  With fewer than 8 row tiles (small batch: the weights are the traffic) the roles swap: the
// row tiles that share a WEIGHT tile get the same XCD and the column tiles spread over all 8 XCDs -- otherwise a layer
// with one row tile would run on a single XCD.  Placement only affects speed, never results.
__device__ __host__ __forceinline__ bool tile_order_swapped(int mtiles, int ntiles) { return mtiles < 8 && ntiles > mtiles; }

// `rot` (the split-K index) rotates which tile of a group of 8 lands on which XCD, so that a partly filled last group
// does not load the same XCDs in every split.
__device__ __forceinline__ bool decode_tile(int b, int mtiles, int ntiles, int& mt, int& nt, int rot = 0) {
    if (tile_order_swapped(mtiles, ntiles)) {
        const int per = 8 * mtiles;
        const int g = b / per, r = b - g * per;
        nt = g * 8 + ((r + rot) & 7);
        mt = r >> 3;
        return nt < ntiles;
    }
    const int per = 8 * ntiles;
    const int g = b / per, r = b - g * per;
    mt = g * 8 + ((r + rot) & 7);
    nt = r >> 3;
    return mt < mtiles;
}

inline unsigned grid_1d(int mtiles, int ntiles) {
    if (tile_order_swapped(mtiles, ntiles)) return (unsigned)(((ntiles + 7) / 8) * 8 * mtiles);
    return (unsigned)(((mtiles + 7) / 8) * 8 * ntiles);
}

// Parameters of one split's partial-tile store: plain [M][N] matrix, no fused epilogue terms.
__device__ __forceinline__ KParams split_params(const KParams& p, int split) {
    KParams q = p;
    q.out = p.part + (size_t)split * p.M * p.N; q.ldo = p.N;
    q.colbias = nullptr; q.rowbias = nullptr; q.cbias = nullptr; q.res = nullptr; q.scale = 1.f; q.act = DS_ACT_NONE;
    q.vec_ok = p.vec_part; q.out_planar = 0; q.stats = nullptr;
    return q;
}

// Split-K / tile-shape cost model (microseconds), calibrated on MI355X with tools/sweep_splits.py.
//   A 128x128 workgroup contracts one 32-deep K tile in ~2.1 us when it has a CU to itself and two co-resident ones take
//   ~3.4 us for one tile each (the CU's fp32-MFMA rate); a 256x128 workgroup (8 waves, one per CU) takes 3.3 us.
//   `blocks` equal workgroups spread over 256 CUs => the busiest CU runs n = ceil(blocks / 256) of them back to back.
//   Splitting K by S multiplies the workgroups and divides their length; it costs a reduce launch (~8 us) plus writing and
//   re-reading S partial tiles (~2 TB/s effective).
inline double cu_time_us(long long n, bool big_tile) {
    return big_tile ? 3.3 * (double)n : 3.4 * (double)(n / 2) + 2.1 * (double)(n & 1);
}
inline double layer_cost_us(long long blocks, bool big_tile, int ktiles, int s, long long mn) {
    const long long n = (blocks * s + 255) / 256;
    double t = (double)((ktiles + s - 1) / s) * cu_time_us(n, big_tile);
    if (s > 1) t += 8.0 + (double)s * (double)mn * 8.0 / 2.0e6;
    return t;
}
// Best split count for a layer of `blocks` tiles whose K loop has `units` splittable units of `tiles_per_unit` K tiles.
inline int choose_splits(long long blocks, bool big_tile, int units, int tiles_per_unit, long long part_capacity_floats,
                         long long mn, double* cost_out = nullptr, int forced = 0) {     // forced > 0: ds_conv_args.tune.splits
    const int ktiles = units * tiles_per_unit;
    double best_c = layer_cost_us(blocks, big_tile, ktiles, 1, mn);
    int best = 1;
    // down to ONE unit per split (round 4; until then units / 2): the layers this matters for are the 8x8 / 16x16 layers of small batches
    // (CIFAR-10 at 8 images: 8 slabs -> 8 splits of one slab = 9 K tiles each instead of 4 x 18), where every serial K tile is ~2 us
    long long smax = units;
    if (smax > 16) smax = 16;
    if (mn > 0 && smax * mn > part_capacity_floats) smax = part_capacity_floats / mn;
    if (units >= 4 && mn > 0) {
        if (forced > 0) {
            best = (int)(forced < smax ? forced : smax);
            if (best < 1) best = 1;
            best_c = layer_cost_us(blocks, big_tile, ((units + best - 1) / best) * tiles_per_unit * best, best, mn);
        } else {
            for (long long s = 2; s <= smax; ++s) {
                const double c = layer_cost_us(blocks, big_tile, ((units + s - 1) / s) * tiles_per_unit * (int)s, (int)s, mn);
                if (c < 0.97 * best_c) { best_c = c; best = (int)s; }
            }
        }
    }
    if (cost_out) *cost_out = best_c;
    return best;
}
int launch_splitk_reduce(const KParams& p, hipStream_t stream);   // gemm_conv.hip (also fills p.stats when requested)
int launch_splitk_reduce_f16(const KParams& p, hipStream_t stream);   // the same for the fp16-activation convolution (fp16 / fp32 residual and output rows)

// gemm_dma8.hip
bool gemm_dma8_applicable(const KParams& p);
int launch_gemm_dma8(KParams& p, hipStream_t stream);

// conv3x3_halo.hip
bool conv3x3_halo_supported(const KParams& p);
int launch_conv3x3_halo(KParams& p, hipStream_t stream);
int conv3x3_halo_choice(const KParams& p);   // 0 / 128 / 256: which halo tile shape the launcher picks

// conv3x3_halo2.hip: second-generation 256 x 128 tile (static tap schedule, double halo buffer)
bool conv3x3_halo2_applicable(const KParams& p, int wide, int mode);   // mode 0 fp32 / 1 fp16 / 2 split-fp16
int launch_conv3x3_halo2(KParams& p, int wide, int mode, hipStream_t stream);

// conv3x3_f16dma.hip: 3x3 on fp16 activations, both operands by LDS-DMA, 256-pixel x 64/128/192/256-channel tiles
// conv3x3_thin.hip: 3x3 layers with at most four output channels (the network heads)
bool conv3x3_thin_applicable(const KParams& p);
int launch_conv3x3_thin(KParams& p, hipStream_t stream);
bool conv3x3_f16dma_applicable(const KParams& p);
int launch_conv3x3_f16dma(KParams& p, hipStream_t stream);

bool conv3x3_f16dma_use_half(const KParams& p);   // true: the layer runs on conv3x3_f16dmah_kernel (four waves, two workgroups per CU)
// conv3x3_f16dmah.hip: the same convolution on 128-pixel tiles with 32-channel half slabs, four waves, two workgroups per CU
bool conv3x3_f16dmah_applicable(const KParams& p);
int conv3x3_f16dmah_max_nb(int W);
int launch_conv3x3_f16dmah_tiles(const KParams& p, int nb, int n_begin, int ntiles, hipStream_t stream);

// gemm_f16dma.hip: 1x1 / Linear on fp16 activations (both operands by LDS-DMA)
bool gemm_f16dma_applicable(const KParams& p);
bool gemm_f16dma_gather_applicable(const KParams& p);                 // 3x3 stride-2 convolution on fp16 rows (the gather form of the same kernel)
int launch_gemm_f16dma(KParams& p, hipStream_t stream, bool gather = false);

// gemm_f16.hip: 1x1 / Linear with fp16 operands (A rounded while staged, W packed fp16)
bool gemm_f16_applicable(const KParams& p);
int launch_gemm_f16(KParams& p, hipStream_t stream);

}  // namespace igemm
