// Epilogue of the fp16-activation kernels (conv3x3_f16dma.hip, gemm_f16dma.hip) that stores straight from the accumulators.
#pragma once
#include "pipe_common.h"

namespace igemm {
namespace {

// ---- The epilogue WITHOUT the LDS transpose (round 4; conv3x3_f16dma / gemm_f16dma) --------------------------------------------------
// Those kernels multiply SWAPPED (weights as the MFMA's first operand, pixels as its second), so a 32 x 32 accumulator block has
//   lane = pixel (lane & 31) of the 32-row block, register r = channel 8 (r >> 2) + 4 h + (r & 3), h = lane >> 5
// -- four consecutive channels of ONE output row per register quad instead of four rows of one channel.  v_permlane32_swap (gfx950) trades
// the odd quads of the lower half-wave for the even quads of the upper one: afterwards registers 8 u + k (u = 0, 1; k = 0 .. 7) hold channels
// 16 u + 8 h + k, eight consecutive fp16 outputs = one 16-byte store per (lane, u); a store instruction covers 32 rows x 32 B.  No staging
// writes, no LDS reads, no waits on either (the staged epilogue spends 40 LDS instructions and two dependent waits per 32 x 64 group), and
// bias / residual / activation / rounding are the SAME operations in the same order on the same values: the output rows are bit-identical
// to epilogue_pipe's.  Column sums for the consumer's GroupNorm: per register over the two row blocks, then across the 32 pixel lanes
// (v_permlane16_swap folds registers e and e + 8 while adding the two 16-lane rows, four DPP adds finish the row) -- the order of the fp32
// additions differs from the staged epilogue's, the values summed (the stored fp16 numbers) do not.
// DIRECT is a template parameter of the kernels (one instantiation per epilogue: a kernel that holds both allocates registers for the
// worse of the two), chosen by the launcher.  Taken when the output rows are fp16, the residual (if any) is the fp16 stream, there is no row bias, the per-image bias is uniform over
// a 32-row block, and not (per-image bias AND residual) -- i.e. every layer of the fp16 engines; anything else goes through
// epilogue_pipe<..., TR = true>.  ds_conv_args.tune.ablate bit 12 forces the staged epilogue (A/B switch for benchmarks and tests).
// NO non-temporal hint on these accesses (NTS = false at both call sites, plain residual loads): a wave instruction here covers 32 rows x
// 32 B, and with the hint such pieces store at 1.5 TB/s and load at 2.5 TB/s against 5.7 / 5.1 TB/s without it (8 rows x 128 B, the
// staged epilogue's pattern: 5.4 / 7.1 with the hint) -- tools/probes/hbm_write_rate.hip, profiles/r4_probe_hbm_write_rate.txt.
// (host side: the launcher picks the kernel instantiation with it; p.t_ablate = ds_conv_args.tune.ablate)
inline bool epi_direct_ok(const KParams& p, bool stats_ok, int nb) {
    if (p.t_ablate & 4096) return false;
    if (!p.out_f16 || p.rowbias) return false;
    if (p.res && (!p.res_f16 || p.cbias)) return false;
    if (p.cbias && !(p.cbias_bcast || p.HW % 32 == 0)) return false;
    if (p.stats && !stats_ok) return false;
    if (p.act == DS_ACT_GEGLU && ((nb & 1) || p.res || p.cbias || p.stats)) return false;
    return true;
}

__device__ __forceinline__ void epi_pair_channels(f32x16& a) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float lo = a[8 * u + t], hi = a[8 * u + 4 + t];          // (named temporaries: __builtin_bit_cast applied to a vector ELEMENT reads element 0)
            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi), false, false);
            const unsigned r0 = r[0], r1 = r[1];
            a[8 * u + t] = __builtin_bit_cast(float, r0);
            a[8 * u + 4 + t] = __builtin_bit_cast(float, r1);
        }
}

template <int CTRL>
__device__ __forceinline__ float epi_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// s[e], e = 0 .. 15: per-lane partial sums of channels 16 (e >> 3) + 8 h + (e & 7) -> w[e], e = 0 .. 7: the sum over the 32 pixel lanes of
// channel 16 x + 8 h + e, x = (lane >> 4) & 1, in every lane of the 16-lane row
__device__ __forceinline__ void epi_fold_lanes(const float (&s)[16], float (&w)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, s[e]), __builtin_bit_cast(unsigned, s[e + 8]), false, false);
        const unsigned r0 = r[0], r1 = r[1];
        float v = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
        v = epi_dpp_add<0xB1>(v);            // quad_perm [1, 0, 3, 2]
        v = epi_dpp_add<0x4E>(v);            // quad_perm [2, 3, 0, 1]
        v = epi_dpp_add<0x124>(v);           // row_ror 4
        v = epi_dpp_add<0x128>(v);           // row_ror 8
        w[e] = v;
    }
}

template <bool NTS, int NB, bool CB, bool RES, bool ST>
__device__ __forceinline__ void epi_direct_run(const KParams& p, f32x16 (&accA)[2][2], f32x16 (&accB)[2][2], int lane, int wm0, int wn0) {
    const int h = lane >> 5, px = lane & 31;
    const float acc_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.acc_scale)));
    const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.scale)));
    const bool silu = p.act == DS_ACT_SILU;
    const bool st = ST && p.stats != nullptr && wm0 < p.M;
    _Float16* out = reinterpret_cast<_Float16*>(p.out);
    const int ch0 = wn0 + 8 * h;                                  // + 32 jb + 16 u + k
    int row[2]; bool ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { row[i] = wm0 + 32 * i + px; ok[i] = row[i] < p.M; }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // residual rows (fp16 stream): the first two column blocks now, block jb + 2 once block jb is stored
    epi_u4 rr[NB][2][2];
    auto request_res = [&](auto jbc) {
        constexpr int jb = decltype(jbc)::value;
        if constexpr (RES && jb < NB) {
            const _Float16* res = reinterpret_cast<const _Float16*>(p.res);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    rr[jb][i][u] = *reinterpret_cast<const epi_u4*>(res + (size_t)min(row[i], p.M - 1) * p.res_ld + ch0 + 32 * jb + 16 * u);
        }
    };
    request_res(IC<0>{}); request_res(IC<1>{});
    // per-image bias rows (uniform over a 32-row block)
    const float* cvp[2] = {nullptr, nullptr};
    if constexpr (CB) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int img = p.cbias_bcast ? 0 : __builtin_amdgcn_readfirstlane(min(wm0 + 32 * i, p.M - 1)) / p.HW;
            cvp[i] = p.cbias + (size_t)img * p.cbias_ld + ch0;
        }
    }
    f32x4 cb[4] = {zero, zero, zero, zero}, cbn[4] = {zero, zero, zero, zero};
    auto load_cb = [&](int jb, f32x4 (&c)[4]) {
        if (p.colbias) { const f32x4* g = reinterpret_cast<const f32x4*>(p.colbias + ch0 + 32 * jb); c[0] = g[0]; c[1] = g[1]; c[2] = g[4]; c[3] = g[5]; }
    };
    load_cb(0, cb);
    static_for<NB>([&](auto jbc) {
        constexpr int jb = decltype(jbc)::value;
        if constexpr (jb + 1 < NB) load_cb(jb + 1, cbn);
        f32x4 cv[2][4];
        if constexpr (CB) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4* g = reinterpret_cast<const f32x4*>(cvp[i] + 32 * jb);
                cv[i][0] = g[0]; cv[i][1] = g[1]; cv[i][2] = g[4]; cv[i][3] = g[5];
            }
        }
        float ss[16], sq[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { ss[e] = 0.f; sq[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16& a = jb < 2 ? accA[i][jb & 1] : accB[i][jb & 1];
            epi_pair_channels(a);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 v[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    v[hh] = f32x4{a[8 * u + 4 * hh], a[8 * u + 4 * hh + 1], a[8 * u + 4 * hh + 2], a[8 * u + 4 * hh + 3]};
                    v[hh] *= acc_scale;
                    v[hh] += cb[2 * u + hh];
                    if constexpr (CB) v[hh] += cv[i][2 * u + hh];
                    if constexpr (RES) {
                        const epi_h8 hr = __builtin_bit_cast(epi_h8, rr[jb][i][u]);
                        v[hh] += f32x4{(float)hr[4 * hh], (float)hr[4 * hh + 1], (float)hr[4 * hh + 2], (float)hr[4 * hh + 3]};
                    }
                    v[hh] *= scale;
                }
                if (silu) {
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[hh][q] = ds_silu(v[hh][q]);
                }
                const epi_h8 hv = {(_Float16)v[0][0], (_Float16)v[0][1], (_Float16)v[0][2], (_Float16)v[0][3],
                                   (_Float16)v[1][0], (_Float16)v[1][1], (_Float16)v[1][2], (_Float16)v[1][3]};
                if (ok[i]) {
                    epi_h8* op = reinterpret_cast<epi_h8*>(out + (size_t)row[i] * p.ldo + ch0 + 32 * jb + 16 * u);
                    if (NTS) __builtin_nontemporal_store(hv, op); else *op = hv;
                }
                if constexpr (ST) {
                    if (st) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float f = ok[i] ? (float)hv[k] : 0.f;
                            ss[8 * u + k] += f; sq[8 * u + k] += f * f;
                        }
                    }
                }
            }
        }
        request_res(IC<jb + 2>{});
        if constexpr (ST) {
            if (st) {
                float ws[8], wq[8];
                epi_fold_lanes(ss, ws);
                epi_fold_lanes(sq, wq);
                if ((lane & 15) == 0) {
                    float* sp = p.stats + (size_t)(wm0 >> 6) * 2 * p.N + ch0 + 32 * jb + 16 * ((lane >> 4) & 1);
                    reinterpret_cast<f32x4*>(sp)[0] = f32x4{ws[0], ws[1], ws[2], ws[3]};
                    reinterpret_cast<f32x4*>(sp)[1] = f32x4{ws[4], ws[5], ws[6], ws[7]};
                    reinterpret_cast<f32x4*>(sp + p.N)[0] = f32x4{wq[0], wq[1], wq[2], wq[3]};
                    reinterpret_cast<f32x4*>(sp + p.N)[1] = f32x4{wq[4], wq[5], wq[6], wq[7]};
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cb[c] = cbn[c];
    });
}

// GEGLU: value columns [0, 32) and gate columns [32, 64) of a 64-column pair of blocks meet in the same lane and register
template <bool NTS, int NB>
__device__ __forceinline__ void epi_direct_geglu(const KParams& p, f32x16 (&accA)[2][2], f32x16 (&accB)[2][2], int lane, int wm0, int wn0) {
    const int h = lane >> 5, px = lane & 31;
    const float acc_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.acc_scale)));
    const float scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.scale)));
    _Float16* out = reinterpret_cast<_Float16*>(p.out);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    static_for<NB / 2>([&](auto jpc) {
        constexpr int jp = decltype(jpc)::value;
        const int c0 = wn0 + 64 * jp + 8 * h;
        f32x4 bv[4] = {zero, zero, zero, zero}, bg[4] = {zero, zero, zero, zero};
        if (p.colbias) {
            const f32x4* g = reinterpret_cast<const f32x4*>(p.colbias + c0);
            bv[0] = g[0]; bv[1] = g[1]; bv[2] = g[4]; bv[3] = g[5];
            bg[0] = g[8]; bg[1] = g[9]; bg[2] = g[12]; bg[3] = g[13];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16& av = jp == 0 ? accA[i][0] : accB[i][0];
            f32x16& ag = jp == 0 ? accA[i][1] : accB[i][1];
            epi_pair_channels(av);
            epi_pair_channels(ag);
            const int row = wm0 + 32 * i + px;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 v[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    f32x4 val = f32x4{av[8 * u + 4 * hh], av[8 * u + 4 * hh + 1], av[8 * u + 4 * hh + 2], av[8 * u + 4 * hh + 3]};
                    f32x4 gt = f32x4{ag[8 * u + 4 * hh], ag[8 * u + 4 * hh + 1], ag[8 * u + 4 * hh + 2], ag[8 * u + 4 * hh + 3]};
                    val = (val * acc_scale + bv[2 * u + hh]) * scale;
                    gt += bg[2 * u + hh];
#pragma unroll
                    for (int q = 0; q < 4; ++q) val[q] *= ds_gelu_gate_fast(gt[q]);
                    v[hh] = val;
                }
                const epi_h8 hv = {(_Float16)v[0][0], (_Float16)v[0][1], (_Float16)v[0][2], (_Float16)v[0][3],
                                   (_Float16)v[1][0], (_Float16)v[1][1], (_Float16)v[1][2], (_Float16)v[1][3]};
                if (row < p.M) {
                    epi_h8* op = reinterpret_cast<epi_h8*>(out + (size_t)row * p.ldo + ((wn0 + 64 * jp) >> 1) + 8 * h + 16 * u);
                    if (NTS) __builtin_nontemporal_store(hv, op); else *op = hv;
                }
            }
        }
    });
}

// ST: the kernel's layers may carry GroupNorm column sums (the convolution); false: such a launch takes the staged epilogue
template <bool NTS, int NB, bool ST>
__device__ __forceinline__ void epilogue_direct(const KParams& p, f32x16 (&accA)[2][2], f32x16 (&accB)[2][2], int lane, int wm0, int wn0) {
    if (p.act == DS_ACT_GEGLU) {
        if constexpr (NB % 2 == 0) epi_direct_geglu<NTS, NB>(p, accA, accB, lane, wm0, wn0);
        return;
    }
    if (p.res) epi_direct_run<NTS, NB, false, true, ST>(p, accA, accB, lane, wm0, wn0);
    else if (p.cbias) epi_direct_run<NTS, NB, true, false, ST>(p, accA, accB, lane, wm0, wn0);
    else epi_direct_run<NTS, NB, false, false, ST>(p, accA, accB, lane, wm0, wn0);
}

}  // namespace
}  // namespace igemm
