// 1x1 convolution / Linear on fp16 ACTIVATIONS: the matrix kernel of conv3x3_f16dma.hip without the halo (gfx950,
// v_mfma_f32_32x32x16_f16).  A tile = 256 rows x 64 channels (32 KB) and W tile = NB * 64 rows x 64 channels of one K tile, both staged
// by LDS-DMA into swizzled pixel-major images, double buffered two K tiles deep; eight waves of 64 x (NB * 32), NB = 1 .. 4; one barrier
// per K tile; the explicit fragment pipeline and the fused epilogues (bias, residual, scale, SiLU, GEGLU gate, GroupNorm column sums,
// optional fp16 output rows) are those of the convolution.
// Why: gemm_f16_kernel reads fp32 activations, converts them in registers and writes fp32 -- on SD-1.5 its K = 320 layers are bound by
// exactly those bytes (0.135 of the fp16 matrix peak, 40 % of the sampler's time).  In the reference's fp16 / autocast mode the operands
// of these layers ARE fp16 tensors (networks_edm.py:486; torch.autocast casts nn.Linear inputs): LayerNorm, the GroupNorm pass, the
// attention output and the GEGLU gate write fp16 rows, and this kernel streams them.
// Scope: taps == 1, ONE fp16 source [M][K] (K % 64 == 0, ld % 8 == 0), cout % 64 == 0 (GEGLU: NB in {2, 4}), any M (tail rows read a
// zero page and are masked by the epilogue).
// GATHER (round 4): the same kernel as the latent-diffusion `Downsample` -- a 3x3, stride-2, pad-1 convolution (openaimodel.py:146-148) on the
// fp16 residual stream.  The A tile of K tile kt = (64-channel slab, tap) is then a GATHER of 128-byte row pieces: output pixel (img, oy, ox)
// reads input pixel (img, 2 oy + ty - 1, 2 ox + tx - 1), channels [64 slab, 64 slab + 64), or the zero page outside the image -- the LDS-DMA
// takes a per-lane global address, so the gather costs a handful of VALU per K tile and nothing else; weights are the [slab][tap][64] packing of
// the stride-1 kernel (ops.pack_conv_weight_f16), so the weight stream is unchanged.  Replaces widen-to-fp32 + igemm_f32_kernel<0> in fp16 mode.
#include "pipe_common.h"
#include "epi_direct.h"

namespace igemm {
namespace {

__device__ __attribute__((aligned(128))) _Float16 g_zero_halfs_g[64];

// NW = 8 waves: 256-row tiles, one workgroup per CU.  NW = 4 waves (2 x 2): 128-row tiles whose LDS footprint (NB <= 3) lets TWO workgroups
// share a CU -- for the short-K, output-heavy projections of the transformer blocks (K = 320 ... 1 280: five to twenty K tiles, then an
// epilogue that moves as many bytes as the whole main loop) the epilogue of one workgroup then runs under the loads and MFMAs of the other.
template <int NB, int NW>
constexpr unsigned gemm16_smem() { return 2u * (NW * 4096u + NB * 8192u); }

template <int NB, int NW, bool DIRECT, bool GATHER = false>
__global__ void __launch_bounds__(NW * 64, 2) gemm_f16dma_kernel(const KParams p) {
    constexpr unsigned AB = NW * 4096u, WB = NB * 8192u;       // bytes of one A / W stage
    constexpr int BM = NW * 32, NT = NW * 64;                  // rows per tile, threads
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);                 // [A 0 | A 1 | W 0 | W 1]
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt, 0)) return;
    const int m0 = mt * BM, n0 = p.n_begin + nt * (NB * 64);
    const _Float16* a0 = reinterpret_cast<const _Float16*>(p.a0);
    const _Float16* wgt = reinterpret_cast<const _Float16*>(p.b);
    const size_t ldbh = (size_t)p.ldb * 2;
    const int KT = p.K / 64;

    // DMA: thread tid owns 16-B unit j * NT + tid of round j: row j * (NT / 8) + (tid >> 3), LDS chunk slot tid & 7 = source chunk ^ ((row >> 1) & 7)
    const int sw = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
    const _Float16* asrc[4];
    int pixb[4];                         // GATHER: input pixel index of tap (0, 0) of the row's output pixel (may lie outside the image)
    unsigned vmask[4];                   // GATHER: bit t = tap t of this row reads inside the image (0 for rows >= M)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = m0 + j * (NT / 8) + (tid >> 3);
        if constexpr (GATHER) {
            asrc[j] = nullptr; pixb[j] = 0; vmask[j] = 0u;
            if (row < p.M) {
                const int img = row / p.HW, rem = row - img * p.HW;
                const int oy = rem / p.W, ox = rem - oy * p.W;
                const int iy0 = 2 * oy - 1, ix0 = 2 * ox - 1;
                pixb[j] = (img * p.IH + iy0) * p.IW + ix0;
                unsigned m = 0u;
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    if ((unsigned)(iy0 + t / 3) < (unsigned)p.IH && (unsigned)(ix0 + t % 3) < (unsigned)p.IW) m |= 1u << t;
                vmask[j] = m;
            }
        } else {
            asrc[j] = row < p.M ? a0 + (size_t)row * p.lda0 + sw : nullptr;
        }
    }
    const _Float16* wsrc = wgt + (size_t)(n0 + (tid >> 3)) * ldbh + sw;
    const int abl = p.coef_lds & 0x7fff; // timing ablations (ds_conv_args.tune.ablate; results are WRONG when bits 0 / 2 are set): bit 0 = no DMA after
                                         // the prologue, bit 2 = no epilogue
    auto dma = [&](int kt, int buf) {
        if ((abl & 1) && kt > 1) return;
        int slab = 0, tap = 0, toff = 0;                       // GATHER: K tile kt = (slab, tap), tap-minor like the weight packing
        if constexpr (GATHER) {
            slab = kt / 9; tap = kt - slab * 9;
            const int ty = tap / 3;
            toff = ty * p.IW + (tap - ty * 3);
        }
        DS_RACE_SKEW(wave);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16* g;
            if constexpr (GATHER)
                g = ((vmask[j] >> tap) & 1u) ? a0 + (size_t)(pixb[j] + toff) * p.lda0 + (slab * 64 + sw) : g_zero_halfs_g;
            else
                g = asrc[j] ? asrc[j] + (size_t)kt * 64 : g_zero_halfs_g;
            __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + buf * AB + (j * NT + wave * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB * 512 / NT; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (size_t)i * (NT / 8) * ldbh + (size_t)kt * 64),
                                             (lptr_t)(lds + 2 * AB + buf * WB + (i * (NT / 8) + wave * 8) * 128), 16, 0, 0);
    };

    const unsigned lds0 = lds_addr2(smem);
    const unsigned gsel = (unsigned)(lane >> 5);
    unsigned abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const unsigned r = (unsigned)(wr * 64 + i * 32 + (lane & 31));
        abase[i] = lds0 + r * 128u + 16u * (((r >> 1) & 7u) ^ gsel);
    }
    const unsigned brow = (unsigned)(wc * (NB * 32) + (lane & 31));
    const unsigned bbase = lds0 + 2 * AB + brow * 128u + 16u * (((brow >> 1) & 7u) ^ gsel);

    f32x16 accA[2][2], accB[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accA[i][j][r] = 0.f; accB[i][j][r] = 0.f; }

    struct Frag { f32x4 a0, a1, b0, b1, b2, b3; };
    auto frag_read = [&](Frag& f, unsigned va0, unsigned va1, unsigned vb) {
        f.a0 = lds_rd<0>(va0);
        f.a1 = lds_rd<0>(va1);
        f.b0 = lds_rd<0>(vb);
        if constexpr (NB > 1) f.b1 = lds_rd<4096>(vb);
        if constexpr (NB > 2) f.b2 = lds_rd<8192>(vb);
        if constexpr (NB > 3) f.b3 = lds_rd<12288>(vb);
    };
    auto frag_wait = [&](Frag& f, auto nc) {
        constexpr int N = decltype(nc)::value;
        if constexpr (NB == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0) : "n"(N));
        if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1) : "n"(N));
        if constexpr (NB == 3) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2) : "n"(N));
        if constexpr (NB == 4)
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2), "+v"(f.b3) : "n"(N));
    };
    // SWAPPED product (round 4): the weight fragment is the MFMA's first operand, so an accumulator block holds lane = row (pixel), registers =
    // columns (channels) -- what epilogue_direct (igemm_common.h) stores without an LDS transpose
#define DSG16_MM(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, b_), __builtin_bit_cast(h8, a_), acc_, 0, 0, 0)
    auto mfma_group = [&](Frag& f) {
        DSG16_MM(accA[0][0], f.a0, f.b0); DSG16_MM(accA[1][0], f.a1, f.b0);
        if constexpr (NB > 1) { DSG16_MM(accA[0][1], f.a0, f.b1); DSG16_MM(accA[1][1], f.a1, f.b1); }
        if constexpr (NB > 2) { DSG16_MM(accB[0][0], f.a0, f.b2); DSG16_MM(accB[1][0], f.a1, f.b2); }
        if constexpr (NB > 3) { DSG16_MM(accB[0][1], f.a0, f.b3); DSG16_MM(accB[1][1], f.a1, f.b3); }
    };
    constexpr int NR = 2 + NB;

    DS_TL(p.part, p.coef_lds, 0, blockIdx.x);                  // (phase stamps: the 'timeline' diagnostics build only, csrc/ds_common.h)
    dma(0, 0);
    if (KT > 1) dma(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DS_TL(p.part, p.coef_lds, 1, blockIdx.x);
    Frag P, Q;
    frag_read(P, abase[0], abase[1], bbase);
    for (int kt = 0; kt < KT; ++kt) {
        const unsigned ao = (unsigned)(kt & 1) * AB, wo = (unsigned)(kt & 1) * WB;
        const unsigned a_0 = abase[0] + ao, a_1 = abase[1] + ao, vb = bbase + wo;
        frag_read(Q, a_0 ^ 32u, a_1 ^ 32u, vb ^ 32u);
        frag_wait(P, IC<NR>{});
        DS2_FENCE(); mfma_group(P); DS2_FENCE();
        frag_read(P, a_0 ^ 64u, a_1 ^ 64u, vb ^ 64u);
        frag_wait(Q, IC<NR>{});
        DS2_FENCE(); mfma_group(Q); DS2_FENCE();
        frag_read(Q, a_0 ^ 96u, a_1 ^ 96u, vb ^ 96u);
        frag_wait(P, IC<NR>{});
        DS2_FENCE(); mfma_group(P); DS2_FENCE();
        frag_wait(Q, IC<0>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // own DMAs of K tile kt + 1 landed
        __builtin_amdgcn_s_barrier();                        // stage kt & 1 is free, K tile kt + 1 is in LDS
        DS2_FENCE();
        if (kt + 1 < KT) {
            const unsigned no = (unsigned)((kt + 1) & 1);
            frag_read(P, abase[0] + no * AB, abase[1] + no * AB, bbase + no * WB);
        }
        DS2_FENCE(); mfma_group(Q); DS2_FENCE();
        if (kt + 2 < KT) dma(kt + 2, kt & 1);                // in the shadow of the last K step's MFMAs
        DS2_FENCE();
    }
#undef DSG16_MM
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    DS_TL(p.part, p.coef_lds, 2, blockIdx.x);

    if (abl & 4) {                       // no epilogue: every accumulator block (and so every MFMA) is kept alive
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { asm volatile("" :: "v"(accA[i][j])); asm volatile("" :: "v"(accB[i][j])); }
        return;
    }
    float* stage = smem + wave * 32 * EPI_LD;
    const int wn0 = n0 + wc * (NB * 32);
    if constexpr (DIRECT) epilogue_direct<false, NB, GATHER>(p, accA, accB, lane, m0 + wr * 64, wn0);      // GATHER layers carry GroupNorm column sums
    else epilogue_pipe<0, false, (NB == 1 ? 32 : 64), (NB == 3 ? 32 : (NB == 4 ? 64 : 0)), true>(p, accA, accB, stage, lane, m0 + wr * 64, wn0, p.out);
#ifdef DS_TIMELINE
    DS_TL(p.part, p.coef_lds, 3, blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DS_TL(p.part, p.coef_lds, 4, blockIdx.x);
#endif
}

template <int NB, int NW, bool GATHER = false>
int launch_nb(KParams p, int n_begin, int ntiles, hipStream_t stream) {
    p.mtiles = (p.M + NW * 32 - 1) / (NW * 32);
    p.ntiles = ntiles;
    p.n_begin = n_begin;
    p.splits = 1;
    p.coef_lds = p.t_ablate;
    int smem = (int)gemm16_smem<NB, NW>();
    const int epi = NW * 32 * EPI_LD * (int)sizeof(float);
    if (smem < epi) smem = epi;
    if (epi_direct_ok(p, GATHER, NB)) {
        DS_ENSURE_DYN_LDS((&gemm_f16dma_kernel<NB, NW, true, GATHER>), 160 * 1024);
        hipLaunchKernelGGL((gemm_f16dma_kernel<NB, NW, true, GATHER>), dim3(grid_1d(p.mtiles, p.ntiles), 1), dim3(NW * 64), smem, stream, p);
    } else {
        DS_ENSURE_DYN_LDS((&gemm_f16dma_kernel<NB, NW, false, GATHER>), 160 * 1024);
        hipLaunchKernelGGL((gemm_f16dma_kernel<NB, NW, false, GATHER>), dim3(grid_1d(p.mtiles, p.ntiles), 1), dim3(NW * 64), smem, stream, p);
    }
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace

bool gemm_f16dma_applicable(const KParams& p) {
    if (p.taps != 1 || p.stride > 1 || p.norm != nullptr || p.ec0 || p.ec1 || p.c1) return false;
    if (p.K < 64 || p.K % 64 || p.N % 64 || !p.vec_ok || p.nrows_b < p.N || p.M < 1) return false;
    return true;
}

// The strided 3x3 convolution on fp16 rows (GATHER): H, W = OUTPUT size, IH = 2 H, IW = 2 W; bias / per-image bias / column sums / fp16 or fp32
// output rows through the family's epilogues, no residual, no extra 1x1 sources.
bool gemm_f16dma_gather_applicable(const KParams& p) {
    if (p.taps != 9 || p.stride != 2 || p.norm != nullptr || p.ec0 || p.ec1 || p.c1 || p.res || p.act == DS_ACT_GEGLU) return false;
    if (p.c0 < 64 || p.c0 % 64 || p.K != 9 * p.c0 || p.N % 64 || !p.vec_ok || p.nrows_b < p.N || p.M < 1) return false;
    if (p.IH != 2 * p.H || p.IW != 2 * p.W || p.HW != p.H * p.W || p.M % p.HW) return false;
    if ((long long)p.M * 4 > 0x7fffffffLL) return false;
    return true;
}

// Column tiling as in conv3x3_f16dma.hip (cost 1 + nb per round of resident workgroups); the GEGLU epilogue pairs 32 value columns with
// their 32 gate columns inside a 64-column half of a wave tile, so it takes even widths only.  Projections with K <= 2 560 and N <= 1 280
// take the four-wave, 128-row variant (two workgroups per CU, NB <= 3): 5 - 10 % faster there, slower on wide outputs (A/B per shape in
// profiles/r3_gemm_f16dma_epilogue.txt); ds_conv_args.tune.f16dma_nw (benchmarks) forces 4 or 8.
int launch_gemm_f16dma(KParams& p, hipStream_t stream, bool gather) {
    const bool geglu = p.act == DS_ACT_GEGLU;
    if (geglu && (p.N % 128)) return DS_E_SHAPE;
    int nw = (p.K <= 2560 && p.N <= 1280) ? 4 : 8;
    if (p.t_nw == 4 || p.t_nw == 8) nw = p.t_nw;
    if (gather) nw = 8;                                        // the gather is instantiated for the eight-wave tile only
    const int mtiles = (p.M + nw * 32 - 1) / (nw * 32), slots = nw == 4 ? 512 : 256, max_nb = nw == 4 ? 3 : 4;
    auto tiling = [&](int nb0, int (*out)[3], int* cost) {
        int n = 0, col = 0, c = 0;
        for (int w = nb0; w >= 1 && col < p.N; --w) {
            if (geglu && (w & 1)) continue;
            const int t = (p.N - col) / (64 * w);
            if (t > 0) { out[n][0] = col; out[n][1] = t; out[n][2] = w; ++n; col += t * 64 * w; c += (int)(((long long)mtiles * t + slots - 1) / slots) * (1 + w); }
        }
        *cost = c;
        return n;
    };
    int best_nb = max_nb, best_cost = 0x7fffffff, best_n = 99, cost, tmp[4][3];
    for (int nb = max_nb; nb >= 1; --nb) {
        if (geglu && (nb & 1)) continue;
        const int n = tiling(nb, tmp, &cost);
        if (cost < best_cost || (cost == best_cost && n < best_n)) { best_cost = cost; best_n = n; best_nb = nb; }
    }
    if (p.t_nb > 0 && !(geglu && (p.t_nb & 1))) best_nb = p.t_nb < max_nb ? p.t_nb : max_nb;
    int plan[4][3];
    const int n = tiling(best_nb, plan, &cost);
    for (int i = 0; i < n; ++i) {
        int rc;
        switch (plan[i][2] + (nw == 4 ? 10 : 0) + (gather ? 20 : 0)) {
            case 21: rc = launch_nb<1, 8, true>(p, plan[i][0], plan[i][1], stream); break;
            case 22: rc = launch_nb<2, 8, true>(p, plan[i][0], plan[i][1], stream); break;
            case 23: rc = launch_nb<3, 8, true>(p, plan[i][0], plan[i][1], stream); break;
            case 24: rc = launch_nb<4, 8, true>(p, plan[i][0], plan[i][1], stream); break;
            case 1: rc = launch_nb<1, 8>(p, plan[i][0], plan[i][1], stream); break;
            case 2: rc = launch_nb<2, 8>(p, plan[i][0], plan[i][1], stream); break;
            case 3: rc = launch_nb<3, 8>(p, plan[i][0], plan[i][1], stream); break;
            case 4: rc = launch_nb<4, 8>(p, plan[i][0], plan[i][1], stream); break;
            case 11: rc = launch_nb<1, 4>(p, plan[i][0], plan[i][1], stream); break;
            case 12: rc = launch_nb<2, 4>(p, plan[i][0], plan[i][1], stream); break;
            default: rc = launch_nb<3, 4>(p, plan[i][0], plan[i][1], stream); break;
        }
        if (rc) return rc;
    }
    return DS_OK;
}

}  // namespace igemm
