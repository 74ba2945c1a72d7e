// 3x3 convolution on fp16 ACTIVATIONS, third generation of the LDS-halo implicit GEMM (gfx950, v_mfma_f32_32x32x16_f16):
// a pure matrix kernel.  Both operands reach LDS by LDS-DMA (global_load_lds_dwordx4) -- no VGPR round trip, no conversion, no
// ds_write -- and the tile is 256 pixels x (NB * 64) channels, NB = 1 .. 4 (64 x (NB * 32) per wave, eight waves).
//
// Why (round 3, profiles/r3_fp16_conv_counters.json): conv3x3_halo2_kernel<., 1> -- fp32 activations, GroupNorm affine + SiLU + RNE
// rounding done by the halo loader -- issues 6 - 8 non-MFMA VALU instructions per MFMA (two transcendentals per input element, repeated
// for every 128-column tile and for the 25 - 55 % halo overlap), keeps the fp16 matrix pipe busy 30 % of the time (ImageNet-64, 64x64
// layers) and throws a quarter of that away on the ragged second tile of 192-channel layers.  Here the normalised, activated tensor
// is an fp16 NHWC tensor in HBM (written once per element by ds_norm_act with out_f16 -- the reference's own storage type in this
// mode, networks_edm.py:486) and the convolution only multiplies:
//
//   * halo of a 64-channel slab: NP pixels x 128 B, pixel-major, the 16-B chunk index XOR-swizzled by (v >> 1) & 7 (same
//     involution on the DMA source address and on the fragment read), v = the halo pixel index on 32- / 64-column images -- the 32
//     lanes of a fragment read are one image row: 16 consecutive pixels of one chunk hit 16 different bank quads -- and, round 4, the
//     pixel's COLUMN x in its halo row on 16-column images, x + 8 (row & 1) on 8-column images: there the 32 lanes are two / four image
//     rows whose halo pitch (18 / 10 pixels) is 2 / 10 mod 16, so a swizzle on the pixel index made every 16-lane group of a
//     ds_read_b128 hit two (three) addresses in the same bank quad (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.25 - 0.29 resp. 0.43 on
//     those instantiations, 0 on the others: profiles/r4_*_fp16_sq_counters.json; model: tools/probes/halo_bank_conflicts.py);
//     out-of-image pixels fetch a zero page.  Two halo buffers: slab s+1 streams in (one 8-KB DMA round per tap) while slab s
//     is multiplied;
//   * weights of a tap: NB * 64 rows x 128 B, swizzled the same way, in a ring of D = 2 .. 4 buffers (f16dma_ring), D - 1 taps ahead;
//   * per tap: one barrier; the A-fragment addresses of the 9 taps are 18 precomputed registers, a K step is `base ^ (ks << 5)`;
//   * 192-channel layers (ADM: 192 / 384 / 576 / 768) get 192-column tiles (NB = 3), 256-multiples NB = 4 where two halo
//     buffers + 64 KB of weights fit the 160 KB LDS (images of at most 32 columns), tails NB = 2 / 1;
//   * the fused epilogue of the family (bias, per-image embedding bias, residual, scale, SiLU, GroupNorm column sums) is unchanged:
//     outputs, norms and the residual stream stay fp32.
// Scope: taps == 9, stride 1, ONE fp16 source [M][c0] (c0 % 64 == 0; the decoder's concatenation is materialised by the norm pass),
// optional fused 1x1 skip projection on ONE fp16 source [M][ec0], square power-of-two images W in {8, 16, 32, 64} with 256-pixel
// tiles (one image per tile, or four 8x8 images -- fewer in the last tile of a batch that is not a multiple of four), cout % 64 == 0.
// Split-K (round 4, second part): layers whose widest tiling leaves half of the chip idle (the 8x8 stages: 56 - 64 tiles of 108 - 220 serial
// taps at the bench batches) contract S contiguous ranges of 64-channel slabs in S workgroups per tile (blockIdx.y), each writing its raw
// fp32 partial tile to the caller's workspace; splitk_reduce_f16_kernel (gemm_conv.hip) sums them in split order and applies the epilogue
// (bias, per-image bias, fp16 / fp32 residual, activation, fp16 / fp32 rows, GroupNorm column sums of the stored values).
// Fused input normalisation (round 5; NORM instantiations, ds_conv_args.norm_coefs with in_f16): the operand may be the RAW fp16 tensor
// (the block input / conv0's output as the producing epilogue stored it) instead of the activated copy a ds_norm_act pass writes.  The raw
// halo still arrives by LDS-DMA; once a DMA round of slab s + 1 has landed (two taps after it was issued) every thread rewrites the 16-byte
// unit IT fetched in place: h = silu((x - mu[c]) * A[c] + B[c]) -- the same fp32 expression and the same RNE rounding as norm_act_kernel, so
// the convolution's output bits equal those of the two-launch form -- while slab s is multiplied.  Out-of-image halo pixels are zero pages
// and are skipped (they must stay zero AFTER the affine).  The {mu, A, B} rows of the slab's 64 channels (768 B per image of the tile) ride
// in the unused tail of the halo buffer, fetched by one more LDS-DMA per slab; the appended 1x1 skip slabs stay raw.  With the
// normalisation in the kernel the decoder's channel concatenation needs no materialising pass either: a slab takes source 0 or source 1
// (both channel counts multiples of 64).  What it buys and costs per layer class: profiles/r5_conv_f16dma_fused_norm_ab.txt.
#include "pipe_common.h"
#include "epi_direct.h"

namespace igemm {
namespace {

__device__ __attribute__((aligned(128))) _Float16 g_zero_halfs[64];        // zero-initialised: the 128-B row of an out-of-image pixel

template <int W>
struct GeoD {
    static constexpr int NIMG = (W * W >= 256) ? 1 : 256 / (W * W);        // image slots per tile (8x8 images: 4)
    static constexpr int TH = 256 / (W * NIMG), WP = W + 2, HP = TH + 2, NP = NIMG * HP * WP;
    static constexpr int NDMA = (NP * 8 + 511) / 512;                      // DMA rounds per halo (512 threads x 16 B = 8 KB each)
    static constexpr unsigned HALO_B = NDMA * 8192u;
    // fused input normalisation: {mu, A, B} x 64 channels x NIMG images in the tail of the halo buffer, behind its NP pixels
    static constexpr unsigned COEF_OFF = NP * 128u, COEF_UNITS = NIMG * 48u;
    static_assert(COEF_OFF + NIMG * 768u <= HALO_B, "coefficient rows must fit behind the halo pixels");
};

// Weight ring depth (round 4): a tap's weights are requested D - 1 taps ahead.  D = 2 is the double buffer of round 3 -- enough for NB >= 3,
// where a tap holds >= 24 MFMAs per wave (1.2 us at the clock the part sustains) and the request of tap kt + 2 has that long to land.  A
// 64-column tile (NB = 1: every layer of the 8x8 stages, 10 % of ImageNet-64 fp16 and 6 % of SD-1.5 fp16, and the tail tiles) holds 8 MFMAs
// per tap: 0.4 us of work against ~1 us of L2 -> LDS latency, measured 1.05 us per tap = 0.2 of the matrix rate of the wide tiles.  Such
// tiles leave LDS unused, so the ring takes what fits next to the two halo buffers, at most four taps.  Measured (session gpurun_out/r6e,
// against the same library with D = 2 everywhere): nothing on a layer benchmarked alone, where the weights stay in L2 / MALL between
// launches (8x8 768 -> 768: 0.077 - 0.086 ms both ways), but +2.5 % on ImageNet-64 fp16 and +1.5 % on SD-1.5 fp16 as samplers, where every
// layer's weights come from HBM: profiles/r4_conv_f16dma_ring_ab.txt.
template <int W, int NB>
constexpr int f16dma_ring() {
    const int fit = (int)((160u * 1024u - 2u * GeoD<W>::HALO_B) / (NB * 8192u));
    return NB >= 3 ? 2 : (fit > 4 ? 4 : (fit < 2 ? 2 : fit));
}
template <int W, int NB>
constexpr unsigned f16dma_smem() { return (unsigned)f16dma_ring<W, NB>() * NB * 8192u + 2u * GeoD<W>::HALO_B; }

// DIRECT: the epilogue that stores straight from the accumulators (epi_direct.h) or the staged one (epilogue_pipe) -- one instantiation
// each, chosen by the launcher (epi_direct_ok): a kernel body holding both allocates registers for the worse of the two.
// (The DMA requests of a tap are issued between the MFMAs of its last K step, round 4; the round-3 order "behind them" was kept as an
// A/B instantiation until the comparison was recorded in docs/HISTORY.md section E.6.)
template <int W, int NB, bool DIRECT, bool NORM>
__global__ void __launch_bounds__(512, 2) conv3x3_f16dma_kernel(const KParams p) {
    using G = GeoD<W>;
    constexpr int WP = G::WP, HP = G::HP, TH = G::TH, NIMG = G::NIMG, NP = G::NP, NDMA = G::NDMA;
    constexpr unsigned WB = NB * 8192u, HB = G::HALO_B;
    constexpr int D = f16dma_ring<W, NB>();                    // weight ring: taps kt .. kt + D - 1 are in LDS or in flight
    static_assert(f16dma_smem<W, NB>() <= 160u * 1024u, "LDS");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);                 // [weights 0 | ... | weights D - 1 | halo 0 | halo 1]
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt, 0)) return;
    const int m0 = mt * 256, n0 = p.n_begin + nt * (NB * 64);
    const _Float16* a0 = reinterpret_cast<const _Float16*>(p.a0);
    const _Float16* e0 = reinterpret_cast<const _Float16*>(p.e0);
    const _Float16* wgt = reinterpret_cast<const _Float16*>(p.b);
    const size_t ldbh = (size_t)p.ldb * 2;                     // weight row pitch in halfs

    const int img0 = m0 / p.HW;
    const int r0 = NIMG == 1 ? (m0 - img0 * p.HW) / W : 0;
    // 8x8 images: four image slots per tile; the last tile of a batch that is not a multiple of four has empty slots (round 6) -- their halo
    // pixels are zero pages, their output rows lie beyond M and are masked by the epilogue
    const int nimgs = p.M / p.HW;

    // ---- halo DMA: thread tid owns 16-B unit j * 512 + tid of round j: pixel (unit >> 3), LDS chunk slot tid & 7 -----------------
    int hpix[NDMA];                                            // source pixel (-1: zero page)
    // source channel offset (halfs): chunk slot ^ swizzle(pixel).  32- / 64-column images: (pixel >> 1) & 7 with pixel = j * 64 + (tid >> 3),
    // the same for every round j; 8- / 16-column images: by the pixel's column (and row parity) in the halo, recomputed per round in halo_dma
    // (a handful of VALU in the shadow of the tap's MFMAs instead of a register kept across the K loop)
    const int hch = ((tid & 7) ^ ((tid >> 4) & 7)) * 8;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int hp = j * 64 + (tid >> 3);
        const int sl = hp / (HP * WP), rem = hp - sl * (HP * WP);
        const int hr = rem / WP, hc = rem - hr * WP;
        const int y = r0 + hr - 1, x = hc - 1;
        const bool ok = hp < NP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)W && (NIMG == 1 || img0 + sl < nimgs);
        hpix[j] = ok ? ((img0 + sl) * p.H + y) * W + x : -1;
        if (NORM && hp >= NP) hpix[j] = -2;                    // no such halo pixel: nothing is fetched (the buffer's tail holds the coefficient rows)
    }
    const int n0ch = p.c0 / 64, ne0ch = p.ec0 / 64;            // NORM: slabs of source 0; the rest come from source 1
    const int nchunks = NORM ? (p.c0 + p.c1) / 64 : p.c0 / 64; // 3x3 slabs (9 taps each)
    const int nextra = NORM ? (p.ec0 + p.ec1) / 64 : p.ec0 / 64;   // appended 1x1 slabs (centre tap only)
    // This workgroup contracts slabs [cb, NCH) = taps [kt0, KT) of the layer's K: everything, or -- split-K, blockIdx.y = split -- the
    // range whose boundaries are the slab boundaries nearest to the equal-tap cuts (a 3x3 slab weighs nine taps, an appended 1x1 slab one)
    int cb = 0, NCH = nchunks + nextra;
    if (p.splits > 1) {
        const int kt_all = nchunks * 9 + nextra, sp = (int)blockIdx.y;
        auto cut = [&](int i) {
            const int t = (int)(((long long)kt_all * i) / p.splits);
            return t <= nchunks * 9 ? (t + 4) / 9 : nchunks + (t - nchunks * 9);
        };
        cb = cut(sp);
        if (sp + 1 < p.splits) NCH = cut(sp + 1);
    }
    const int kt0 = cb <= nchunks ? cb * 9 : nchunks * 9 + (cb - nchunks);
    const int KT = NCH <= nchunks ? NCH * 9 : nchunks * 9 + (NCH - nchunks);
    const int abl = p.coef_lds;                                 // timing ablations (ds_conv_args.tune.ablate; results are wrong when set)
    auto halo_dma = [&](int chunk, int hbuf, auto jc) {          // DMA round j of slab `chunk` into halo buffer hbuf
        constexpr int j = decltype(jc)::value;
        if ((abl & 2) && chunk > 0) return;
        const bool extra = chunk >= nchunks;
        const _Float16* base = extra ? e0 + (size_t)(chunk - nchunks) * 64 : a0 + (size_t)chunk * 64;
        int ld = extra ? p.elda0 : p.lda0;
        if constexpr (NORM) {                                  // two sources: the decoder's concatenation is never materialised
            if (hpix[j] == -2) return;
            if (!extra && chunk >= n0ch) { base = reinterpret_cast<const _Float16*>(p.a1) + (size_t)(chunk - n0ch) * 64; ld = p.lda1; }
            if (extra && chunk - nchunks >= ne0ch) { base = reinterpret_cast<const _Float16*>(p.e1) + (size_t)(chunk - nchunks - ne0ch) * 64; ld = p.elda1; }
        }
        int hcj = hch;
        if constexpr (W <= 16) {                               // halo pixel of this unit: row hp / WP (counted over the tile's image slots), column hp % WP
            const unsigned hp = (unsigned)(j * 64 + (tid >> 3)), hrow = hp / (unsigned)WP, hcol = hp - hrow * (unsigned)WP;
            hcj = (int)(((unsigned)(tid & 7) ^ (((hcol + (W == 8 ? 8u * (hrow & 1u) : 0u)) >> 1) & 7u)) * 8u);
        }
        const _Float16* g = hpix[j] >= 0 ? base + (size_t)hpix[j] * ld + hcj : g_zero_halfs;
        DS_RACE_SKEW(wave);
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + D * WB + hbuf * HB + (j * 512 + wave * 64) * 16), 16, 0, 0);
    };
    // ---- NORM: the slab's {mu, A, B} rows -> tail of halo buffer hbuf (unit u = tid < NIMG * 48: image slot u / 48, plane (u % 48) / 16) ----
    auto coef_dma = [&](int chunk, int hbuf) {
        if constexpr (NORM) {
            if (wave * 64 < (int)G::COEF_UNITS) {
                if (tid < (int)G::COEF_UNITS) {
                    const int sl = tid / 48, rem = tid - sl * 48;
                    const int im = NIMG == 1 ? img0 : min(img0 + sl, nimgs - 1);          // (an empty slot's rows are never used: any valid address)
                    const float* g = p.norm + ((size_t)im * 3 + (rem >> 4)) * (size_t)(p.c0 + p.c1) + (size_t)chunk * 64 + (rem & 15) * 4;
                    DS_RACE_SKEW(wave);
                    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + D * WB + hbuf * HB + G::COEF_OFF + wave * 64 * 16), 16, 0, 0);
                }
            }
        }
    };
    // ---- NORM: rewrite the unit this thread fetched in round j of a 3x3 slab's halo (buffer hbuf) in place: silu((x - mu) * A + B), fp32
    // arithmetic, RNE to fp16 -- the expressions of norm_act_kernel (norm_act.hip: xf / to_h4).  Callers guarantee the round has landed. ----
    auto norm_round = [&](int hbuf, auto jc) {
        if constexpr (NORM) {
            constexpr int j = decltype(jc)::value;
            if (hpix[j] < 0) return;                           // zero page (must stay zero) or no pixel
            const unsigned hbase = lds_addr2(smem) + D * WB + (unsigned)hbuf * HB;
            const unsigned ua = hbase + (unsigned)(j * 512 + tid) * 16u;
            unsigned oct = (unsigned)hch >> 3;                 // channel octet of the unit inside the slab = DMA source offset / 8
            const unsigned hp = (unsigned)(j * 64 + (tid >> 3));
            if constexpr (W <= 16) {
                const unsigned hrow = hp / (unsigned)WP, hcol = hp - hrow * (unsigned)WP;
                oct = (unsigned)(tid & 7) ^ (((hcol + (W == 8 ? 8u * (hrow & 1u) : 0u)) >> 1) & 7u);
            }
            unsigned ca = hbase + G::COEF_OFF + oct * 32u;
            if constexpr (NIMG > 1) ca += (hp / (unsigned)(HP * WP)) * 768u;
            f32x4 xr = lds_rd<0>(ua);
            f32x4 m0 = lds_rd<0>(ca), m1 = lds_rd<16>(ca), g0 = lds_rd<256>(ca), g1 = lds_rd<272>(ca), b0 = lds_rd<512>(ca), b1 = lds_rd<528>(ca);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xr), "+v"(m0), "+v"(m1), "+v"(g0), "+v"(g1), "+v"(b0), "+v"(b1));
            const h8 xh = __builtin_bit_cast(h8, xr);
            const bool act = p.norm_act == DS_ACT_SILU;
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float mu = e < 4 ? m0[e & 3] : m1[e & 3], ga = e < 4 ? g0[e & 3] : g1[e & 3], be = e < 4 ? b0[e & 3] : b1[e & 3];
                const float v = (float)xh[e];
                const float u = (v - mu) * ga + be;
                t[e] = act ? ds_silu(u) : u;
            }
            const f32x4 o = {pack_h2(t[0], t[1]), pack_h2(t[2], t[3]), pack_h2(t[4], t[5]), pack_h2(t[6], t[7])};
            DS_RACE_SKEW(wave);
            lds_wr<0>(ua, o);
        }
    };

    // ---- weight DMA of K tile (tap) kt: rows i * 64 + (tid >> 3), i < NB; the source chunk is pre-swizzled -----------------------
    const _Float16* wsrc = wgt + (size_t)(n0 + (tid >> 3)) * ldbh + (((tid & 7) ^ ((tid >> 4) & 7)) * 8);
    auto w_dma = [&](int kt, int wbuf) {
        if ((abl & 1) && kt >= D) return;
        DS_RACE_SKEW(wave);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (size_t)i * 64 * ldbh + (size_t)kt * 64),
                                             (lptr_t)(lds + wbuf * WB + (i * 64 + wave * 8) * 128), 16, 0, 0);
    };
    auto w_dma_row = [&](int kt, int wbuf, auto ic) {            // one 64-row block of it (issued between MFMAs, see the tap)
        constexpr int i = decltype(ic)::value;
        if ((abl & 1) && kt >= D) return;
        if (i == 0) DS_RACE_SKEW(wave);
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (size_t)i * 64 * ldbh + (size_t)kt * 64),
                                         (lptr_t)(lds + wbuf * WB + (i * 64 + wave * 8) * 128), 16, 0, 0);
    };

    // ---- fragment addresses (LDS byte addresses relative to a halo buffer / a weight buffer) ------------------------------------------
    // A fragments: row block i covers 32 consecutive output pixels; tap (ty, tx) reads halo pixel hp0[i] + ty * WP + tx, chunk
    // (2 ks + g) ^ ((hp >> 1) & 7).  Only hp0 is kept; the 5 VALU per (tap, block) that rebuild the address are noise next to 16 NB MFMAs.
    int hp0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wr * 64 + i * 32 + (lane & 31);
        const int sl = m / (TH * W), rem = m - sl * (TH * W);
        const int r = rem / W, c = rem - r * W;
        hp0[i] = (sl * HP + r) * WP + c;
    }
    const unsigned gsel = (unsigned)(lane >> 5);
    auto a_addr = [&](int i, int tt) -> unsigned {             // byte offset inside a halo buffer, K step 0
        const unsigned hp = (unsigned)(hp0[i] + (tt / 3) * WP + (tt % 3));
        unsigned sw = (hp >> 1) & 7u;
        if constexpr (W <= 16) {                               // swizzle by the column (and row parity) of the halo pixel: see the header
            const unsigned hrow = hp / (unsigned)WP, hcol = hp - hrow * (unsigned)WP;
            sw = ((hcol + (W == 8 ? 8u * (hrow & 1u) : 0u)) >> 1) & 7u;
        }
        return hp * 128u + 16u * (sw ^ gsel);
    };
    const int brow = wc * (NB * 32) + (lane & 31);
    const unsigned lds0 = lds_addr2(smem);
    const unsigned bbase = lds0 + (unsigned)brow * 128u + 16u * (unsigned)(((brow >> 1) & 7) ^ (lane >> 5));

    f32x16 accA[2][2], accB[2][2];                             // output columns [0, 64) and [64, 128) of the wave tile
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { accA[i][j][r] = 0.f; accB[i][j][r] = 0.f; }

    // Fragment sets of one K step (16 channels): two A blocks (32 pixels each), NB weight blocks.  Reads are volatile asm in program
    // order; the wait names the set it releases ("+v"), so no use can move above it (cdna_hip_programming.md 5.7, form (ii)).
    struct Frag { f32x4 a0, a1, b0, b1, b2, b3; };
    auto frag_read = [&](Frag& f, unsigned va0, unsigned va1, unsigned vb) {
        if (abl & 32) return;                                  // (timing ablation: no fragment reads)
        f.a0 = lds_rd<0>(va0);
        f.a1 = lds_rd<0>(va1);
        f.b0 = lds_rd<0>(vb);
        if constexpr (NB > 1) f.b1 = lds_rd<4096>(vb);
        if constexpr (NB > 2) f.b2 = lds_rd<8192>(vb);
        if constexpr (NB > 3) f.b3 = lds_rd<12288>(vb);
    };
    auto frag_wait = [&](Frag& f, auto nc) {                   // wait until at most N younger LDS operations are outstanding
        constexpr int N = decltype(nc)::value;
        if constexpr (NB == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0) : "n"(N));
        if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1) : "n"(N));
        if constexpr (NB == 3) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2) : "n"(N));
        if constexpr (NB == 4)
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2), "+v"(f.b3) : "n"(N));
    };
    // SWAPPED product (round 4): the weight fragment is the MFMA's first operand, so an accumulator block holds lane = pixel, registers =
    // channels -- what epilogue_direct (igemm_common.h) stores without an LDS transpose
#define DSD_MM(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, b_), __builtin_bit_cast(h8, a_), acc_, 0, 0, 0)
    auto mfma_group = [&](Frag& f) {                           // consecutive MFMAs never touch the same accumulator
        DSD_MM(accA[0][0], f.a0, f.b0); DSD_MM(accA[1][0], f.a1, f.b0);
        if constexpr (NB > 1) { DSD_MM(accA[0][1], f.a0, f.b1); DSD_MM(accA[1][1], f.a1, f.b1); }
        if constexpr (NB > 2) { DSD_MM(accB[0][0], f.a0, f.b2); DSD_MM(accB[1][0], f.a1, f.b2); }
        if constexpr (NB > 3) { DSD_MM(accB[0][1], f.a0, f.b3); DSD_MM(accB[1][1], f.a1, f.b3); }
    };
    constexpr int NR = 2 + NB;                                 // LDS reads per fragment set
    const unsigned halo0 = lds0 + D * WB;

    // ---- prologue: halo of slab 0, the part of slab 1's halo that is due (see the tap), weights of taps 0 and 1 -------------------
    DS_TL(p.splits > 1 ? nullptr : p.part, abl, 0, blockIdx.x);        // (phase stamps: the 'timeline' diagnostics build only, csrc/ds_common.h)
    static_for<NDMA>([&](auto jc) { halo_dma(cb, cb & 1, jc); });
    if (cb < nchunks) coef_dma(cb, cb & 1);
    if (cb + 1 < NCH) {
        if (cb >= nchunks) static_for<NDMA>([&](auto jc) { halo_dma(cb + 1, (cb + 1) & 1, jc); });      // the first slab is a one-tap slab
        else halo_dma(cb + 1, (cb + 1) & 1, IC<0>{});
        if (cb + 1 < nchunks) coef_dma(cb + 1, (cb + 1) & 1);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (kt0 + d < KT) w_dma(kt0 + d, d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (NORM) {
        // the first slab's halo is normalised here.  A thread rewrites only the unit it fetched itself -- covered by its own vmcnt(0) -- but the
        // COEFFICIENT rows were fetched by wave 0 alone: every other wave needs wave 0's wait + a barrier before it reads them.  (Without this
        // barrier the kernel passed every test and whole-network comparison for a day and then produced four wrong images in one first run at
        // 64 images per call: tools/diag_fuse_norm.py, docs/HISTORY.md G.5.  In the steady state the taps' own barriers separate the two.)
#ifndef DS_TEST_DROP_G5_BARRIER                                // tests only (build.py 'stress_g5'): the race re-introduced, to prove that the stress build catches it
        __builtin_amdgcn_s_barrier();
#endif
        if (cb < nchunks) static_for<NDMA>([&](auto jc) { norm_round(cb & 1, jc); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    DS_TL(p.splits > 1 ? nullptr : p.part, abl, 1, blockIdx.x);
    Frag P_, Q_;
    {
        const unsigned ctr = cb < nchunks ? 0u : 4u;           // first tap: (0, 0) of a 3x3 slab, or the centre tap of a 1x1 slab
        const unsigned h0 = halo0 + (unsigned)(cb & 1) * HB;
        frag_read(P_, h0 + a_addr(0, (int)ctr), h0 + a_addr(1, (int)ctr), bbase);
    }

    int kt = kt0, slot = 0;                                    // slot = (kt - kt0) % D: the ring buffer of tap kt
    // One tap: T9 = tap of a 3x3 slab (0..8) or 9 = the centre tap of an appended 1x1 slab.  P holds the fragments of its K step 0
    // (read after the previous tap's barrier).
    //   K steps 0..2 : reads of step k+1 in flight under the MFMAs of step k
    //   then         : all reads of this tap done, own DMAs landed (counted, see below), barrier: ring buffer kt % D -- and, at a slab end, the
    //                  halo buffer -- are free and the operands of tap kt+1 are in LDS
    //   K step 3     : behind the barrier: the first fragment reads of tap kt+1, the step's MFMAs, then -- in their shadow -- the DMA
    //                  issue of tap kt+2's weights and of the next halo round
    // Halo schedule: slab s+1 lives in buffer (s+1) & 1, free once slab s-1 is done; its DMA rounds are issued one per barrier from the
    // last tap of slab s-1 on (all of them at once when slab s is a one-tap slab).
    auto tap = [&](auto t9c, int chunk) {
        Frag &P = P_, &Q = Q_;
        constexpr int T9 = decltype(t9c)::value;
        constexpr bool X = (T9 == 9);
        constexpr int TT = X ? 4 : T9;
        constexpr bool SLAB_END = X || T9 == 8;
        const unsigned hoff = halo0 + (unsigned)(chunk & 1) * HB;
        const unsigned woff = (unsigned)slot * WB;
        const int nslot = slot + 1 == D ? 0 : slot + 1;
        const unsigned a_0 = a_addr(0, TT) + hoff, a_1 = a_addr(1, TT) + hoff;
        const unsigned vb = bbase + woff;
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
        // own DMAs landed: everything at a slab end (the next slab's halo) and in the last D - 1 taps; otherwise all but the youngest
        // (D - 2) * NB requests -- the weights of taps kt + 2 .. kt + D - 1 (loads complete in order; halo rounds issued in between only
        // make this wait for more than it needs)
        if (D == 2 || SLAB_END || kt + D > KT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NB) : "memory");
        if (!(abl & 16)) __builtin_amdgcn_s_barrier();         // (timing ablation: no per-tap barrier)
        DS2_FENCE();
        // ---- behind the barrier: first the fragment reads of tap kt+1 and the last K step's MFMAs, THEN the DMA issue (address
        // arithmetic, M0 writes, NB + 1 LDS-DMA instructions: ~150-300 cycles per wave) in the shadow of those MFMAs -------------------
        if (kt + 1 < KT) {
            const unsigned nwoff = (unsigned)nslot * WB;
            if constexpr (SLAB_END) {
                const unsigned nh = halo0 + (unsigned)((chunk + 1) & 1) * HB;
                const int nt9 = chunk + 1 >= nchunks ? 4 : 0;
                frag_read(P, nh + a_addr(0, nt9), nh + a_addr(1, nt9), bbase + nwoff);
            } else {
                frag_read(P, a_addr(0, T9 + 1) + hoff, a_addr(1, T9 + 1) + hoff, bbase + nwoff);
            }
        }
        auto halo_issue = [&]() {
            if constexpr (SLAB_END) {
                if (chunk + 2 < NCH) {
                    if (chunk + 1 >= nchunks) static_for<NDMA>([&](auto jc) { halo_dma(chunk + 2, chunk & 1, jc); });   // next slab has one tap
                    else halo_dma(chunk + 2, chunk & 1, IC<0>{});
                    if (chunk + 2 < nchunks) coef_dma(chunk + 2, chunk & 1);
                }
            } else if constexpr (T9 + 1 < NDMA) {
                if (chunk + 1 < NCH) halo_dma(chunk + 1, (chunk + 1) & 1, IC<T9 + 1>{});
            }
        };
        {
            // Round 4: the requests are issued BETWEEN the MFMAs of the group.  A wave issues in order: behind the group's last MFMA only
            // its own 32 cycles shelter anything, and both waves of a SIMD reach this point together (they left the same barrier), so the
            // 150 - 300 cycles of address arithmetic, M0 writes and LDS-DMA issue left the matrix pipe idle.  Behind MFMA pair i go weight
            // rows [i * 64, i * 64 + 64) of tap kt + D; the halo round goes behind the first pair.
            const bool wd = kt + D < KT;
            DS2_FENCE();
            DSD_MM(accA[0][0], Q.a0, Q.b0); DSD_MM(accA[1][0], Q.a1, Q.b0);
            DS2_FENCE(); halo_issue(); if (wd) w_dma_row(kt + D, slot, IC<0>{}); DS2_FENCE();
            if constexpr (NB > 1) {
                DSD_MM(accA[0][1], Q.a0, Q.b1); DSD_MM(accA[1][1], Q.a1, Q.b1);
                DS2_FENCE(); if (wd) w_dma_row(kt + D, slot, IC<1>{}); DS2_FENCE();
            }
            if constexpr (NB > 2) {
                DSD_MM(accB[0][0], Q.a0, Q.b2); DSD_MM(accB[1][0], Q.a1, Q.b2);
                DS2_FENCE(); if (wd) w_dma_row(kt + D, slot, IC<2>{}); DS2_FENCE();
            }
            if constexpr (NB > 3) {
                DSD_MM(accB[0][1], Q.a0, Q.b3); DSD_MM(accB[1][1], Q.a1, Q.b3);
                DS2_FENCE(); if (wd) w_dma_row(kt + D, slot, IC<3>{}); DS2_FENCE();
            }
        }
        DS2_FENCE();
        if constexpr (NORM && !X && T9 >= 1 && T9 <= NDMA) {
            // round T9 - 1 of the NEXT 3x3 slab's halo was issued TWO taps ago (round 0: at the previous slab's last tap, or in the prologue),
            // so this tap's DMA wait covered it under every ring depth: with D = 4 the wait leaves the youngest 2 NB requests outstanding, and
            // a round issued only one tap ago sits among them (behind that tap's NB weight rows).  Normalise it now, in the registers the last
            // K step's fragments have just released; rounds 0 .. NDMA - 1 <= 6 take taps 1 .. 7, the slab's last barrier publishes them.
            if (chunk + 1 < nchunks && chunk + 1 < NCH) norm_round((chunk + 1) & 1, IC<T9 - 1>{});
            DS2_FENCE();
        }
        ++kt; slot = nslot;
    };
    int chunk = cb;
    const int n33 = nchunks < NCH ? nchunks : NCH;             // end of this range's 3x3 slabs
    for (; chunk < n33; ++chunk) {
        tap(IC<0>{}, chunk); tap(IC<1>{}, chunk); tap(IC<2>{}, chunk);
        tap(IC<3>{}, chunk); tap(IC<4>{}, chunk); tap(IC<5>{}, chunk);
        tap(IC<6>{}, chunk); tap(IC<7>{}, chunk); tap(IC<8>{}, chunk);
    }
    for (; chunk < NCH; ++chunk) tap(IC<9>{}, chunk);
#undef DSD_MM
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (no fragment read is pending after the last tap; cheap insurance)
    DS_TL(p.splits > 1 ? nullptr : p.part, abl, 2, blockIdx.x);

    if (abl & 4) {                                             // no epilogue: every accumulator block (and so every MFMA) is kept alive
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { asm volatile("" :: "v"(accA[i][j])); asm volatile("" :: "v"(accB[i][j])); }
        return;
    }
    float* stage = smem + wave * 32 * EPI_LD;
    const int wn0 = n0 + wc * (NB * 32);
    if (p.splits > 1) {                                        // split-K: the raw partial tile, fp32 rows of plane blockIdx.y (staged instantiation only)
        if constexpr (!DIRECT) {
            KParams q = p;
            q.out = p.part + (size_t)blockIdx.y * p.M * p.N; q.ldo = p.N;
            q.colbias = nullptr; q.rowbias = nullptr; q.cbias = nullptr; q.res = nullptr; q.scale = 1.f; q.act = DS_ACT_NONE;
            q.stats = nullptr; q.out_f16 = 0; q.res_f16 = 0; q.out_planar = 0;
            epilogue_pipe<0, true, (NB == 1 ? 32 : 64), (NB == 3 ? 32 : (NB == 4 ? 64 : 0)), true>(q, accA, accB, stage, lane, m0 + wr * 64, wn0, q.out);
        }
        return;
    }
    // non-temporal residual loads / output stores: +1 ... 2 % (A/B, profiles/r3_conv_f16dma_ablations.txt); the launcher only takes this
    // kernel on the vector path (vec_ok, cout a multiple of the tile width)
    if constexpr (DIRECT) epilogue_direct<false, NB, true>(p, accA, accB, lane, m0 + wr * 64, wn0);
    else epilogue_pipe<0, true, (NB == 1 ? 32 : 64), (NB == 3 ? 32 : (NB == 4 ? 64 : 0)), true>(p, accA, accB, stage, lane, m0 + wr * 64, wn0, p.out);
#ifdef DS_TIMELINE
    DS_TL(p.part, abl, 3, blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DS_TL(p.part, abl, 4, blockIdx.x);
#endif
}

// p.t_ablate (ds_conv_args.tune.ablate), benchmarks only: bit 0 = no weight DMA after the prologue, 1 = no halo DMA after slab 0,
// 2 = no epilogue, 4 = no per-tap barrier, 5 = no fragment reads

template <int W, int NB>
int launch_w_nb(KParams p, int n_begin, int ntiles, hipStream_t stream) {
    p.mtiles = (p.M + 255) / 256;
    p.ntiles = ntiles;
    p.n_begin = n_begin;
    if (p.splits < 1) p.splits = 1;                            // > 1: set by launch_conv3x3_f16dma (the reduce follows the last column range)
    p.coef_lds = p.t_ablate;
    int smem = (int)f16dma_smem<W, NB>();
    const int epi = 8 * 32 * EPI_LD * (int)sizeof(float);
    if (smem < epi) smem = epi;
    const bool staged = p.splits > 1 || !epi_direct_ok(p, true, NB);          // partial tiles leave through the staged epilogue
    const dim3 grid(grid_1d(p.mtiles, p.ntiles), staged ? p.splits : 1);
#define DSD_LAUNCH(DIRECT_, NORM_)                                                                                             \
    do {                                                                                                                       \
        DS_ENSURE_DYN_LDS((&conv3x3_f16dma_kernel<W, NB, DIRECT_, NORM_>), 160 * 1024);                                        \
        hipLaunchKernelGGL((conv3x3_f16dma_kernel<W, NB, DIRECT_, NORM_>), grid, dim3(512), smem, stream, p);                  \
    } while (0)
    if (p.norm) {
        if constexpr (NB < 4) { if (staged) DSD_LAUNCH(false, true); else DSD_LAUNCH(true, true); }        // (max_nb: no 256-column NORM tile)
        else return DS_E_SHAPE;
    } else { if (staged) DSD_LAUNCH(false, false); else DSD_LAUNCH(true, false); }
#undef DSD_LAUNCH
    DS_CHECK_LAUNCH();
    return DS_OK;
}

template <int W>
int launch_w(const KParams& p, int nb, int n_begin, int ntiles, hipStream_t stream) {
    switch (nb) {
        case 1: return launch_w_nb<W, 1>(p, n_begin, ntiles, stream);
        case 2: return launch_w_nb<W, 2>(p, n_begin, ntiles, stream);
        case 3: return launch_w_nb<W, 3>(p, n_begin, ntiles, stream);
        default:
            if constexpr (f16dma_smem<W, 4>() <= 160u * 1024u) { if (!p.norm) return launch_w_nb<W, 4>(p, n_begin, ntiles, stream); }
            return DS_E_SHAPE;
    }
}

}  // namespace

// p.t_nb (ds_conv_args.tune.f16dma_nb), benchmarks / tests: > 0 forces the column-tile width of the main launch (64 * nb columns)

// widest column tile the LDS holds next to two halo buffers; with the fused input normalisation 192 columns (the 256-column instantiations have
// no registers left for the normalisation's temporaries: 6 - 137 spilled, tools/kernel_resources.py)
static int max_nb(const KParams& p) { return (!p.norm && (p.W == 16 || p.W == 32)) ? 4 : 3; }

bool conv3x3_f16dma_applicable(const KParams& p) {
    if (p.taps != 9 || p.stride > 1) return false;
    if (!(p.W == 8 || p.W == 16 || p.W == 32 || p.W == 64) || p.H != p.W) return false;
    if (p.HW != p.H * p.W || p.M % p.HW) return false;
    if (p.M % 256 && !(p.W == 8 && p.M % 64 == 0)) return false;      // whole 256-pixel tiles; 8x8 images: a last tile with one to three images (round 6)
    if (p.c0 <= 0 || p.c0 % 64 || p.ec0 % 64) return false;
    if (p.norm) {           // fused input normalisation (NORM instantiations): second sources allowed, whole 64-channel slabs each
        if ((p.norm_act != DS_ACT_NONE && p.norm_act != DS_ACT_SILU) || p.c1 % 64 || p.ec1 % 64 || (p.ec1 && !p.ec0)) return false;
    } else if (p.c1 != 0 || p.ec1 != 0) return false;
    if (p.N % 64 || !p.vec_ok || p.nrows_b < p.N) return false;
    return true;
}

// Column tiling of a layer: a list of (first column, tiles, NB), widest tiles first (e.g. 320 = 192 + 128, 640 = 2 x 256 + 128).
// The starting width is chosen by a small cost model: a launch of t workgroups takes ceil(t / 256) rounds of tiles, and a tile of
// 64 * nb columns costs about 1 + nb (the halo stream, the A-fragment reads, prologue and epilogue do not shrink with the width), so
// a layer with few pixel tiles takes narrower column tiles to cover the 256 CUs.
// `half`: the four-wave half-slab variant (conv3x3_f16dmah.hip): 128-pixel tiles, two workgroups per CU = 512 tile slots per round.
static int tiling(const KParams& p, int nb0, int (*out)[3], int* cost, bool half = false) {
    const int mtiles = (p.M + (half ? 127 : 255)) / (half ? 128 : 256), slots = half ? 512 : 256;
    int n = 0, col = 0, c = 0;
    for (int w = nb0; w >= 1 && col < p.N; --w) {
        const int t = (p.N - col) / (64 * w);
        if (t > 0) {
            out[n][0] = col; out[n][1] = t; out[n][2] = w; ++n;
            col += t * 64 * w;
            c += (int)(((long long)mtiles * t + slots - 1) / slots) * (1 + w);
        }
    }
    *cost = c;
    return n;
}

int conv3x3_f16dma_plan(const KParams& p, int (*out)[3], bool half = false) {
#ifdef DS_BUILD_EXPERIMENTS
    const int cap = half ? conv3x3_f16dmah_max_nb(p.W) : max_nb(p);
#else
    const int cap = max_nb(p);
#endif
    int cost;
    if (p.t_nb > 0) return tiling(p, p.t_nb < cap ? p.t_nb : cap, out, &cost, half);
    int best_nb = cap, best_cost = 0x7fffffff, best_n = 99;
    for (int nb = cap; nb >= 1; --nb) {
        int tmp[4][3];
        const int n = tiling(p, nb, tmp, &cost, half);
        if (cost < best_cost || (cost == best_cost && n < best_n)) { best_cost = cost; best_n = n; best_nb = nb; }
    }
    return tiling(p, best_nb, out, &cost, half);
}

// Split-K (see the header): S > 1 when the WIDEST tiling of the layer (fewest LDS operand bytes per MFMA) fills at most half of the 256 CUs and
// every split keeps at least two 3x3 slabs' worth of taps (18): S = 256 / tiles, at most 16, within the workspace.  S is a function of the
// layer alone -- never of a forced tile width (ds_conv_args.tune.f16dma_nb) -- so the order of the fp32 sums, and with it every output bit,
// is the same under any tile shape the planner picks; a split layer takes the widest tiling unless the width is forced.
// ds_conv_args.tune.splits forces S (1 = never).
static int conv3x3_f16dma_splits(const KParams& p, int (*plan)[3], int* n) {
    if (!p.part || !p.vec_part || p.t_splits == 1) return 1;
    int wide[4][3], cost;
    const int nw = tiling(p, max_nb(p), wide, &cost);
    long long tiles = 0;
    for (int i = 0; i < nw; ++i) tiles += (long long)((p.M + 255) / 256) * wide[i][1];
    long long s = p.t_splits > 1 ? p.t_splits : (tiles <= 128 ? 256 / tiles : 1);
    const long long kt_all = (long long)((p.c0 + p.c1) / 64) * 9 + (p.ec0 + p.ec1) / 64, mn = (long long)p.M * p.N;
    if (s > 16) s = 16;
    if (s > kt_all / 18) s = kt_all / 18;
    if (s * mn > p.part_cap) s = p.part_cap / mn;
    if (s < 2) return 1;
    if (p.t_nb <= 0) {
        for (int i = 0; i < nw; ++i) { plan[i][0] = wide[i][0]; plan[i][1] = wide[i][1]; plan[i][2] = wide[i][2]; }
        *n = nw;
    }
    return (int)s;
}

// Which layers take the four-wave half-slab variant (two workgroups per CU, conv3x3_f16dmah.hip).  ds_conv_args.tune.f16dma_nw forces it
// (4) or the eight-wave kernel (8); otherwise by layer class, from the A/B of profiles/r4_conv_f16dmah_ab.txt.
bool conv3x3_f16dma_use_half(const KParams& p) {
#ifdef DS_BUILD_EXPERIMENTS                                    // a recorded negative result: built with DS_BUILD_EXPERIMENTS=1 only, never a default
    return p.t_nw == 4 && conv3x3_f16dmah_applicable(p);
#else
    (void)p;
    return false;
#endif
}

int launch_conv3x3_f16dma(KParams& p, hipStream_t stream) {
    int plan[4][3];
#ifdef DS_BUILD_EXPERIMENTS
    if (conv3x3_f16dma_use_half(p)) {
        const int n = conv3x3_f16dma_plan(p, plan, true);
        for (int i = 0; i < n; ++i) {
            const int rc = launch_conv3x3_f16dmah_tiles(p, plan[i][2], plan[i][0], plan[i][1], stream);
            if (rc) return rc;
        }
        return DS_OK;
    }
#endif
    int n = conv3x3_f16dma_plan(p, plan);
    p.splits = conv3x3_f16dma_splits(p, plan, &n);
    for (int i = 0; i < n; ++i) {
        int rc;
        switch (p.W) {
            case 8: rc = launch_w<8>(p, plan[i][2], plan[i][0], plan[i][1], stream); break;
            case 16: rc = launch_w<16>(p, plan[i][2], plan[i][0], plan[i][1], stream); break;
            case 32: rc = launch_w<32>(p, plan[i][2], plan[i][0], plan[i][1], stream); break;
            default: rc = launch_w<64>(p, plan[i][2], plan[i][0], plan[i][1], stream); break;
        }
        if (rc) return rc;
    }
    if (p.splits > 1) return launch_splitk_reduce_f16(p, stream);
    return DS_OK;
}

}  // namespace igemm
