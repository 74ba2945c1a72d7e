// 3x3 convolution on fp16 activations, HALF-SLAB / TWO-WORKGROUPS-PER-CU variant of conv3x3_f16dma.hip (round 4).
//
// Why: conv3x3_f16dma_kernel (eight waves, 256-pixel tiles, 160 KB of LDS) owns its CU alone, so everything it waits for is exposed: the
// per-tap barrier, the prologue, and above all the epilogue -- 20 - 36 % of a tile on the K <= 3 456 layers (profiles/r3_conv_f16dma_ablations.txt),
// a per-wave chain (residual rows -> LDS transpose -> rows -> stores) during which the matrix pipe of that CU is idle.  Here a workgroup
// is FOUR waves on a 128-pixel x (NB * 64)-channel tile (64 x NB*32 per wave, as before) and stages 32-channel HALF slabs, so that its LDS
// (two halo buffers + THREE weight buffers) fits 80 KB and TWO independent workgroups share a CU: each SIMD hosts one wave of each, and one
// workgroup's epilogue / barrier / prologue time is the other's K loop.
//
//   * halo of a half slab: NP pixels x 64 B (32 channels), pixel-major, the 16-B chunk index (0..3) XOR-swizzled by (pixel >> 2) & 3 -- the
//     same involution on the DMA source address and on the fragment read; any 16 lanes of a ds_read_b128 lane group then hit 16 different
//     bank quads, whatever the tap offset; out-of-image pixels fetch a zero page.  Two buffers, half slab s+1 streams in one 4-KB round per tap;
//   * weights of a (half slab, tap): NB * 64 rows x 64 B, swizzled the same way, THREE buffers, requested TWO taps ahead (a tap is 12 MFMAs
//     per wave, ~0.2 - 0.4 us: one tap of flight time does not cover an L2 round trip); the wait before a tap's barrier leaves exactly the
//     newest request group in flight (counted vmcnt);
//   * the packed weights are those of conv3x3_f16dma.hip ([cout_pad][K] halfs, K = (slab64 * 9 + tap) * 64 + c): the half slab h of slab s
//     is the 64-byte half of each 128-byte row segment, no repacking;
//   * per tap: two K steps of 16 channels, one barrier; fragment pipeline and fused epilogue (epilogue_pipe) as in the eight-wave kernel.
// Costs: every 128-pixel tile re-reads the layer's weights (twice the L2 -> LDS weight traffic of 256-pixel tiles) and the halo overlap
// of a 2-row tile of a 64-column image is 2x instead of 1.5x.  Which layers take this variant is decided per layer in
// conv3x3_f16dma.hip (conv3x3_f16dma_use_half) from measurements (profiles/r4_conv_f16dmah_ab.txt).
// Scope: as conv3x3_f16dma.hip, with M % 128 == 0 and c0 % 32 == 0 (the engines have c0 % 64 == 0).
#include "pipe_common.h"

namespace igemm {
namespace {

__device__ __attribute__((aligned(128))) _Float16 g_zero_halfs_h[64];      // zero-initialised: the row of an out-of-image pixel

template <int W>
struct GeoH {
    static constexpr int NIMG = (W * W >= 128) ? 1 : 128 / (W * W);        // image slots per tile (8x8 images: 2)
    static constexpr int TH = 128 / (W * NIMG), WP = W + 2, HP = TH + 2, NP = NIMG * HP * WP;
    static constexpr int NDMA = (NP * 4 + 255) / 256;                      // DMA rounds per halo (256 threads x 16 B = 4 KB each)
    static constexpr unsigned HALO_B = NDMA * 4096u;
};

template <int W, int NB>
constexpr unsigned f16dmah_smem() { return 3u * NB * 4096u + 2u * GeoH<W>::HALO_B; }

template <int W, int NB>
__global__ void __launch_bounds__(256, 2) conv3x3_f16dmah_kernel(const KParams p) {
    using G = GeoH<W>;
    constexpr int WP = G::WP, HP = G::HP, TH = G::TH, NIMG = G::NIMG, NP = G::NP, NDMA = G::NDMA;
    constexpr unsigned WB = NB * 4096u, HB = G::HALO_B;
    static_assert(f16dmah_smem<W, NB>() <= 80u * 1024u, "two workgroups per CU");
    static_assert(NDMA <= 8, "one halo round per tap");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);                 // [weights 0 | weights 1 | weights 2 | halo 0 | halo 1]
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt, 0)) return;
    const int m0 = mt * 128, n0 = p.n_begin + nt * (NB * 64);
    const _Float16* a0 = reinterpret_cast<const _Float16*>(p.a0);
    const _Float16* e0 = reinterpret_cast<const _Float16*>(p.e0);
    const _Float16* wgt = reinterpret_cast<const _Float16*>(p.b);
    const size_t ldbh = (size_t)p.ldb * 2;                     // weight row pitch in halfs

    const int img0 = m0 / p.HW;
    const int r0 = NIMG == 1 ? (m0 - img0 * p.HW) / W : 0;

    // ---- halo DMA: thread tid owns 16-B unit j * 256 + tid of round j: pixel (unit >> 2), LDS chunk slot tid & 3 -----------------
    int hpix[NDMA];                                            // source pixel (-1: zero page)
    // source channel offset (halfs): chunk slot ^ ((pixel >> 2) & 3); pixel = j * 64 + (tid >> 2), so the same for every round j
    const int hch = ((tid & 3) ^ ((tid >> 4) & 3)) * 8;
#pragma unroll
    for (int j = 0; j < NDMA; ++j) {
        const int hp = j * 64 + (tid >> 2);
        const int sl = hp / (HP * WP), rem = hp - sl * (HP * WP);
        const int hr = rem / WP, hc = rem - hr * WP;
        const int y = r0 + hr - 1, x = hc - 1;
        const bool ok = hp < NP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)W;
        hpix[j] = ok ? ((img0 + sl) * p.H + y) * W + x : -1;
    }
    const int nchunks = p.c0 / 32;                             // 3x3 half slabs (9 taps each)
    const int nextra = p.ec0 / 32;                             // appended 1x1 half slabs (centre tap only)
    const int NCH = nchunks + nextra;
    const int KT = nchunks * 9 + nextra;
    const int abl = p.coef_lds;                                 // timing ablations (ds_conv_args.tune.ablate; results are wrong when set)
    auto halo_dma = [&](int chunk, int hbuf, auto jc) {          // DMA round j of half slab `chunk` into halo buffer hbuf
        constexpr int j = decltype(jc)::value;
        const bool extra = chunk >= nchunks;
        const _Float16* base = extra ? e0 + (size_t)(chunk - nchunks) * 32 : a0 + (size_t)chunk * 32;
        const int ld = extra ? p.elda0 : p.lda0;
        const _Float16* g = hpix[j] >= 0 ? base + (size_t)hpix[j] * ld + hch : g_zero_halfs_h;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + 3 * WB + hbuf * HB + (j * 256 + wave * 64) * 16), 16, 0, 0);
    };
    // ---- weight DMA of K tile kt = (half slab, tap): rows i * 64 + (tid >> 2), i < NB; the source chunk is pre-swizzled ---------------
    // K offset (halfs) of K tile kt in the packed weights: 3x3 half slab c, tap t -> ((c >> 1) * 9 + t) * 64 + (c & 1) * 32; appended 1x1
    // half slab e -> (c0 / 64) * 9 * 64 + e * 32
    auto k_off = [&](int kt) -> size_t {
        if (kt < nchunks * 9) { const int c = kt / 9, t = kt - c * 9; return (size_t)(((c >> 1) * 9 + t) * 64 + (c & 1) * 32); }
        return (size_t)(nchunks / 2) * 9 * 64 + (size_t)(kt - nchunks * 9) * 32;
    };
    const _Float16* wsrc = wgt + (size_t)(n0 + (tid >> 2)) * ldbh + (((tid & 3) ^ ((tid >> 4) & 3)) * 8);
    auto w_dma = [&](int kt, int wbuf) {
        const size_t ko = k_off(kt);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (size_t)i * 64 * ldbh + ko),
                                             (lptr_t)(lds + wbuf * WB + (i * 256 + wave * 64) * 16), 16, 0, 0);
    };

    // ---- fragment addresses (LDS byte addresses relative to a halo buffer / a weight buffer) ------------------------------------------
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
        return hp * 64u + 16u * (((hp >> 2) & 3u) ^ gsel);
    };
    const int brow = wc * (NB * 32) + (lane & 31);
    const unsigned lds0 = lds_addr2(smem);
    const unsigned bbase = lds0 + (unsigned)brow * 64u + 16u * (unsigned)(((brow >> 2) & 3) ^ (lane >> 5));

    f32x16 accA[2][2], accB[2][2];                             // output columns [0, 64) and [64, 128) of the wave tile
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
        if constexpr (NB > 1) f.b1 = lds_rd<2048>(vb);          // weight rows + 32: 32 x 64 B (the swizzle term (row >> 2) & 3 is unchanged)
        if constexpr (NB > 2) f.b2 = lds_rd<4096>(vb);
        if constexpr (NB > 3) f.b3 = lds_rd<6144>(vb);
    };
    auto frag_wait = [&](Frag& f, auto nc) {                   // wait until at most N younger LDS operations are outstanding
        constexpr int N = decltype(nc)::value;
        if constexpr (NB == 1) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0) : "n"(N));
        if constexpr (NB == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1) : "n"(N));
        if constexpr (NB == 3) asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2) : "n"(N));
        if constexpr (NB == 4)
            asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(f.a0), "+v"(f.a1), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2), "+v"(f.b3) : "n"(N));
    };
#define DSH_MM(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a_), __builtin_bit_cast(h8, b_), acc_, 0, 0, 0)
    auto mfma_group = [&](Frag& f) {                           // consecutive MFMAs never touch the same accumulator
        DSH_MM(accA[0][0], f.a0, f.b0); DSH_MM(accA[1][0], f.a1, f.b0);
        if constexpr (NB > 1) { DSH_MM(accA[0][1], f.a0, f.b1); DSH_MM(accA[1][1], f.a1, f.b1); }
        if constexpr (NB > 2) { DSH_MM(accB[0][0], f.a0, f.b2); DSH_MM(accB[1][0], f.a1, f.b2); }
        if constexpr (NB > 3) { DSH_MM(accB[0][1], f.a0, f.b3); DSH_MM(accB[1][1], f.a1, f.b3); }
    };
    constexpr int NR = 2 + NB;                                 // LDS reads per fragment set
    const unsigned halo0 = lds0 + 3 * WB;

    // ---- prologue: halo of half slab 0, the part of half slab 1's halo that is due, weights of K tiles 0, 1, 2 -------------------------
    static_for<NDMA>([&](auto jc) { halo_dma(0, 0, jc); });
    if (NCH > 1) {
        if (nchunks == 0) static_for<NDMA>([&](auto jc) { halo_dma(1, 1, jc); });      // half slab 0 is a one-tap slab
        else halo_dma(1, 1, IC<0>{});
    }
    w_dma(0, 0);
    if (KT > 1) w_dma(1, 1);
    if (KT > 2) w_dma(2, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    Frag P_, Q_;
    {
        const unsigned ctr = nchunks > 0 ? 0u : 4u;            // first tap: (0, 0) of a 3x3 half slab, or the centre tap of a 1x1 one
        frag_read(P_, halo0 + a_addr(0, (int)ctr), halo0 + a_addr(1, (int)ctr), bbase);
    }

    int kt = 0;
    int wb = 0;                                                // weight buffer of K tile kt (= kt % 3)
    int young = 0;                                             // DMA instructions of the newest request group (issued behind the previous barrier)
    // One tap: T9 = tap of a 3x3 half slab (0..8) or 9 = the centre tap of an appended 1x1 half slab.  P holds the fragments of its K step 0
    // (read behind the previous tap's barrier).
    //   K step 0     : reads of step 1 in flight under the MFMAs of step 0
    //   then         : all reads of this tap done; every request group but the newest has landed (counted vmcnt); barrier: weight buffer
    //                  kt % 3 -- and, at a half slab's end, its halo buffer -- are free, the operands of tap kt + 1 are in LDS
    //   K step 1     : behind the barrier: the first fragment reads of tap kt + 1, the step's MFMAs, then -- in their shadow -- the DMA
    //                  requests of tap kt + 3's weights (into the buffer just freed) and of the next halo round
    // Halo schedule as in conv3x3_f16dma.hip: half slab s+1 lives in buffer (s+1) & 1, free once half slab s-1 is done; its DMA rounds are
    // issued one per barrier from the last tap of half slab s-1 on (all of them at once when half slab s is a one-tap slab).
    auto tap = [&](auto t9c, int chunk) {
        Frag &P = P_, &Q = Q_;
        constexpr int T9 = decltype(t9c)::value;
        constexpr bool X = (T9 == 9);
        constexpr int TT = X ? 4 : T9;
        constexpr bool SLAB_END = X || T9 == 8;
        const unsigned hoff = halo0 + (unsigned)(chunk & 1) * HB;
        const unsigned woff = (unsigned)wb * WB;
        const unsigned a_0 = a_addr(0, TT) + hoff, a_1 = a_addr(1, TT) + hoff;
        const unsigned vb = bbase + woff;
        frag_read(Q, a_0 ^ 32u, a_1 ^ 32u, vb ^ 32u);
        frag_wait(P, IC<NR>{});
        DS2_FENCE(); mfma_group(P); DS2_FENCE();
        frag_wait(Q, IC<0>{});
        // everything but the newest request group (tap kt + 2's weights [+ one halo round]) must have landed: the weights of tap kt + 1 were
        // requested two taps ago.  LDS-DMA requests complete in order, so "at most `young` outstanding" is exactly that.
        if (young == NB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
        else if (young == NB + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(abl & 16)) __builtin_amdgcn_s_barrier();         // (timing ablation: no per-tap barrier)
        DS2_FENCE();
        const int wb1 = wb == 2 ? 0 : wb + 1;
        if (kt + 1 < KT) {
            const unsigned nwoff = (unsigned)wb1 * WB;
            if constexpr (SLAB_END) {
                const unsigned nh = halo0 + (unsigned)((chunk + 1) & 1) * HB;
                const int nt9 = chunk + 1 >= nchunks ? 4 : 0;
                frag_read(P, nh + a_addr(0, nt9), nh + a_addr(1, nt9), bbase + nwoff);
            } else {
                frag_read(P, a_addr(0, T9 + 1) + hoff, a_addr(1, T9 + 1) + hoff, bbase + nwoff);
            }
        }
        DS2_FENCE(); mfma_group(Q); DS2_FENCE();
        int issued = 0;
        if (kt + 3 < KT) { w_dma(kt + 3, wb); issued += NB; }  // weights first, the halo round behind them (see the counted wait)
        if constexpr (SLAB_END) {
            if (chunk + 2 < NCH) {
                if (chunk + 1 >= nchunks) { static_for<NDMA>([&](auto jc) { halo_dma(chunk + 2, chunk & 1, jc); }); issued += NDMA; }   // next half slab has one tap
                else { halo_dma(chunk + 2, chunk & 1, IC<0>{}); issued += 1; }
            }
        } else if constexpr (T9 + 1 < NDMA) {
            if (chunk + 1 < NCH) { halo_dma(chunk + 1, (chunk + 1) & 1, IC<T9 + 1>{}); issued += 1; }
        }
        young = issued;
        DS2_FENCE();
        ++kt;
        wb = wb1;
    };
    int chunk = 0;
    for (; chunk < nchunks; ++chunk) {
        tap(IC<0>{}, chunk); tap(IC<1>{}, chunk); tap(IC<2>{}, chunk);
        tap(IC<3>{}, chunk); tap(IC<4>{}, chunk); tap(IC<5>{}, chunk);
        tap(IC<6>{}, chunk); tap(IC<7>{}, chunk); tap(IC<8>{}, chunk);
    }
    for (; chunk < NCH; ++chunk) tap(IC<9>{}, chunk);
#undef DSH_MM
    // (the last tap's wait was a full vmcnt(0) -- nothing is requested behind tap KT - 3 -- and no fragment read follows its barrier:
    // the LDS is free for the epilogue's staging rows)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    if (abl & 4) {                                             // no epilogue: every accumulator block (and so every MFMA) is kept alive
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { asm volatile("" :: "v"(accA[i][j])); asm volatile("" :: "v"(accB[i][j])); }
        return;
    }
    float* stage = smem + wave * 32 * EPI_LD;
    const int wn0 = n0 + wc * (NB * 32);
    epilogue_pipe<0, true, (NB == 1 ? 32 : 64), (NB == 3 ? 32 : (NB == 4 ? 64 : 0))>(p, accA, accB, stage, lane, m0 + wr * 64, wn0, p.out);
}

template <int W, int NB>
int launch_h_nb(KParams p, int n_begin, int ntiles, hipStream_t stream) {
    p.mtiles = p.M / 128;
    p.ntiles = ntiles;
    p.n_begin = n_begin;
    p.splits = 1;
    p.coef_lds = p.t_ablate;
    int smem = (int)f16dmah_smem<W, NB>();
    const int epi = 4 * 32 * EPI_LD * (int)sizeof(float);
    if (smem < epi) smem = epi;
    DS_ENSURE_DYN_LDS((&conv3x3_f16dmah_kernel<W, NB>), 80 * 1024);
    hipLaunchKernelGGL((conv3x3_f16dmah_kernel<W, NB>), dim3(grid_1d(p.mtiles, p.ntiles), 1), dim3(256), smem, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

template <int W>
int launch_h(const KParams& p, int nb, int n_begin, int ntiles, hipStream_t stream) {
    switch (nb) {
        case 1: return launch_h_nb<W, 1>(p, n_begin, ntiles, stream);
        case 2: return launch_h_nb<W, 2>(p, n_begin, ntiles, stream);
        case 3: return launch_h_nb<W, 3>(p, n_begin, ntiles, stream);
        default:
            if constexpr (f16dmah_smem<W, 4>() <= 80u * 1024u) return launch_h_nb<W, 4>(p, n_begin, ntiles, stream);
            else return DS_E_SHAPE;
    }
}

}  // namespace

// widest column tile whose three weight buffers fit 80 KB next to two halo buffers
int conv3x3_f16dmah_max_nb(int W) { return W == 64 ? 3 : 4; }

bool conv3x3_f16dmah_applicable(const KParams& p) {
    return conv3x3_f16dma_applicable(p) && p.M % 128 == 0;
}

int launch_conv3x3_f16dmah_tiles(const KParams& p, int nb, int n_begin, int ntiles, hipStream_t stream) {
    switch (p.W) {
        case 8: return launch_h<8>(p, nb, n_begin, ntiles, stream);
        case 16: return launch_h<16>(p, nb, n_begin, ntiles, stream);
        case 32: return launch_h<32>(p, nb, n_begin, ntiles, stream);
        default: return launch_h<64>(p, nb, n_begin, ntiles, stream);
    }
}

}  // namespace igemm
