// 3x3 convolution, LDS-halo implicit GEMM, second generation of the 256-pixel x 128-channel tile (gfx950, fp32 MFMA).
//
// What round 2 measured on conv3x3_halo_kernel<4> (profiles/r2a_*): the matrix pipe is busy 82 % of the time at 2.4 GHz, every wave
// is parked ~20 % of its life, and the losses are ADDITIVE pieces of serialised non-MFMA work -- the slab-boundary phase that converts
// (GroupNorm affine + SiLU) and publishes the next halo between two barriers (4.5 %), the clustered per-tap scalar/address work (part
// of a 7 % "tap structure" loss; a software-pipelined fragment read alone changed nothing), the weight-DMA issue (2 %), the epilogue
// (3.5-7.5 %).  This kernel keeps the tile shape and data layout of the first one and removes the serialisation:
//
//   * compile-time geometry (template W: 16 / 32 / 64 columns, one image per tile): every LDS address of the tap loop is a
//     per-lane base register plus an instruction immediate -- no per-tap address arithmetic;
//   * the 9 taps of a slab are unrolled with static tap index; fragment reads of K step g+1 are in flight under the MFMAs of step g
//     (asm-ordered reads, counted lgkmcnt);
//   * TWO halo buffers: the next slab's halo is converted and written while the current slab is multiplied, one float4 slot per tap,
//     each slot cut into small pieces that issue in the shadow of individual MFMAs (sched_barrier fences pin the interleave).  There
//     is no slab-boundary phase and no second barrier;
//   * raw halo loads are asm global loads issued one slab ahead with hand-counted vmcnt, so that neither they nor the LDS-DMA weight
//     stream is ever drained by a compiler-inserted vmcnt(0);
//   * the fused 1x1 skip-projection slabs (one tap each) run through the same pipeline.
// Template F16 = the reference's reduced-precision mode (networks_edm.py:486 `use_fp16`, sample.py:296 autocast): the SAME kernel with
// fp16 operands on v_mfma_f32_32x32x16_f16 (16x the fp32 matrix rate, fp32 accumulation).  Activations stay fp32 in HBM; the halo writer
// rounds them to fp16 (RNE, after the fp32 GroupNorm affine + SiLU) when it stages them, the weights are packed to fp16 once.  A slab is
// 64 channels, so every LDS byte offset of the fp32 kernel keeps its meaning (halo row = 64 halfs + pad = 144 B, weight row = 128 B,
// a 16-B slot = 8 channels) and a K step of 16 is four MFMAs.  With the matrix pipe 16x faster the kernel is bound by the halo
// conversion (VALU), the LDS fragment reads and the per-tile prologue/epilogue instead -- see DESIGN.md.
// Scope: taps == 9, stride 1, square power-of-two images: W in {16, 32, 64} with H*W a multiple of 256 (one image per tile), or 8x8
// images with four whole images per tile and no fused input normalisation; 128-column tiles, no split-K.  Everything else stays on
// conv3x3_halo.hip.
#include "pipe_common.h"

namespace igemm {
namespace {

__device__ __attribute__((aligned(16))) float g_zero_page2[64];                                                     // zero-initialised
__device__ __attribute__((aligned(16))) float g_ident_page2[16] = {0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0};  // {mu, A, B} = identity affine

template <int W>
struct Geo2 {
    static constexpr int T = 512;
    static constexpr int NIMG = (W * W >= 256) ? 1 : 256 / (W * W);      // image slots per tile (8x8 images: 4)
    static constexpr int TH = 256 / (W * NIMG), WP = W + 2, HP = TH + 2, NP = NIMG * HP * WP;
    static constexpr int NS = (NP * 8 + T - 1) / T;          // float4 halo slots per thread (6 or 7)
    static constexpr unsigned ROW = WP * 144;                // bytes per halo row (36 floats per pixel)
    static constexpr unsigned HALO_B = NP * 144;             // bytes per halo buffer
    static constexpr unsigned BS_B = 2 * 128 * 32 * 4;       // two weight buffers [128][32] floats (LDS-DMA image, unpadded)
    static constexpr unsigned SMEM = BS_B + 2 * HALO_B;
    static_assert(NS <= 7, "halo slots");
    static_assert(6 * 9216 + 15 < 65536 && 2 * ROW + 288 + 96 < 65536, "immediates");
};

template <int W, int MODE>
__global__ void __launch_bounds__(512, 2) conv3x3_halo2_kernel(const KParams p) {
    constexpr bool F16 = (MODE == 1), SPLIT = (MODE == 2);          // MODE 0: fp32 operands (exact fp32 MFMA)
    using G = Geo2<W>;
    constexpr int T = G::T, WP = G::WP, HP = G::HP, TH = G::TH, NIMG = G::NIMG, NP = G::NP, NS = G::NS;
    constexpr unsigned ROW = G::ROW, HALO_B = G::HALO_B, BS_B = G::BS_B;
    constexpr int BKC = F16 ? 64 : 32;            // channels per slab
    constexpr int H = F16 ? 2 : 1;                // float4 loads per 16-B halo slot (8 or 4 channels)
    constexpr int NLOAD = (NS + 3) * H;           // asm global loads per slab (raw slots + the three coefficient vectors)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lds0 = lds_addr2(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt, 0)) return;
    const int m0 = mt * 256, n0 = nt * 128;
    const int ld_row = tid >> 3, ld_col = (tid & 7) * 4 * H;
    const float* zero = g_zero_page2;
    const float* ident = g_ident_page2;

    const int img0 = m0 / p.HW;
    const int r0 = NIMG == 1 ? (m0 - img0 * p.HW) / W : 0;

    // ---- per-thread halo slots (fixed for the whole K loop) ---------------------------------------------------------------------
    int h_pix[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int hp = (tid >> 3) + j * 64;
        const int sl = hp / (HP * WP), rem = hp - sl * (HP * WP);         // image slot of the tile, pixel inside its halo
        const int hr = rem / WP, hc = rem - hr * WP;
        const int y = r0 + hr - 1, x = hc - 1;
        const bool ok = hp < NP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)W;
        h_pix[j] = ok ? ((img0 + sl) * p.H + y) * W + x : -1;
    }
    const bool last_slot_valid = (tid >> 3) + (NS - 1) * 64 < NP;
    const unsigned st_base = lds0 + BS_B + (unsigned)(tid >> 3) * 144 + (unsigned)(tid & 7) * (SPLIT ? 8 : 16);   // + j * 9216 (+ HALO_B)

    // ---- fragment addresses -----------------------------------------------------------------------------------------------------
    unsigned abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wr * 64 + i * 32 + (lane & 31);
        const int sl = m / (TH * W), rem = m - sl * (TH * W);
        const int r = rem / W, c = rem - r * W;
        abase[i] = lds0 + BS_B + (unsigned)(((sl * HP + r) * WP + c) * 144) + (unsigned)(lane >> 5) * 16;   // halo buffer 0, tap (0,0), ks 0
    }
    const int b_row = wc * 64 + (lane & 31);
    const unsigned c0 = (unsigned)((lane >> 5) ^ (((lane & 31) >> 1) & 7));
    unsigned bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[ks] = lds0 + (unsigned)b_row * 128 + ((c0 ^ (2u * ks)) * 16);

    // ---- weight DMA: this thread's two source rows (16-B chunk pre-swizzled), wave-uniform LDS destinations ---------------------
    const float* bsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
        bsrc[i] = p.b + (size_t)(n0 + ld_row + 64 * i) * p.ldb + (((tid & 7) ^ ((ld_row >> 1) & 7)) * 4);
    auto b_dma = [&](int kt, int buf) {
        DS_RACE_SKEW(wave);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* dst = smem + buf * 4096 + (wave * 8 + 64 * i) * 32;
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i] + (size_t)kt * 32), (lptr_t)(dst), 16, 0, 0);
        }
    };

    // ---- slab bookkeeping -------------------------------------------------------------------------------------------------------
    const int nchunks = (p.c0 + p.c1) / BKC;         // 3x3 slabs (9 taps)
    const int nextra = (p.ec0 + p.ec1) / BKC;        // appended 1x1 slabs (centre tap only)
    const int NCH = nchunks + nextra;
    const int KT = nchunks * 9 + nextra;
    const int Ctot = p.c0 + p.c1;
    const bool silu = p.norm_act == DS_ACT_SILU;

    f32x4 hreg[NS][H];                                // raw halo of the slab that is converted next
    f32x4 cmu[H], cga[H], cbe[H];                     // its {mu, A, B} vectors (identity when there is nothing to normalise)
    // asm global loads of slab `chunk` (clamped to the last slab: the loads are unconditional so that their count is static)
    auto slab_src = [&](int chunk, const float*& src, int& ld) {
        const bool extra = chunk >= nchunks;
        const int c = (extra ? chunk - nchunks : chunk) * BKC;
        const int cc0 = extra ? p.ec0 : p.c0;
        const bool first = c < cc0;
        const float* s0 = extra ? p.e0 : p.a0;
        const float* s1 = extra ? p.e1 : p.a1;
        src = first ? s0 + c + ld_col : s1 + (c - cc0) + ld_col;
        ld = first ? (extra ? p.elda0 : p.lda0) : (extra ? p.elda1 : p.lda1);
    };
    const float* nsrc = zero; int nld = 0;            // source of the raw loads in progress (set once per slab by load_coefs)
    auto load_slot = [&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float* ptr = h_pix[j] >= 0 ? nsrc + (size_t)h_pix[j] * nld : zero;
#pragma unroll
        for (int h = 0; h < H; ++h) hreg[j][h] = gld16(ptr + 4 * h);
    };
    auto load_coefs = [&](int chunk) {
        const int ch = min(chunk, NCH - 1);
        slab_src(ch, nsrc, nld);
        const bool on = p.norm != nullptr && ch < nchunks;
        const float* cp = on ? p.norm + (size_t)img0 * 3 * Ctot + ch * BKC + ld_col : ident;
        const int st = on ? Ctot : 4;
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const int o = on ? 4 * h : 0;
            cmu[h] = gld16(cp + o);
            cga[h] = gld16(cp + st + o);
            cbe[h] = gld16(cp + 2 * st + o);
        }
    };
    // conversion of one element of slot j (GroupNorm affine + SiLU, networks_edm.py:160,167; fp16: then RNE rounding, two
    // elements per dword), then the 16-B store of the slot
    f32x4 cvt;
    float cvt_even = 0.f, cvt_even_lo = 0.f;
    auto convert_elem = [&](auto jc, auto ec, bool act) {
        constexpr int j = decltype(jc)::value, e = decltype(ec)::value;
        float v = fmaf(hreg[j][e >> 2][e & 3] - cmu[e >> 2][e & 3], cga[e >> 2][e & 3], cbe[e >> 2][e & 3]);
        if (act) v = ds_silu(v);
        v = h_pix[j] >= 0 ? v : 0.f;
        if constexpr (SPLIT) {
            const float hi = (float)(_Float16)v;
            const float lo = v - hi;                     // exact in fp32
            if constexpr ((e & 1) == 0) { cvt_even = hi; cvt_even_lo = lo; }
            else { cvt[e >> 1] = pack_h2(cvt_even, hi); cvt[2 + (e >> 1)] = pack_h2(cvt_even_lo, lo); }
        } else if constexpr (!F16) cvt[e] = v;
        else if constexpr ((e & 1) == 0) cvt_even = v;
        else cvt[e >> 1] = pack_h2(cvt_even, v);
    };
    auto store_slot = [&](auto jc, unsigned st_addr) {
        constexpr int j = decltype(jc)::value;
        DS_RACE_SKEW(wave);
        if (j < NS - 1 || last_slot_valid) {
            if constexpr (SPLIT) {
                const f32x2 hi = {cvt[0], cvt[1]}, lo = {cvt[2], cvt[3]};
                lds_wr64<j * 9216>(st_addr, hi);          // channels ld_col .. ld_col+3 of the hi half-row
                lds_wr64<j * 9216 + 64>(st_addr, lo);     // and of the lo half-row
            } else {
                lds_wr<j * 9216>(st_addr, cvt);
            }
        }
    };
    constexpr int NE = 4 * H;                         // elements (channels) per slot
    // "+v" pseudo-uses that tie the asm-loaded registers to the point where their data has landed (asm operands inside lambdas do
    // not capture: bind references first)
    auto touch_slot = [&](auto jc) {
#pragma unroll
        for (int h = 0; h < H; ++h) { f32x4& r = hreg[decltype(jc)::value][h]; asm volatile("" : "+v"(r)); }
    };
    auto touch_coefs = [&]() {
#pragma unroll
        for (int h = 0; h < H; ++h) { f32x4 &m_ = cmu[h], &a_ = cga[h], &b_ = cbe[h]; asm volatile("" : "+v"(m_), "+v"(a_), "+v"(b_)); }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: first halo (exposed, once per tile), raw data of the second slab, first two weight tiles --------------------
    b_dma(0, 0);
    if (KT > 1) b_dma(1, 1);
    {
        load_coefs(0);
        static_for<NS>([&](auto jc) { load_slot(jc); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        touch_coefs();
        const bool act0 = silu && p.norm != nullptr && nchunks > 0;
        static_for<NS>([&](auto jc) {
            touch_slot(jc);
            static_for<NE>([&](auto ec) { convert_elem(jc, ec, act0); });
            store_slot(jc, st_base);
        });
        load_coefs(1);
        static_for<NS>([&](auto jc) { load_slot(jc); });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // first halo and both weight tiles are in LDS (vmcnt(0) above)

    int kt = 0;                      // global tap counter = weight K tile
    unsigned hb = 0;                 // byte offset of the CURRENT halo buffer (0 or HALO_B); the other one is being filled
    const unsigned first_off = nchunks > 0 ? 0u : ROW + 144u;
    if constexpr (!SPLIT) {
    Frag2 P_, Q_;
    frag_read2<0>(P_, abase[0] + first_off, abase[1] + first_off, bq[0]);

#define DS2_M(i, j, r, f) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32((f).a##i[r], (f).b##j[r], acc[i][j], 0, 0, 0)
#define DS2_MH(i, j, f) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (f).a##i), __builtin_bit_cast(h8, (f).b##j), acc[i][j], 0, 0, 0)
    // The MFMAs of one K step (fp32: 16 x 32x32x2, K = 8 per step; fp16: 4 x 32x32x16, K = 16 per step) with a hook after each of
    // them; hook(IC<k>) emits its own sched_barrier fences when it does anything.
#define DS2_GROUP(f, hook)                                                                                                        \
    if constexpr (F16) {                                                                                                          \
        DS2_MH(0, 0, f); hook(IC<0>{}); DS2_MH(0, 1, f); hook(IC<1>{}); DS2_MH(1, 0, f); hook(IC<2>{}); DS2_MH(1, 1, f); hook(IC<3>{});  \
    } else {                                                                                                                      \
    DS2_M(0, 0, 0, f); hook(IC<0>{});  DS2_M(0, 1, 0, f); hook(IC<1>{});  DS2_M(1, 0, 0, f); hook(IC<2>{});  DS2_M(1, 1, 0, f); hook(IC<3>{});   \
    DS2_M(0, 0, 1, f); hook(IC<4>{});  DS2_M(0, 1, 1, f); hook(IC<5>{});  DS2_M(1, 0, 1, f); hook(IC<6>{});  DS2_M(1, 1, 1, f); hook(IC<7>{});   \
    DS2_M(0, 0, 2, f); hook(IC<8>{});  DS2_M(0, 1, 2, f); hook(IC<9>{});  DS2_M(1, 0, 2, f); hook(IC<10>{}); DS2_M(1, 1, 2, f); hook(IC<11>{});  \
    DS2_M(0, 0, 3, f); hook(IC<12>{}); DS2_M(0, 1, 3, f); hook(IC<13>{}); DS2_M(1, 0, 3, f); hook(IC<14>{}); DS2_M(1, 1, 3, f); hook(IC<15>{});  \
    }

    // One tap.  T9 = tap index inside a 3x3 slab (0..8) or 9 = the single centre tap of a 1x1 slab.
    //   K steps 0..2 : fragments double-buffered in P / Q; hooks convert halo slots of the NEXT slab (3x3 slab: slot T9-1 during
    //                  tap T9 = 1..NS; 1x1 slab: all NS slots, as late in the tap as they fit so that the raw loads issued one tap
    //                  earlier have landed)
    //   then         : lgkmcnt(0) + vmcnt(weights of tap kt+1 landed) + barrier
    //   K step 3     : hooks issue the weight DMA of tap kt+2, the first fragment reads of tap kt+1 and -- on tap 7 of a 3x3 slab
    //                  and on every 1x1 slab -- the raw loads of the slab after next (NLOAD asm loads, the newest VMEM operations
    //                  of the wave, so the barrier of tap 8 waits with vmcnt(NLOAD) and leaves them in flight)
    // Conversion work is cut into NE + 1 steps per slot (one element each, then the store).  fp32: one step per hook position (the
    // 64-cycle shadow of one MFMA hides a step); fp16: the MFMAs are 32 cycles and four per K step, so steps are grouped SPP per
    // position and overlap with the OTHER wave of the SIMD rather than with this wave's own MFMAs.
    auto tap = [&](auto t9c, int chunk) {
        Frag2 &P = P_, &Q = Q_;                                         // (asm operands do not capture: bind references first)
        constexpr int T9 = decltype(t9c)::value;
        constexpr bool X = (T9 == 9);                                   // 1x1 slab
        constexpr int TY = X ? 1 : T9 / 3, TX = X ? 1 : T9 % 3;
        constexpr int AOFF = TY * (int)ROW + TX * 144;
        constexpr bool SLAB_END = X || T9 == 8;
        constexpr bool LOADS = X || T9 == 7;
        constexpr int PG = F16 ? 4 : 16;                                // hook positions per K step
        constexpr int SE = NE + 1;                                      // conversion steps per slot
        // positions available for conversion in K steps 0..2 and how many steps each takes
        constexpr int NPOS = X ? (F16 ? 12 : 40) : (F16 ? 12 : 8);
        constexpr int NSTEP = (X ? NS : 1) * SE;
        constexpr int SPP = (NSTEP + NPOS - 1) / NPOS;
        const unsigned va0 = abase[0] + hb, va1 = abase[1] + hb;
        const unsigned cb = (unsigned)(kt & 1) * 16384u;
        const unsigned st_addr = st_base + (hb ^ HALO_B);               // the buffer being filled (hb is 0 or HALO_B: the xor is a select)
        const bool conv_act = silu && p.norm != nullptr && (chunk + 1) < nchunks;
        auto conv_step = [&](auto hc) {                                 // h-th conversion step of this tap
            constexpr int h = decltype(hc)::value;
            constexpr int j = X ? h / SE : T9 - 1, s = X ? h % SE : h;
            if constexpr (h >= 0 && h < NSTEP && j >= 0 && j < NS) {
                if constexpr (h == 0 || (!X && s == 0 && j == 0)) {
                    if constexpr (X) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    touch_coefs();
                }
                if constexpr (s == 0) touch_slot(IC<j>{});
                if constexpr (s < NE) convert_elem(IC<j>{}, IC<s>{}, conv_act);
                else store_slot(IC<j>{}, st_addr);
            }
        };
        auto conv_pos = [&](auto qc) {                                  // q-th conversion position of this tap
            constexpr int q = decltype(qc)::value;
            constexpr int first = X ? NSTEP - (NPOS - q) * SPP : q * SPP;      // 1x1 slabs: packed towards the END of the tap
            if constexpr (q >= 0 && q < NPOS && first + SPP > 0 && first < NSTEP) {
                DS2_FENCE();
                static_for<SPP>([&](auto ic) { conv_step(IC<first + decltype(ic)::value>{}); });
                DS2_FENCE();
            }
        };
        // hook position -> conversion position.  fp32 3x3 slab: the odd positions of K step 0; fp32 1x1 slab: from the 9th MFMA
        // of K step 0 on; fp16: every position of K steps 0..2.
        auto hookA = [&](auto kc) { constexpr int k = decltype(kc)::value;
            if constexpr (F16) conv_pos(IC<k>{}); else if constexpr (X) conv_pos(IC<k - 8>{}); else if constexpr (k % 2 == 1) conv_pos(IC<k / 2>{}); };
        auto hookB = [&](auto kc) { constexpr int k = decltype(kc)::value;
            if constexpr (F16) conv_pos(IC<PG + k>{}); else if constexpr (X) conv_pos(IC<k + 8>{}); };
        auto hookC = [&](auto kc) { constexpr int k = decltype(kc)::value;
            if constexpr (F16) conv_pos(IC<2 * PG + k>{}); else if constexpr (X) conv_pos(IC<k + 24>{}); };

        frag_read2<AOFF + 32>(Q, va0, va1, bq[1] + cb);
        DS2_FRAG_WAIT(4, P);
        DS2_GROUP(P, hookA)
        DS2_FENCE();
        frag_read2<AOFF + 64>(P, va0, va1, bq[2] + cb);
        DS2_FRAG_WAIT(4, Q);
        DS2_GROUP(Q, hookB)
        DS2_FENCE();
        frag_read2<AOFF + 96>(Q, va0, va1, bq[3] + cb);
        DS2_FRAG_WAIT(4, P);
        DS2_GROUP(P, hookC)
        DS2_FENCE();
        DS2_FRAG_WAIT(0, Q);
        if constexpr (!X && T9 == 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // raw barrier: every LDS access of this loop is volatile asm with hand-placed waits (above); __syncthreads() would add a
        // vmcnt(0) of its own and drain the raw-halo loads that are meant to stay in flight across the barrier of tap 8
        __builtin_amdgcn_s_barrier();
        // ---- after the barrier: buffer kt & 1 and (at a slab end) the current halo are dead ------------------------------------
        if constexpr (SLAB_END) hb ^= HALO_B;
        const bool next_is_x = SLAB_END ? (chunk + 1 >= nchunks) : false;
        auto issue_dma = [&]() {
            DS2_FENCE();
            if (kt + 2 < KT) b_dma(kt + 2, kt & 1);
            DS2_FENCE();
        };
        auto issue_prefetch = [&]() {
            DS2_FENCE();
            const unsigned nb = (unsigned)((kt + 1) & 1) * 16384u;
            if constexpr (SLAB_END) {
                const unsigned off = next_is_x ? ROW + 144u : 0u;
                frag_read2<0>(P, abase[0] + hb + off, abase[1] + hb + off, bq[0] + nb);
            } else {
                constexpr int NY = (T9 + 1) / 3, NX = (T9 + 1) % 3;
                frag_read2<NY * (int)ROW + NX * 144>(P, va0, va1, bq[0] + nb);
            }
            DS2_FENCE();
        };
        auto hookD = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k == 0) issue_dma();
            else if constexpr (k == 1) issue_prefetch();
            else if constexpr (LOADS && !F16 && k >= 2 && k < 2 + NS + 1) {
                DS2_FENCE();
                if constexpr (k == 2) load_coefs(chunk + 2);
                else load_slot(IC<k - 3>{});
                DS2_FENCE();
            } else if constexpr (LOADS && F16 && k == 2) {
                DS2_FENCE();
                load_coefs(chunk + 2);
                static_for<NS>([&](auto jc) { load_slot(jc); });
                DS2_FENCE();
            }
        };
        DS2_GROUP(Q, hookD)
        DS2_FENCE();
        ++kt;
    };

    int chunk = 0;
    for (; chunk < nchunks; ++chunk) {
        tap(IC<0>{}, chunk); tap(IC<1>{}, chunk); tap(IC<2>{}, chunk);
        tap(IC<3>{}, chunk); tap(IC<4>{}, chunk); tap(IC<5>{}, chunk);
        tap(IC<6>{}, chunk); tap(IC<7>{}, chunk); tap(IC<8>{}, chunk);
    }
    for (; chunk < NCH; ++chunk) tap(IC<9>{}, chunk);
#undef DS2_M
#undef DS2_MH
#undef DS2_GROUP
    DS2_FRAG_WAIT(0, P_);                             // drain the last (discarded) fragment prefetch

    } else {
    Frag2S P_, Q_;
    frag_read2s<0>(P_, abase[0] + first_off, abase[1] + first_off, bq[0], bq[2]);
#define DS2_MS(i, j, x, y, f) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (f).a##i##x), __builtin_bit_cast(h8, (f).b##j##y), acc[i][j], 0, 0, 0)
    // 12 MFMAs of one K step of 16 channels: hi*hi, hi*lo, lo*hi for the 2 x 2 tiles (the accumulators rotate, so that consecutive
    // MFMAs never depend on each other), a hook after each
#define DS2_GROUP_S(f, hook)                                                                                                      \
    DS2_MS(0, 0, h, h, f); hook(IC<0>{}); DS2_MS(0, 1, h, h, f); hook(IC<1>{}); DS2_MS(1, 0, h, h, f); hook(IC<2>{}); DS2_MS(1, 1, h, h, f); hook(IC<3>{});   \
    DS2_MS(0, 0, h, l, f); hook(IC<4>{}); DS2_MS(0, 1, h, l, f); hook(IC<5>{}); DS2_MS(1, 0, h, l, f); hook(IC<6>{}); DS2_MS(1, 1, h, l, f); hook(IC<7>{});   \
    DS2_MS(0, 0, l, h, f); hook(IC<8>{}); DS2_MS(0, 1, l, h, f); hook(IC<9>{}); DS2_MS(1, 0, l, h, f); hook(IC<10>{}); DS2_MS(1, 1, l, h, f); hook(IC<11>{});
    // One tap of the split mode = two K steps of 16 channels.  Step 0 (fragments P, prefetched after the previous barrier) carries
    // the halo-conversion hooks; then lgkmcnt(0) + vmcnt + barrier; step 1 (fragments Q) carries the weight DMA of tap kt+2, the
    // fragment prefetch of tap kt+1 and the raw loads, exactly as the other modes' last K step does.
    auto tap = [&](auto t9c, int chunk) {
        Frag2S &P = P_, &Q = Q_;
        constexpr int T9 = decltype(t9c)::value;
        constexpr bool X = (T9 == 9);
        constexpr int TY = X ? 1 : T9 / 3, TX = X ? 1 : T9 % 3;
        constexpr int AOFF = TY * (int)ROW + TX * 144;
        constexpr bool SLAB_END = X || T9 == 8;
        constexpr bool LOADS = X || T9 == 7;
        constexpr int SE = NE + 1, NPOS = 12;
        constexpr int NSTEP = (X ? NS : 1) * SE;
        constexpr int SPP = (NSTEP + NPOS - 1) / NPOS;
        const unsigned va0 = abase[0] + hb, va1 = abase[1] + hb;
        const unsigned cb = (unsigned)(kt & 1) * 16384u;
        const unsigned st_addr = st_base + (hb ^ HALO_B);
        const bool conv_act = silu && p.norm != nullptr && (chunk + 1) < nchunks;
        auto conv_step = [&](auto hc) {
            constexpr int h = decltype(hc)::value;
            constexpr int j = X ? h / SE : T9 - 1, s = X ? h % SE : h;
            if constexpr (h >= 0 && h < NSTEP && j >= 0 && j < NS) {
                if constexpr (h == 0 || (!X && s == 0 && j == 0)) {
                    if constexpr (X) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    touch_coefs();
                }
                if constexpr (s == 0) touch_slot(IC<j>{});
                if constexpr (s < NE) convert_elem(IC<j>{}, IC<s>{}, conv_act);
                else store_slot(IC<j>{}, st_addr);
            }
        };
        auto hookA = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr int first = X ? NSTEP - (NPOS - q) * SPP : q * SPP;
            if constexpr (first + SPP > 0 && first < NSTEP) {
                DS2_FENCE();
                static_for<SPP>([&](auto ic) { conv_step(IC<first + decltype(ic)::value>{}); });
                DS2_FENCE();
            }
        };
        DS2_FRAG_WAIT_S(0, P);
        frag_read2s<AOFF + 32>(Q, va0, va1, bq[1] + cb, bq[3] + cb);
        DS2_GROUP_S(P, hookA)
        DS2_FENCE();
        DS2_FRAG_WAIT_S(0, Q);
        if constexpr (!X && T9 == 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (SLAB_END) hb ^= HALO_B;
        const bool next_is_x = SLAB_END ? (chunk + 1 >= nchunks) : false;
        auto hookD = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k == 0) {
                DS2_FENCE();
                if (kt + 2 < KT) b_dma(kt + 2, kt & 1);
                DS2_FENCE();
            } else if constexpr (k == 1) {
                DS2_FENCE();
                const unsigned nb = (unsigned)((kt + 1) & 1) * 16384u;
                if constexpr (SLAB_END) {
                    const unsigned off = next_is_x ? ROW + 144u : 0u;
                    frag_read2s<0>(P, abase[0] + hb + off, abase[1] + hb + off, bq[0] + nb, bq[2] + nb);
                } else {
                    constexpr int NY = (T9 + 1) / 3, NX = (T9 + 1) % 3;
                    frag_read2s<NY * (int)ROW + NX * 144>(P, va0, va1, bq[0] + nb, bq[2] + nb);
                }
                DS2_FENCE();
            } else if constexpr (LOADS && k >= 2 && k < 2 + NS + 1) {
                DS2_FENCE();
                if constexpr (k == 2) load_coefs(chunk + 2);
                else load_slot(IC<k - 3>{});
                DS2_FENCE();
            }
        };
        DS2_GROUP_S(Q, hookD)
        DS2_FENCE();
        ++kt;
    };
    int chunk = 0;
    for (; chunk < nchunks; ++chunk) {
        tap(IC<0>{}, chunk); tap(IC<1>{}, chunk); tap(IC<2>{}, chunk);
        tap(IC<3>{}, chunk); tap(IC<4>{}, chunk); tap(IC<5>{}, chunk);
        tap(IC<6>{}, chunk); tap(IC<7>{}, chunk); tap(IC<8>{}, chunk);
    }
    for (; chunk < NCH; ++chunk) tap(IC<9>{}, chunk);
#undef DS2_MS
#undef DS2_GROUP_S
    DS2_FRAG_WAIT_S(0, P_);                           // drain the last (discarded) fragment prefetch
    }

    // drain the unconditional raw loads of the clamped "slab after next"
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    touch_coefs();
    static_for<NS>([&](auto jc) { touch_slot(jc); });
    __syncthreads();
    epilogue<0, true>(p, acc, smem + wave * 32 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 64, p.out);
}

template <int W, int MODE>
int launch_halo2_w(KParams p, int wide, hipStream_t stream) {
    using G = Geo2<W>;
    p.TH = G::TH; p.nimg = G::NIMG; p.HP = G::HP; p.WP = G::WP; p.NP = G::NP;
    p.mtiles = p.M / 256;
    p.ntiles = wide;
    p.n_begin = 0;
    p.splits = 1;
    int smem = (int)G::SMEM;
    const int epi = 8 * 32 * EPI_LD * (int)sizeof(float);
    if (smem < epi) smem = epi;
    DS_ENSURE_DYN_LDS((&conv3x3_halo2_kernel<W, MODE>), 160 * 1024);
    hipLaunchKernelGGL((conv3x3_halo2_kernel<W, MODE>), dim3(grid_1d(p.mtiles, p.ntiles), 1), dim3(512), smem, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace

// Layers this kernel takes: see the header comment.  `wide` = number of 128-column tiles the caller wants from it (fp32: the full
// tiles, a ragged 64-column tail goes to the first-generation kernel; fp16: ALL tiles, a ragged last tile multiplies the zero rows the
// weight packing pads to 128 and its epilogue guards the columns).  `f16`: the weights at p.b are fp16 in the 64-channel K order.
bool conv3x3_halo2_applicable(const KParams& p, int wide, int mode) {
#ifndef DS_BUILD_EXPERIMENTS
    if (mode != 2) return false;            // modes 0 / 1 are A/B records (docs/HISTORY.md B, D): built with DS_BUILD_EXPERIMENTS=1 only
#endif
    const int bkc = mode == 1 ? 64 : 32;
    if (p.taps != 9 || wide < 1) return false;
    if (!(p.W == 8 || p.W == 16 || p.W == 32 || p.W == 64)) return false;
    if (p.HW != p.H * p.W || p.M % 256) return false;
    if (p.W == 8 ? (p.H != 8 || p.norm != nullptr) : (p.HW % 256 != 0)) return false;   // 8x8: four whole images per tile, and no fused
                                                                                       // input normalisation (its planes are per image)
    if ((p.c0 + p.c1) % bkc || (p.ec0 + p.ec1) % bkc || (p.c1 > 0 && p.c0 % bkc) || (p.ec1 > 0 && p.ec0 % bkc)) return false;
    if ((p.c0 + p.c1) == 0) return false;
    if (p.nrows_b < wide * 128) return false;                 // every weight row of the tiles exists (rows are padded to 128)
    return true;
}

template <int MODE>
static int launch_mode(KParams& p, int wide, hipStream_t stream) {
    switch (p.W) {
        case 8: return launch_halo2_w<8, MODE>(p, wide, stream);
        case 16: return launch_halo2_w<16, MODE>(p, wide, stream);
        case 32: return launch_halo2_w<32, MODE>(p, wide, stream);
        default: return launch_halo2_w<64, MODE>(p, wide, stream);
    }
}

// mode: 0 = fp32 operands, 1 = fp16 operands (64-channel slabs), 2 = split fp16 hi/lo operands (fp32-emulated, 32-channel slabs)
int launch_conv3x3_halo2(KParams& p, int wide, int mode, hipStream_t stream) {
    if (mode == 2) return launch_mode<2>(p, wide, stream);
#ifdef DS_BUILD_EXPERIMENTS
    if (mode == 1) return launch_mode<1>(p, wide, stream);
    return launch_mode<0>(p, wide, stream);
#else
    return DS_E_SHAPE;
#endif
}

}  // namespace igemm
