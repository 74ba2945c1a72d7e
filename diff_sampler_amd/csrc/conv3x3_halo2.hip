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
// Scope: taps == 9, stride 1, H*W a multiple of 256 and W in {16, 32, 64} (the 16x16, 32x32 and 64x64 layers at any batch), full
// 128-column tiles, no split-K.  Everything else stays on conv3x3_halo.hip.
#include <type_traits>

#include "igemm_common.h"

namespace igemm {
namespace {

__device__ __attribute__((aligned(16))) float g_zero_page2[64];                                                     // zero-initialised
__device__ __attribute__((aligned(16))) float g_ident_page2[16] = {0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0};  // {mu, A, B} = identity affine

struct Frag2 { f32x4 a0, a1, b0, b1; };
template <int K> using IC = std::integral_constant<int, K>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int OFF>
__device__ __forceinline__ f32x4 lds_rd(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_wr(unsigned addr, const f32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ f32x4 gld16(const float* ptr) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ unsigned lds_addr2(const float* p) {
    typedef __attribute__((address_space(3))) const void* lcptr_t;
    return (unsigned)(size_t)(lcptr_t)(p);
}
#define DS2_FRAG_WAIT(N, f) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"((f).a0), "+v"((f).a1), "+v"((f).b0), "+v"((f).b1))
#define DS2_FENCE() __builtin_amdgcn_sched_barrier(0)

// A operand: two 32-row MFMA tiles per wave, base address + immediate; B operand: swizzled 16-B chunk of the LDS-DMA weight image
template <int AOFF>
__device__ __forceinline__ void frag_read2(Frag2& f, unsigned va0, unsigned va1, unsigned vb) {
    f.a0 = lds_rd<AOFF>(va0);
    f.a1 = lds_rd<AOFF>(va1);
    f.b0 = lds_rd<0>(vb);
    f.b1 = lds_rd<4096>(vb);
}

template <int W>
struct Geo2 {
    static constexpr int T = 512, TH = 256 / W, WP = W + 2, HP = TH + 2, NP = HP * WP;
    static constexpr int NS = (NP * 8 + T - 1) / T;          // float4 halo slots per thread (6 or 7)
    static constexpr unsigned ROW = WP * 144;                // bytes per halo row (36 floats per pixel)
    static constexpr unsigned HALO_B = NP * 144;             // bytes per halo buffer
    static constexpr unsigned BS_B = 2 * 128 * 32 * 4;       // two weight buffers [128][32] floats (LDS-DMA image, unpadded)
    static constexpr unsigned SMEM = BS_B + 2 * HALO_B;
    static_assert(NS <= 7, "halo slots");
    static_assert(6 * 9216 + 15 < 65536 && 2 * ROW + 288 + 96 < 65536, "immediates");
};

template <int W>
__global__ void __launch_bounds__(512, 2) conv3x3_halo2_kernel(const KParams p) {
    using G = Geo2<W>;
    constexpr int T = G::T, WP = G::WP, NP = G::NP, NS = G::NS;
    constexpr unsigned ROW = G::ROW, HALO_B = G::HALO_B, BS_B = G::BS_B;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lds0 = lds_addr2(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt, 0)) return;
    const int m0 = mt * 256, n0 = nt * 128;
    const int ld_row = tid >> 3, ld_col = (tid & 7) * 4;
    const float* zero = g_zero_page2;
    const float* ident = g_ident_page2;

    const int img0 = m0 / p.HW;
    const int r0 = (m0 - img0 * p.HW) / W;

    // ---- per-thread halo slots (fixed for the whole K loop) ---------------------------------------------------------------------
    int h_pix[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int hp = (tid >> 3) + j * 64;
        const int hr = hp / WP, hc = hp - hr * WP;
        const int y = r0 + hr - 1, x = hc - 1;
        const bool ok = hp < NP && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)W;
        h_pix[j] = ok ? (img0 * p.H + y) * W + x : -1;
    }
    const bool last_slot_valid = (tid >> 3) + (NS - 1) * 64 < NP;
    const unsigned st_base = lds0 + BS_B + (unsigned)(tid >> 3) * 144 + (unsigned)(tid & 7) * 16;   // + j * 9216 (+ HALO_B)

    // ---- fragment addresses -----------------------------------------------------------------------------------------------------
    unsigned abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = wr * 64 + i * 32 + (lane & 31);
        const int r = m / W, c = m - r * W;
        abase[i] = lds0 + BS_B + (unsigned)((r * WP + c) * 144) + (unsigned)(lane >> 5) * 16;      // halo buffer 0, tap (0,0), ks 0
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
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* dst = smem + buf * 4096 + (wave * 8 + 64 * i) * 32;
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)(bsrc[i] + (size_t)kt * 32), (lptr_t)(dst), 16, 0, 0);
        }
    };

    // ---- slab bookkeeping -------------------------------------------------------------------------------------------------------
    const int nchunks = (p.c0 + p.c1) / BK;          // 3x3 slabs (9 taps)
    const int nextra = (p.ec0 + p.ec1) / BK;         // appended 1x1 slabs (centre tap only)
    const int NCH = nchunks + nextra;
    const int KT = nchunks * 9 + nextra;
    const int Ctot = p.c0 + p.c1;
    const bool silu = p.norm_act == DS_ACT_SILU;

    f32x4 hreg[NS];                                   // raw halo of the slab that is converted next
    f32x4 cmu, cga, cbe;                              // its {mu, A, B} quads (identity when there is nothing to normalise)
    // asm global loads of slab `chunk` (clamped to the last slab: the loads are unconditional so that their count is static)
    auto slab_src = [&](int chunk, const float*& src, int& ld) {
        const bool extra = chunk >= nchunks;
        const int c = (extra ? chunk - nchunks : chunk) * BK;
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
        hreg[j] = gld16(h_pix[j] >= 0 ? nsrc + (size_t)h_pix[j] * nld : zero);
    };
    auto load_coefs = [&](int chunk) {
        const int ch = min(chunk, NCH - 1);
        slab_src(ch, nsrc, nld);
        const bool on = p.norm != nullptr && ch < nchunks;
        const float* cp = on ? p.norm + (size_t)img0 * 3 * Ctot + ch * BK + ld_col : ident;
        const int st = on ? Ctot : 4;
        cmu = gld16(cp);
        cga = gld16(cp + st);
        cbe = gld16(cp + 2 * st);
    };
    // conversion of one element of slot j (GroupNorm affine + SiLU, networks_edm.py:160,167), then the 16-B store of the slot
    f32x4 cvt;
    auto convert_elem = [&](auto jc, auto ec, bool act) {
        constexpr int j = decltype(jc)::value, e = decltype(ec)::value;
        float v = fmaf(hreg[j][e] - cmu[e], cga[e], cbe[e]);
        if (act) v = ds_silu(v);
        cvt[e] = h_pix[j] >= 0 ? v : 0.f;
    };
    auto store_slot = [&](auto jc, unsigned st_addr) {
        constexpr int j = decltype(jc)::value;
        if (j < NS - 1 || last_slot_valid) lds_wr<j * 9216>(st_addr, cvt);
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
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(cmu), "+v"(cga), "+v"(cbe));   // (function scope: plain operands)
        const bool act0 = silu && p.norm != nullptr && nchunks > 0;
        static_for<NS>([&](auto jc) {
            f32x4& hj = hreg[decltype(jc)::value];
            asm volatile("" : "+v"(hj));
            convert_elem(jc, IC<0>{}, act0); convert_elem(jc, IC<1>{}, act0); convert_elem(jc, IC<2>{}, act0); convert_elem(jc, IC<3>{}, act0);
            store_slot(jc, st_base);
        });
        load_coefs(1);
        static_for<NS>([&](auto jc) { load_slot(jc); });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // first halo and both weight tiles are in LDS (vmcnt(0) above)

    Frag2 P_, Q_;
    int kt = 0;                      // global tap counter = weight K tile
    unsigned hb = 0;                 // byte offset of the CURRENT halo buffer (0 or HALO_B); the other one is being filled
    const unsigned first_off = nchunks > 0 ? 0u : ROW + 144u;
    frag_read2<0>(P_, abase[0] + first_off, abase[1] + first_off, bq[0]);

#define DS2_M(i, j, r, f) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32((f).a##i[r], (f).b##j[r], acc[i][j], 0, 0, 0)
    // 16 MFMAs of one K step with a hook after each of them; hook(IC<k>) emits its own sched_barrier fences when it does anything
#define DS2_GROUP(f, hook)                                                                                                        \
    DS2_M(0, 0, 0, f); hook(IC<0>{});  DS2_M(0, 1, 0, f); hook(IC<1>{});  DS2_M(1, 0, 0, f); hook(IC<2>{});  DS2_M(1, 1, 0, f); hook(IC<3>{});   \
    DS2_M(0, 0, 1, f); hook(IC<4>{});  DS2_M(0, 1, 1, f); hook(IC<5>{});  DS2_M(1, 0, 1, f); hook(IC<6>{});  DS2_M(1, 1, 1, f); hook(IC<7>{});   \
    DS2_M(0, 0, 2, f); hook(IC<8>{});  DS2_M(0, 1, 2, f); hook(IC<9>{});  DS2_M(1, 0, 2, f); hook(IC<10>{}); DS2_M(1, 1, 2, f); hook(IC<11>{});  \
    DS2_M(0, 0, 3, f); hook(IC<12>{}); DS2_M(0, 1, 3, f); hook(IC<13>{}); DS2_M(1, 0, 3, f); hook(IC<14>{}); DS2_M(1, 1, 3, f); hook(IC<15>{});

    // One tap.  T9 = tap index inside a 3x3 slab (0..8) or 9 = the single centre tap of a 1x1 slab.
    //   K steps 0..2 : fragments double-buffered in P / Q; hooks convert halo slots of the NEXT slab (3x3 slab: slot T9-1 during
    //                  tap T9 = 1..NS; 1x1 slab: all NS slots, starting at the 9th MFMA so that the raw loads of the previous tap
    //                  have landed)
    //   then         : lgkmcnt(0) + vmcnt(weights of tap kt+1 landed) + barrier
    //   K step 3     : hooks issue the weight DMA of tap kt+2, the first fragment reads of tap kt+1 and -- on tap 7 of a 3x3 slab
    //                  and on every 1x1 slab -- the raw loads of the slab after next (NS + 3 asm loads, the newest VMEM operations
    //                  of the wave, so the barrier of tap 8 waits with vmcnt(NS + 3) and leaves them in flight)
    auto tap = [&](auto t9c, int chunk) {
        Frag2 &P = P_, &Q = Q_;                                         // (asm operands do not capture: bind references first)
        constexpr int T9 = decltype(t9c)::value;
        constexpr bool X = (T9 == 9);                                   // 1x1 slab
        constexpr int TY = X ? 1 : T9 / 3, TX = X ? 1 : T9 % 3;
        constexpr int AOFF = TY * (int)ROW + TX * 144;
        constexpr bool SLAB_END = X || T9 == 8;
        constexpr bool LOADS = X || T9 == 7;
        const unsigned va0 = abase[0] + hb, va1 = abase[1] + hb;
        const unsigned cb = (unsigned)(kt & 1) * 16384u;
        const unsigned st_addr = st_base + (hb ^ HALO_B);               // the buffer being filled (HALO_B is not a power of two:
                                                                        // hb is 0 or HALO_B, so the xor is a select)
        const bool conv_act = silu && p.norm != nullptr && (chunk + 1) < nchunks;
        auto conv_hook = [&](auto hc) {                                 // h-th conversion step of this tap (5 steps per slot)
            constexpr int h = decltype(hc)::value;
            constexpr int j = X ? h / 5 : T9 - 1, s = X ? h % 5 : h;
            if constexpr (j >= 0 && j < NS && s >= 0 && s < 5) {
                DS2_FENCE();
                if constexpr (s == 0 && (j == 0)) {
                    f32x4 &m_ = cmu, &a_ = cga, &b_ = cbe;              // asm operands do not capture: bind references first
                    if constexpr (X) asm volatile("s_waitcnt vmcnt(0)" : "+v"(m_), "+v"(a_), "+v"(b_));
                    else asm volatile("" : "+v"(m_), "+v"(a_), "+v"(b_));
                }
                if constexpr (s == 0) { f32x4& hj = hreg[j]; asm volatile("" : "+v"(hj)); }
                if constexpr (s < 4) convert_elem(IC<j>{}, IC<s>{}, conv_act);
                else store_slot(IC<j>{}, st_addr);
                DS2_FENCE();
            }
        };
        auto hook0 = [&](auto kc) { constexpr int k = decltype(kc)::value; if constexpr (X) { if constexpr (k >= 8) conv_hook(IC<k - 8>{}); } else { if constexpr (k % 2 == 1) conv_hook(IC<k / 2>{}); } };
        auto hook1 = [&](auto kc) { constexpr int k = decltype(kc)::value; if constexpr (X) conv_hook(IC<k + 8>{}); };
        auto hook2 = [&](auto kc) { constexpr int k = decltype(kc)::value; if constexpr (X) conv_hook(IC<k + 24>{}); };

        frag_read2<AOFF + 32>(Q, va0, va1, bq[1] + cb);
        DS2_FRAG_WAIT(4, P);
        DS2_GROUP(P, hook0)
        DS2_FENCE();
        frag_read2<AOFF + 64>(P, va0, va1, bq[2] + cb);
        DS2_FRAG_WAIT(4, Q);
        DS2_GROUP(Q, hook1)
        DS2_FENCE();
        frag_read2<AOFF + 96>(Q, va0, va1, bq[3] + cb);
        DS2_FRAG_WAIT(4, P);
        DS2_GROUP(P, hook2)
        DS2_FENCE();
        DS2_FRAG_WAIT(0, Q);
        if constexpr (!X && T9 == 8) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS + 3) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // raw barrier: every LDS access of this loop is volatile asm with hand-placed waits (above); __syncthreads() would add a
        // vmcnt(0) of its own and drain the raw-halo loads that are meant to stay in flight across the barrier of tap 8
        __builtin_amdgcn_s_barrier();
        // ---- after the barrier: buffer kt & 1 and (at a slab end) the current halo are dead ------------------------------------
        if constexpr (SLAB_END) hb ^= HALO_B;
        const bool next_is_x = SLAB_END ? (chunk + 1 >= nchunks) : false;
        auto hook3 = [&](auto kc) {
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
                    frag_read2<0>(P, abase[0] + hb + off, abase[1] + hb + off, bq[0] + nb);
                } else {
                    constexpr int NY = (T9 + 1) / 3, NX = (T9 + 1) % 3;
                    frag_read2<NY * (int)ROW + NX * 144>(P, va0, va1, bq[0] + nb);
                }
                DS2_FENCE();
            } else if constexpr (LOADS && k >= 2 && k < 2 + NS + 1) {
                DS2_FENCE();
                if constexpr (k == 2) load_coefs(chunk + 2);
                else load_slot(IC<k - 3>{});
                DS2_FENCE();
            }
        };
        DS2_GROUP(Q, hook3)
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
#undef DS2_GROUP

    // drain: the last (discarded) fragment prefetch and the unconditional raw loads of the clamped "slab after next"
    DS2_FRAG_WAIT(0, P_);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cmu), "+v"(cga), "+v"(cbe));
    static_for<NS>([&](auto jc) { f32x4& hj = hreg[decltype(jc)::value]; asm volatile("" : "+v"(hj)); });
    __syncthreads();
    epilogue<0, true>(p, acc, smem + wave * 32 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 64, p.out);
}

template <int W>
int launch_halo2_w(KParams p, int wide, hipStream_t stream) {
    using G = Geo2<W>;
    p.TH = G::TH; p.nimg = 1; p.HP = G::HP; p.WP = G::WP; p.NP = G::NP;
    p.mtiles = p.M / 256;
    p.ntiles = wide;
    p.n_begin = 0;
    p.splits = 1;
    int smem = (int)G::SMEM;
    const int epi = 8 * 32 * EPI_LD * (int)sizeof(float);
    if (smem < epi) smem = epi;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo2_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_halo2_kernel<W>), dim3(grid_1d(p.mtiles, p.ntiles), 1), dim3(512), smem, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace

// Layers this kernel takes: see the header comment.  `wide` = number of full 128-column tiles (the caller launches a ragged
// 64-column tail, if any, on the first-generation kernel).
bool conv3x3_halo2_applicable(const KParams& p, int wide) {
    if (p.taps != 9 || wide < 1) return false;
    if (!(p.W == 16 || p.W == 32 || p.W == 64)) return false;
    if (p.HW % 256 || p.M % 256 || p.HW != p.H * p.W) return false;
    if ((p.c0 + p.c1) % BK || (p.ec0 + p.ec1) % BK || (p.c1 > 0 && p.c0 % BK) || (p.ec1 > 0 && p.ec0 % BK)) return false;
    if ((p.c0 + p.c1) == 0) return false;
    if (p.nrows_b < wide * 128) return false;                 // every weight row of a full tile exists (rows are padded to 128)
    if ((long long)(p.M / 256) * wide < 1) return false;
    return true;
}

long long g_halo2_launches = 0;     // how many launches went to this kernel (tests assert the routing)

int launch_conv3x3_halo2(KParams& p, int wide, hipStream_t stream) {
    ++g_halo2_launches;
    switch (p.W) {
        case 16: return launch_halo2_w<16>(p, wide, stream);
        case 32: return launch_halo2_w<32>(p, wide, stream);
        default: return launch_halo2_w<64>(p, wide, stream);
    }
}

}  // namespace igemm
