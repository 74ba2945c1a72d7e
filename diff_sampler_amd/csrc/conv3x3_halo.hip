// 3x3 convolution as an implicit GEMM with an LDS-resident input halo (gfx950, fp32 MFMA).
//
// Round-1 ablations of the generic gather kernel (gemm_conv.hip) showed it is bound by the per-CU global->LDS staging
// path, not by the matrix pipe: with the staging loads removed it runs at 147-152 TFLOP/s (93-96 % of the 157.3 peak),
// with them at 112-122 -- even when the loads hit L1 and even with prefetch distance 2 (so it is issue/throughput of
// the vector-memory path, ~32 KB per 128x128x32 tile, not latency).  The 9 taps of a 3x3 conv re-read the SAME
// activations shifted by one pixel, so this kernel stages each 32-channel slab of the input ONCE per M tile as a
// halo tile in LDS and derives all 9 tap operands from it:
//
//   M tile  = BM output pixels = nimg image slots x TH rows x W columns (TH*W*nimg = BM)
//   halo    = nimg x (TH+2) x (W+2) pixels x 32 channels, rows padded to 36 floats (same conflict-free ds_read_b128
//             pattern as the generic kernel); out-of-image halo pixels are zero (loaded from a zero page)
//   tap (dy,dx) operand of output pixel (s,r,c) = halo[(s*(TH+2) + r+1+dy)*(W+2) + c+1+dx]  -> one uniform LDS
//             offset per tap added to a per-lane base: no per-tap address arithmetic, no per-tap global loads for A
//   weights = [Cout_pad][K], K = (chunk*9 + tap)*32 + cc, double-buffered per tap exactly like the generic kernel
//
// Two tile shapes (template WM = wave rows; every wave owns 64x64 outputs = 2x2 MFMA 32x32x2 tiles):
//   WM = 2: BM = 128, 4 waves, 2 workgroups per CU.  Global bytes per slab and tile: halo 26 KB (W=32) + weights
//           9 x 16 KB = 170 KB (the generic kernel moves 288 KB) -> 124-137 TFLOP/s.
//   WM = 4: BM = 256, 8 waves, 1 workgroup per CU.  The weight tile is shared by twice as many pixels:
//           (43.5 + 144) KB per 2x the FLOPs = 94 KB per 128x128-equivalent.  Used when the layer has enough 256-pixel
//           tiles to fill the chip.
// Round 2 added two wave-tile shapes on top of them (template WN = wave columns, NT = 32-column MFMA blocks per wave; see the kernel):
//   WM = 4, WN = 2, NT = 4: 256 pixels x 256 channels, eight waves of 64 x 128 -- the dominant kernel of the headline (0.87-0.88 of the
//           fp32 matrix peak), with scalar-addressed weight DMA and a non-temporal epilogue;
//   WM = 2, WN = 4, NT = 1: the 128 x 128 tile on eight waves of 64 x 32, for layers with at most one tile per CU.
// The halo for slab c+1 is loaded into registers while slab c is multiplied and written to LDS at the slab boundary
// (one extra barrier per 576 MFMAs per wave).
#include "igemm_common.h"

namespace igemm {
namespace {

constexpr int ns_max(int wn) { return wn == 4 ? 5 : (wn == 2 ? 10 : 18); }    // halo float4 slots per thread (upper bound per shape)

__device__ float g_zero_page_halo[64];   // zero-initialised

// ---- hand-ordered LDS fragment reads (VAR bit 0, "PIPE") ---------------------------------------------------------------------
// hipcc emits `4-6 x ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 16 x v_mfma` per K step of the tap loop: the LDS latency of every
// fragment group is exposed to the issuing wave and is hidden only while the SIMD's other wave happens to be in its MFMA phase;
// source-level double buffering is re-clustered by the machine scheduler.  Here the reads are inline asm (volatile asm statements
// keep their order), the fragments of K step g+1 are requested BEFORE the MFMAs of step g, and the wait is a counted
// `s_waitcnt lgkmcnt(4)` that leaves exactly those four newer reads in flight.  The wait statement names the fragment registers
// "+v", so every consumer is data-dependent on it (cdna_hip_programming.md section 5.7, form (ii)); LDS operations return in
// order, so compiler-issued LDS/SMEM traffic in between can only make a counted wait stricter, never wrong.
struct Frag { f32x4 a0, a1, b0, b1; };

template <int OFF>
__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int KS, int BOFF1>
__device__ __forceinline__ void frag_read(Frag& f, unsigned va0, unsigned va1, unsigned vb) {
    f.a0 = lds_read_b128<KS * 32>(va0);
    f.a1 = lds_read_b128<KS * 32>(va1);
    f.b0 = lds_read_b128<0>(vb);
    f.b1 = lds_read_b128<BOFF1>(vb);
}
#define DS_FRAG_WAIT(N, f) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"((f).a0), "+v"((f).a1), "+v"((f).b0), "+v"((f).b1))

__device__ __forceinline__ unsigned lds_addr(const float* p) {
    typedef __attribute__((address_space(3))) const void* lcptr_t;
    return (unsigned)(size_t)(lcptr_t)(p);
}


// Kernel variants (template VAR, selected per call by ds_conv_args.tune.variant; 0 = the round-1 kernel, 1 = PIPE).  The other
// bits are TIMING ABLATIONS that produce wrong results on purpose (tools/bench_conv.py --variants): 2 = no GroupNorm/SiLU
// arithmetic in the halo writer, 4 = the halo is staged once and never again, 8 = the weight DMA is issued once and never
// again, 16 = the epilogue stores nothing.
constexpr int VAR_PIPE = 1, VAR_NO_NORM = 2, VAR_NO_HALO = 4, VAR_NO_DMA = 8, VAR_NO_EPI = 16;
// Round-2 options of the 256 x 256 tile (NT = 4).  VAR_LEAN | VAR_NTEPI is the DEFAULT kernel of that tile (+1.3 % on the main
// shapes, +0.7 ... 1.1 % on the headline, profiles/r2_conv_tile_options.txt); tune.variant = 256 selects the plain one (A/B runs):
//   VAR_LEAN  the weight DMA of a tap is addressed as (uniform SGPR base of the tap and row group) + (ONE constant 32-bit per-lane
//             offset) with the LDS destination computed on the scalar unit: the compiled loop otherwise spends 22 VALU instructions
//             per tap on 64-bit pointer arithmetic, zero-page selects and v_readfirstlane of a wave-uniform value, and on this chip
//             a VALU instruction costs ~3.3 cycles of fp32-matrix issue time (profiles/r2_probe_mfma_valu.txt).
//   VAR_NTEPI residual loads / output stores of the epilogue carry the non-temporal hint.
// Measured and dropped (same file): issuing the weight DMA of tap kt+1 in the middle of tap kt instead of at its top (-2.3 %: an
// LDS-DMA instruction between MFMA groups costs more issue time than one next to the tap's first fragment reads); starting the first
// round's workgroups 2 ... 8 us apart so that the CUs are not all in their epilogue -- the tile's only HBM-heavy phase -- at once
// (what the de-synchronised epilogues gain, the late finishers lose: +-0.1 %).
constexpr int VAR_LEAN = 32, VAR_NTEPI = 128;
constexpr int VAR_TILE_OPTS = VAR_LEAN | VAR_NTEPI;

// GLDS = weight tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write).  The DMA
// writes lane-linear (wave base + lane*16 B), so the LDS image is unpadded [row][32 floats] and the bank-conflict fix
// moves to an XOR swizzle of the 16-B chunk index with (row >> 1) & 7, applied to the per-lane SOURCE address when
// staging and to the fragment read address (same involution on both sides; 16 distinct slots per ds_read_b128 lane group).
// WN = wave columns: 2 = 128 output channels per tile (the normal shape), 1 = 64 (launched only for the ragged last
// column tile of layers whose channel count is not a multiple of 128 -- 192, 320, 576 ... -- and for the few-channel
// output conv, instead of multiplying a half-empty 128-wide tile; it starts at column p.n_begin).
// WN = 4 with NT = 1: the 128-pixel x 128-channel tile on EIGHT waves of 64 x 32 (2 wave rows x 4 wave columns).  A layer with at most
// one 128 x 128 tile per CU (the 8x8 layers at the benchmark batch: 256 tiles) leaves every SIMD with ONE 64 x 64 wave: the fp32 matrix
// pipe then issues at 91 % at best (profiles/r2_probe_mfma_valu.txt) and nothing hides that wave's LDS latency -- those layers ran at
// 0.70 of the peak against 0.87 for the same kernel with two waves per SIMD.  Half-size wave tiles double the waves instead (3 fragment
// reads per 8 MFMAs instead of 4 per 16).
// NT = 32-column MFMA tiles per wave: 2 = 64 x 64 per wave (the normal shape), 4 = 64 x 128 per wave, i.e. a 256-pixel x 256-channel
// tile for the 8-wave shape: the halo of a slab is staged (normalised, SiLU'd) once for 256 output channels instead of twice, a K step
// reads 6 fragments for 32 MFMAs instead of 4 for 16, and a tap has 128 MFMAs per wave between barriers (128 accumulator registers).
template <int WM, bool GLDS, int WN, int VAR = 0, int NT = 2>
__global__ void __launch_bounds__(64 * WM * WN, WN == 4 ? 2 : WN) conv3x3_halo_kernel(const KParams p) {
    static_assert(NT != 1 || (WM == 2 && WN == 4 && GLDS && VAR == 0), "half-size wave tiles: 128 x 128 tile on 8 waves, LDS-DMA weights");
    static_assert(WN != 4 || NT == 1, "four wave columns: 32-column wave tiles only");
    static_assert(NT == 1 || NT == 2 || ((NT == 4 || NT == 3) && WM == 4 && WN == 2 && GLDS && (VAR & ~(VAR_TILE_OPTS | VAR_NO_NORM | VAR_NO_HALO | VAR_NO_DMA | VAR_NO_EPI)) == 0), "wide-N tiles: 8-wave LDS-DMA shape only");
    static_assert(NT >= 3 || (VAR & VAR_TILE_OPTS) == 0 || (NT == 2 && WN == 2 && GLDS),
                  "lean addressing / non-temporal epilogue: the 256 x 256 tile, and (round 3) the 128-column LDS-DMA tiles");
    constexpr bool LEAN = (VAR & VAR_LEAN) != 0, NTEPI = (VAR & VAR_NTEPI) != 0;
    constexpr bool PIPE = (VAR & VAR_PIPE) != 0;
    static_assert(!PIPE || GLDS, "the pipelined tap loop reads the LDS-DMA weight image");
    constexpr int T = 64 * WM * WN;        // threads
    constexpr int TBM = 64 * WM;           // output pixels per tile
    constexpr int BNT = 32 * NT * WN;      // output channels per tile
    constexpr int BROWS = BNT * 8 / T;     // weight float4 per thread per tap (4 or 2)
    constexpr int BLD = GLDS ? 32 : LDSK;  // floats per weight row in LDS
    constexpr int NS_MAX = (NT >= 3) ? 7 : ns_max(WN);      // wide-N tiles: one image per tile, at most 7 slots per thread (64-column images)
    constexpr int B_FLOATS = 2 * BNT * LDSK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Bs = smem;                               // [2][BNT][LDSK]
    float* Ah = smem + B_FLOATS;                    // [NP][LDSK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt, blockIdx.y)) return;
    const int m0 = mt * TBM, n0 = p.n_begin + nt * BNT;
    const int ld_row = tid >> 3, ld_col = (tid & 7) * 4;
    const float* zero = g_zero_page_halo;

    // ---- tile geometry -----------------------------------------------------------------------------------------
    const int img0 = m0 / p.HW;
    const int r0 = (p.nimg == 1) ? (m0 - img0 * p.HW) / p.W : 0;
    const int n_images = p.M / p.HW;
    const int ns = (p.NP * 8 + T - 1) / T;          // halo slots per thread actually used (uniform)

    // per-thread halo slots: pixel index in the image tensor (or -1: zero) -- fixed for the whole K loop
    // (tiles of several small images, e.g. 8x8: also the slot's offset into the coefficient planes staged in LDS, see coef_stage)
    int h_pix[NS_MAX];
    int h_cof[NS_MAX];
#pragma unroll
    for (int j = 0; j < NS_MAX; ++j) {
        const int q = tid + j * T;
        const int hp = q >> 3;
        const int s = hp / (p.HP * p.WP);
        const int rem = hp - s * p.HP * p.WP;
        const int hr = rem / p.WP, hc = rem - hr * p.WP;
        const int img = img0 + s, y = r0 + hr - 1, x = hc - 1;
        const bool ok = hp < p.NP && img < n_images && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        h_pix[j] = ok ? (img * p.H + y) * p.W + x : -1;
        h_cof[j] = s * 3 * (p.c0 + p.c1);
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
    size_t b_off[BROWS];
    bool b_ok[BROWS];
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
        const int n = n0 + ld_row + (T / 8) * i;
        b_ok[i] = n < p.nrows_b;
        // GLDS: this lane fills LDS chunk (tid & 7) of its row, which must hold source chunk (tid & 7) ^ swizzle(row);
        // (T / 8) is a multiple of 16, so the swizzle depends on ld_row only
        b_off[i] = (size_t)n * p.ldb + (GLDS ? (((tid & 7) ^ ((ld_row >> 1) & 7)) * 4) : ld_col);
    }

    f32x4 hreg[NS_MAX];
    f32x4 rb[BROWS];
    const int nchunks = (p.c0 + p.c1) / BK;         // 3x3 slabs
    const int nextra = (p.ec0 + p.ec1) / BK;        // appended 1x1 slabs (fused skip projection)
    const int Ctot = p.c0 + p.c1;
    auto halo_load = [&](int chunk) {
        const bool extra = chunk >= nchunks;
        const int c = (extra ? chunk - nchunks : chunk) * BK;
        const int cc0 = extra ? p.ec0 : p.c0;
        const bool first = c < cc0;
        const float* s0 = extra ? p.e0 : p.a0;
        const float* s1 = extra ? p.e1 : p.a1;
        const float* src = first ? s0 + c + ld_col : s1 + (c - cc0) + ld_col;
        const int ld = first ? (extra ? p.elda0 : p.lda0) : (extra ? p.elda1 : p.lda1);
#pragma unroll
        for (int j = 0; j < NS_MAX; ++j) {
            if (j < ns) {
                const float* ptr = h_pix[j] >= 0 ? src + (size_t)h_pix[j] * ld : zero;
                hreg[j] = *reinterpret_cast<const f32x4*>(ptr);
            }
        }
    };
    // LDS <- registers; the GroupNorm affine + SiLU of the consumer layer is applied here, once per element and slab
    // (the reference materialises silu(norm(x)) as a tensor, networks_edm.py:160,167); padding pixels stay zero.
    // When the tile lies in one image (nimg == 1) the three coefficient quads of this thread's channels are
    // prefetched together with the halo (coef_load), so nothing at the slab boundary waits on memory.
    const bool norm_on = p.norm != nullptr;
    const bool one_img = p.nimg == 1;
    f32x4 cmu = {0.f, 0.f, 0.f, 0.f}, cga = {1.f, 1.f, 1.f, 1.f}, cbe = {0.f, 0.f, 0.f, 0.f};
    auto coef_load = [&](int chunk) {
        const bool on = norm_on && one_img && chunk < nchunks;
        const float* cp = on ? p.norm + (size_t)img0 * 3 * Ctot + chunk * BK + ld_col : zero;
        const int st = on ? Ctot : 0;
        cmu = *reinterpret_cast<const f32x4*>(cp);
        cga = *reinterpret_cast<const f32x4*>(cp + st);
        cbe = *reinterpret_cast<const f32x4*>(cp + 2 * st);
    };
    // A tile of several images (8x8 layers: 2 or 4 images per tile) needs the coefficients of each slot's own image.  Loading them
    // from global memory inside halo_store put an L2 latency plus an integer division per slot on the slab boundary (the 8x8 layers
    // lost 11 % to it, profiles/r2_conv_ablations.txt v2); instead the {mu, A, B} planes of the tile's images are copied to LDS once,
    // behind the halo, and halo_store reads them from there.
    float* Cs = Ah + p.NP * LDSK;                   // [nimg][3][Ctot]
    auto coef_stage = [&]() {
        const int per = 3 * Ctot;
        DS_RACE_SKEW(wave);
        for (int i = tid * 4; i < p.nimg * per; i += T * 4) {
            const int s = i / per;
            f32x4 v = {0.f, 1.f, 0.f, 0.f};
            if (img0 + s < n_images) v = *reinterpret_cast<const f32x4*>(p.norm + (size_t)img0 * per + i);
            *reinterpret_cast<f32x4*>(Cs + i) = v;
        }
    };
    auto halo_store = [&](int chunk) {
        const bool do_norm = norm_on && chunk < nchunks && !(VAR & VAR_NO_NORM);
        const int cq = chunk * BK + ld_col;
        DS_RACE_SKEW(wave);
#pragma unroll
        for (int j = 0; j < NS_MAX; ++j) {
            if (j < ns) {
                const int q = tid + j * T;
                f32x4 v = hreg[j];
                if (do_norm && h_pix[j] >= 0) {
                    f32x4 mu = cmu, ga = cga, be = cbe;
                    if (!one_img) {
                        const float* cp = (p.coef_lds ? Cs : p.norm + (size_t)img0 * 3 * Ctot) + h_cof[j] + cq;
                        mu = *reinterpret_cast<const f32x4*>(cp);
                        ga = *reinterpret_cast<const f32x4*>(cp + Ctot);
                        be = *reinterpret_cast<const f32x4*>(cp + 2 * Ctot);
                    }
                    v = (v - mu) * ga + be;
                    if (p.norm_act == DS_ACT_SILU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ds_silu(v[e]);
                    }
                }
                if ((q >> 3) < p.NP) *reinterpret_cast<f32x4*>(Ah + (q >> 3) * LDSK + (q & 7) * 4) = v;
            }
        }
    };
    auto b_addr = [&](int kt, int i) -> const float* { return b_ok[i] ? p.b + b_off[i] + kt * BK : zero; };
    auto b_store = [&](int buf) {
        float* bs = Bs + buf * BNT * LDSK + ld_row * LDSK + ld_col;
        DS_RACE_SKEW(wave);
#pragma unroll
        for (int i = 0; i < BROWS; ++i) *reinterpret_cast<f32x4*>(bs + (T / 8) * i * LDSK) = rb[i];
    };
    // LDS-DMA of weight tile kt into buffer buf: one 1-KiB wave instruction covers 8 rows x 128 B
    auto b_dma = [&](int kt, int buf) {
        DS_RACE_SKEW(wave);
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            float* dst = Bs + buf * BNT * 32 + (wave * 8 + (T / 8) * i) * 32;      // wave-uniform base
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)(b_addr(kt, i)), (lptr_t)(dst), 16, 0, 0);
        }
    };

    // VAR_LEAN: every row of the tile exists (the launcher only takes 256-column tiles below N <= nrows_b), so there is no zero-page
    // select; row group i of a tap = scalar base + i * 64 rows, lane offset constant for the whole kernel
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned b_voff = (unsigned)((ld_row * p.ldb + (((tid & 7) ^ ((ld_row >> 1) & 7)) * 4)) * (int)sizeof(float));
    auto b_dma_lean = [&](int kt, int buf) {
        const char* tap = reinterpret_cast<const char*>(p.b + (size_t)n0 * p.ldb + (size_t)kt * BK);
        DS_RACE_SKEW(wave_u);
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
            float* dst = Bs + buf * BNT * 32 + (wave_u * 8 + (T / 8) * i) * 32;
            const char* src = tap + (size_t)i * (T / 8) * p.ldb * sizeof(float) + (size_t)b_voff;
            typedef const __attribute__((address_space(1))) void* gptr_t;
            typedef __attribute__((address_space(3))) void* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)(src), (lptr_t)(dst), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];                                // columns 0..63 of the wave tile
    f32x16 acc_hi[2][2];                             // columns 64..127 (NT == 4 only; two plain arrays so that both stay in registers)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc_hi[i][j][r] = 0.f; }

    const int NCH_all = nchunks + nextra;            // slabs: 3x3 ones (9 taps each) then 1x1 ones (centre tap only)
    // split-K: this block contracts slabs [c_begin, NCH) only (splits == 1: everything)
    const int c_begin = (int)((long long)blockIdx.y * NCH_all / p.splits);
    const int NCH = (int)((long long)(blockIdx.y + 1) * NCH_all / p.splits);
    auto kt_of = [&](int chunk) { return chunk < nchunks ? chunk * 9 : nchunks * 9 + (chunk - nchunks); };
    const int kt0 = kt_of(c_begin);
    const int KT = kt_of(NCH);

    // ---- prologue ----------------------------------------------------------------------------------------------
    halo_load(c_begin);
    coef_load(c_begin);
    if (LEAN) {
        b_dma_lean(kt0, kt0 & 1);
    } else if (GLDS) {
        b_dma(kt0, kt0 & 1);
        if (PIPE && kt0 + 1 < KT) b_dma(kt0 + 1, (kt0 + 1) & 1);   // the pipelined loop runs the weight DMA two taps ahead
    } else {
#pragma unroll
        for (int i = 0; i < BROWS; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_addr(kt0, i));
    }
    if (norm_on && !one_img && p.coef_lds) { coef_stage(); __syncthreads(); }
    halo_store(c_begin);
    if (!GLDS) b_store(kt0 & 1);
    if (PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first two weight tiles (see the note in the tap loop)
    if (c_begin + 1 < NCH) { halo_load(c_begin + 1); coef_load(c_begin + 1); }
    __syncthreads();

    // weight fragment offsets: padded rows (register staging) or swizzled 16-B chunks (LDS-DMA)
    const int b_row = wc * 32 * NT + (lane & 31);
    const int b_swz = ((lane & 31) >> 1) & 7;
    auto b_frag = [&](int ks) -> int {
        return GLDS ? b_row * 32 + (((ks * 2 + (lane >> 5)) ^ b_swz) * 4) : b_row * LDSK + (lane >> 5) * 4 + ks * 8;
    };
    int kt = kt0;
    if constexpr (PIPE) {
        // Software-pipelined tap loop.  Per tap: K steps 0..3 alternate between fragment sets P and Q; the reads of step g+1 are in
        // flight under the 16 MFMAs of step g.  The workgroup barrier sits BEFORE the last step's MFMAs: by then every wave has
        // finished its LDS reads of this tap (weight buffer `cur`, and -- on a slab's last tap -- the halo), so right after it the
        // weight tile of tap kt+2 is DMA'd into `cur` (a full tap of flight time before its `vmcnt(0)`), and the first fragments of
        // tap kt+1 are requested under the last 16 MFMAs.
        constexpr int BOFF1 = 32 * 32 * 4;                       // rows +32 of the weight tile
        constexpr unsigned BBUF = BNT * 32 * 4;                  // bytes per weight buffer
        const unsigned a_b0 = lds_addr(Ah) + (unsigned)a_foff[0] * 4, a_b1 = lds_addr(Ah) + (unsigned)a_foff[1] * 4;
        const unsigned c0 = (unsigned)((lane >> 5) ^ b_swz);
        unsigned bq[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bq[ks] = lds_addr(Bs) + (unsigned)b_row * 128 + ((c0 ^ (2u * ks)) * 16);
        auto tap_off = [&](int chunk, int t9) -> unsigned {      // byte offset of the tap inside the halo
            const int tap = chunk < nchunks ? t9 : 4;
            const int ty = tap / 3;
            return (unsigned)(((ty - 1) * p.WP + (tap - ty * 3 - 1)) * LDSK * 4);
        };
        Frag P, Q;
        {
            const unsigned to = tap_off(c_begin, 0);
            frag_read<0, BOFF1>(P, a_b0 + to, a_b1 + to, bq[0] + (unsigned)(kt & 1) * BBUF);
        }
        for (int chunk = c_begin; chunk < NCH; ++chunk) {
            const int ntaps = chunk < nchunks ? 9 : 1;
            for (int t9 = 0; t9 < ntaps; ++t9, ++kt) {
                const unsigned to = tap_off(chunk, t9);
                const unsigned va0 = a_b0 + to, va1 = a_b1 + to;
                const unsigned cb = (unsigned)(kt & 1) * BBUF;
#define DS_MFMA16(f)                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                             \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32((f).a0[r], (f).b0[r], acc[0][0], 0, 0, 0);             \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32((f).a0[r], (f).b1[r], acc[0][1], 0, 0, 0);             \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32((f).a1[r], (f).b0[r], acc[1][0], 0, 0, 0);             \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32((f).a1[r], (f).b1[r], acc[1][1], 0, 0, 0);             \
    }
                frag_read<1, BOFF1>(Q, va0, va1, bq[1] + cb);
                DS_FRAG_WAIT(4, P);
                DS_MFMA16(P)
                __builtin_amdgcn_sched_barrier(0);
                frag_read<2, BOFF1>(P, va0, va1, bq[2] + cb);
                DS_FRAG_WAIT(4, Q);
                DS_MFMA16(Q)
                __builtin_amdgcn_sched_barrier(0);
                frag_read<3, BOFF1>(Q, va0, va1, bq[3] + cb);
                DS_FRAG_WAIT(4, P);
                DS_MFMA16(P)
                __builtin_amdgcn_sched_barrier(0);
                DS_FRAG_WAIT(0, Q);
                // hipcc waits for an LDS-DMA only in front of an LDS read it can see; this loop's reads are asm, so say it here
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                 // all LDS reads of this tap done; weight tile kt+1 landed
                if (!(VAR & VAR_NO_DMA) && kt + 2 < KT) b_dma(kt + 2, kt & 1);
                {
                    // first fragments of the next tap of this slab (past the slab's last tap: a harmless read of the dying halo
                    // that is discarded -- the read is unconditional so that no fragment register is written on one side of a branch)
                    const unsigned tn = tap_off(chunk, min(t9 + 1, ntaps - 1));
                    frag_read<0, BOFF1>(P, a_b0 + tn, a_b1 + tn, bq[0] + (unsigned)((kt + 1) & 1) * BBUF);
                }
                __builtin_amdgcn_sched_barrier(0);
                DS_MFMA16(Q)
#undef DS_MFMA16
            }
            if (chunk + 1 < NCH) {
                // slab boundary: the halo died at the barrier above; publish the next one (its conversion overlaps the last 16 MFMAs)
                DS_FRAG_WAIT(0, P);
                if (!(VAR & VAR_NO_HALO)) {
                    halo_store(chunk + 1);
                    if (chunk + 2 < NCH) { halo_load(chunk + 2); coef_load(chunk + 2); }
                }
                __syncthreads();
                const unsigned tn = tap_off(chunk + 1, 0);
                frag_read<0, BOFF1>(P, a_b0 + tn, a_b1 + tn, bq[0] + (unsigned)(kt & 1) * BBUF);
            }
        }
        DS_FRAG_WAIT(0, P);                                      // drain the last (discarded) prefetch before LDS is reused
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
    for (int chunk = c_begin; chunk < NCH; ++chunk) {
        const int ntaps = chunk < nchunks ? 9 : 1;
        for (int t9 = 0; t9 < ntaps; ++t9, ++kt) {
            const int tap = ntaps == 9 ? t9 : 4;
            const int ty = tap / 3;
            const int toff = ((ty - 1) * p.WP + (tap - ty * 3 - 1)) * LDSK;
            const int cur = kt & 1;
            const int nxt = min(kt + 1, KT - 1);            // past the end: re-stage the last tile (branch-free body)
            const float* as0 = Ah + a_foff[0] + toff;
            const float* as1 = Ah + a_foff[1] + toff;
            const float* bs = Bs + cur * BNT * BLD;
            if (LEAN) b_dma_lean(nxt, cur ^ 1);
            else if (GLDS && !(VAR & VAR_NO_DMA)) b_dma(nxt, cur ^ 1);   // buffer cur^1 was last read in tap kt-1 (barrier passed)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(as0 + ks * 8);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(as1 + ks * 8);
                // register staging: the weight loads ride in the shadow of the MFMA groups
                if (!GLDS) {
                    if (BROWS == 4) rb[ks] = *reinterpret_cast<const f32x4*>(b_addr(nxt, ks));
                    else if (ks < BROWS) rb[ks] = *reinterpret_cast<const f32x4*>(b_addr(nxt, ks));
                }
                if constexpr (NT == 1) {                        // 64 x 32 per wave: one column block
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs + b_frag(ks));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b0[r], acc[0][0], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b0[r], acc[1][0], 0, 0, 0);
                    }
                } else {
                    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs + b_frag(ks));
                    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bs + 32 * BLD + b_frag(ks));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b0[r], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b1[r], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b0[r], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b1[r], acc[1][1], 0, 0, 0);
                    }
                }
                if constexpr (NT == 3) {                        // 64 x 96 per wave (192-column tiles, round 3): one more 32-column block
                    const f32x4 b2 = *reinterpret_cast<const f32x4*>(bs + 64 * BLD + b_frag(ks));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc_hi[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b2[r], acc_hi[0][0], 0, 0, 0);
                        acc_hi[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b2[r], acc_hi[1][0], 0, 0, 0);
                    }
                }
                if constexpr (NT == 4) {                        // second 64-column half: 2 + 2 fragments live at a time
                    const f32x4 b2 = *reinterpret_cast<const f32x4*>(bs + 64 * BLD + b_frag(ks));
                    const f32x4 b3 = *reinterpret_cast<const f32x4*>(bs + 96 * BLD + b_frag(ks));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc_hi[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b2[r], acc_hi[0][0], 0, 0, 0);
                        acc_hi[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[r], b3[r], acc_hi[0][1], 0, 0, 0);
                        acc_hi[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b2[r], acc_hi[1][0], 0, 0, 0);
                        acc_hi[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[r], b3[r], acc_hi[1][1], 0, 0, 0);
                    }
                }
            }
            if (!GLDS) b_store(cur ^ 1);
            __syncthreads();                                // with LDS-DMA in flight the compiler drains vmcnt(0) here
        }
        if (chunk + 1 < NCH && !(VAR & VAR_NO_HALO)) {
            // every wave has passed the barrier of the slab's last tap: its halo is dead, publish the next one
            halo_store(chunk + 1);
            if (chunk + 2 < NCH) { halo_load(chunk + 2); coef_load(chunk + 2); }
            __syncthreads();
        }
    }
    }

    // epilogue staging overlays the LDS allocation (the launcher sizes it for 4 x 64 or 8 x 32 staging rows per wave)
    constexpr bool HALF = (WM * WN == 8);            // 8 waves: 32-row staging halves; up to 4 waves: 64 rows each
    if constexpr ((VAR & VAR_NO_EPI) != 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {                                         // keep the accumulators (and the K loop) alive
                asm volatile("" :: "v"(acc[i][j]));
                if constexpr (NT >= 3) asm volatile("" :: "v"(acc_hi[i][j]));
            }
        return;
    }
    if constexpr (NT == 1) {                         // 64 x 32 wave tiles (also the raw partial tile of a split)
        if (p.splits > 1) {
            const KParams q = split_params(p, blockIdx.y);
            epilogue32<HALF>(q, acc, smem + wave * 32 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 32, q.out);
        } else {
            epilogue32<HALF>(p, acc, smem + wave * 32 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 32, p.out);
        }
        return;
    }
    if (p.splits > 1) {
        const KParams q = split_params(p, blockIdx.y);
        epilogue<0, HALF>(q, acc, smem + wave * (HALF ? 32 : 64) * EPI_LD, lane, m0 + wr * 64, n0 + wc * 64, q.out);
        return;
    }
    // one 64-column half of the wave tile at a time through the wave's private staging rows
    epilogue<0, HALF, NTEPI>(p, acc, smem + wave * (HALF ? 32 : 64) * EPI_LD, lane, m0 + wr * 64, n0 + wc * 32 * NT, p.out);
    if constexpr (NT == 4)
        epilogue<0, HALF, NTEPI>(p, acc_hi, smem + wave * (HALF ? 32 : 64) * EPI_LD, lane, m0 + wr * 64, n0 + wc * 32 * NT + 64, p.out);
    if constexpr (NT == 3)
        epilogue32<HALF>(p, acc_hi, smem + wave * 32 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 32 * NT + 64, p.out);
}

struct Geo { int TH, nimg, NP; bool ok; };

Geo geometry(const KParams& p, int tbm, int wn = 2) {
    Geo g{0, 0, 0, false};
    if (p.taps != 9 || p.W < 4 || p.W > 64) return g;
    if (p.HW >= tbm) { if (tbm % p.W || p.HW % tbm) return g; g.TH = tbm / p.W; g.nimg = 1; }
    else { if (tbm % p.HW) return g; g.TH = p.H; g.nimg = tbm / p.HW; }
    g.NP = g.nimg * (g.TH + 2) * (p.W + 2);
    const int threads = tbm * wn;                     // 64 * WM * WN
    g.ok = (g.NP * 8 + threads - 1) / threads <= ns_max(wn);
    return g;
}

// Per-call overrides (ds_conv_args.tune -> KParams.t_mode / t_variant; 0 = the library's choice):
//   t_mode 128 / 256 = forced M tile, 2 = weights staged through registers instead of LDS-DMA, 4 = no 64-column tail tiles (A/B switches)
//   t_variant        = kernel variant of the LDS-DMA tiles (see VAR_*; benchmarks / ablations)
inline int tile_override(const KParams& p) { return (p.t_mode == 128 || p.t_mode == 256) ? p.t_mode : 0; }
inline bool use_glds(const KParams& p) { return p.t_mode != 2; }
inline bool use_tail64(const KParams& p) { return p.t_mode != 4; }

// 256-pixel x 256-channel tiles (64 x 128 per wave, NT = 4) instead of 256 x 128: taken where the channel count reaches 256, the
// tile lies in one image (16-, 32- or 64-column images: at most 7 halo slots per thread -- the kernel sits at 256 VGPRs), there is
// no split-K, and the tiles still give every CU a workgroup.  Measured +4.1 ... +4.5 % on the CIFAR-10 / FFHQ 32x32 and 16x16 layers (129 -> 136 TFLOP/s
// network average, profiles/r2_conv_wide_n.txt).  p.t_variant 7 switches it off (A/B runs), 6 forces it regardless of the tile count.
bool wide_n_tiles(const KParams& p, const Geo& g) {
    const int v = p.t_variant & 31;                    // bits 5.. select options of the 256 x 256 tile itself
#ifdef DS_CONV_ABLATIONS
    if (p.t_variant & 0x10000) { if (p.splits != 1 || p.N < 256 || g.NP * 8 > 7 * 512 || g.nimg != 1) return false; return true; }
#endif
    if (v != 0 && v != 6) return false;
    if (p.splits != 1 || p.N < 256 || g.NP * 8 > 7 * 512 || g.nimg != 1 || p.nrows_b < (p.N / 256) * 256) return false;
    const long long blocks = (long long)((p.M + 255) / 256) * (p.N / 256);       // the 256-column tiles (a remainder keeps 128 / 64-column tiles)
    return v == 6 || blocks >= 256;
}

bool wide192_tiles(const KParams& p, const Geo& g) {
    if ((p.t_variant & 31) != 0 || (p.t_variant & 8192)) return false;
    if (p.splits != 1 || p.N % 192 || p.N % 256 == 0 || g.NP * 8 > 7 * 512 || g.nimg != 1 || p.nrows_b < p.N || !p.vec_ok || p.out_planar) return false;
    return (p.t_variant & 16384) || (long long)((p.M + 255) / 256) * (p.N / 192) >= 256;          // the tiles still cover the 256 CUs (bit 14: forced, tests)
}

template <int WM, bool GLDS, int WN, int VAR = 0, int NT = 2>
int launch_one(KParams p, const Geo& g, int n_begin, int ntiles, hipStream_t stream) {
    constexpr int TBM = 64 * WM;
    constexpr int B_BYTES = 2 * 32 * NT * WN * LDSK * (int)sizeof(float);
    p.TH = g.TH; p.nimg = g.nimg; p.HP = g.TH + 2; p.WP = p.W + 2; p.NP = g.NP;
    p.mtiles = (p.M + TBM - 1) / TBM;
    p.ntiles = ntiles;
    p.n_begin = n_begin;
    int smem = B_BYTES + p.NP * LDSK * (int)sizeof(float);
    // coefficient planes of a multi-image tile's images in LDS, where they fit without costing the 4-wave shape its second
    // workgroup per CU (80 KB each) or the 8-wave shape the 128 KB it may ask for; otherwise halo_store reads them from global memory
    const int coef_bytes = (p.norm && g.nimg > 1) ? g.nimg * 3 * (p.c0 + p.c1) * (int)sizeof(float) : 0;
    p.coef_lds = coef_bytes > 0 && smem + coef_bytes <= (WM == 2 ? 80 : 128) * 1024 && !(p.t_variant & 512);     // 512: A/B switch
    if (p.coef_lds) smem += coef_bytes;
    const int epi = 4 * 64 * EPI_LD * (int)sizeof(float);       // = 8 x 32 x EPI_LD for the 8-wave shape
    if (smem < epi) smem = epi;
    DS_ENSURE_DYN_LDS((&conv3x3_halo_kernel<WM, GLDS, WN, VAR, NT>), 128 * 1024);
    hipLaunchKernelGGL((conv3x3_halo_kernel<WM, GLDS, WN, VAR, NT>), dim3(grid_1d(p.mtiles, p.ntiles), p.splits), dim3(64 * WM * WN), smem, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

// Half-size wave tiles (conv3x3_halo_kernel<2, true, 4, 0, 1>) for the `wide` full 128-column tiles of a layer on 128-pixel tiles: taken
// where the whole launch is at most one workgroup per CU (with more, two 4-wave workgroups share a CU and every SIMD has its two waves
// anyway), the float4 epilogue is legal and every one of those tiles is complete.  Variant bit 11: off (A/B runs).
bool half_wave_tiles(const KParams& p, int n_begin, int wide) {
    if ((p.t_variant & 31) != 0 || (p.t_variant & 2048) || !use_glds(p) || wide < 1) return false;
    if (!p.vec_ok || p.out_planar || p.act == DS_ACT_GEGLU || n_begin + wide * BN > p.N || p.nrows_b < n_begin + wide * BN) return false;
    if (p.splits > 1 && !p.vec_part) return false;
    if ((long long)((p.M + 127) / 128) * wide * p.splits > 256) return false;
    return geometry(p, 128, 4).ok;
}

// One layer = the full 128-column tiles (WN = 2) plus, when the channel count leaves 1..64 columns over, one launch of
// 64-column tiles for them (WN = 1); the split-K partial planes are shared and reduced once.
template <int WM, bool GLDS>
int launch_wm(KParams& p, hipStream_t stream) {
    // columns [0, n256): 256-column tiles of the 8-wave kernel where they apply (channel counts such as 384 = 256 + 128,
    // 576 = 2 x 256 + 64, 320 = 256 + 64 get them for their first 256-multiples); the rest as before from column n256 on
    int n256 = 0;
    if constexpr (WM == 4 && GLDS) {
        const Geo g4 = geometry(p, 256, 2);
        // round 3: channel counts that are multiples of 192 but not of 256 (ADM: 192 / 384 / 576) take 192-column tiles (64 x 96 per wave)
        // for ALL their columns instead of 256 / 128-column tiles plus a 64-column tail; variant bit 13 switches it off (A/B runs)
        if (wide192_tiles(p, g4)) {
            KParams q = p;
            const int rc = launch_one<4, true, 2, VAR_TILE_OPTS, 3>(q, g4, 0, p.N / 192, stream);
            if (rc) return rc;
            return DS_OK;
        }
        if (wide_n_tiles(p, g4)) {
            n256 = (p.N / 256) * 256;
            KParams q = p;
            int rc;
#ifdef DS_CONV_ABLATIONS
            if (p.t_variant & 0x10000) {                 // timing ablations of the 256 x 256 tile (wrong results on purpose)
                switch (p.t_variant & 31) {
                    case 4: rc = launch_one<4, true, 2, 4, 4>(q, g4, 0, n256 / 256, stream); break;
                    case 16: rc = launch_one<4, true, 2, 16, 4>(q, g4, 0, n256 / 256, stream); break;
                    case 20: rc = launch_one<4, true, 2, 20, 4>(q, g4, 0, n256 / 256, stream); break;
                    case 28: rc = launch_one<4, true, 2, 28, 4>(q, g4, 0, n256 / 256, stream); break;
                    default: rc = launch_one<4, true, 2, 0, 4>(q, g4, 0, n256 / 256, stream); break;
                }
                if (rc) return rc;
            } else
#endif
            {
            if (p.t_variant & 256) rc = launch_one<4, true, 2, 0, 4>(q, g4, 0, n256 / 256, stream);                  // plain (A/B)
            else rc = launch_one<4, true, 2, VAR_TILE_OPTS, 4>(q, g4, 0, n256 / 256, stream);
            if (rc) return rc;
            }
        }
    }
    const int nrest = p.N - n256;
    const int full = nrest / BN, rem = nrest - full * BN;
    const bool tail64 = rem > 0 && rem <= 64 && geometry(p, 64 * WM, 1).ok && use_tail64(p);
    const int wide = tail64 ? full : (nrest + BN - 1) / BN;
    if (wide > 0) {
        const Geo g = geometry(p, 64 * WM, 2);
        int rc;
        if (WM == 4 && GLDS && (p.t_variant & 31) == 3 && p.splits == 1 && n256 == 0 && conv3x3_halo2_applicable(p, wide, 0)) {
            KParams q = p;
            rc = launch_conv3x3_halo2(q, wide, 0, stream);
        } else if (WM == 2 && GLDS && half_wave_tiles(p, n256, wide)) {
            if constexpr (WM == 2 && GLDS) rc = launch_one<2, true, 4, 0, 1>(p, geometry(p, 128, 4), n256, wide, stream);
            else rc = DS_E_ARG;
        } else if constexpr (GLDS) {
            // (routing the layers that leave a SIMD with ONE wave -- 8x8 layers at the benchmark batch -- to the software-pipelined
            // VAR_PIPE kernel gains 2.8 % on those layers in isolation and nothing measurable on the network: not done)
            switch (p.t_variant & 31) {
                case 1: rc = launch_one<WM, GLDS, 2, 1>(p, g, n256, wide, stream); break;
#ifdef DS_CONV_ABLATIONS
                case 2: rc = launch_one<WM, GLDS, 2, 2>(p, g, 0, wide, stream); break;
                case 4: rc = launch_one<WM, GLDS, 2, 4>(p, g, 0, wide, stream); break;
                case 8: rc = launch_one<WM, GLDS, 2, 8>(p, g, 0, wide, stream); break;
                case 16: rc = launch_one<WM, GLDS, 2, 16>(p, g, 0, wide, stream); break;
                case 28: rc = launch_one<WM, GLDS, 2, 28>(p, g, 0, wide, stream); break;
                case 5: rc = launch_one<WM, GLDS, 2, 5>(p, g, 0, wide, stream); break;
                case 17: rc = launch_one<WM, GLDS, 2, 17>(p, g, 0, wide, stream); break;
                case 29: rc = launch_one<WM, GLDS, 2, 29>(p, g, 0, wide, stream); break;
#endif
                default:
                    // round 3: the 128-column tiles take the 256 x 256 tile's scalar-addressed weight DMA and non-temporal epilogue
                    // (every weight row of a full tile exists: rows are padded to 128); variant bit 12 switches it off (A/B runs)
                    if (p.splits == 1 && !(p.t_variant & 4096) && n256 + wide * BN <= p.nrows_b)
                        rc = launch_one<WM, GLDS, 2, VAR_LEAN | VAR_NTEPI>(p, g, n256, wide, stream);
                    else
                        rc = launch_one<WM, GLDS, 2, 0>(p, g, n256, wide, stream);
                    break;
            }
        } else {
            rc = launch_one<WM, GLDS, 2>(p, g, n256, wide, stream);
        }
        if (rc) return rc;
    }
    if (tail64) {
        int rc = launch_one<WM, GLDS, 1>(p, geometry(p, 64 * WM, 1), n256 + full * BN, 1, stream);
        if (rc) return rc;
    }
    if (p.splits > 1) return launch_splitk_reduce(p, stream);
    return DS_OK;
}

}  // namespace

bool conv3x3_halo_supported(const KParams& p) { return geometry(p, 128).ok; }


// Tile shape and split-K factor of a layer: the cheaper of the two tile shapes under the cost model (igemm_common.h); the
// 256-pixel tile gets a 3 % bonus where both fill the chip (weights shared by twice the pixels).
struct HaloPlan { int tile, splits; };

HaloPlan plan_halo(const KParams& p) {
    const Geo g128 = geometry(p, 128), g256 = geometry(p, 256);
    HaloPlan hp{0, 1};
    if (!g128.ok) return hp;
    const int nt = (p.N + BN - 1) / BN;
    const int units = (p.c0 + p.c1 + p.ec0 + p.ec1) / BK;
    const long long mn = (long long)p.M * p.N, cap = p.part ? p.part_cap : 0;
    double c128 = 0.0, c256 = 0.0;
    const int s128 = choose_splits((long long)((p.M + 127) / 128) * nt, false, units, 9, cap, mn, &c128, p.t_splits);
    hp.tile = 128; hp.splits = s128;
    const long long blocks256 = (long long)((p.M + 255) / 256) * nt;
    // the 8-wave shape only where its tiles alone cover the 256 CUs twice (below that the model is optimistic about it)
    if (g256.ok && tile_override(p) != 128 && (blocks256 >= 512 || tile_override(p) == 256)) {
        const int s256 = choose_splits(blocks256, true, units, 9, cap, mn, &c256, p.t_splits);
        if (tile_override(p) == 256 || 0.97 * c256 < c128) { hp.tile = 256; hp.splits = s256; }
    }
    if (((p.t_variant & 31) == 6 || (p.t_variant & 16384)) && hp.tile == 256) hp.splits = 1;      // forced wide tiles (tests at small sizes): no split-K
    return hp;
}

// 0 = unsupported, 128 / 256 = M tile of the 128-column kernels, 2565 = the 256 x 256-tile kernel, 1284 = 128-pixel tiles on 8 half-size waves
int conv3x3_halo_choice(const KParams& p) {
    const HaloPlan hp = plan_halo(p);
    if (hp.tile == 128 && use_glds(p)) {
        KParams q = p;
        q.splits = hp.splits;
        const int full = p.N / BN, rem = p.N - full * BN;
        const bool tail64 = rem > 0 && rem <= 64 && geometry(p, 128, 1).ok && use_tail64(p);
        if (half_wave_tiles(q, 0, tail64 ? full : (p.N + BN - 1) / BN)) return 1284;
    }
    if (hp.tile == 256 && use_glds(p)) {
        KParams q = p;
        q.splits = hp.splits;
        if ((p.t_variant & 31) == 3 && hp.splits == 1) {          // tune.variant 3: the second-generation 256 x 128 kernel where it applies
            const int full = p.N / BN, rem = p.N - full * BN;
            const bool tail64 = rem > 0 && rem <= 64 && geometry(p, 256, 1).ok && use_tail64(p);
            if (conv3x3_halo2_applicable(q, tail64 ? full : (p.N + BN - 1) / BN, 0)) return 2560;
        }
        if (wide192_tiles(q, geometry(p, 256, 2))) return 2568;        // 192-column tiles for every column (ADM channel counts)
        if (wide_n_tiles(q, geometry(p, 256, 2))) return 2565;         // (its first N / 256 column tiles; a remainder stays on 128 / 64 columns)
    }
    return hp.tile;
}

int launch_conv3x3_halo(KParams& p, hipStream_t stream) {
    const HaloPlan hp = plan_halo(p);
    p.splits = hp.splits;
    if (hp.tile == 256) return use_glds(p) ? launch_wm<4, true>(p, stream) : launch_wm<4, false>(p, stream);
    return use_glds(p) ? launch_wm<2, true>(p, stream) : launch_wm<2, false>(p, stream);
}

}  // namespace igemm
