// 1x1 convolution / Linear layer with fp16 operands on v_mfma_f32_32x32x16_f16 (the reference's use_fp16 / autocast mode for the
// projections: networks_edm.py:155-156 qkv / proj, ldm/modules/attention.py every nn.Linear of the SpatialTransformer).
//
//   out[M][N] = epilogue( A[M][K] * W[N][K]^T ),   A fp32 in HBM (rounded to fp16, RNE, while it is staged), W fp16 (packed once)
//
// Same wave layout, LDS byte layout and hand-ordered pipeline as the fp16 mode of conv3x3_halo2.hip, of which this is the "1x1 slab"
// path made into its own kernel: 256 x 128 tile, 8 waves of 64 x 64 (2 x 2 MFMA tiles), a tap = 64 channels = four K steps of 16.
//   * W tiles ([128 rows][64 halfs] = 16 KB) by LDS-DMA into the XOR-swizzled unpadded image, two taps ahead;
//   * A tiles through registers: each thread owns 4 slots (row, 8 channels) per tap = 8 global_load_dwordx4; TWO register sets, so that
//     the loads of tap t+3 are issued while tap t is multiplied and consumed (converted + written to LDS, rows padded to 144 B) during
//     tap t+2: about two taps of flight time.  With the matrix pipe at fp16 speed a tap is ~0.5 us, so one tap of prefetch distance (the
//     conv kernel's 1x1 slabs) is shorter than the memory latency;
//   * every load is unconditional (clamped tile index) and every wait is a counted vmcnt: at the barrier of tap t the 8 A loads
//     issued after the weight DMA stay in flight (vmcnt(8)); before converting, the DMA pair and the next 8 loads do (vmcnt(10)).
// Scope: taps == 1, M % 256 == 0, K % 64 == 0 (each source), weights padded to 128 rows.  Fused epilogue = the shared one (bias,
// residual, scale, SiLU, GEGLU gate, GroupNorm column sums).
#include "pipe_common.h"

namespace igemm {
namespace {

__global__ void __launch_bounds__(512, 2) gemm_f16_kernel(const KParams p) {
    constexpr int NS = 4;                                   // (row, 8-channel) slots per thread and tap
    constexpr unsigned BS_B = 2 * 128 * 32 * 4;             // two weight buffers of [128][64 halfs]
    constexpr unsigned A_B = 256 * 144;                     // one A buffer: 256 rows of 64 halfs + 16 B pad
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned lds0 = lds_addr2(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    int mt, nt;
    if (!decode_tile(blockIdx.x, p.mtiles, p.ntiles, mt, nt, 0)) return;
    const int m0 = mt * 256, n0 = nt * 128;
    const int ld_row = tid >> 3, ld_col = (tid & 7) * 8;
    const int KT = p.K / 64;

    // ---- addresses ---------------------------------------------------------------------------------------------------------------
    const unsigned st_base = lds0 + BS_B + (unsigned)ld_row * 144 + (unsigned)(tid & 7) * 16;          // + j * 9216 (+ A_B)
    unsigned abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = lds0 + BS_B + (unsigned)(wr * 64 + i * 32 + (lane & 31)) * 144 + (unsigned)(lane >> 5) * 16;
    const int b_row = wc * 64 + (lane & 31);
    const unsigned c0 = (unsigned)((lane >> 5) ^ (((lane & 31) >> 1) & 7));
    unsigned bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bq[ks] = lds0 + (unsigned)b_row * 128 + ((c0 ^ (2u * ks)) * 16);
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

    // ---- A staging: two register sets (tap s uses set s & 1) --------------------------------------------------------------------
    f32x4 hreg[2][NS][2];
    auto load_tap = [&](auto setc, int s) {
        constexpr int S = decltype(setc)::value;
        const int k = min(s, KT - 1) * 64;
        const bool first = k < p.c0;
        const float* src = (first ? p.a0 + k : p.a1 + (k - p.c0)) + ld_col;
        const int ld = first ? p.lda0 : p.lda1;
        static_for<NS>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float* ptr = src + (size_t)(m0 + ld_row + 64 * j) * ld;
            hreg[S][j][0] = gld16(ptr);
            hreg[S][j][1] = gld16(ptr + 4);
        });
    };
    auto convert_slot = [&](auto setc, auto jc, unsigned st_addr) {
        constexpr int S = decltype(setc)::value, j = decltype(jc)::value;
        f32x4 &lo = hreg[S][j][0], &hi = hreg[S][j][1];
        asm volatile("" : "+v"(lo), "+v"(hi));
        f32x4 cvt;
        cvt[0] = pack_h2(lo[0], lo[1]); cvt[1] = pack_h2(lo[2], lo[3]);
        cvt[2] = pack_h2(hi[0], hi[1]); cvt[3] = pack_h2(hi[2], hi[3]);
        if (j == 0) DS_RACE_SKEW(wave);
        lds_wr<j * 9216>(st_addr, cvt);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue.  VMEM issue order (the counted waits below rely on it): DMA(0), loads(0) | wait | loads(1), DMA(1), loads(2) ----
    b_dma(0, 0);
    load_tap(IC<0>{}, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<NS>([&](auto jc) { convert_slot(IC<0>{}, jc, st_base); });
    load_tap(IC<1>{}, 1);
    b_dma(min(1, KT - 1), 1);
    load_tap(IC<0>{}, 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // tap 0 needs A tile 0 (LDS stores, here) and W tile 0 (waited above)
    __builtin_amdgcn_s_barrier();

    Frag2 P_, Q_;
    int kt = 0;
    frag_read2<0>(P_, abase[0], abase[1], bq[0]);
#define DSG_MH(i, j, f) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (f).a##i), __builtin_bit_cast(h8, (f).b##j), acc[i][j], 0, 0, 0)
#define DSG_GROUP(f, hook) DSG_MH(0, 0, f); hook(IC<0>{}); DSG_MH(0, 1, f); hook(IC<1>{}); DSG_MH(1, 0, f); hook(IC<2>{}); DSG_MH(1, 1, f); hook(IC<3>{});
    // One tap (64 channels).  PAR = kt & 1 (compile time: it selects the register set): tap kt multiplies A buffer PAR / W buffer PAR,
    // converts register set PAR ^ 1 (tap kt + 1) into A buffer PAR ^ 1, and after the barrier refills that set with tap kt + 3.
    auto tap = [&](auto parc) {
        Frag2 &P = P_, &Q = Q_;
        constexpr int PAR = decltype(parc)::value, OTH = PAR ^ 1;
        const unsigned va0 = abase[0] + PAR * A_B, va1 = abase[1] + PAR * A_B;
        const unsigned cb = PAR * 16384u;
        const unsigned st_addr = st_base + OTH * A_B;
        auto nohook = [&](auto) {};
        auto hookA = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            DS2_FENCE();
            if constexpr (k == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");     // set OTH landed; DMA pair + 8 newer loads in flight
            convert_slot(IC<OTH>{}, IC<k>{}, st_addr);
            DS2_FENCE();
        };
        frag_read2<32>(Q, va0, va1, bq[1] + cb);
        DS2_FRAG_WAIT(4, P);
        DSG_GROUP(P, nohook)
        DS2_FENCE();
        frag_read2<64>(P, va0, va1, bq[2] + cb);
        DS2_FRAG_WAIT(4, Q);
        DSG_GROUP(Q, nohook)
        DS2_FENCE();
        frag_read2<96>(Q, va0, va1, bq[3] + cb);
        DS2_FRAG_WAIT(4, P);
        DSG_GROUP(P, hookA)                                  // conversions as late as the tap allows: the most flight time for the loads
        DS2_FENCE();
        DS2_FRAG_WAIT(0, Q);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // W tile kt + 1 landed; the 8 A loads issued after it stay in flight
        __builtin_amdgcn_s_barrier();
        auto hookD = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
            DS2_FENCE();
            if constexpr (k == 0) b_dma(min(kt + 2, KT - 1), PAR);
            else if constexpr (k == 1) frag_read2<0>(P, abase[0] + OTH * A_B, abase[1] + OTH * A_B, bq[0] + OTH * 16384u);
            else if constexpr (k == 2) load_tap(IC<OTH>{}, kt + 3);
            DS2_FENCE();
        };
        DSG_GROUP(Q, hookD)
        DS2_FENCE();
        ++kt;
    };
    int t = 0;
    for (; t + 1 < KT; t += 2) { tap(IC<0>{}); tap(IC<1>{}); }
    if (t < KT) tap(IC<0>{});
#undef DSG_MH
#undef DSG_GROUP
    DS2_FRAG_WAIT(0, P_);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<NS>([&](auto jc) {
        f32x4 &a = hreg[0][decltype(jc)::value][0], &b = hreg[0][decltype(jc)::value][1], &c = hreg[1][decltype(jc)::value][0],
              &d = hreg[1][decltype(jc)::value][1];
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    });
    __syncthreads();
    epilogue<0, true>(p, acc, smem + wave * 32 * EPI_LD, lane, m0 + wr * 64, n0 + wc * 64, p.out);
}

}  // namespace

bool gemm_f16_applicable(const KParams& p) {
    if (p.taps != 1 || p.stride != 1 || p.ec0 || p.ec1 || p.norm) return false;
    if (p.M % 256 || p.K % 64 || (p.c1 > 0 && p.c0 % 64) || p.K < 64) return false;
    const int ntiles = (p.N + 127) / 128;
    return p.nrows_b >= ntiles * 128;
}

int launch_gemm_f16(KParams& p, hipStream_t stream) {
    constexpr int SMEM = 2 * 128 * 32 * 4 + 2 * 256 * 144;          // 106,496 B
    DS_ENSURE_DYN_LDS((&gemm_f16_kernel), SMEM);
    p.mtiles = p.M / 256;
    p.ntiles = (p.N + 127) / 128;
    p.splits = 1;
    hipLaunchKernelGGL(gemm_f16_kernel, dim3(grid_1d(p.mtiles, p.ntiles)), dim3(512), SMEM, stream, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace igemm
