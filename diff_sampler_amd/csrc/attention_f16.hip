// Fused softmax(Q K^T * scale) V with fp16 operands on v_mfma_f32_32x32x16_f16 -- the attention of the reference's fp16 / autocast
// mode: networks_edm.py:98-110 (AttentionOp: fp16 q, k multiplied with fp32 accumulation, fp32 softmax, weights cast back to fp16,
// fp16 w v product) and ldm/modules/attention.py:168-194 under torch.autocast (diff-solvers-main/sample.py:296).
//
// q / k / v are fp32 tensors rounded to fp16 (RNE) while they are staged, or -- ds_attn_args.in_f16, the reference's own storage: its
// qkv projection emits fp16 -- fp16 tensors staged as they are (bit 0: q, bit 1: k and v); the scores, the softmax statistics and both
// accumulators are fp32.  With fp32 q the factor scale * log2(e) is folded into q before it is rounded; with fp16 q (already rounded by
// its producer) it multiplies the fp32 scores instead, inside the exponent's fma -- no second rounding of q.  Same transposed online-softmax formulation as attention.hip
// (a query is a lane; the C/D layout of S^T is directly the B operand of the P V product), re-tiled for the 16-deep fp16 MFMA:
//   * workgroup = 256 queries of one (image, head), 8 waves x 32 queries; K/V stream through LDS in 64-key tiles, DOUBLE buffered
//     (one barrier per tile), register-prefetched one tile ahead;
//   * S^T[key, q] = sum_d K[key, d] Q[q, d]:   A = K tile rows (ds_read_b128 = 8 halfs of one key), B = Q (registers, scale * log2 e
//     folded in before rounding); d is zero padded to a multiple of 16 (d = 40 -> 48: 3 MFMAs per 32 keys instead of 20 fp32 ones);
//   * O^T[d, q] += sum_key V[key, d] P^T[key, q]:  the contraction index must be contiguous in the A operand's registers, so V is
//     TRANSPOSED while it is staged (each thread transposes a 4-key x 4-channel patch in registers, four ds_write_b64) into
//     Vt[d][position(key)], where position() swaps key bits 2 and 3: with that order the 8 keys lane-half hb contracts in MFMA
//     step s are exactly the registers 8 (s & 1) ... + 7 of the S^T accumulator it already holds (keys (r & 3) + 8 (r >> 2) + 4 hb),
//     so P goes from the exp2 to the P V product through one v_cvt_pk_f16_f32 per pair, no shuffles and no LDS round trip;
//   * the O^T rescale runs only on tiles where some lane's running maximum moved (wave-uniform branch);
//   * XCD-aware 1-D grid: linear workgroup id % 8 is the XCD, so the (image, head) index is id % 8 + 8 * (...) and all query blocks
//     of one (image, head) share an L2 (its K/V, <= 1.3 MB at 4096 x 40, are fetched from HBM once, not once per XCD).
// Round 5 (SQ counters of round 4: 207 VALU per 14 MFMAs per tile, matrix pipe busy 0.25 -- the kernel is VALU-bound, not matrix-bound):
//   * the softmax DENOMINATOR comes out of the matrix pipe: head sizes that are not a multiple of 32 (d = 40, 80: every SD-1.5 self- and
//     cross-attention at 64x64 / 32x32) pad O^T to whole 32-row blocks anyway, so row d of V^T is set to ones once and the P V product
//     accumulates sum_key P[key, q] in that row for free -- the 32 v_add_f32 per tile and lane of the running sum are gone, and the sum is
//     of the fp16-rounded weights the product actually uses (the reference normalises in fp32 and rounds the weights afterwards,
//     networks_edm.py:108-109: either way the weights sum to 1 within 2**-11);
//   * the operand types (fp16 or fp32 q; fp16 or fp32 k / v) are template parameters: the staging code holds no run-time selects and the
//     registers of the unused path are not allocated (the four-waves-per-SIMD allocation at d <= 64 spilled 10 - 23 registers);
//   * waves 4 - 7 run at s_setprio 1 (MI355X_MICROARCH.md, "Two waves per SIMD", item 4: the second-dispatched half of an eight-wave
//     workgroup otherwise loses every VALU arbitration).  -DDS_ATTN_NOPRIO builds without it (A/B runs).
// Head sizes: d % 8 == 0, d <= 160 (registers / LDS); other sizes keep the fp32 kernel (ds_attention_f16_supported).
#include <type_traits>

#include "ds_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk2(float x, float y) {          // two fp32 -> two fp16 (round to nearest even) in one dword
    const h2 p = {(_Float16)x, (_Float16)y};
    return __builtin_bit_cast(unsigned, p);
}

template <int D, int INF16>
// Waves per SIMD the register allocation must leave room for.  Head sizes <= 64 at 4: TWO 512-thread workgroups per CU.  With one (the
// round-3 allocation: 144 / 147 VGPRs at d = 40 / 64) the two waves of a SIMD belong to the same workgroup and leave the same per-tile
// barrier together: they run their MFMA segments (Q K^T, P V) and their VALU segment (softmax: ~207 VALU per 14 MFMAs, SQ counters in
// profiles/r4_attn_f16_counters.json) in lockstep; four waves per SIMD also hide more of each other's latencies.  A second, independent
// workgroup on the CU is out of phase with the first.  (-DDS_ATTN_OCC2 builds the round-3 allocation: A/B runs.)
#ifdef DS_ATTN_OCC2
#define DS_ATTN_WAVES(D) 2
#else
#define DS_ATTN_WAVES(D) ((D) <= 64 ? 4 : 2)
#endif
__global__ void __launch_bounds__(512, DS_ATTN_WAVES(D)) flash_attn_f16_kernel(const ds_attn_args a, const int qblocks, const int pairs) {
    static_assert(!(D % 32) || ((D % 32) % 4 == 0), "ones row");
    constexpr int DP = (D + 15) / 16 * 16;           // contraction length of S^T, zero padded
    constexpr int NKS = DP / 16;
    constexpr int DB = (D + 31) / 32;                // 32-row blocks of O^T
    constexpr bool ONES = (D % 32) != 0;             // row D of V^T is all ones: O^T row D accumulates the softmax denominator
    constexpr int LB = D / 32, LR = ((D % 32) & 3) + 4 * ((D % 32) >> 3), LH = ((D % 32) >> 2) & 1;   // ... in ot[LB][LR] of lane half LH
    constexpr int KT = 64;                           // keys per tile
    constexpr int KLD = DP + 8;                      // halfs; row stride / 16 B odd: conflict-free ds_read_b128 over 32 rows
    constexpr int VLD = KT + 8;
    constexpr int KBYTES = KT * KLD * 2, VBYTES = DB * 32 * VLD * 2, TILE_B = KBYTES + VBYTES;
    constexpr int D8 = D / 8, D4 = D / 4;
    constexpr int KCH = KT * D8;                     // (key, 8 channels) chunks of a K tile
    constexpr int NLK = (KCH + 511) / 512;
    constexpr int VTS = (KT / 4) * D4;               // (4 keys, 4 channels) patches of a V tile
    constexpr int NLV = (VTS + 511) / 512;
    constexpr bool PREFETCH = D <= 96;               // larger heads: no registers left to hold a staged tile across the MFMAs
                                                     // (dropping the prefetch at d <= 64 to avoid the 10 - 23 spilled registers of the
                                                     // four-waves-per-SIMD allocation: 3 - 6 % slower, profiles/r4_attn_f16_occupancy_ab.txt)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    float* Es = reinterpret_cast<float*>(smem_b + 2 * TILE_B);      // epilogue transposition patches, 32 x 33 floats per wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = lane >> 5, l31 = lane & 31;
    const int L = blockIdx.x, slot = L >> 3;
    const int qb = slot % qblocks, pair = (slot / qblocks) * 8 + (L & 7);
    if (pair >= pairs) return;
    const int b = pair / a.heads, h = pair - b * a.heads;
    const int q0 = qb * 256 + wave * 32;
    const bool active = q0 < a.sq;
    const float* qp = a.q + (size_t)b * a.q_bs + h * D;
    const float* kp = a.k + (size_t)b * a.k_bs + h * D;
    const float* vp = a.v + (size_t)b * a.v_bs + h * D;

    if (DP != D) {          // the zero padding of K's contraction columns (never overwritten by the staging below)
        if (tid < 2 * KT) *reinterpret_cast<u32x4*>(smem_b + (tid >> 6) * TILE_B + ((tid & 63) * KLD + D) * 2) = u32x4{0u, 0u, 0u, 0u};
    }
    if (ONES) {             // V^T row D = 1.0 for all 64 key positions of both tile buffers (the staging writes rows < D only); 16 B per thread
        if (tid < 16) *reinterpret_cast<u32x4*>(smem_b + (tid >> 3) * TILE_B + KBYTES + (D * VLD + (tid & 7) * 8) * 2) =
            u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    }

    const float sc = a.scale * 1.4426950408889634f;
    constexpr bool q16 = INF16 & 1, kv16 = INF16 & 2;            // fp16 sources: leading dimensions / batch strides in halfs
    const float ss = q16 ? sc : 1.0f;                             // factor applied to the fp32 scores (fp32 q carries it already)
    h8 qf[NKS];
    {
        const int qrow = min(q0 + l31, a.sq - 1);
        const float* qr = qp + (size_t)qrow * a.ldq + 8 * hb;
        const _Float16* qr16 = reinterpret_cast<const _Float16*>(a.q) + (size_t)b * a.q_bs + h * D + (size_t)qrow * a.ldq + 8 * hb;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 w = {0u, 0u, 0u, 0u};
            if (16 * ks + 8 * hb < D) {
                if constexpr (q16) w = *reinterpret_cast<const u32x4*>(qr16 + 16 * ks);
                else {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(qr + 16 * ks) * sc;
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(qr + 16 * ks + 4) * sc;
                    w = u32x4{pk2(lo[0], lo[1]), pk2(lo[2], lo[3]), pk2(hi[0], hi[1]), pk2(hi[2], hi[3])};
                }
            }
            qf[ks] = __builtin_bit_cast(h8, w);
        }
    }
    const _Float16* kp16 = reinterpret_cast<const _Float16*>(a.k) + (size_t)b * a.k_bs + h * D;
    const _Float16* vp16 = reinterpret_cast<const _Float16*>(a.v) + (size_t)b * a.v_bs + h * D;
    f32x16 ot[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    float m = -1e30f, l = 0.f;

    f32x4 kr[NLK][kv16 ? 1 : 2], vr[NLV][4];
    auto gload = [&](int t) {
#pragma unroll
        for (int j = 0; j < NLK; ++j) {
            const int idx = tid + 512 * j;
            if (NLK * 512 == KCH || idx < KCH) {
                const int row = idx / D8, c8 = idx - row * D8;
                const size_t off = (size_t)min(t * KT + row, a.skv - 1) * a.ldk + c8 * 8;
                if constexpr (kv16) kr[j][0] = *reinterpret_cast<const f32x4*>(kp16 + off);          // eight halfs, staged as they are
                else {
                    kr[j][0] = *reinterpret_cast<const f32x4*>(kp + off);
                    kr[j][1] = *reinterpret_cast<const f32x4*>(kp + off + 4);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NLV; ++j) {
            const int idx = tid + 512 * j;
            if (NLV * 512 == VTS || idx < VTS) {
                const int g = idx & 15, d4 = idx >> 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const size_t off = (size_t)min(t * KT + 4 * g + i, a.skv - 1) * a.ldv + 4 * d4;
                    if constexpr (kv16) {          // four halfs of one key in the first two dwords
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        const f32x2_t w = *reinterpret_cast<const f32x2_t*>(vp16 + off);
                        vr[j][i] = f32x4{w[0], w[1], 0.f, 0.f};
                    } else vr[j][i] = *reinterpret_cast<const f32x4*>(vp + off);
                }
            }
        }
    };
    auto sstore = [&](int buf) {
        unsigned char* base = smem_b + buf * TILE_B;
        DS_RACE_SKEW(wave);
#pragma unroll
        for (int j = 0; j < NLK; ++j) {
            const int idx = tid + 512 * j;
            if (NLK * 512 == KCH || idx < KCH) {
                const int row = idx / D8, c8 = idx - row * D8;
                const f32x4 lo = kr[j][0], hi = kr[j][kv16 ? 0 : 1];
                if constexpr (kv16) *reinterpret_cast<u32x4*>(base + (row * KLD + 8 * c8) * 2) = __builtin_bit_cast(u32x4, lo);
                else *reinterpret_cast<u32x4*>(base + (row * KLD + 8 * c8) * 2) =
                    u32x4{pk2(lo[0], lo[1]), pk2(lo[2], lo[3]), pk2(hi[0], hi[1]), pk2(hi[2], hi[3])};
            }
        }
#pragma unroll
        for (int j = 0; j < NLV; ++j) {
            const int idx = tid + 512 * j;
            if (NLV * 512 == VTS || idx < VTS) {
                const int g = idx & 15, d4 = idx >> 4;
                const int pos = 4 * ((g & ~3) | ((g & 1) << 1) | ((g & 2) >> 1));      // key bits 2 and 3 swapped
                if constexpr (kv16) {          // the same 4-key x 4-channel transposition on halfs: byte permutes of the keys' dwords
                    unsigned kd[4][2];                                               // key i: dword 0 = channels (0, 1), dword 1 = channels (2, 3)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {          // (through scalars: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0 with this compiler)
                        const float f0 = vr[j][i][0], f1 = vr[j][i][1];
                        kd[i][0] = __float_as_uint(f0); kd[i][1] = __float_as_uint(f1);
                    }
                    auto lo2 = [](unsigned x, unsigned y) { return (x & 0xffffu) | (y << 16); };          // low halfs of two keys
                    auto hi2 = [](unsigned x, unsigned y) { return (x >> 16) | (y & 0xffff0000u); };      // high halfs
                    unsigned char* vt = base + KBYTES + (4 * d4 * VLD + pos) * 2;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        *reinterpret_cast<u32x2*>(vt + (2 * e) * VLD * 2) = u32x2{lo2(kd[0][e], kd[1][e]), lo2(kd[2][e], kd[3][e])};
                        *reinterpret_cast<u32x2*>(vt + (2 * e + 1) * VLD * 2) = u32x2{hi2(kd[0][e], kd[1][e]), hi2(kd[2][e], kd[3][e])};
                    }
                } else
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *reinterpret_cast<u32x2*>(base + KBYTES + ((4 * d4 + c) * VLD + pos) * 2) =
                        u32x2{pk2(vr[j][0][c], vr[j][1][c]), pk2(vr[j][2][c], vr[j][3][c])};
            }
        }
    };

    const int ntiles = (a.skv + KT - 1) / KT;
#ifndef DS_ATTN_NOPRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    gload(0);
    sstore(0);
    if (PREFETCH && ntiles > 1) gload(1);
    __syncthreads();
    const unsigned kfrag = (unsigned)(l31 * KLD + 8 * hb) * 2;
    const unsigned vfrag = (unsigned)KBYTES + (unsigned)(l31 * VLD + 8 * hb) * 2;
    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* base = smem_b + (t & 1) * TILE_B;
        f32x16 st[2];
        h8 pf[4];
        if (active) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const h8 kv = *reinterpret_cast<const h8*>(base + kfrag + (kb * 32 * KLD + 16 * ks) * 2);
                    st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kv, qf[ks], st[kb], 0, 0, 0);
                }
            }
            if (t == ntiles - 1 && (a.skv & (KT - 1))) {
                // a REAL (wave-uniform) branch: without the asm statement the compiler if-converts this block into 32 v_cndmask per tile -- a
                // fifth of the loop's VALU instructions, on every tile, for a mask that can only matter on a ragged LAST tile (round 6)
                asm volatile("; ragged last key tile" ::: "memory");
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + kb * 32 + 4 * hb + (r & 3) + 8 * (r >> 2) >= a.skv) st[kb][r] = -1e30f;
            }
            float mx = st[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m, mx * ss);                  // running maximum in scaled units (ss > 0)
            const bool moved = mn > m;
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kb][r], ss, -mn));
                    if (!ONES) rs += st[kb][r];
                }
            if (!ONES) l = l * alpha + rs;
            if (__any(moved)) {
#pragma unroll
                for (int i = 0; i < DB; ++i) ot[i] *= alpha;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x16& p = st[s >> 1];
                const int r0 = 8 * (s & 1);
                pf[s] = __builtin_bit_cast(h8, (u32x4{pk2(p[r0], p[r0 + 1]), pk2(p[r0 + 2], p[r0 + 3]), pk2(p[r0 + 4], p[r0 + 5]),
                                                      pk2(p[r0 + 6], p[r0 + 7])}));
            }
        }
        if (t + 1 < ntiles) {            // next tile into the other buffer (nobody reads it before the barrier below)
            if (!PREFETCH) gload(t + 1);
            sstore((t + 1) & 1);
            if (PREFETCH && t + 2 < ntiles) gload(t + 2);
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const h8 vv = *reinterpret_cast<const h8*>(base + vfrag + (i * 32 * VLD + 16 * s) * 2);
                    ot[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vv, pf[s], ot[i], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    if (!active) return;

    if (ONES) {             // the denominator sits in O^T row D: register LR of block LB in the lanes of half LH; both halves of a query need it
        l = __shfl(ot[LB][LR], l31 + 32 * LH);
    } else l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    float* patch = Es + wave * (32 * 33);
    float* op = a.out + (size_t)b * a.o_bs + h * D;
#pragma unroll
    for (int i = 0; i < DB; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) patch[l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * hb] = ot[i][r] * inv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int c4 = (lane & 7) * 4;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int q = pass * 8 + (lane >> 3);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = patch[q * 33 + c4 + j];
            if (q0 + q < a.sq && i * 32 + c4 < D) {
                if (a.out_f16) {
                    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
                    const h4_t hv = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                    *reinterpret_cast<h4_t*>(reinterpret_cast<_Float16*>(a.out) + (size_t)b * a.o_bs + h * D + (size_t)(q0 + q) * a.ldo + i * 32 + c4) = hv;
                } else
                *reinterpret_cast<f32x4*>(op + (size_t)(q0 + q) * a.ldo + i * 32 + c4) = o;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- Two query blocks per wave (round 6) -----------------------------------------------------------------------------------------------
// flash_attn_f16_kernel gives a wave ONE 32-query S^T block: per 64-key tile it issues 6 - 8 MFMAs (Q K^T), then ~110 VALU of softmax that
// depend on them, then 8 MFMAs (P V) that depend on those -- inside a wave nothing overlaps, and the SQ counters of round 5 read exactly
// like the SUM of the two phases (matrix pipe busy 0.43 with four waves per SIMD; profiles/r5_fp16_sq_counters_summary.txt).
// Here a wave owns TWO 32-query blocks a / b of the same (image, head) and runs them skewed by half a phase on the same key tile:
//     S_a, S_b    12 - 16 MFMAs on four independent accumulator chains; each K fragment is read from LDS ONCE and multiplies both blocks
//     softmax(a)  VALU
//     P V (a)     8 MFMAs issued ...            |  ... and while they execute:  softmax(b)  VALU
//     P V (b)     8 MFMAs issued ...            |  ... staging of the next key tile (conversions / transposition, LDS writes)
// Workgroup = 4 waves x 64 queries = 256 queries (the same query blocks per workgroup and the same K / V staging per query as the one-block
// kernel's 8 x 32), ~50 KB of LDS and < 256 VGPRs: two workgroups per CU, i.e. the two waves of a SIMD belong to DIFFERENT workgroups and are
// out of phase.  Same arithmetic per query as the one-block kernel (the same MFMA operands in the same order, the same softmax
// expressions): results are bit-identical to it -- tests/test_hip_kernels.py compares the two.  Head sizes 32 / 40 / 64 (registers).
// MEASURED SLOWER (use_x2 below): never the library's choice, reachable through ds_attn_args.variant = 2 only.
template <int D, int INF16>
__global__ void __launch_bounds__(256, 2) flash_attn_f16x2_kernel(const ds_attn_args a, const int qblocks, const int pairs) {
    static_assert(D <= 64, "two query blocks per wave: head sizes up to 64");
    constexpr int DP = (D + 15) / 16 * 16, NKS = DP / 16, DB = (D + 31) / 32;
    constexpr bool ONES = (D % 32) != 0;
    constexpr int LB = D / 32, LR = ((D % 32) & 3) + 4 * ((D % 32) >> 3), LH = ((D % 32) >> 2) & 1;
    constexpr int KT = 64, KLD = DP + 8, VLD = KT + 8;
    constexpr int KBYTES = KT * KLD * 2, VBYTES = DB * 32 * VLD * 2, TILE_B = KBYTES + VBYTES;
    constexpr int D8 = D / 8, D4 = D / 4, NT = 256;
    constexpr int KCH = KT * D8, NLK = (KCH + NT - 1) / NT;
    constexpr int VTS = (KT / 4) * D4, NLV = (VTS + NT - 1) / NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    float* Es = reinterpret_cast<float*>(smem_b + 2 * TILE_B);      // epilogue transposition patches, 32 x 33 floats per wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = lane >> 5, l31 = lane & 31;
    const int L = blockIdx.x, slot = L >> 3;
    const int qb = slot % qblocks, pair = (slot / qblocks) * 8 + (L & 7);
    if (pair >= pairs) return;
    const int b = pair / a.heads, h = pair - b * a.heads;
    const int q0 = qb * 256 + wave * 64;                            // block x covers queries q0 + 32 x .. + 31
    const bool active = q0 < a.sq;
    const float* qp = a.q + (size_t)b * a.q_bs + h * D;
    const float* kp = a.k + (size_t)b * a.k_bs + h * D;
    const float* vp = a.v + (size_t)b * a.v_bs + h * D;

    if (DP != D) {          // zero padding of K's contraction columns (never overwritten by the staging)
        if (tid < 2 * KT) *reinterpret_cast<u32x4*>(smem_b + (tid >> 6) * TILE_B + ((tid & 63) * KLD + D) * 2) = u32x4{0u, 0u, 0u, 0u};
    }
    if (ONES) {             // V^T row D = 1.0 for all 64 key positions of both tile buffers
        if (tid < 16) *reinterpret_cast<u32x4*>(smem_b + (tid >> 3) * TILE_B + KBYTES + (D * VLD + (tid & 7) * 8) * 2) =
            u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    }
    const float sc = a.scale * 1.4426950408889634f;
    constexpr bool q16 = INF16 & 1, kv16 = INF16 & 2;
    const float ss = q16 ? sc : 1.0f;
    h8 qf[2][NKS];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int qrow = min(q0 + 32 * x + l31, a.sq - 1);
        const float* qr = qp + (size_t)qrow * a.ldq + 8 * hb;
        const _Float16* qr16 = reinterpret_cast<const _Float16*>(a.q) + (size_t)b * a.q_bs + h * D + (size_t)qrow * a.ldq + 8 * hb;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            u32x4 w = {0u, 0u, 0u, 0u};
            if (16 * ks + 8 * hb < D) {
                if constexpr (q16) w = *reinterpret_cast<const u32x4*>(qr16 + 16 * ks);
                else {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(qr + 16 * ks) * sc;
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(qr + 16 * ks + 4) * sc;
                    w = u32x4{pk2(lo[0], lo[1]), pk2(lo[2], lo[3]), pk2(hi[0], hi[1]), pk2(hi[2], hi[3])};
                }
            }
            qf[x][ks] = __builtin_bit_cast(h8, w);
        }
    }
    const _Float16* kp16 = reinterpret_cast<const _Float16*>(a.k) + (size_t)b * a.k_bs + h * D;
    const _Float16* vp16 = reinterpret_cast<const _Float16*>(a.v) + (size_t)b * a.v_bs + h * D;
    f32x16 ot[2][DB];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[x][i][r] = 0.f;
    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};

    f32x4 kr[NLK][kv16 ? 1 : 2], vr[NLV][4];
    auto gload = [&](int t) {
#pragma unroll
        for (int j = 0; j < NLK; ++j) {
            const int idx = tid + NT * j;
            if (NLK * NT == KCH || idx < KCH) {
                const int row = idx / D8, c8 = idx - row * D8;
                const size_t off = (size_t)min(t * KT + row, a.skv - 1) * a.ldk + c8 * 8;
                if constexpr (kv16) kr[j][0] = *reinterpret_cast<const f32x4*>(kp16 + off);
                else {
                    kr[j][0] = *reinterpret_cast<const f32x4*>(kp + off);
                    kr[j][1] = *reinterpret_cast<const f32x4*>(kp + off + 4);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NLV; ++j) {
            const int idx = tid + NT * j;
            if (NLV * NT == VTS || idx < VTS) {
                const int g = idx & 15, d4 = idx >> 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const size_t off = (size_t)min(t * KT + 4 * g + i, a.skv - 1) * a.ldv + 4 * d4;
                    if constexpr (kv16) {
                        typedef float f32x2_t __attribute__((ext_vector_type(2)));
                        const f32x2_t w = *reinterpret_cast<const f32x2_t*>(vp16 + off);
                        vr[j][i] = f32x4{w[0], w[1], 0.f, 0.f};
                    } else vr[j][i] = *reinterpret_cast<const f32x4*>(vp + off);
                }
            }
        }
    };
    auto sstore = [&](int buf) {
        unsigned char* base = smem_b + buf * TILE_B;
        DS_RACE_SKEW(wave);
#pragma unroll
        for (int j = 0; j < NLK; ++j) {
            const int idx = tid + NT * j;
            if (NLK * NT == KCH || idx < KCH) {
                const int row = idx / D8, c8 = idx - row * D8;
                const f32x4 lo = kr[j][0], hi = kr[j][kv16 ? 0 : 1];
                if constexpr (kv16) *reinterpret_cast<u32x4*>(base + (row * KLD + 8 * c8) * 2) = __builtin_bit_cast(u32x4, lo);
                else *reinterpret_cast<u32x4*>(base + (row * KLD + 8 * c8) * 2) =
                    u32x4{pk2(lo[0], lo[1]), pk2(lo[2], lo[3]), pk2(hi[0], hi[1]), pk2(hi[2], hi[3])};
            }
        }
#pragma unroll
        for (int j = 0; j < NLV; ++j) {
            const int idx = tid + NT * j;
            if (NLV * NT == VTS || idx < VTS) {
                const int g = idx & 15, d4 = idx >> 4;
                const int pos = 4 * ((g & ~3) | ((g & 1) << 1) | ((g & 2) >> 1));      // key bits 2 and 3 swapped
                if constexpr (kv16) {
                    unsigned kd[4][2];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float f0 = vr[j][i][0], f1 = vr[j][i][1];
                        kd[i][0] = __float_as_uint(f0); kd[i][1] = __float_as_uint(f1);
                    }
                    auto lo2 = [](unsigned x, unsigned y) { return (x & 0xffffu) | (y << 16); };
                    auto hi2 = [](unsigned x, unsigned y) { return (x >> 16) | (y & 0xffff0000u); };
                    unsigned char* vt = base + KBYTES + (4 * d4 * VLD + pos) * 2;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        *reinterpret_cast<u32x2*>(vt + (2 * e) * VLD * 2) = u32x2{lo2(kd[0][e], kd[1][e]), lo2(kd[2][e], kd[3][e])};
                        *reinterpret_cast<u32x2*>(vt + (2 * e + 1) * VLD * 2) = u32x2{hi2(kd[0][e], kd[1][e]), hi2(kd[2][e], kd[3][e])};
                    }
                } else
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    *reinterpret_cast<u32x2*>(base + KBYTES + ((4 * d4 + c) * VLD + pos) * 2) =
                        u32x2{pk2(vr[j][0][c], vr[j][1][c]), pk2(vr[j][2][c], vr[j][3][c])};
            }
        }
    };

    const int ntiles = (a.skv + KT - 1) / KT;
    gload(0);
    sstore(0);
    if (ntiles > 1) gload(1);
    __syncthreads();
    const unsigned kfrag = (unsigned)(l31 * KLD + 8 * hb) * 2;
    const unsigned vfrag = (unsigned)KBYTES + (unsigned)(l31 * VLD + 8 * hb) * 2;
    for (int t = 0; t < ntiles; ++t) {
        const unsigned char* base = smem_b + (t & 1) * TILE_B;
        f32x16 st[2][2];                                           // [block][32-key half of the tile]
        h8 pf[4];
        // softmax of one block on its scores: running maximum / rescale, un-normalised weights -> the fp16 operand registers of P V
        auto softmax = [&](auto xc) {
            constexpr int x = decltype(xc)::value;
            if (t == ntiles - 1 && (a.skv & (KT - 1))) {
                asm volatile("; ragged last key tile" ::: "memory");        // a real branch, not 32 v_cndmask per tile (see flash_attn_f16_kernel)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (t * KT + kb * 32 + 4 * hb + (r & 3) + 8 * (r >> 2) >= a.skv) st[x][kb][r] = -1e30f;
            }
            float mx = st[x][0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[x][kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float mn = fmaxf(m[x], mx * ss);
            const bool moved = mn > m[x];
            const float alpha = __builtin_amdgcn_exp2f(m[x] - mn);
            m[x] = mn;
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    st[x][kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[x][kb][r], ss, -mn));
                    if (!ONES) rs += st[x][kb][r];
                }
            if (!ONES) l[x] = l[x] * alpha + rs;
            if (__any(moved)) {
#pragma unroll
                for (int i = 0; i < DB; ++i) ot[x][i] *= alpha;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const f32x16& p = st[x][s4 >> 1];
                const int r0 = 8 * (s4 & 1);
                pf[s4] = __builtin_bit_cast(h8, (u32x4{pk2(p[r0], p[r0 + 1]), pk2(p[r0 + 2], p[r0 + 3]), pk2(p[r0 + 4], p[r0 + 5]),
                                                       pk2(p[r0 + 6], p[r0 + 7])}));
            }
        };
        auto pv = [&](auto xc) {
            constexpr int x = decltype(xc)::value;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < DB; ++i) {                     // the DB accumulator chains alternate
                    const h8 vv = *reinterpret_cast<const h8*>(base + vfrag + (i * 32 * VLD + 16 * s4) * 2);
                    ot[x][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vv, pf[s4], ot[x][i], 0, 0, 0);
                }
        };
        if (active) {
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[x][kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {                   // one K fragment, both blocks: four accumulator chains in rotation
                    const h8 kv = *reinterpret_cast<const h8*>(base + kfrag + (kb * 32 * KLD + 16 * ks) * 2);
                    st[0][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kv, qf[0][ks], st[0][kb], 0, 0, 0);
                    st[1][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kv, qf[1][ks], st[1][kb], 0, 0, 0);
                }
            softmax(std::integral_constant<int, 0>{});
            pv(std::integral_constant<int, 0>{});                  // issued; executes under the softmax of block b
            softmax(std::integral_constant<int, 1>{});
            pv(std::integral_constant<int, 1>{});                  // issued; executes under the staging below
        }
        if (t + 1 < ntiles) {            // next tile into the other buffer (nobody reads it before the barrier below)
            sstore((t + 1) & 1);
            if (t + 2 < ntiles) gload(t + 2);
        }
        __syncthreads();
    }
    if (!active) return;

    float* patch = Es + wave * (32 * 33);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        float lx = l[x];
        if (ONES) lx = __shfl(ot[x][LB][LR], l31 + 32 * LH);
        else lx += __shfl_xor(lx, 32);
        const float inv = 1.0f / lx;
        float* op = a.out + (size_t)b * a.o_bs + h * D;
        const int qx = q0 + 32 * x;
#pragma unroll
        for (int i = 0; i < DB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[l31 * 33 + (r & 3) + 8 * (r >> 2) + 4 * hb] = ot[x][i][r] * inv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int c4 = (lane & 7) * 4;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int q = pass * 8 + (lane >> 3);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = patch[q * 33 + c4 + j];
                if (qx + q < a.sq && i * 32 + c4 < D) {
                    if (a.out_f16) {
                        typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
                        const h4_t hv = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                        *reinterpret_cast<h4_t*>(reinterpret_cast<_Float16*>(a.out) + (size_t)b * a.o_bs + h * D + (size_t)(qx + q) * a.ldo + i * 32 + c4) = hv;
                    } else
                    *reinterpret_cast<f32x4*>(op + (size_t)(qx + q) * a.ldo + i * 32 + c4) = o;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int D, int INF16>
int launch_x2(const ds_attn_args* a, hipStream_t stream) {
    if constexpr (D <= 64) {
        constexpr int DP = (D + 15) / 16 * 16, DB = (D + 31) / 32;
        constexpr int bytes = 2 * (64 * (DP + 8) * 2 + DB * 32 * 72 * 2) + 4 * 32 * 33 * (int)sizeof(float);
        DS_ENSURE_DYN_LDS((&flash_attn_f16x2_kernel<D, INF16>), bytes);
        const int qblocks = (a->sq + 255) / 256, pairs = a->batch * a->heads;
        const long long blocks = (long long)qblocks * ((pairs + 7) / 8) * 8;
        if (blocks > 0x7fffffffLL) return DS_E_SHAPE;
        hipLaunchKernelGGL((flash_attn_f16x2_kernel<D, INF16>), dim3((unsigned)blocks), dim3(256), bytes, stream, *a, qblocks, pairs);
        DS_CHECK_LAUNCH();
        return DS_OK;
    }
    return DS_E_SHAPE;
}

template <int D, int INF16>
int launch_t(const ds_attn_args* a, hipStream_t stream) {
    constexpr int DP = (D + 15) / 16 * 16, DB = (D + 31) / 32;
    constexpr int bytes = 2 * (64 * (DP + 8) * 2 + DB * 32 * 72 * 2) + 8 * 32 * 33 * (int)sizeof(float);
    static_assert(bytes <= 160 * 1024, "LDS");
    DS_ENSURE_DYN_LDS((&flash_attn_f16_kernel<D, INF16>), bytes);
    const int qblocks = (a->sq + 255) / 256, pairs = a->batch * a->heads;
    const long long blocks = (long long)qblocks * ((pairs + 7) / 8) * 8;
    if (blocks > 0x7fffffffLL) return DS_E_SHAPE;
    hipLaunchKernelGGL((flash_attn_f16_kernel<D, INF16>), dim3((unsigned)blocks), dim3(512), bytes, stream, *a, qblocks, pairs);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

// Two query blocks per wave: ONLY on request (ds_attn_args.variant = 2).  Measured slower than the one-block kernel on every shape of the two
// fp16 lines (profiles/r6_attn_f16_two_blocks_ab.txt: d = 40, S = 4 096, 32 latents 1.35 -> 1.44 ms; d = 64, S = 1 024 0.078 -> 0.085 ms; SD-1.5
// fp16 78.3 -> 76.9 images/s): at 210 - 242 VGPRs it runs two waves per SIMD against four, and what the in-wave skew hides is less than what the
// two extra waves hid.  Kept, bit-identical and tested, as the record of that experiment (docs/HISTORY.md H.3).
static bool use_x2(const ds_attn_args* a) { return a->d <= 64 && a->variant == 2; }

template <int D>
int launch(const ds_attn_args* a, hipStream_t stream) {
    if (a->variant == 2 && D > 64) return DS_E_SHAPE;
    if (use_x2(a)) {
        switch (a->in_f16 & 3) {
            case 0: return launch_x2<D, 0>(a, stream);
            case 1: return launch_x2<D, 1>(a, stream);
            case 2: return launch_x2<D, 2>(a, stream);
            default: return launch_x2<D, 3>(a, stream);
        }
    }
    switch (a->in_f16 & 3) {           // bit 0: fp16 q, bit 1: fp16 k and v
        case 0: return launch_t<D, 0>(a, stream);
        case 1: return launch_t<D, 1>(a, stream);
        case 2: return launch_t<D, 2>(a, stream);
        default: return launch_t<D, 3>(a, stream);
    }
}

}  // namespace

extern "C" int ds_attention_f16_supported(int d) {
    switch (d) {
        case 32: case 40: case 64: case 80: case 96: case 128: case 160: return 1;
        default: return 0;
    }
}

extern "C" int ds_attention_f16(const ds_attn_args* a, void* stream) {
    (void)hipGetLastError();
    if (!a || !a->q || !a->k || !a->v || !a->out) return DS_E_ARG;
    if (a->batch <= 0 || a->heads <= 0 || a->sq <= 0 || a->skv <= 0) return DS_E_ARG;
    if ((a->ldq & 3) || (a->ldk & 3) || (a->ldv & 3) || (a->ldo & 3) || (a->q_bs & 3) || (a->k_bs & 3) || (a->v_bs & 3) || (a->o_bs & 3))
        return DS_E_ALIGN;
    if ((a->in_f16 & ~3) || a->variant < 0 || a->variant > 2) return DS_E_ARG;
    if ((a->in_f16 & 1) && ((a->ldq & 7) || (a->q_bs & 7))) return DS_E_ALIGN;          // fp16 rows are read in 16-byte (q, k) / 8-byte (v) pieces
    if ((a->in_f16 & 2) && ((a->ldk & 7) || (a->k_bs & 7))) return DS_E_ALIGN;
    if (!ds_aligned16(a->q) || !ds_aligned16(a->k) || !ds_aligned16(a->v) || !ds_aligned16(a->out)) return DS_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    switch (a->d) {
        case 32: return launch<32>(a, s);
        case 40: return launch<40>(a, s);
        case 64: return launch<64>(a, s);
        case 80: return launch<80>(a, s);
        case 96: return launch<96>(a, s);
        case 128: return launch<128>(a, s);
        case 160: return launch<160>(a, s);
        default: return DS_E_SHAPE;
    }
}
