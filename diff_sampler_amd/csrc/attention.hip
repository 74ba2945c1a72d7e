// Fused softmax(Q K^T * scale) V on the gfx950 fp32 matrix pipe (online-softmax / "flash" formulation).
//
// Replaces, per attention layer, the reference's  einsum -> softmax -> einsum  chain (networks_edm.py:98-110 AttentionOp,
// :171-176 UNetBlock; ldm/modules/attention.py:168-194 CrossAttention) and the four launches + S x S score tensor of the
// unfused path (Q K^T GEMM, row softmax, V^T GEMM, P V GEMM): the scores never leave the CU.
//
// Work split: block = 128 queries of one (image, head), 4 waves x 32 queries; K/V are streamed in tiles of 32 or 64 keys
// through LDS (register-prefetched one tile ahead).  Everything is computed TRANSPOSED so that a query is a lane:
//     S^T[key, q] = sum_d K[key, d] Q[q, d]              A operand = K tile (LDS, ds_read_b128), B = Q (registers)
//     O^T[d,   q] += sum_key V[key, d] P^T[key, q]       A operand = V tile (LDS, ds_read_b32),  B = P^T (registers)
// In the 32x32 MFMA C/D layout the column is lane & 31, so lane (q, hb) holds 16 of the 32 scores of ITS query: the
// row maximum / sum are 16 register operations plus one cross-half shuffle, the O^T rescale is a per-lane scalar, and
// register r of P^T is directly the B operand of the P V MFMA whose two contracted keys are
// key(r, hb) = (r & 3) + 8 (r >> 2) + 4 hb  -- no data movement between the two GEMMs.
// exp is evaluated as exp2 with scale * log2(e) folded into Q at load time.
#include "ds_common.h"

namespace {

// DSPLIT (round 6; head sizes that are multiples of 128): the block is 32 queries and its four waves split the CHANNELS, not the queries --
// wave w contracts channels [w D/4, (w + 1) D/4) of Q K^T, the four partial S^T blocks are summed through LDS in wave order (every wave ends up
// with the same bits), every wave runs the softmax of the same 32 queries, and wave w accumulates and stores rows [w D/4, (w + 1) D/4) of O^T.
// A quarter of the MFMAs per wave on four times the waves: for launches that leave most SIMDs without a wave -- the single 256-wide head of the
// SongUNet nets at 8 images was 16 blocks = 64 waves, each 131 000 cycles of dependent fp32 MFMAs (92 us per launch, tools/time_plan_ops.py);
// the launcher takes it while the query-split grid has fewer than 1 024 waves.  The scores are summed in a different order (four 64-channel
// partial sums): results agree with the query-split kernel to fp32 rounding, not bit for bit -- ds_attn_args.variant 1 / 2 force either one.
template <int D, bool DSPLIT = false>
__global__ void __launch_bounds__(256) flash_attn_kernel(const ds_attn_args a) {
    // A head size that leaves 8 channels beyond the last full 32-row block of O^T (d = 40: SD-1.5's 64x64 stage) would spend
    // a whole MFMA block (16 MFMAs per key tile) on 8 useful rows; those VREM channels are accumulated by the vector ALU
    // instead (16 keys x 8 channels = 128 FMAs per lane and tile, V read as LDS broadcasts), in the shadow of the MFMAs.
    constexpr int VREM = (D > 32 && (D % 32) == 8) ? 8 : 0;
    constexpr int DB = VREM ? D / 32 : (D + 31) / 32;        // 32-row blocks of O^T on the matrix pipe
    constexpr int KLD = D + 4;               // (D + 4) / 4 odd: conflict-free ds_read_b128 of 32 distinct rows
    constexpr int VLD = DB * 32 + 8;         // rows 4 apart land 32 banks apart (and room for the VREM channels)
    constexpr int NQ4 = D / 8;
    constexpr int D4 = D / 4;
    static_assert(!DSPLIT || (D % 128 == 0), "channel split: four waves x a multiple of 32 channels");
    constexpr int DBW = DSPLIT ? DB / 4 : DB;               // 32-row blocks of O^T per wave
    constexpr int NQW = DSPLIT ? NQ4 / 4 : NQ4;             // Q fragments (8 channels each) per wave
    constexpr bool PREFETCH = D <= 96 || DSPLIT;            // larger heads: the O^T / Q registers leave no room for a staged tile (the channel split has)
    // 32-key blocks per K/V tile: 64-key tiles halve the barriers and the per-tile softmax bookkeeping (max / rescale) where
    // the registers still allow two waves per SIMD (d = 64: 7.07 -> 6.85 ms on ImageNet-64; d = 40 with 64-key tiles needs
    // 288 registers, drops to one wave per SIMD and loses 13 %)
    constexpr int KB = (D <= 64 && VREM == 0) ? 2 : 1;
    constexpr int KT = 32 * KB;
    constexpr int NLD = (KT / 4 * D + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + KT * KLD;
    float* Es = smem + KT * KLD + KT * VLD + 32;   // epilogue transposition patches, 32 x 33 floats per wave
    float* Rs = Es + 4 * 32 * 33;                  // DSPLIT: the four waves' partial S^T blocks, [wave][register][lane]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hb = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = DSPLIT ? blockIdx.x * 32 : blockIdx.x * 128 + wave * 32;
    const int d0 = DSPLIT ? wave * (D / 4) : 0;    // this wave's channels: the contraction range of Q K^T and the rows of O^T
    const bool active = q0 < a.sq;
    const float* qp = a.q + (size_t)b * a.q_bs + h * D;
    const float* kp = a.k + (size_t)b * a.k_bs + h * D;
    const float* vp = a.v + (size_t)b * a.v_bs + h * D;

    const float sc = a.scale * 1.4426950408889634f;
    f32x4 qf[NQW];
    {
        const int qrow = min(q0 + l31, a.sq - 1);
        const float* qr = qp + (size_t)qrow * a.ldq + 4 * hb + d0;
#pragma unroll
        for (int ks = 0; ks < NQW; ++ks) qf[ks] = *reinterpret_cast<const f32x4*>(qr + 8 * ks) * sc;
    }
    f32x16 ot[DBW];
#pragma unroll
    for (int i = 0; i < DBW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[i][r] = 0.f;
    float m = -1e30f, l = 0.f;
    float oe[VREM ? VREM : 1];
#pragma unroll
    for (int j = 0; j < (VREM ? VREM : 1); ++j) oe[j] = 0.f;

    f32x4 kr[NLD], vr[NLD];
    auto gload = [&](int t) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + 256 * j;
            if (NLD * 256 == KT / 4 * D || idx < KT / 4 * D) {
                const int row = idx / D4, c4 = idx - row * D4;
                const int key = min(t * KT + row, a.skv - 1);
                kr[j] = *reinterpret_cast<const f32x4*>(kp + (size_t)key * a.ldk + c4 * 4);
                vr[j] = *reinterpret_cast<const f32x4*>(vp + (size_t)key * a.ldv + c4 * 4);
            }
        }
    };
    auto sstore = [&]() {
        DS_RACE_SKEW(tid >> 6);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + 256 * j;
            if (NLD * 256 == KT / 4 * D || idx < KT / 4 * D) {
                const int row = idx / D4, c4 = idx - row * D4;
                *reinterpret_cast<f32x4*>(Ks + row * KLD + c4 * 4) = kr[j];
                *reinterpret_cast<f32x4*>(Vs + row * VLD + c4 * 4) = vr[j];
            }
        }
    };

    const int ntiles = (a.skv + KT - 1) / KT;
    if (PREFETCH) gload(0);
    const float* kfrag = Ks + l31 * KLD + 4 * hb + d0;
    for (int t = 0; t < ntiles; ++t) {
        if (!PREFETCH) gload(t);
        __syncthreads();                     // every wave is done with the previous tile
        sstore();
        __syncthreads();
        if (PREFETCH && t + 1 < ntiles) gload(t + 1);    // in flight during the MFMAs below
        if (!active) continue;

        f32x16 st[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < NQW; ++ks) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(kfrag + kb * 32 * KLD + 8 * ks);
#pragma unroll
                for (int r = 0; r < 4; ++r) st[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[r], qf[ks][r], st[kb], 0, 0, 0);
            }
        }
        if constexpr (DSPLIT) {
            // S^T = the four waves' partial blocks, summed in wave order by every wave (same lane, same register: same layout).  The buffer is
            // rewritten in the next tile only behind that tile's barriers, which every wave reaches after these reads.
            static_assert(!DSPLIT || KB == 1, "channel split: 32-key tiles");
            DS_RACE_SKEW(wave);
#pragma unroll
            for (int r = 0; r < 16; ++r) Rs[(wave * 16 + r) * 64 + lane] = st[0][r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                st[0][r] = ((Rs[(0 * 16 + r) * 64 + lane] + Rs[(1 * 16 + r) * 64 + lane]) + Rs[(2 * 16 + r) * 64 + lane]) + Rs[(3 * 16 + r) * 64 + lane];
        }
        if (t == ntiles - 1) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * KT + kb * 32 + 4 * hb + (r & 3) + 8 * (r >> 2) >= a.skv) st[kb][r] = -1e30f;
        }
        float mx = st[0][0];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kb][r] = __builtin_amdgcn_exp2f(st[kb][r] - mn); rs += st[kb][r]; }
        l = l * alpha + rs;
#pragma unroll
        for (int i = 0; i < DBW; ++i) ot[i] *= alpha;
        if (VREM) {
#pragma unroll
            for (int j = 0; j < VREM; ++j) oe[j] *= alpha;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const float* vrow = Vs + (kb * 32 + 4 * hb) * VLD + DB * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(vrow + ((r & 3) + 8 * (r >> 2)) * VLD);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(vrow + ((r & 3) + 8 * (r >> 2)) * VLD + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { oe[j] += st[kb][r] * v0[j]; oe[4 + j] += st[kb][r] * v1[j]; }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < DBW; ++i) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const float* vcol = Vs + (kb * 32 + 4 * hb) * VLD + d0 + i * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float vv = vcol[((r & 3) + 8 * (r >> 2)) * VLD];
                    ot[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, st[kb][r], ot[i], 0, 0, 0);
                }
            }
        }
    }
    if (!active) return;

    const float inv = 1.0f / (l + __shfl_xor(l, 32));
    float* patch = Es + wave * (32 * 33);
    float* op = a.out + (size_t)b * a.o_bs + h * D;
#pragma unroll
    for (int i = 0; i < DBW; ++i) {
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
            if (q0 + q < a.sq && d0 + i * 32 + c4 < D)
                *reinterpret_cast<f32x4*>(op + (size_t)(q0 + q) * a.ldo + d0 + i * 32 + c4) = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (VREM) {
        // the two lane halves hold the sums over their own keys: combine, normalise, 32 B per query row
        f32x4 e0, e1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e0[j] = (oe[j] + __shfl_xor(oe[j], 32)) * inv;
            e1[j] = (oe[4 + j] + __shfl_xor(oe[4 + j], 32)) * inv;
        }
        if (hb == 0 && q0 + l31 < a.sq) {
            float* dst = op + (size_t)(q0 + l31) * a.ldo + DB * 32;
            *reinterpret_cast<f32x4*>(dst) = e0;
            *reinterpret_cast<f32x4*>(dst + 4) = e1;
        }
    }
}

template <int D>
int launch(const ds_attn_args* a, hipStream_t stream) {
    constexpr int DB = (D > 32 && (D % 32) == 8) ? D / 32 : (D + 31) / 32;
    constexpr int KT = (D <= 64 && !(D > 32 && (D % 32) == 8)) ? 64 : 32;
    constexpr int bytes = (KT * (D + 4) + KT * (DB * 32 + 8) + 32 + 4 * 32 * 33) * (int)sizeof(float);
    if constexpr (D % 128 == 0) {
        // the channel-split block (32 queries, four waves x D / 4 channels) while the query-split grid has fewer than 1 024 waves (one per SIMD);
        // ds_attn_args.variant: 1 = query split, 2 = channel split (tests, A/B runs)
        const long long waves = (long long)((a->sq + 127) / 128) * a->heads * a->batch * 4;
        if (a->variant == 2 || (a->variant != 1 && waves < 1024)) {
            constexpr int bytes_d = bytes + 4 * 16 * 64 * (int)sizeof(float);
            DS_ENSURE_DYN_LDS((&flash_attn_kernel<D, true>), bytes_d);
            dim3 grid((a->sq + 31) / 32, a->heads, a->batch);
            hipLaunchKernelGGL((flash_attn_kernel<D, true>), grid, dim3(256), bytes_d, stream, *a);
            DS_CHECK_LAUNCH();
            return DS_OK;
        }
    }
    DS_ENSURE_DYN_LDS((&flash_attn_kernel<D>), bytes);
    dim3 grid((a->sq + 127) / 128, a->heads, a->batch);
    hipLaunchKernelGGL(flash_attn_kernel<D>, grid, dim3(256), bytes, stream, *a);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace

extern "C" int ds_attention_supported(int d) {
    switch (d) {
        case 8: case 16: case 32: case 40: case 64: case 80: case 96: case 128: case 160: case 256: return 1;
        default: return 0;
    }
}

extern "C" int ds_attention(const ds_attn_args* a, void* stream) {
    (void)hipGetLastError();
    if (a && (a->out_f16 || a->in_f16)) return DS_E_ARG;          // fp16 tensors: ds_attention_f16 only
    if (!a || !a->q || !a->k || !a->v || !a->out) return DS_E_ARG;
    if (a->batch <= 0 || a->heads <= 0 || a->sq <= 0 || a->skv <= 0 || a->batch > 65535 || a->heads > 65535) return DS_E_ARG;
    if ((a->ldq & 3) || (a->ldk & 3) || (a->ldv & 3) || (a->ldo & 3) || (a->q_bs & 3) || (a->k_bs & 3) || (a->v_bs & 3) || (a->o_bs & 3))
        return DS_E_ALIGN;
    if (!ds_aligned16(a->q) || !ds_aligned16(a->k) || !ds_aligned16(a->v) || !ds_aligned16(a->out)) return DS_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    switch (a->d) {
        case 8: return launch<8>(a, s);
        case 16: return launch<16>(a, s);
        case 32: return launch<32>(a, s);
        case 40: return launch<40>(a, s);
        case 64: return launch<64>(a, s);
        case 80: return launch<80>(a, s);
        case 96: return launch<96>(a, s);
        case 128: return launch<128>(a, s);
        case 160: return launch<160>(a, s);
        case 256: return launch<256>(a, s);
        default: return DS_E_SHAPE;
    }
}
