// HBM-bound kernels of the denoiser: GroupNorm statistics, fused normalise+affine+SiLU+resample, row softmax,
// noise embedding, stem im2col, channel mean.  NHWC fp32, float4 (16 B / lane) accesses everywhere, channels are the
// fastest dimension so a wave reads whole 1 KiB pixel rows.
#include "ds_common.h"

namespace {

// ONE definition of the normalisation's per-channel coefficients and of its per-element arithmetic (explicit fused multiply-adds: no compiler
// contraction choice), shared by the statistics kernels that write the {mu, A, B} planes, the pass kernels that apply them and the pass
// that computes them itself (norm_act16_kernel<FIN>): the paths must agree bit for bit.
//     A = rstd * gamma * (1 + scale),  B = beta * (1 + scale) + shift,  y = act((x - mu) * A + B)
__device__ __forceinline__ void gn_coefs(float r, float gm, float bt, float sc1, float sh, float& A, float& B) {
    A = r * gm * sc1;
    B = __builtin_fmaf(bt, sc1, sh);
}
__device__ __forceinline__ float gn_affine(float x, float mu, float A, float B) { return __builtin_fmaf(x - mu, A, B); }

// --------------------------------------------------------------------------------------------------------------
// GroupNorm statistics.  One block per image; thread t owns channel quad (t % CQ) and walks pixels t / CQ, t / CQ + PL, ...
// Sums are kept in fp64 (one pass, no cancellation problem in E[x^2] - E[x]^2), reduced through LDS atomics per group.
__device__ __forceinline__ void gn_finalize(const ds_norm_args& a, double* s_sum, double* s_sq);

__global__ void __launch_bounds__(1024) gn_stats_kernel(const ds_norm_args a, int CQ, int PL) {
    __shared__ double s_sum[64];
    __shared__ double s_sq[64];
    const int tid = threadIdx.x;
    const int n = blockIdx.x;
    const int P = gridDim.y;                 // pixel chunks per image (> 1: small batches, needs a.partial / a.counters)
    if (tid < 64) { s_sum[tid] = 0.0; s_sq[tid] = 0.0; }
    __syncthreads();
    const int C = a.c0 + a.c1;
    const int cpg = C / a.groups;
    const int HW = a.h * a.w;
    const int cq = tid % CQ, pl = tid / CQ;
    if (pl < PL) {
        int c = cq * 4;
        const float* src; int ld; bool half;
        if (c < a.c0) { src = a.x0; ld = a.ld0; half = a.in_f16 & 1; } else { src = a.x1; ld = a.ld1; half = a.in_f16 & 2; c -= a.c0; }
        const _Float16* src16 = reinterpret_cast<const _Float16*>(src) + (size_t)n * HW * ld + c;       // fp16 source: ld in halfs
        src += (size_t)n * HW * ld + c;
        c = cq * 4;
        typedef _Float16 h4s_t __attribute__((ext_vector_type(4)));
        double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
        for (int p = blockIdx.y * PL + pl; p < HW; p += PL * P) {
            f32x4 v;
            if (half) { const h4s_t hv = *reinterpret_cast<const h4s_t*>(src16 + (size_t)p * ld); v = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]}; }
            else v = *reinterpret_cast<const f32x4*>(src + (size_t)p * ld);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += (double)v[j]; q[j] += (double)v[j] * (double)v[j]; }
        }
        // merge the channels of this quad that fall in the same group before touching LDS
        int g_prev = c / cpg;
        double ss = s[0], qq = q[0];
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const int g = (c + j) / cpg;
            if (g != g_prev) {
                atomicAdd(&s_sum[g_prev], ss); atomicAdd(&s_sq[g_prev], qq);
                ss = 0.0; qq = 0.0; g_prev = g;
            }
            ss += s[j]; qq += q[j];
        }
        atomicAdd(&s_sum[g_prev], ss); atomicAdd(&s_sq[g_prev], qq);
    }
    __syncthreads();
    if (P > 1) {
        // small batches: publish this chunk's partial sums; gn_finalize_kernel (next launch) adds them in chunk order
        double* part = a.partial + ((size_t)n * P + blockIdx.y) * 128;
        if (tid < a.groups) { part[tid] = s_sum[tid]; part[64 + tid] = s_sq[tid]; }
        return;
    }
    gn_finalize(a, s_sum, s_sq);
}

__device__ __forceinline__ void gn_finalize(const ds_norm_args& a, double* s_sum, double* s_sq) {
    const int tid = threadIdx.x;
    const int n = blockIdx.x;
    const int C = a.c0 + a.c1;
    const int cpg = C / a.groups;
    const int HW = a.h * a.w;
    if (tid < a.groups) {
        const double cnt = (double)cpg * (double)HW;
        const double mean = s_sum[tid] / cnt;
        double var = s_sq[tid] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        a.mean[(size_t)n * a.groups + tid] = (float)mean;
        a.rstd[(size_t)n * a.groups + tid] = (float)(1.0 / sqrt(var + (double)a.eps));
        s_sum[tid] = mean;                                        // reuse as broadcast slots for the coefficient pass
        s_sq[tid] = 1.0 / sqrt(var + (double)a.eps);
    }
    if (a.coefs) {
        // per-(image, channel) planes {mu, A, B} for the convolution's fused input normalisation
        __syncthreads();
        float* cp = a.coefs + (size_t)n * 3 * C;
        for (int c = tid; c < C; c += blockDim.x) {
            const int g = c / cpg;
            const float m = (float)s_sum[g], r = (float)s_sq[g];
            const float gm = a.gamma ? a.gamma[c] : 1.f;
            const float bt = a.beta ? a.beta[c] : 0.f;
            float sc1 = 1.f, sh = 0.f;
            if (a.scale) {
                const size_t row = (a.ss_rows == 1) ? 0 : (size_t)n;
                sc1 = a.scale[row * a.ss_ld + c] + 1.f;
                sh = a.shift[row * a.ss_ld + c];
            }
            float A_, B_;
            gn_coefs(r, gm, bt, sc1, sh, A_, B_);
            cp[c] = m; cp[C + c] = A_; cp[2 * C + c] = B_;
        }
    }
}

__global__ void __launch_bounds__(256) gn_finalize_kernel(const ds_norm_args a, int P) {
    __shared__ double s_sum[64];
    __shared__ double s_sq[64];
    const int tid = threadIdx.x, n = blockIdx.x;
    if (tid < a.groups) {
        double ss = 0.0, qq = 0.0;
        const double* pp = a.partial + (size_t)n * P * 128;
        for (int k = 0; k < P; ++k) { ss += pp[k * 128 + tid]; qq += pp[k * 128 + 64 + tid]; }
        s_sum[tid] = ss; s_sq[tid] = qq;
    }
    __syncthreads();
    gn_finalize(a, s_sum, s_sq);
}

// --------------------------------------------------------------------------------------------------------------
// y = resample(act((x - mean) * A + B)).  grid = (pixel chunks, images); thread = (channel quad, pixel lane).
__global__ void __launch_bounds__(1024) norm_act_kernel(const ds_norm_args a, int CQ, int PL, int chunk) {
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int cq = tid % CQ, pl = tid / CQ;
    if (pl >= PL) return;
    const int C = a.c0 + a.c1;
    const int c = cq * 4;
    const float* src; int ld; bool half;          // half: this thread's source is an fp16 tensor (in_f16 bit 0: x0, bit 1: x1), ld in halfs
    int cs = c;                                   // channel inside its source
    if (c < a.c0) { src = a.x0; ld = a.ld0; half = a.in_f16 & 1; } else { src = a.x1; ld = a.ld1; half = a.in_f16 & 2; cs = c - a.c0; }
    const int H = a.h, W = a.w;
    const _Float16* src16 = reinterpret_cast<const _Float16*>(src) + (size_t)n * H * W * ld + cs;
    src += (size_t)n * H * W * ld + cs;

    float mu[4], A[4], Bc[4];
    const int cpg = a.mean ? C / a.groups : 1;
    const bool planes = a.coefs != nullptr && a.out_f16;      // {mu, A, B} planes [n][3][C] written by ds_gn_finalize / ds_gn_stats: three
    if (planes) {                                             // independent 16-B loads instead of 24 dependent scalar ones
        const float* cp = a.coefs + (size_t)n * 3 * C + c;
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(cp), a4 = *reinterpret_cast<const f32x4*>(cp + C), b4 = *reinterpret_cast<const f32x4*>(cp + 2 * C);
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = m4[j]; A[j] = a4[j]; Bc[j] = b4[j]; }
    } else
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float m = 0.f, r = 1.f;
        if (a.mean) { const int g = (c + j) / cpg; m = a.mean[(size_t)n * a.groups + g]; r = a.rstd[(size_t)n * a.groups + g]; }
        const float gm = a.gamma ? a.gamma[c + j] : 1.f;
        const float bt = a.beta ? a.beta[c + j] : 0.f;
        float sc1 = 1.f, sh = 0.f;
        if (a.scale) {
            const size_t row = (a.ss_rows == 1) ? 0 : (size_t)n;
            sc1 = a.scale[row * a.ss_ld + c + j] + 1.f;
            sh = a.shift[row * a.ss_ld + c + j];
        }
        mu[j] = m;
        gn_coefs(r, gm, bt, sc1, sh, A[j], Bc[j]);
    }
    const bool identity = !planes && !a.mean && !a.gamma && !a.scale;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    auto to_h4 = [](const f32x4 v) { h4 o = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]}; return o; };   // RNE, like .to(float16)
    auto xf = [&](const f32x4 v) {
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = identity ? v[j] : gn_affine(v[j], mu[j], A[j], Bc[j]);
            o[j] = (a.act == DS_ACT_SILU) ? ds_silu(t) : t;
        }
        return o;
    };

    const int OH = (a.resample == DS_RESAMPLE_DOWN) ? H / 2 : (a.resample == DS_RESAMPLE_UP ? H * 2 : H);
    const int OW = (a.resample == DS_RESAMPLE_DOWN) ? W / 2 : (a.resample == DS_RESAMPLE_UP ? W * 2 : W);
    float* dst = a.out + (size_t)n * OH * OW * a.out_ld + c;
    _Float16* dst16 = reinterpret_cast<_Float16*>(a.out) + (size_t)n * OH * OW * a.out_ld + c;
    _Float16* raw16 = a.raw_out ? reinterpret_cast<_Float16*>(a.raw_out) + (size_t)n * OH * OW * a.raw_ld + c : nullptr;
    auto ld4 = [&](size_t pix) -> f32x4 {
        if (half) {
            const h4 v = __builtin_nontemporal_load(reinterpret_cast<const h4*>(src16 + pix * ld));
            f32x4 o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            return o;
        }
        return *reinterpret_cast<const f32x4*>(src + pix * ld);
    };
    auto st4 = [&](int p, const f32x4 o, const f32x4 raw) {
        if (a.out_f16) {
            *reinterpret_cast<h4*>(dst16 + (size_t)p * a.out_ld) = to_h4(o);
            if (raw16) *reinterpret_cast<h4*>(raw16 + (size_t)p * a.raw_ld) = to_h4(raw);
        } else {
            *reinterpret_cast<f32x4*>(dst + (size_t)p * a.out_ld) = o;
        }
    };
    const int p_begin = blockIdx.x * chunk;
    const int p_end = min(p_begin + chunk, OH * OW);
    int p = p_begin + pl;
    if (a.resample == DS_RESAMPLE_NONE) {
        // four pixels per iteration: four independent 16-B (8-B) loads in flight per thread -- with one, the pass ran at 2.2 TB/s
        for (; p + 3 * PL < p_end; p += 4 * PL) {
            f32x4 r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = ld4((size_t)(p + q * PL));
#pragma unroll
            for (int q = 0; q < 4; ++q) st4(p + q * PL, xf(r[q]), r[q]);
        }
    }
    for (; p < p_end; p += PL) {
        f32x4 o, raw;
        if (a.resample == DS_RESAMPLE_NONE) {
            raw = ld4((size_t)p);
            o = xf(raw);
        } else if (a.resample == DS_RESAMPLE_UP) {
            const int oh = p / OW, ow = p - oh * OW;
            raw = ld4((size_t)((oh >> 1) * W + (ow >> 1)));
            o = xf(raw);
        } else {
            const int oh = p / OW, ow = p - oh * OW;
            const size_t s0 = (size_t)((2 * oh) * W + 2 * ow);
            const f32x4 r00 = ld4(s0), r01 = ld4(s0 + 1), r10 = ld4(s0 + W), r11 = ld4(s0 + W + 1);
            const f32x4 v00 = xf(r00), v01 = xf(r01), v10 = xf(r10), v11 = xf(r11);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = ((v00[j] + v01[j]) + (v10[j] + v11[j])) * 0.25f;
                raw[j] = ((r00[j] + r01[j]) + (r10[j] + r11[j])) * 0.25f;      // the same box filter on the raw input (the skip path's resample)
            }
        }
        st4(p, o, raw);
    }
}

// --------------------------------------------------------------------------------------------------------------
// The fp16-mode pass at 16 bytes per lane (round 6): y = silu((x - mu[c]) * A[c] + B[c]) on fp16 rows -> fp16 rows, both sources of a
// concatenation, optional raw copy, no resampling.  norm_act_kernel moves 8 bytes per lane and access (four channels: the fp32 geometry
// applied to halfs) and ran at 0.45 of the 8 TB/s on the ImageNet-64 mix (profiles/r5_*): here a thread owns a channel OCTET (one
// global_load_dwordx4 / global_store_dwordx4 per pixel), four pixels in flight, the same arithmetic through gn_affine / ds_silu / RNE.
// MODE 2: mean / rstd per (image, group) + gamma / beta [+ adaptive scale / shift] instead of planes (the attention blocks' norm2, the
// SpatialTransformer's GroupNorm): the coefficients are formed in the prologue exactly as norm_act_kernel forms them.
// MODE 1 (FIN): the pass computes the GroupNorm statistics ITSELF from the per-(64-row block, channel) sums the producing convolutions left behind
// (what ds_gn_finalize does in a launch of its own, ~6 us each, 950 per ImageNet-64 sampler call): channel sums in fp64 over the image's row
// blocks in order, group sums in channel order, mean / rstd and the coefficients by gn_coefs -- every workgroup of an image repeats that for
// the whole image, so the launcher takes this form only where the sums are small next to the tensor (images of at most 32 x 32 pixels) and
// caps the workgroups per image.  MEASURED (profiles/r6_norm_pass_ab.txt): the launch it saves (~6 us) is about what every workgroup's own
// reduction costs -- +-1 us per layer either way, a wash on both fp16 lines -- so the engines keep ds_gn_finalize (plan.FOLD_FINALIZE off).
typedef unsigned n16_u4 __attribute__((ext_vector_type(4)));
typedef _Float16 n16_h8 __attribute__((ext_vector_type(8)));

// MODE 0: {mu, A, B} planes (a.coefs); 1: FIN -- the producers' column sums (a.stats0 / stats1); 2: a.mean / a.rstd + gamma / beta [/ scale / shift].
// RS: the pass resamples by 2 (a.resample; its own instantiation: the box filter's registers would cost the plain pass two waves per SIMD).
template <int MODE, bool RS>
__global__ void __launch_bounds__(512) norm_act16_kernel(const ds_norm_args a, int CO, int PL, int chunk) {
    constexpr bool FIN = MODE == 1;
    static_assert(!(FIN && RS), "the self-finalising pass does not resample");
    extern __shared__ __attribute__((aligned(16))) double fin_lds[];          // FIN: [2][C] channel sums, then [2][64] floats mean / rstd
    const int tid = threadIdx.x, n = blockIdx.y;
    const int C = a.c0 + a.c1, HW = a.h * a.w;                     // HW: INPUT pixels per image (statistics, source rows)
    const int rs = RS ? a.resample : DS_RESAMPLE_NONE, W = a.w;
    const int OW = rs == DS_RESAMPLE_DOWN ? a.w / 2 : (rs == DS_RESAMPLE_UP ? a.w * 2 : a.w);
    const int OHW = rs == DS_RESAMPLE_DOWN ? HW / 4 : (rs == DS_RESAMPLE_UP ? HW * 4 : HW);      // output pixels per image: the loop below walks these
    const int co = tid % CO, pl = tid / CO;
    const bool live = pl < PL;
    const int c = co * 8;
    const bool first = c < a.c0;
    const _Float16* src = reinterpret_cast<const _Float16*>(first ? a.x0 : a.x1) + (size_t)n * HW * (first ? a.ld0 : a.ld1) + (first ? c : c - a.c0);
    const int ld = first ? a.ld0 : a.ld1;
    const int p_begin = blockIdx.x * chunk, p_end = min(p_begin + chunk, OHW);
    int p = p_begin + pl;
    // FIN: everything that does not depend on the statistics is requested BEFORE they are reduced -- the thread's first four pixels and its
    // channels' gamma / beta / scale / shift -- so that the reduction's memory round trip is the only one in front of the first store
    n16_u4 r[4];
    bool have = false;
    f32x4 gmv[2], btv[2], scv[2], shv[2];
    if constexpr (MODE != 0) {
        if (live) {
            if (rs == DS_RESAMPLE_NONE && p + 3 * PL < p_end) {
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = __builtin_nontemporal_load(reinterpret_cast<const n16_u4*>(src + (size_t)(p + q * PL) * ld));
                have = true;
            }
            const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gmv[h] = a.gamma ? *reinterpret_cast<const f32x4*>(a.gamma + c + 4 * h) : one;
                btv[h] = a.beta ? *reinterpret_cast<const f32x4*>(a.beta + c + 4 * h) : zero;
                scv[h] = zero; shv[h] = zero;
                if (a.scale) {
                    const size_t row = (a.ss_rows == 1) ? 0 : (size_t)n;
                    scv[h] = *reinterpret_cast<const f32x4*>(a.scale + row * a.ss_ld + c + 4 * h);
                    shv[h] = *reinterpret_cast<const f32x4*>(a.shift + row * a.ss_ld + c + 4 * h);
                }
            }
        }
    }
    if constexpr (FIN) {
        float* g_mean = reinterpret_cast<float*>(fin_lds + 2 * C);
        float* g_rstd = g_mean + 64;
        double* ch_s = fin_lds;
        double* ch_q = fin_lds + C;
        const int nrb = HW >> 6;
        for (int cc_ = tid; cc_ < C; cc_ += blockDim.x) {
            const bool f0 = cc_ < a.c0;
            const float* sp = f0 ? a.stats0 : a.stats1;
            const int cs = f0 ? a.c0 : a.c1, cc = f0 ? cc_ : cc_ - a.c0;
            const float* base = sp + ((size_t)n * nrb * 2) * cs + cc;
            // up to 16 row blocks (images of at most 32 x 32 pixels): ALL loads first -- a loop of load / add pairs costs one memory round trip
            // per row block (~1 us each) --, then the additions in row-block order
            float vs[16], vq[16];
#pragma unroll
            for (int rb = 0; rb < 16; ++rb) {
                const int rc = rb < nrb ? rb : nrb - 1;
                vs[rb] = base[(size_t)rc * 2 * cs]; vq[rb] = base[((size_t)rc * 2 + 1) * cs];
            }
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int rb = 0; rb < 16; ++rb)
                if (rb < nrb) { s += (double)vs[rb]; q += (double)vq[rb]; }
            ch_s[cc_] = s; ch_q[cc_] = q;
        }
        __syncthreads();
        if (tid < a.groups) {
            const int cpg = C / a.groups;
            double s = 0.0, q = 0.0;
            for (int k = 0; k < cpg; ++k) { s += ch_s[tid * cpg + k]; q += ch_q[tid * cpg + k]; }
            const double cnt = (double)cpg * (double)HW;
            const double mean = s / cnt;
            double var = q / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            g_mean[tid] = (float)mean;
            g_rstd[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
        }
        __syncthreads();
    }
    if (!live) return;
    float mu[8], A[8], Bc[8];
    if constexpr (FIN) {
        const float* g_mean = reinterpret_cast<const float*>(fin_lds + 2 * C);
        const float* g_rstd = g_mean + 64;
        const int cpg = C / a.groups;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (c + j) / cpg;
            mu[j] = g_mean[g];
            gn_coefs(g_rstd[g], gmv[j >> 2][j & 3], btv[j >> 2][j & 3], a.scale ? scv[j >> 2][j & 3] + 1.f : 1.f, a.scale ? shv[j >> 2][j & 3] : 0.f, A[j], Bc[j]);
        }
    } else if constexpr (MODE == 2) {
        // statistics from ds_gn_stats / ds_gn_finalize as mean / rstd per (image, group): the attention blocks' norm2, the transformer's GroupNorm
        const int cpg = a.mean ? C / a.groups : 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m_ = 0.f, r_ = 1.f;
            if (a.mean) { const int g = (c + j) / cpg; m_ = a.mean[(size_t)n * a.groups + g]; r_ = a.rstd[(size_t)n * a.groups + g]; }
            mu[j] = m_;
            gn_coefs(r_, gmv[j >> 2][j & 3], btv[j >> 2][j & 3], a.scale ? scv[j >> 2][j & 3] + 1.f : 1.f, a.scale ? shv[j >> 2][j & 3] : 0.f, A[j], Bc[j]);
        }
    } else {
        const float* cp = a.coefs + (size_t)n * 3 * C + c;
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(cp), m1 = *reinterpret_cast<const f32x4*>(cp + 4);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(cp + C), a1 = *reinterpret_cast<const f32x4*>(cp + C + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(cp + 2 * C), b1 = *reinterpret_cast<const f32x4*>(cp + 2 * C + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = m0[j]; mu[4 + j] = m1[j]; A[j] = a0[j]; A[4 + j] = a1[j]; Bc[j] = b0[j]; Bc[4 + j] = b1[j]; }
    }
    const bool silu = a.act == DS_ACT_SILU;
    auto xf = [&](const n16_u4 raw) -> n16_u4 {
        const n16_h8 x = __builtin_bit_cast(n16_h8, raw);
        n16_h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = gn_affine((float)x[j], mu[j], A[j], Bc[j]);
            o[j] = (_Float16)(silu ? ds_silu(t) : t);                       // RNE, like .to(float16)
        }
        return __builtin_bit_cast(n16_u4, o);
    };
    _Float16* dst = reinterpret_cast<_Float16*>(a.out) + (size_t)n * OHW * a.out_ld + c;
    _Float16* raw16 = a.raw_out ? reinterpret_cast<_Float16*>(a.raw_out) + (size_t)n * OHW * a.raw_ld + c : nullptr;
    if constexpr (RS) if (rs == DS_RESAMPLE_DOWN) {
        // 2x2 box filter, stride 2 (networks_edm.py:77 with resample_filter [1, 1]): the four ACTIVATED pixels are averaged in fp32 and rounded once,
        // the raw copy is the same filter on the widened input -- norm_act_kernel's expressions; two output pixels = eight loads in flight
        auto xf32 = [&](const n16_u4 raw, float (&o)[8], float (&w)[8]) {
            const n16_h8 x = __builtin_bit_cast(n16_h8, raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                w[j] = (float)x[j];
                const float t = gn_affine(w[j], mu[j], A[j], Bc[j]);
                o[j] = silu ? ds_silu(t) : t;
            }
        };
        auto down = [&](const n16_u4 (&q4)[4], int po) {
            float v[4][8], w[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) xf32(q4[k], v[k], w[k]);
            n16_h8 o, rw;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = (_Float16)(((v[0][j] + v[1][j]) + (v[2][j] + v[3][j])) * 0.25f);
                rw[j] = (_Float16)(((w[0][j] + w[1][j]) + (w[2][j] + w[3][j])) * 0.25f);
            }
            *reinterpret_cast<n16_u4*>(dst + (size_t)po * a.out_ld) = __builtin_bit_cast(n16_u4, o);
            if (raw16) *reinterpret_cast<n16_u4*>(raw16 + (size_t)po * a.raw_ld) = __builtin_bit_cast(n16_u4, rw);
        };
        auto load4 = [&](int po, n16_u4 (&q4)[4]) {
            const int oh = po / OW, ow = po - oh * OW;
            const _Float16* s0 = src + (size_t)((2 * oh) * W + 2 * ow) * ld;
            q4[0] = __builtin_nontemporal_load(reinterpret_cast<const n16_u4*>(s0));
            q4[1] = __builtin_nontemporal_load(reinterpret_cast<const n16_u4*>(s0 + ld));
            q4[2] = __builtin_nontemporal_load(reinterpret_cast<const n16_u4*>(s0 + (size_t)W * ld));
            q4[3] = __builtin_nontemporal_load(reinterpret_cast<const n16_u4*>(s0 + (size_t)(W + 1) * ld));
        };
        for (; p + PL < p_end; p += 2 * PL) {
            n16_u4 qa[4], qb[4];
            load4(p, qa); load4(p + PL, qb);
            down(qa, p); down(qb, p + PL);
        }
        for (; p < p_end; p += PL) {
            n16_u4 qa[4];
            load4(p, qa);
            down(qa, p);
        }
        return;
    }
    // no resampling, or nearest-neighbour x2 (networks_edm.py:75): output pixel (oh, ow) reads input pixel (oh / 2, ow / 2)
    auto sidx = [&](int po) -> size_t {
        if (rs == DS_RESAMPLE_NONE) return (size_t)po;
        const int oh = po / OW, ow = po - oh * OW;
        return (size_t)((oh >> 1) * W + (ow >> 1));
    };
    for (; p + 3 * PL < p_end; p += 4 * PL) {
        if (!have) {
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = __builtin_nontemporal_load(reinterpret_cast<const n16_u4*>(src + sidx(p + q * PL) * ld));
        }
        have = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<n16_u4*>(dst + (size_t)(p + q * PL) * a.out_ld) = xf(r[q]);
            if (raw16) *reinterpret_cast<n16_u4*>(raw16 + (size_t)(p + q * PL) * a.raw_ld) = r[q];
        }
    }
    for (; p < p_end; p += PL) {
        const n16_u4 r1 = __builtin_nontemporal_load(reinterpret_cast<const n16_u4*>(src + sidx(p) * ld));
        *reinterpret_cast<n16_u4*>(dst + (size_t)p * a.out_ld) = xf(r1);
        if (raw16) *reinterpret_cast<n16_u4*>(raw16 + (size_t)p * a.raw_ld) = r1;
    }
}

// Which ds_norm_act calls take norm_act16_kernel: fp16 rows in (every source present) and out, whole channel octets; 2x resampling included
// (not in the self-finalising form).
static bool norm16_ok(const ds_norm_args* a) {
    if (!a->out_f16 || (a->resample != DS_RESAMPLE_NONE && a->stats0 != nullptr)) return false;
    if (!(a->in_f16 & 1) || (a->c1 && !(a->in_f16 & 2))) return false;
    const int C = a->c0 + a->c1;
    if ((C & 7) || (a->c0 & 7) || (a->ld0 & 7) || (a->c1 && (a->ld1 & 7)) || (a->out_ld & 7) || C > 4096) return false;
    if (!ds_aligned16(a->x0) || (a->c1 && !ds_aligned16(a->x1)) || !ds_aligned16(a->out)) return false;
    if (a->raw_out && ((a->raw_ld & 7) || !ds_aligned16(a->raw_out))) return false;
    if (a->coefs != nullptr || a->stats0 != nullptr) return true;
    // mean / rstd form: statistics present and an affine map to apply (the identity / raw-resampler uses of ds_norm_act keep the generic kernel)
    return a->mean != nullptr && a->groups > 0 && (a->c0 + a->c1) % a->groups == 0;
}

// --------------------------------------------------------------------------------------------------------------
// Row softmax: one wave per row, 4 rows per block.
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows,
                                                           int cols, int ld) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld;
    float* yr = y + row * ld;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xr[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int c = lane; c < cols; c += 64) sum += __expf(xr[c] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int c = lane; c < cols; c += 64) yr[c] = __expf(xr[c] - mx) * inv;
}

// --------------------------------------------------------------------------------------------------------------
__global__ void noise_embed_kernel(const float* __restrict__ sigma, int bs, const float* __restrict__ freqs, int nch, int swap,
                                   float* __restrict__ out, int out_ld) {
    const int half = nch / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= bs * half) return;
    const int b = idx / half, i = idx - b * half;
    const float cn = (swap & 2) ? sigma[b] : logf(sigma[b]) / 4.0f;       // c_noise (networks_edm.py:491) or given
    const float ang = cn * freqs[i];
    const float cs = cosf(ang), sn = sinf(ang);
    float* o = out + (size_t)b * out_ld;
    if (swap & 1) { o[i] = sn; o[half + i] = cs; } else { o[i] = cs; o[half + i] = sn; }
}

// rows = pixels; k = tap*c + ch, zero-padded to kpad.
__global__ void stem_im2col_kernel(const float* __restrict__ x, const float* __restrict__ sigma, int sigma_rows, float sd,
                                   int n, int c, int h, int w, float* __restrict__ out, int kpad) {
    const long long total = (long long)n * h * w * kpad;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx % kpad);
        const long long pix = idx / kpad;
        const int ow = (int)(pix % w);
        const int oh = (int)((pix / w) % h);
        const int img = (int)(pix / ((long long)w * h));
        float v = 0.f;
        if (k < 9 * c) {
            const int tap = k / c, ch = k - tap * c;
            const int ih = oh + tap / 3 - 1, iw = ow + tap % 3 - 1;
            if ((unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w) {
                const float s = sigma[sigma_rows == 1 ? 0 : img];
                v = ds_c_in(s, sd) * x[(((size_t)img * c + ch) * h + ih) * w + iw];
            }
        }
        out[idx] = v;
    }
}

__global__ void channel_mean_kernel(const float* __restrict__ x, int ld, int c, long long rows, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ld;
    float s = 0.f;
    for (int i = lane; i < c; i += 64) s += xr[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = s / (float)c;
}

// --------------------------------------------------------------------------------------------------------------
// LayerNorm over the channel dimension of token rows: one wave per row, the row lives in registers (<= 8 float4 per
// lane), two-pass mean / variance like ATen's row-wise moments.
template <bool OUT16, int LPR, bool IN16 = false>
__global__ void __launch_bounds__(256) layernorm_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float* __restrict__ y, int ldy,
                                                             long long rows, int cols) {
    // LPR lanes share one row (16 for 320 columns: five 16-byte loads per lane, no idle lanes); a block holds 256 / LPR rows and
    // walks the row set with a grid stride so that a wave lives for many rows.
    constexpr int RPB = 256 / LPR;
    const int lane = threadIdx.x & (LPR - 1);
    const int n4 = cols >> 2;
    const float inv_n = 1.0f / (float)cols;
    for (long long row = (long long)blockIdx.x * RPB + threadIdx.x / LPR; row < rows; row += (long long)gridDim.x * RPB) {
        const float* xr = x + row * ldx;
        f32x4 v[8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + LPR * i;
            if (idx < n4) {
                if (IN16) {          // x is an fp16 tensor (ldx in halfs): the fp16 residual stream, widened before any arithmetic
                    typedef _Float16 h4i_t __attribute__((ext_vector_type(4)));
                    const h4i_t hv = *reinterpret_cast<const h4i_t*>(reinterpret_cast<const _Float16*>(x) + row * ldx + idx * 4);
                    v[i] = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
                } else v[i] = *reinterpret_cast<const f32x4*>(xr + idx * 4);
                s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * inv_n;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + LPR * i;
            if (idx < n4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q * inv_n + eps);
        float* yr = y + row * ldy;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = lane + LPR * i;
            if (idx < n4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + idx * 4);
                const f32x4 b = *reinterpret_cast<const f32x4*>(beta + idx * 4);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mean) * rstd * g[j] + b[j];
                if (OUT16) {
                    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
                    const h4_t hv = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                    *reinterpret_cast<h4_t*>(reinterpret_cast<_Float16*>(y) + row * ldy + idx * 4) = hv;
                } else
                    *reinterpret_cast<f32x4*>(yr + idx * 4) = o;
            }
        }
    }
}

template <bool OUT16, bool IN16 = false>
static void launch_layernorm_rows(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* y, int ldy, long long rows,
                                  int cols, hipStream_t stream) {
    const int n4 = cols >> 2;
    const int lpr = n4 <= 128 ? 16 : (n4 <= 256 ? 32 : 64);
    const long long rpb = 256 / lpr;
    long long blocks = (rows + rpb - 1) / rpb;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (lpr == 16)
        hipLaunchKernelGGL((layernorm_rows_kernel<OUT16, 16, IN16>), dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, gamma, beta, eps, y, ldy, rows, cols);
    else if (lpr == 32)
        hipLaunchKernelGGL((layernorm_rows_kernel<OUT16, 32, IN16>), dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, gamma, beta, eps, y, ldy, rows, cols);
    else
        hipLaunchKernelGGL((layernorm_rows_kernel<OUT16, 64, IN16>), dim3((unsigned)blocks), dim3(256), 0, stream, x, ldx, gamma, beta, eps, y, ldy, rows, cols);
}

// y[r, c] = x[r, c] * gelu(x[r, inner + c]), exact GELU 0.5 g (1 + erf(g / sqrt 2)).
__global__ void __launch_bounds__(256) geglu_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, long long rows,
                                                    int inner4) {
    const long long total = rows * inner4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / inner4;
        const int c4 = (int)(idx - r * inner4) * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + r * ldx + c4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(x + r * ldx + (size_t)inner4 * 4 + c4);
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = a[j] * (0.5f * g[j] * (1.0f + erff(g[j] * 0.70710678118654752440f)));
        *reinterpret_cast<f32x4*>(y + r * ldy + c4) = o;
    }
}

// GroupNorm statistics from the convolutions' epilogue partial sums.  Groups are independent, so the grid is
// (image, chunk of GPB groups): a block sums its channels' per-row-block partials in fp64 (thread = channel lane x row-block
// lane), merges them into its groups through LDS atomics and finalises exactly like gn_stats_kernel (mean / rstd and the
// {mu, A, B} planes of its channels).  GPB is chosen by the launcher so that small batches still get >= 64 blocks.
__global__ void __launch_bounds__(256) gn_from_partials_kernel(const ds_gn_finalize_args f, int GPB) {
    __shared__ double s_sum[64];
    __shared__ double s_sq[64];
    const int tid = threadIdx.x, n = blockIdx.x;
    const int g0 = blockIdx.y * GPB, g1 = min(g0 + GPB, f.groups);
    if (tid < 64) { s_sum[tid] = 0.0; s_sq[tid] = 0.0; }
    __syncthreads();
    const int C = f.c0 + f.c1;
    const int cpg = C / f.groups;
    const int nrb = f.hw >> 6;
    const int cb0 = g0 * cpg, cb1 = g1 * cpg;
    const int cl = tid & 63, rl = tid >> 6;
    // The affine operands of this thread's FIRST channel are requested before the sums (round 6): the kernel is two dependent memory round trips
    // long (partial sums, then gamma / beta / scale / shift) and nothing else; the second one now overlaps the first.  Same values, same arithmetic.
    const int c_first = cb0 + tid;
    float pf_gm = 1.f, pf_bt = 0.f, pf_sc = 0.f, pf_sh = 0.f;
    if (f.coefs && c_first < cb1) {
        if (f.gamma) pf_gm = f.gamma[c_first];
        if (f.beta) pf_bt = f.beta[c_first];
        if (f.scale) {
            const size_t row = (f.ss_rows == 1) ? 0 : (size_t)n;
            pf_sc = f.scale[row * f.ss_ld + c_first];
            pf_sh = f.shift[row * f.ss_ld + c_first];
        }
    }
    for (int c = cb0 + cl; c < cb1; c += 64) {
        const bool first = c < f.c0;
        const float* sp = first ? f.stats0 : f.stats1;
        const int cs = first ? f.c0 : f.c1, cc = first ? c : c - f.c0;
        const float* base = sp + ((size_t)n * nrb * 2) * cs + cc;
        double s = 0.0, q = 0.0;
        for (int rb = rl; rb < nrb; rb += 4) { s += (double)base[(size_t)rb * 2 * cs]; q += (double)base[((size_t)rb * 2 + 1) * cs]; }
        atomicAdd(&s_sum[c / cpg - g0], s); atomicAdd(&s_sq[c / cpg - g0], q);
    }
    __syncthreads();
    if (tid < g1 - g0) {
        const double cnt = (double)cpg * (double)f.hw;
        const double mean = s_sum[tid] / cnt;
        double var = s_sq[tid] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)f.eps);
        f.mean[(size_t)n * f.groups + g0 + tid] = (float)mean;
        f.rstd[(size_t)n * f.groups + g0 + tid] = (float)rstd;
        s_sum[tid] = mean; s_sq[tid] = rstd;
    }
    if (f.coefs) {
        __syncthreads();
        float* cp = f.coefs + (size_t)n * 3 * C;
        for (int c = cb0 + tid; c < cb1; c += blockDim.x) {
            const int g = c / cpg - g0;
            const float m = (float)s_sum[g], r = (float)s_sq[g];
            float gm, bt, sc1 = 1.f, sh = 0.f;
            if (c == c_first) {
                gm = pf_gm; bt = pf_bt;
                if (f.scale) { sc1 = pf_sc + 1.f; sh = pf_sh; }
            } else {
                gm = f.gamma ? f.gamma[c] : 1.f;
                bt = f.beta ? f.beta[c] : 0.f;
                if (f.scale) {
                    const size_t row = (f.ss_rows == 1) ? 0 : (size_t)n;
                    sc1 = f.scale[row * f.ss_ld + c] + 1.f;
                    sh = f.shift[row * f.ss_ld + c];
                }
            }
            float A_, B_;
            gn_coefs(r, gm, bt, sc1, sh, A_, B_);
            cp[c] = m; cp[C + c] = A_; cp[2 * C + c] = B_;
        }
    }
}

int norm_geometry(const ds_norm_args* a, int* CQ, int* PL) {
    const int C = a->c0 + a->c1;
    if (C <= 0 || (C & 3) || (a->c0 & 3)) return DS_E_SHAPE;
    if (C / 4 > 1024) return DS_E_SHAPE;
    *CQ = C / 4;
    *PL = 1024 / *CQ;
    if (*PL > a->h * a->w) *PL = a->h * a->w;
    if (*PL < 1) *PL = 1;
    return DS_OK;
}

}  // namespace

extern "C" int ds_gn_stats(const ds_norm_args* a, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!a || !a->x0 || !a->mean || !a->rstd) return DS_E_ARG;
    if (a->groups <= 0 || a->groups > 64 || (a->c0 + a->c1) % a->groups) return DS_E_SHAPE;
    if ((a->ld0 & 3) || (a->c1 && (a->ld1 & 3))) return DS_E_ALIGN;
    if ((a->in_f16 & ~3) || ((a->in_f16 & 2) && !a->c1)) return DS_E_ARG;
    int CQ, PL;
    int rc = norm_geometry(a, &CQ, &PL);
    if (rc) return rc;
    int threads = CQ * PL;
    threads = ((threads + 63) / 64) * 64;
    if (threads < 64) threads = 64;
    // small batches: split every image over P pixel chunks so that the launch still covers the chip
    int P = 1;
    if (a->partial && a->n < 256) {
        P = (512 + a->n - 1) / a->n;
        const int maxp = (a->h * a->w) / (PL * 4);          // at least 4 pixel iterations per thread
        if (P > maxp) P = maxp;
        if (P > DS_GN_MAX_CHUNKS) P = DS_GN_MAX_CHUNKS;
        if (P < 1) P = 1;
    }
    hipLaunchKernelGGL(gn_stats_kernel, dim3(a->n, P), dim3(threads), 0, (hipStream_t)stream, *a, CQ, PL);
    if (P > 1) hipLaunchKernelGGL(gn_finalize_kernel, dim3(a->n), dim3(256), 0, (hipStream_t)stream, *a, P);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_norm_act(const ds_norm_args* a, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!a || !a->x0 || !a->out) return DS_E_ARG;
    if ((a->mean == nullptr) != (a->rstd == nullptr)) return DS_E_ARG;
    if ((a->scale == nullptr) != (a->shift == nullptr)) return DS_E_ARG;
    if ((a->ld0 & 3) || (a->c1 && (a->ld1 & 3)) || (a->out_ld & 3)) return DS_E_ALIGN;
    if ((a->in_f16 & ~3) || ((a->in_f16 & 1) && (reinterpret_cast<uintptr_t>(a->x0) & 7u)) || ((a->in_f16 & 2) && (!a->c1 || (reinterpret_cast<uintptr_t>(a->x1) & 7u)))) return DS_E_ARG;
    if (a->raw_out && (!a->out_f16 || (a->raw_ld & 3) || (reinterpret_cast<uintptr_t>(a->raw_out) & 7u))) return DS_E_ARG;
    if (a->out_f16 && (reinterpret_cast<uintptr_t>(a->out) & 7u)) return DS_E_ALIGN;
    if (a->resample == DS_RESAMPLE_DOWN && ((a->h | a->w) & 1)) return DS_E_SHAPE;
    int CQ, PL;
    int rc = norm_geometry(a, &CQ, &PL);
    if (rc) return rc;
    const int OH = (a->resample == DS_RESAMPLE_DOWN) ? a->h / 2 : (a->resample == DS_RESAMPLE_UP ? a->h * 2 : a->h);
    const int OW = (a->resample == DS_RESAMPLE_DOWN) ? a->w / 2 : (a->resample == DS_RESAMPLE_UP ? a->w * 2 : a->w);
    if (a->coefs && a->out_f16 && !ds_aligned16(a->coefs)) return DS_E_ALIGN;
    if (a->stats0) {
        // the pass computes its own statistics from the producers' column sums (norm_act16_kernel<FIN>): no ds_gn_finalize launch in front of it
        if (!norm16_ok(a) || a->coefs || (a->c1 && !a->stats1) || a->groups <= 0 || a->groups > 64 || (a->c0 + a->c1) % a->groups) return DS_E_SHAPE;
        if ((a->h * a->w) & 63 || (a->h * a->w) > 1024) return DS_E_SHAPE;
        if ((a->scale == nullptr) != (a->shift == nullptr)) return DS_E_ARG;
    }
    if (norm16_ok(a) && !(a->tune_variant & 1)) {
        const bool fin = a->stats0 != nullptr;
        const int C = a->c0 + a->c1, CO = C / 8, HW = OH * OW;          // the OUTPUT pixels of an image are what the workgroups share out
        const int T = CO <= 256 ? 256 : 512;
        int PL16 = T / CO;
        if (PL16 > HW) PL16 = HW;
        // workgroups per image.  Plain: >= 16 pixels per thread on the large tensors, 4 on the small ones (see below).  FIN: every workgroup re-reads
        // the image's column sums (HW / 64 x 2 x C floats): at most as many workgroups as keep that below a workgroup's own rows
        const long long elems = (long long)a->n * HW * C;
        const int ppt = elems < (16ll << 20) ? 4 : 16;
        int chunks = (HW + ppt * PL16 - 1) / (ppt * PL16);
        if (fin) {
            const long long sums = (long long)(HW / 64) * 2 * C * 4, rows = (long long)HW * C * 2;
            const int div = (a->tune_variant & 2) ? 4 : ((a->tune_variant & 4) ? 0 : 1);          // (benchmarks: tune_variant bits 1 / 2 change the cap; 1x measured best)
            int cap = div ? (int)(rows / (div * sums)) : chunks;
            if (cap < 1) cap = 1;
            if (chunks > cap) chunks = cap;
        }
        if (chunks < 1) chunks = 1;
        int chunk = (HW + chunks - 1) / chunks;
        chunk = ((chunk + PL16 - 1) / PL16) * PL16;
        chunks = (HW + chunk - 1) / chunk;
        const size_t lds = fin ? (size_t)2 * C * sizeof(double) + 128 * sizeof(float) : 0;
        const bool rsm = a->resample != DS_RESAMPLE_NONE;
#define DS_N16(MODE_, RS_, LDS_) hipLaunchKernelGGL((norm_act16_kernel<MODE_, RS_>), dim3(chunks, a->n), dim3(T), LDS_, (hipStream_t)stream, *a, CO, PL16, chunk)
        if (fin) DS_N16(1, false, lds);
        else if (a->coefs) { if (rsm) DS_N16(0, true, 0); else DS_N16(0, false, 0); }
        else { if (rsm) DS_N16(2, true, 0); else DS_N16(2, false, 0); }
#undef DS_N16
        DS_CHECK_LAUNCH();
        return DS_OK;
    }
    if (a->out_f16) {
        // the fp16-mode pass streams whole tensors (it runs twice per block): 256-thread workgroups, and few enough of them that a thread
        // walks >= 16 pixels with four loads in flight -- the 1024-thread / 2048-workgroup geometry below left ~6 pixels per thread behind
        // a 2-us coefficient prologue and ran at 2.1 TB/s
        PL = 256 / CQ;
        if (PL < 1) PL = 1;
    }
    if (PL > OH * OW) PL = OH * OW;
    int threads = ((CQ * PL + 63) / 64) * 64;
    // enough blocks per image to cover the chip even at small batch: aim for >= 2048 blocks in total
    int chunks = (2048 + a->n - 1) / a->n;
    int max_chunks = (OH * OW + PL - 1) / PL;
    if (a->out_f16) {
        // at least 16 pixels per thread on the large tensors (amortises the ~2-us coefficient prologue of a workgroup); on the SMALL ones
        // (8x8 / 16x16 stages: a few MB, the whole pass is one or two memory round trips) 4 pixels per thread = one round of four loads and
        // four times the workgroups -- the 16-pixel geometry left them at 13 - 19 us per launch, ~1 TB/s (profiles/r4_norm_small.txt)
        const long long elems = (long long)a->n * OH * OW * (a->c0 + a->c1);
        const int ppt = elems < (16ll << 20) ? 4 : 16;
        const int by_work = (OH * OW + ppt * PL - 1) / (ppt * PL);
        if (max_chunks > by_work) max_chunks = by_work;
    }
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    int chunk = (OH * OW + chunks - 1) / chunks;
    chunk = ((chunk + PL - 1) / PL) * PL;
    chunks = (OH * OW + chunk - 1) / chunk;
    hipLaunchKernelGGL(norm_act_kernel, dim3(chunks, a->n), dim3(threads), 0, (hipStream_t)stream, *a, CQ, PL, chunk);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_softmax_rows(const float* x, float* y, long long rows, int cols, int ld, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!x || !y || rows <= 0 || cols <= 0 || ld < cols) return DS_E_ARG;
    const long long blocks = (rows + 3) / 4;
    if (blocks > 0x7fffffffLL) return DS_E_SHAPE;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, ld);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_noise_embed(const float* sigma, int bs, const float* freqs, int nch, int swap, float* out, int out_ld,
                              void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!sigma || !freqs || !out || bs <= 0 || nch <= 0 || (nch & 1)) return DS_E_ARG;
    const int total = bs * (nch / 2);
    hipLaunchKernelGGL(noise_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, sigma, bs, freqs, nch,
                       swap, out, out_ld);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_stem_im2col(const float* x, const float* sigma, int sigma_rows, float sigma_data, int n, int c, int h, int w,
                              float* out, int kpad, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!x || !sigma || !out || n <= 0 || c <= 0 || kpad < 9 * c || (kpad % 32)) return DS_E_ARG;
    const long long total = (long long)n * h * w * kpad;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(stem_im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, sigma, sigma_rows,
                       sigma_data, n, c, h, w, out, kpad);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_channel_mean(const float* x, int ld, int c, long long rows, float* out, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!x || !out || rows <= 0 || c <= 0) return DS_E_ARG;
    hipLaunchKernelGGL(channel_mean_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ld, c, rows, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_layernorm_rows(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* y, int ldy,
                                 long long rows, int cols, void* stream) {
    (void)hipGetLastError();
    if (!x || !gamma || !beta || !y || rows <= 0 || cols <= 0) return DS_E_ARG;
    if ((cols & 3) || cols > 2048) return DS_E_SHAPE;
    if ((ldx & 3) || (ldy & 3) || !ds_aligned16(x) || !ds_aligned16(y) || !ds_aligned16(gamma) || !ds_aligned16(beta)) return DS_E_ALIGN;
    launch_layernorm_rows<false>(x, ldx, gamma, beta, eps, y, ldy, rows, cols, (hipStream_t)stream);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_layernorm_rows_f16(const float* x, int ldx, const float* gamma, const float* beta, float eps, void* y16, int ldy,
                                     long long rows, int cols, void* stream) {
    (void)hipGetLastError();
    if (!x || !gamma || !beta || !y16 || rows <= 0 || cols <= 0) return DS_E_ARG;
    if ((cols & 3) || cols > 2048) return DS_E_SHAPE;
    if ((ldx & 3) || (ldy & 3) || !ds_aligned16(x) || (reinterpret_cast<uintptr_t>(y16) & 7u) || !ds_aligned16(gamma) || !ds_aligned16(beta)) return DS_E_ALIGN;
    launch_layernorm_rows<true>(x, ldx, gamma, beta, eps, reinterpret_cast<float*>(y16), ldy, rows, cols, (hipStream_t)stream);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_layernorm_rows_f16io(const void* x16, int ldx, const float* gamma, const float* beta, float eps, void* y16, int ldy,
                                       long long rows, int cols, void* stream) {
    (void)hipGetLastError();
    if (!x16 || !gamma || !beta || !y16 || rows <= 0 || cols <= 0) return DS_E_ARG;
    if ((cols & 3) || cols > 2048) return DS_E_SHAPE;
    if ((ldx & 3) || (ldy & 3) || (reinterpret_cast<uintptr_t>(x16) & 7u) || (reinterpret_cast<uintptr_t>(y16) & 7u) || !ds_aligned16(gamma) || !ds_aligned16(beta))
        return DS_E_ALIGN;
    launch_layernorm_rows<true, true>(reinterpret_cast<const float*>(x16), ldx, gamma, beta, eps, reinterpret_cast<float*>(y16), ldy, rows, cols, (hipStream_t)stream);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_geglu(const float* x, int ldx, float* y, int ldy, long long rows, int inner, void* stream) {
    (void)hipGetLastError();
    if (!x || !y || rows <= 0 || inner <= 0) return DS_E_ARG;
    if (inner & 3) return DS_E_SHAPE;
    if ((ldx & 3) || (ldy & 3) || !ds_aligned16(x) || !ds_aligned16(y)) return DS_E_ALIGN;
    long long blocks = (rows * (inner / 4) + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, inner / 4);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_gn_finalize(const ds_gn_finalize_args* a, void* stream) {
    (void)hipGetLastError();
    if (!a || !a->stats0 || !a->mean || !a->rstd || a->c0 <= 0 || a->c1 < 0 || (a->c1 && !a->stats1)) return DS_E_ARG;
    if (a->n <= 0 || a->hw <= 0 || (a->hw & 63)) return DS_E_SHAPE;
    if (a->groups <= 0 || a->groups > 64 || (a->c0 + a->c1) % a->groups) return DS_E_SHAPE;
    if ((a->scale == nullptr) != (a->shift == nullptr)) return DS_E_ARG;
    int gpb = (int)(((long long)a->n * a->groups + 511) / 512);          // groups per block: ~512 blocks at large batch,
    if (gpb < 1) gpb = 1;                                                 // one group per block at small batch
    if (gpb > a->groups) gpb = a->groups;
    hipLaunchKernelGGL(gn_from_partials_kernel, dim3(a->n, (a->groups + gpb - 1) / gpb), dim3(256), 0, (hipStream_t)stream, *a, gpb);
    DS_CHECK_LAUNCH();
    return DS_OK;
}
