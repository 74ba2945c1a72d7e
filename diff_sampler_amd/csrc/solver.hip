// Solver-side kernels: fused EDM-precondition + ODE update (HBM bound, one pass over every operand), dynamic
// thresholding (exact torch.quantile semantics via LDS radix select), latent scaling, uint8 quantisation.
#include <type_traits>

#include "ds_common.h"

namespace {

typedef DsUpdCoefs Coefs;
__device__ __forceinline__ Coefs load_coefs(const ds_update_args& a, int img) { return ds_upd_load_coefs(a, img); }

// One thread = 4 consecutive pixels of one image, all channels.  Every NCHW plane access is a float4 (16 B/lane,
// 1 KiB per wave instruction); the raw NHWC network output (3 or 4 channels, row = f_ld floats) is read as whole rows.
template <int VEC>
__global__ void __launch_bounds__(256) solver_update_kernel(const ds_update_args a) {
    const int HW = a.h * a.w;
    const int groups_per_img = HW / VEC;
    const long long total = (long long)a.n * groups_per_img;
    for (long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; gidx < total; gidx += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(gidx / groups_per_img);
        const int p0 = (int)(gidx - (long long)img * groups_per_img) * VEC;
        const Coefs k = load_coefs(a, img);
        float cskip = 0.f, cout_ = 0.f;
        if (a.raw) { cskip = ds_c_skip(k.sig, a.sigma_data); cout_ = ds_c_out(k.sig, a.sigma_data); }
        const float afs_div = sqrtf(1.0f + k.t * k.t);
        for (int ch = 0; ch < a.c; ++ch) {
            const size_t off = ((size_t)img * a.c + ch) * HW + p0;
            float xe[VEC], xb[VEC], f[VEC], h0[VEC], h1[VEC], h2[VEC];
            if (VEC == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(a.xe + off);
#pragma unroll
                for (int j = 0; j < VEC; ++j) xe[j] = v[j];
            } else {
                xe[0] = a.xe[off];
            }
            if (a.xb != a.xe) {
                if (VEC == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(a.xb + off);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) xb[j] = v[j];
                } else {
                    xb[0] = a.xb[off];
                }
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) xb[j] = xe[j];
            }
            if (!a.afs) {
                if (a.raw && a.f_ld > 0) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) f[j] = a.f[((size_t)img * HW + p0 + j) * a.f_ld + ch];
                } else if (VEC == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(a.f + off);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) f[j] = v[j];
                } else {
                    f[0] = a.f[off];
                }
            }
            auto ldh = [&](const float* hp, float (&dst)[VEC]) {
                if (!hp) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) dst[j] = 0.f;
                } else if (VEC == 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(hp + off);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) dst[j] = v[j];
                } else {
                    dst[0] = hp[off];
                }
            };
            ldh(a.hist[0], h0); ldh(a.hist[1], h1); ldh(a.hist[2], h2);

            float m[VEC], xo[VEC];
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                if (a.afs) {
                    const float d = xe[j] / afs_div;              // solvers.py:77
                    const float D = xe[j] - k.t * d;              // solvers.py:680
                    m[j] = a.store_d ? d : D;
                    float acc = k.cx * xb[j] + k.cm * m[j];
                    if (a.hist[0]) acc += k.ch0 * h0[j];
                    if (a.hist[1]) acc += k.ch1 * h1[j];
                    if (a.hist[2]) acc += k.ch2 * h2[j];
                    xo[j] = acc;
                } else {                                          // the arithmetic shared with the head-fused form (ds_common.h)
                    ds_upd_element(k, cskip, cout_, a.raw != 0, a.store_d != 0, xe[j], xb[j], f[j], a.hist[0] != nullptr, h0[j], a.hist[1] != nullptr, h1[j],
                                   a.hist[2] != nullptr, h2[j], m[j], xo[j]);
                }
            }
            if (VEC == 4) {
                if (a.m_out) { f32x4 v = {m[0], m[1], m[2], m[3]}; *reinterpret_cast<f32x4*>(a.m_out + off) = v; }
                if (a.x_out) { f32x4 v = {xo[0], xo[1], xo[2], xo[3]}; *reinterpret_cast<f32x4*>(a.x_out + off) = v; }
            } else {
                if (a.m_out) a.m_out[off] = m[0];
                if (a.x_out) a.x_out[off] = xo[0];
            }
        }
    }
}

// Streaming variant for the common geometry (CH = 3 or 4 channels, H*W % 4 == 0, raw rows of 4 floats): all loads of
// a pixel quad -- CH planes of up to five operands plus four network-output rows -- are issued before the first use, so
// every lane keeps 13-24 independent 16-B loads in flight (the generic kernel above walks the channels one by one).
template <int CH>
__global__ void __launch_bounds__(256) solver_update_fast_kernel(const ds_update_args a) {
    const int HW = a.h * a.w;
    const int gpi = HW >> 2;
    const long long total = (long long)a.n * gpi;
    const bool has_xb = a.xb != a.xe;
    for (long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; gidx < total; gidx += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(gidx / gpi);
        const int p0 = (int)(gidx - (long long)img * gpi) << 2;
        f32x4 fr[4], xe[CH], xb[CH], fv[CH], h0[CH], h1[CH], h2[CH];
        const bool rows = a.raw && a.f_ld > 0;       // F as NHWC rows of 4 floats; otherwise F / D as NCHW planes
        if (!a.afs && rows) {
            const f32x4* fp = reinterpret_cast<const f32x4*>(a.f) + ((size_t)img * HW + p0);
#pragma unroll
            for (int j = 0; j < 4; ++j) fr[j] = __builtin_nontemporal_load(fp + j);
        }
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const size_t off = ((size_t)img * CH + ch) * HW + p0;
            xe[ch] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.xe + off));
            if (has_xb) xb[ch] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.xb + off));
            if (!a.afs && !rows) fv[ch] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.f + off));
            if (a.hist[0]) h0[ch] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.hist[0] + off));
            if (a.hist[1]) h1[ch] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.hist[1] + off));
            if (a.hist[2]) h2[ch] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.hist[2] + off));
        }
        const Coefs k = load_coefs(a, img);
        float cskip = 0.f, cout_ = 0.f;
        if (a.raw) { cskip = ds_c_skip(k.sig, a.sigma_data); cout_ = ds_c_out(k.sig, a.sigma_data); }
        const float afs_div = sqrtf(1.0f + k.t * k.t);
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            const size_t off = ((size_t)img * CH + ch) * HW + p0;
            f32x4 m, xo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = xe[ch][j];
                if (a.afs) {
                    const float d = x / afs_div;
                    const float D = x - k.t * d;
                    m[j] = a.store_d ? d : D;
                    float acc = k.cx * (has_xb ? xb[ch][j] : x) + k.cm * m[j];
                    if (a.hist[0]) acc += k.ch0 * h0[ch][j];
                    if (a.hist[1]) acc += k.ch1 * h1[ch][j];
                    if (a.hist[2]) acc += k.ch2 * h2[ch][j];
                    xo[j] = acc;
                } else {                                          // the arithmetic shared with the head-fused form (ds_common.h)
                    const float f = rows ? fr[j][ch] : fv[ch][j];
                    float mm, xx;
                    ds_upd_element(k, cskip, cout_, a.raw != 0, a.store_d != 0, x, has_xb ? xb[ch][j] : x, f, a.hist[0] != nullptr, a.hist[0] ? h0[ch][j] : 0.f,
                                   a.hist[1] != nullptr, a.hist[1] ? h1[ch][j] : 0.f, a.hist[2] != nullptr, a.hist[2] ? h2[ch][j] : 0.f, mm, xx);
                    m[j] = mm; xo[j] = xx;
                }
            }
            if (a.m_out) __builtin_nontemporal_store(m, reinterpret_cast<f32x4*>(a.m_out + off));
            if (a.x_out) __builtin_nontemporal_store(xo, reinterpret_cast<f32x4*>(a.x_out + off));
        }
    }
}

__global__ void table_select_kernel(const float* __restrict__ table, int row_floats, int* __restrict__ step, int advance,
                                    float* __restrict__ dst) {
    const int s = *step;
    for (int i = threadIdx.x; i < row_floats; i += blockDim.x) dst[i] = table[(size_t)s * row_floats + i];
    __syncthreads();
    if (advance && threadIdx.x == 0) *step = s + 1;
}

__global__ void __launch_bounds__(256) scale_kernel(const float* __restrict__ x, float a, float* __restrict__ y, long long count) {
    const long long n4 = count >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        v *= a;
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) y[i] = a * x[i];
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ dst, float v, long long count) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) dst[i] = v;
}

__global__ void __launch_bounds__(256) copy_rows_kernel(const float* __restrict__ src, int src_ld, float* __restrict__ dst, int dst_ld,
                                                        long long rows, int cols) {
    const long long total = rows * cols;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}

__global__ void __launch_bounds__(256) quantize_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int n, int c, int h, int w) {
    const long long total = (long long)n * h * w;
    const int HW = h * w;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(pix / HW);
        const int p = (int)(pix - (long long)img * HW);
        for (int ch = 0; ch < c; ++ch) {
            float v = x[((size_t)img * c + ch) * HW + p] * 127.5f + 128.0f;
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            out[pix * c + ch] = (uint8_t)v;        // truncation, like .to(torch.uint8) (sample.py:311)
        }
    }
}

// --------------------------------------------------------------------------------------------------------------
// Dynamic thresholding.  One block per sample.  |x| bit patterns are staged in LDS; an 8-bit-digit radix select
// finds the two order statistics around rank p*(n-1); torch.quantile's linear interpolation (lerp) follows.
__device__ unsigned radix_select(const unsigned* vals, int n, unsigned rank, unsigned* hist) {
    unsigned prefix = 0, mask = 0;
    for (int pass = 3; pass >= 0; --pass) {
        const int shift = pass * 8;
        for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const unsigned v = vals[i];
            if ((v & mask) == prefix) atomicAdd(&hist[(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        // every thread scans the 256-bin histogram (uniform result, no extra barrier data hazards)
        unsigned cum = 0, digit = 0, before = 0;
        for (int b = 0; b < 256; ++b) {
            const unsigned hcount = hist[b];
            if (cum + hcount > rank) { digit = b; before = cum; break; }
            cum += hcount;
        }
        rank -= before;
        prefix |= digit << shift;
        mask |= 255u << shift;
        __syncthreads();
    }
    return prefix;
}

__global__ void __launch_bounds__(512) dynamic_threshold_kernel(const float* __restrict__ x0, float* __restrict__ out, int per, float p) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm[];
    unsigned* vals = sm;
    unsigned* hist = sm + per;
    __shared__ unsigned s_cnt;
    __shared__ unsigned s_min;
    const float* x = x0 + (size_t)blockIdx.x * per;
    float* y = out + (size_t)blockIdx.x * per;
    for (int i = threadIdx.x; i < per; i += blockDim.x) vals[i] = __float_as_uint(fabsf(x[i]));
    if (threadIdx.x == 0) { s_cnt = 0; s_min = 0xffffffffu; }
    __syncthreads();
    const float rank = p * (float)(per - 1);        // fp32, as ATen computes it
    const float lo_f = floorf(rank);
    const float w = rank - lo_f;
    const unsigned lo = (unsigned)lo_f;
    const unsigned hi = (unsigned)ceilf(rank);
    const unsigned v_lo = radix_select(vals, per, lo, hist);
    unsigned v_hi = v_lo;
    if (hi != lo) {
        unsigned cnt = 0, mn = 0xffffffffu;
        for (int i = threadIdx.x; i < per; i += blockDim.x) {
            const unsigned v = vals[i];
            if (v <= v_lo) ++cnt; else mn = min(mn, v);
        }
        atomicAdd(&s_cnt, cnt);
        atomicMin(&s_min, mn);
        __syncthreads();
        v_hi = (s_cnt >= hi + 1) ? v_lo : s_min;
    }
    const float a = __uint_as_float(v_lo), b = __uint_as_float(v_hi);
    // at::lerp: weight < 0.5 ? a + w (b - a) : b - (b - a) (1 - w)
    float s = (w < 0.5f) ? a + w * (b - a) : b - (b - a) * (1.0f - w);
    s = fmaxf(s, 1.0f);
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        const float v = fminf(fmaxf(x[i], -s), s);
        y[i] = v / s;
    }
}

// One DPM-Solver++ step in data-prediction form in ONE launch (solvers.py:674-702 + solver_utils.py:77-86, :102-163): per sample
//   D = c_skip x + c_out F (or the given denoised tensor, or the AFS direction)   -> m0 = clamp(D, -s, s) / s with s = max(q_0.995(|D|), 1)
//   x' = cx xb + cm m0 + ch0 m1 + ch1 m2
// One block per sample: |D| bit patterns go to LDS for the same radix select as dynamic_threshold_kernel, D itself is recomputed
// from x and F (L2-resident) when it is clamped.  Replaces three launches (D pass, threshold, combination).
__global__ void __launch_bounds__(512) dpmpp_x0_step_kernel(const ds_update_args a, float p) {
    extern __shared__ __attribute__((aligned(16))) unsigned sm[];
    const int HW = a.h * a.w, per = a.c * HW;
    unsigned* vals = sm;
    unsigned* hist = sm + per;
    __shared__ unsigned s_cnt;
    __shared__ unsigned s_min;
    const int img = blockIdx.x;
    const Coefs k = load_coefs(a, img);
    float cskip = 0.f, cout_ = 0.f;
    if (a.raw) { cskip = ds_c_skip(k.sig, a.sigma_data); cout_ = ds_c_out(k.sig, a.sigma_data); }
    const float afs_div = sqrtf(1.0f + k.t * k.t);
    const size_t base = (size_t)img * per;
    auto denoised = [&](int i) -> float {
        const float x = a.xe[base + i];
        if (a.afs) return x - k.t * (x / afs_div);                      // solvers.py:77, :680
        const float f = a.f[base + i];
        return a.raw ? cskip * x + cout_ * f : f;                       // networks_edm.py:495
    };
    for (int i = threadIdx.x; i < per; i += blockDim.x) vals[i] = __float_as_uint(fabsf(denoised(i)));
    if (threadIdx.x == 0) { s_cnt = 0; s_min = 0xffffffffu; }
    __syncthreads();
    const float rank = p * (float)(per - 1);
    const float lo_f = floorf(rank);
    const float w = rank - lo_f;
    const unsigned lo = (unsigned)lo_f;
    const unsigned hi = (unsigned)ceilf(rank);
    const unsigned v_lo = radix_select(vals, per, lo, hist);
    unsigned v_hi = v_lo;
    if (hi != lo) {
        unsigned cnt = 0, mn = 0xffffffffu;
        for (int i = threadIdx.x; i < per; i += blockDim.x) {
            const unsigned v = vals[i];
            if (v <= v_lo) ++cnt; else mn = min(mn, v);
        }
        atomicAdd(&s_cnt, cnt);
        atomicMin(&s_min, mn);
        __syncthreads();
        v_hi = (s_cnt >= hi + 1) ? v_lo : s_min;
    }
    const float qa = __uint_as_float(v_lo), qb = __uint_as_float(v_hi);
    float s = (w < 0.5f) ? qa + w * (qb - qa) : qb - (qb - qa) * (1.0f - w);
    s = fmaxf(s, 1.0f);
    for (int i = threadIdx.x; i < per; i += blockDim.x) {
        const float m = fminf(fmaxf(denoised(i), -s), s) / s;
        if (a.m_out) a.m_out[base + i] = m;
        if (a.x_out) {
            float acc = k.cx * a.xb[base + i] + k.cm * m;
            if (a.hist[0]) acc += k.ch0 * a.hist[0][base + i];
            if (a.hist[1]) acc += k.ch1 * a.hist[1][base + i];
            a.x_out[base + i] = acc;
        }
    }
}

// The same step with the sample held in REGISTERS (samples of THREADS * VPT values: 3 x 32 x 32, 3 x 64 x 64, 4 x 64 x 64, tiny test
// nets): every operand is read from HBM exactly once as non-temporal 16-B accesses (x, F, the one or two history tensors -- issued
// before the select so that they are in flight under it) and x', m0 are written once: 3-4 R + 2 W passes per image against the
// 5 R + 2 W of dpmpp_x0_step_kernel, whose dword loads and serial 256-bin histogram scans also make it latency-bound (52 us per
// launch whatever the batch).  The order statistics come from a three-digit (11 + 10 + 10 bit) MSB-first radix select over the
// register-resident |D| bit patterns: LDS histogram by atomics (the 11-bit first digit carries three mantissa bits, so the few
// exponents a sample spans spread over 8x the bins), block-parallel prefix scan to find the digit.  Results are the exact order
// statistics, hence bit-identical to the LDS kernel and to torch.quantile.  9 workgroups of 256 threads fit a CU (16.1 KB LDS each).
template <int THREADS, int VPT>
__global__ void __launch_bounds__(THREADS) dpmpp_x0_step_reg_kernel(const ds_update_args a, float p) {
    constexpr int NV = VPT / 4, NW = THREADS / 64, per = THREADS * VPT;
    __shared__ __attribute__((aligned(16))) unsigned hist[4096];        // digit 0: [0, 2048), digit 1: [2048, 3072), digit 2: [3072, 4096)
    __shared__ unsigned s_wsum[16];
    __shared__ unsigned s_sel[2];
    __shared__ unsigned s_cnt, s_min;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.x;
    const size_t base = (size_t)img * per;
    const bool has_xb = a.xb != a.xe;
    f32x4 xv[NV], dv[NV], h0[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const size_t off = base + (size_t)(j * THREADS + tid) * 4;
        xv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.xe + off));
        if (!a.afs) dv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.f + off));
    }
    // the newest history tensor (every 2M / 3M step) is requested now and lands under the select; a separate xb and the second history
    // tensor (3M steps only) are read in the output loop, which keeps the kernel at <= 80 registers (6 workgroups of 256 per CU)
    if (a.x_out && a.hist[0]) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
            h0[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.hist[0] + base + (size_t)(j * THREADS + tid) * 4));
    }
    {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < 1024; i += THREADS) reinterpret_cast<f32x4*>(hist)[i] = z;
        if (tid == 0) { s_cnt = 0; s_min = 0xffffffffu; }
    }
    const Coefs k = load_coefs(a, img);
    float cskip = 0.f, cout_ = 0.f;
    if (a.raw) { cskip = ds_c_skip(k.sig, a.sigma_data); cout_ = ds_c_out(k.sig, a.sigma_data); }
    const float afs_div = sqrtf(1.0f + k.t * k.t);
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = xv[j][e];
            float D;
            if (a.afs) D = x - k.t * (x / afs_div);                        // solvers.py:77, :680
            else { const float f = dv[j][e]; D = a.raw ? cskip * x + cout_ * f : f; }      // networks_edm.py:495
            dv[j][e] = D;
        }
    auto key = [&](int i) -> unsigned { return __float_as_uint(dv[i >> 2][i & 3]) & 0x7fffffffu; };      // bit pattern of |D|
    __syncthreads();
    const float rank_f = p * (float)(per - 1);
    const float lo_f = floorf(rank_f);
    const float w = rank_f - lo_f;
    const unsigned lo = (unsigned)lo_f;
    const unsigned hi = (unsigned)ceilf(rank_f);

    // block-parallel search of the bin that holds order statistic `rank` in a BINS-bin histogram: each thread owns BINS / THREADS
    // consecutive bins; returns {bin, number of keys before it}
    unsigned rank = lo, prefix = 0;
    auto find_bin = [&](const unsigned* hp, auto binsc) {
        constexpr int BINS = decltype(binsc)::value;
        constexpr int BPT = BINS / THREADS;
        static_assert(BPT >= 1, "bins per thread");
        unsigned hv[BPT], sum = 0;
#pragma unroll
        for (int b = 0; b < BPT; ++b) { hv[b] = hp[tid * BPT + b]; sum += hv[b]; }
        unsigned v = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(v, o); if (lane >= o) v += t; }
        if (lane == 63) s_wsum[wave] = v;
        __syncthreads();
        unsigned excl = v - sum;
#pragma unroll
        for (int q = 0; q < NW; ++q) if (q < wave) excl += s_wsum[q];
        if (excl <= rank && rank < excl + sum) {                           // exactly one thread
            unsigned cum = excl, digit = 0, before = excl;
            bool found = false;
#pragma unroll
            for (int b = 0; b < BPT; ++b) {
                if (!found && cum + hv[b] > rank) { digit = (unsigned)(tid * BPT + b); before = cum; found = true; }
                cum += hv[b];
            }
            s_sel[0] = digit; s_sel[1] = before;
        }
        __syncthreads();
        rank -= s_sel[1];
        return s_sel[0];
    };
    // digit 0: key bits 30..20
#pragma unroll
    for (int i = 0; i < VPT; ++i) atomicAdd(&hist[key(i) >> 20], 1u);
    __syncthreads();
    prefix = find_bin(hist, std::integral_constant<int, (2048 > THREADS ? 2048 : THREADS)>{});
    // digit 1: bits 19..10 of the keys whose top digit matches
#pragma unroll
    for (int i = 0; i < VPT; ++i) if ((key(i) >> 20) == prefix) atomicAdd(&hist[2048 + ((key(i) >> 10) & 1023u)], 1u);
    __syncthreads();
    prefix = (prefix << 10) | find_bin(hist + 2048, std::integral_constant<int, 1024>{});
    // digit 2: bits 9..0
#pragma unroll
    for (int i = 0; i < VPT; ++i) if ((key(i) >> 10) == prefix) atomicAdd(&hist[3072 + (key(i) & 1023u)], 1u);
    __syncthreads();
    const unsigned v_lo = (prefix << 10) | find_bin(hist + 3072, std::integral_constant<int, 1024>{});
    unsigned v_hi = v_lo;
    if (hi != lo) {
        unsigned cnt = 0, mn = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < VPT; ++i) { if (key(i) <= v_lo) ++cnt; else mn = min(mn, key(i)); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { cnt += __shfl_xor(cnt, o); mn = min(mn, (unsigned)__shfl_xor(mn, o)); }
        if (lane == 0) { atomicAdd(&s_cnt, cnt); atomicMin(&s_min, mn); }
        __syncthreads();
        v_hi = (s_cnt >= hi + 1) ? v_lo : s_min;
    }
    const float qa = __uint_as_float(v_lo), qb = __uint_as_float(v_hi);
    float s = (w < 0.5f) ? qa + w * (qb - qa) : qb - (qb - qa) * (1.0f - w);       // at::lerp
    s = fmaxf(s, 1.0f);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const size_t off = base + (size_t)(j * THREADS + tid) * 4;
        f32x4 mv, xo, xb = xv[j], h1 = {0.f, 0.f, 0.f, 0.f};
        if (a.x_out && has_xb) xb = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.xb + off));
        if (a.x_out && a.hist[1]) h1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.hist[1] + off));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float m = fminf(fmaxf(dv[j][e], -s), s) / s;
            mv[e] = m;
            float acc = k.cx * xb[e] + k.cm * m;
            if (a.hist[0]) acc += k.ch0 * h0[j][e];
            if (a.hist[1]) acc += k.ch1 * h1[e];
            xo[e] = acc;
        }
        if (a.m_out) __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(a.m_out + off));
        if (a.x_out) __builtin_nontemporal_store(xo, reinterpret_cast<f32x4*>(a.x_out + off));
    }
}

// CFGPrecond epilogue: D = x - sigma * F, F = Fu + g (Fc - Fu) for a doubled evaluation.  Thread = one pixel (all channels):
// the NHWC row of F is one 16-B load when f_ld == 4.
__global__ void __launch_bounds__(256) cfg_denoise_kernel(const float* __restrict__ x, const float* __restrict__ f, int f_ld,
                                                          const float* __restrict__ sigma, int sigma_rows, float g, int doubled, int n,
                                                          int c, int hw, float* __restrict__ out) {
    const long long total = (long long)n * hw;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(pix / hw);
        const int p = (int)(pix - (long long)img * hw);
        const float sg = sigma[sigma_rows == 1 ? 0 : img];
        const float* fu = f + pix * f_ld;
        const float* fc = f + (pix + total) * f_ld;
        for (int ch = 0; ch < c; ++ch) {
            float fx = fu[ch];
            if (doubled) fx = fx + g * (fc[ch] - fx);
            const size_t o = ((size_t)img * c + ch) * hw + p;
            out[o] = x[o] - sg * fx;
        }
    }
}

}  // namespace

extern "C" int ds_solver_update(const ds_update_args* a, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!a || !a->xe || !a->xb) return DS_E_ARG;
    if (!a->afs && !a->f) return DS_E_ARG;
    if (!a->x_out && !a->m_out) return DS_E_ARG;
    if (a->raw && !a->afs && a->f_ld != 0 && a->f_ld < a->c) return DS_E_ARG;
    if (a->n <= 0 || a->c <= 0 || a->h <= 0 || a->w <= 0) return DS_E_ARG;
    if (a->coefs && a->coef_rows != 1 && a->coef_rows != a->n) return DS_E_ARG;
    const int HW = a->h * a->w;
    bool vec4 = (HW % 4 == 0) && ds_aligned16(a->xe) && ds_aligned16(a->xb) && ((a->raw && a->f_ld > 0) || !a->f || ds_aligned16(a->f)) &&
                (!a->x_out || ds_aligned16(a->x_out)) && (!a->m_out || ds_aligned16(a->m_out));
    for (int i = 0; i < 3; ++i) if (a->hist[i] && !ds_aligned16(a->hist[i])) vec4 = false;
    const long long work = (long long)a->n * (vec4 ? HW / 4 : HW);
    long long blocks = (work + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const bool fast = vec4 && (a->c == 3 || a->c == 4) && (!a->raw || a->afs || a->f_ld == 0 || (a->f_ld == 4 && ds_aligned16(a->f)));
    if (fast && a->c == 3) hipLaunchKernelGGL(solver_update_fast_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    else if (fast) hipLaunchKernelGGL(solver_update_fast_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    else if (vec4) hipLaunchKernelGGL(solver_update_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    else hipLaunchKernelGGL(solver_update_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_table_select(const float* table, int row_floats, int* step, int advance, float* dst, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!table || !step || !dst || row_floats <= 0) return DS_E_ARG;
    hipLaunchKernelGGL(table_select_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, table, row_floats, step, advance, dst);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_dynamic_threshold(const float* x0, float* out, int n, int per, float p, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!x0 || !out || n <= 0 || per <= 1) return DS_E_ARG;
    const size_t smem = ((size_t)per + 256) * sizeof(unsigned);
    if (smem > 150 * 1024) return DS_E_SHAPE;
    DS_ENSURE_DYN_LDS((&dynamic_threshold_kernel), 150 * 1024);
    hipLaunchKernelGGL(dynamic_threshold_kernel, dim3(n), dim3(512), smem, (hipStream_t)stream, x0, out, per, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}


// 1 when ds_dpmpp_x0_step runs a sample of `per` values on the register-resident kernel (given 16-B aligned tensors)
extern "C" int ds_dpmpp_x0_step_in_registers(long long per) {
    return (per == 64 * 12 || per == 256 * 12 || per == 512 * 24 || per == 1024 * 16) ? 1 : 0;
}

extern "C" int ds_dpmpp_x0_step(const ds_update_args* a, float p, void* stream) {
    (void)hipGetLastError();
    if (!a || !a->xe || !a->xb || (!a->afs && !a->f)) return DS_E_ARG;
    if (!a->x_out && !a->m_out) return DS_E_ARG;
    if (a->n <= 0 || a->c <= 0 || a->h <= 0 || a->w <= 0) return DS_E_ARG;
    if (a->raw && !a->afs && a->f_ld != 0) return DS_E_ARG;                  // the raw network output must be channel-planar here
    if (a->coefs && a->coef_rows != 1 && a->coef_rows != a->n) return DS_E_ARG;
    const long long per = (long long)a->c * a->h * a->w;
    bool al = ds_aligned16(a->xe) && ds_aligned16(a->xb) && (a->afs || ds_aligned16(a->f)) && (!a->x_out || ds_aligned16(a->x_out)) &&
              (!a->m_out || ds_aligned16(a->m_out)) && !a->hist[2];
    for (int i = 0; i < 2; ++i) if (a->hist[i] && !ds_aligned16(a->hist[i])) al = false;
    if (a->variant == 0 && al && ds_dpmpp_x0_step_in_registers(per)) {
        const dim3 grid(a->n);
        if (per == 64 * 12) hipLaunchKernelGGL((dpmpp_x0_step_reg_kernel<64, 12>), grid, dim3(64), 0, (hipStream_t)stream, *a, p);
        else if (per == 256 * 12) hipLaunchKernelGGL((dpmpp_x0_step_reg_kernel<256, 12>), grid, dim3(256), 0, (hipStream_t)stream, *a, p);
        else if (per == 512 * 24) hipLaunchKernelGGL((dpmpp_x0_step_reg_kernel<512, 24>), grid, dim3(512), 0, (hipStream_t)stream, *a, p);
        else hipLaunchKernelGGL((dpmpp_x0_step_reg_kernel<1024, 16>), grid, dim3(1024), 0, (hipStream_t)stream, *a, p);
        DS_CHECK_LAUNCH();
        return DS_OK;
    }
    const size_t smem = ((size_t)per + 256) * sizeof(unsigned);
    if (per <= 1 || smem > 150 * 1024) return DS_E_SHAPE;
    DS_ENSURE_DYN_LDS((&dpmpp_x0_step_kernel), 150 * 1024);
    hipLaunchKernelGGL(dpmpp_x0_step_kernel, dim3(a->n), dim3(512), smem, (hipStream_t)stream, *a, p);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_cfg_denoise(const float* x, const float* f, int f_ld, const float* sigma, int sigma_rows, float guidance, int doubled,
                              int n, int c, int h, int w, float* out, void* stream) {
    (void)hipGetLastError();
    if (!x || !f || !sigma || !out || n <= 0 || c <= 0 || h <= 0 || w <= 0 || f_ld < c || (sigma_rows != 1 && sigma_rows != n)) return DS_E_ARG;
    long long blocks = ((long long)n * h * w + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cfg_denoise_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, f, f_ld, sigma, sigma_rows, guidance,
                       doubled, n, c, h * w, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_scale(const float* x, float a, float* y, long long count, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!x || !y || count <= 0) return DS_E_ARG;
    if (!ds_aligned16(x) || !ds_aligned16(y)) return DS_E_ALIGN;
    long long blocks = ((count >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, a, y, count);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_quantize_u8_nhwc(const float* x, uint8_t* out, int n, int c, int h, int w, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!x || !out || n <= 0 || c <= 0) return DS_E_ARG;
    long long blocks = ((long long)n * h * w + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(quantize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, out, n, c, h, w);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_fill(float* dst, float value, long long count, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!dst || count <= 0) return DS_E_ARG;
    long long blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst, value, count);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_copy_rows(const float* src, int src_ld, float* dst, int dst_ld, long long rows, int cols, void* stream) {
    (void)hipGetLastError();   // drop stale errors of unrelated runtime calls
    if (!src || !dst || rows <= 0 || cols <= 0) return DS_E_ARG;
    long long blocks = (rows * cols + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, src_ld, dst, dst_ld, rows, cols);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_build_experiments(void) {          // bit 0: DS_BUILD_EXPERIMENTS (A/B-record kernels present); bit 1: DS_RACE_STRESS (tests-only delayed build)
    int flags = 0;
#ifdef DS_BUILD_EXPERIMENTS
    flags |= 1;
#endif
#ifdef DS_RACE_STRESS
    flags |= 2;
#endif
    return flags;
}
extern "C" int ds_version(void) { return 4; }      // ABI 4: ds_norm_args.stats0 / stats1 / tune_variant appended; ABI 3: ds_conv_args.update appended (head-fused solver update), ds_build_experiments(); ABI 2: ds_conv_args.tune / ds_update_args.variant, ds_fid_moments

extern "C" const char* ds_error_string(int code) {
    switch (code) {
        case DS_OK: return "ok";
        case DS_E_ARG: return "invalid argument";
        case DS_E_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        case DS_E_SHAPE: return "unsupported shape";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}
