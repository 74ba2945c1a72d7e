// Batched per-seed latent generator.
//
// The reference draws every image's latent from its own generator (diff-solvers-main/sample.py:22-36):
//     torch.Generator(device).manual_seed(seed % 2**32)  ->  torch.randn([C, H, W], generator=g)
// i.e. B generator constructions and B tiny launches per batch.  On ROCm that randn is ATen's
// distribution_elementwise_grid_stride_kernel (ATen/native/cuda/DistributionTemplates.h) over hipRAND's Philox4x32-10:
//     threads_total = 256 * min(CUs * (maxThreadsPerCU / 256), ceil(n / 256));   thread idx: rocrand_init(seed, idx, offset)
//     element li = idx + threads_total * q  is component (q % 4) of that thread's (q / 4)-th rocrand_normal4 call
//     (for n <= threads_total -- every latent of the scope -- that is: element i = rocrand_normal4(seed, subsequence i, offset).x)
//     and the generator's offset then advances by ((n - 1) / (threads_total * 4) + 1) * 4.
// This kernel produces the same bits for a whole batch of seeds in ONE launch by evaluating exactly that map with rocRAND's
// Philox4x32-10 device functions (header-only: rocrand_philox4x32_10.h) and a Box-Muller pinned to the math functions of the installed
// torch build (box_muller_torch below).  randint(label_dim, size=[]) of the same
// generators (sample.py:283) is the first 32-bit output of the block at (seed, subsequence 0, offset) modulo the range
// (ATen random_from_to, ranges below 2**32).
#include <rocrand/rocrand_kernel.h>

#include "ds_common.h"

namespace {

// Box-Muller exactly as the INSTALLED torch (2.10 + ROCm 7.0 build) evaluates rocRAND's rocrand_normal4: same formula as
// rocrand_normal.h:53-68, but with the math functions that build resolved them to.  tools/probe_rng.py ran all 96 combinations of
// {logf builtin, OCML log, native log, log2-based} x {4 sqrt} x {3 sin/cos} x {fma-contracted, separate mul+add} against torch.randn
// on the GPU: the OCML library logarithm (__ocml_log_f32), the correctly rounded sqrtf, the native sin/cos and the contracted form
// reproduce it with 0 mismatches of 4096 (profiles/r2_rng_probe.json); this toolchain's own `logf` (an LLVM builtin expansion since
// ROCm 7.2) is 1-2 ulp off in ~40 % of the values.
__device__ __forceinline__ float2 box_muller_torch(unsigned int x, unsigned int y) {
    const float u = __builtin_fmaf((float)x, ROCRAND_2POW32_INV, ROCRAND_2POW32_INV);
    const float v = __builtin_fmaf((float)y, ROCRAND_2POW32_INV_2PI, ROCRAND_2POW32_INV_2PI);
    const float s = sqrtf(-2.0f * __ocml_log_f32(u));
    float2 r;
    r.x = __ocml_native_sin_f32(v) * s;
    r.y = __ocml_native_cos_f32(v) * s;
    return r;
}
__device__ __forceinline__ float4 normal4_torch(rocrand_state_philox4x32_10* st) {
    const uint4 r = rocrand4(st);
    const float2 a = box_muller_torch(r.x, r.y), b = box_muller_torch(r.z, r.w);
    return float4{a.x, a.y, b.x, b.y};
}

__global__ void __launch_bounds__(256) philox_randn_kernel(const unsigned long long* __restrict__ seeds, unsigned long long offset,
                                                           float* __restrict__ out, int batch, long long n, long long threads_total) {
    const long long total = (long long)batch * n;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(g / n);
        const long long li = g - (long long)b * n;
        const long long q = li / threads_total;
        const long long idx = li - q * threads_total;
        rocrand_state_philox4x32_10 st;
        rocrand_init(seeds[b], (unsigned long long)idx, offset + 4ull * (unsigned long long)(q >> 2), &st);
        const float4 v = normal4_torch(&st);
        const int c = (int)(q & 3);
        out[g] = c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
    }
}

__global__ void __launch_bounds__(256) philox_randint_kernel(const unsigned long long* __restrict__ seeds, unsigned long long offset,
                                                             unsigned int range, int* __restrict__ out, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seeds[b], 0ull, offset, &st);
    const uint4 v = rocrand4(&st);
    out[b] = (int)(v.x % range);
}

// ---- diagnostic: Box-Muller with selectable math-function variants (which build of logf / sqrtf / sincos does the installed torch
// use?  -- tools/probe_rng.py).  variant = ((contract * 3 + sincos) * 4 + sqrt) * 4 + log.
__device__ float bm_log(float u, int v) {
    switch (v) {
        case 0: return logf(u);
        case 1: return __ocml_log_f32(u);
        case 2: return __ocml_native_log_f32(u);
        default: return __builtin_amdgcn_logf(u) * 0.69314718055994530942f;
    }
}
__device__ float bm_sqrt(float x, int v) {
    switch (v) {
        case 0: return sqrtf(x);
        case 1: return __ocml_sqrt_f32(x);
        case 2: return __ocml_native_sqrt_f32(x);
        default: return __builtin_amdgcn_sqrtf(x);
    }
}
__global__ void __launch_bounds__(256) philox_probe_kernel(unsigned long long seed, unsigned long long offset, float* __restrict__ out, int n,
                                                           int variant) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int vl = variant & 3, vs = (variant >> 2) & 3, vc = (variant >> 4) % 3, vf = (variant >> 4) / 3;
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, (unsigned long long)i, offset, &st);
    const uint4 r = rocrand4(&st);
    float u, v;
    if (vf == 0) {
        u = __builtin_fmaf((float)r.x, ROCRAND_2POW32_INV, ROCRAND_2POW32_INV);
        v = __builtin_fmaf((float)r.y, ROCRAND_2POW32_INV_2PI, ROCRAND_2POW32_INV_2PI);
    } else {
        volatile float a = (float)r.x * ROCRAND_2POW32_INV, b = (float)r.y * ROCRAND_2POW32_INV_2PI;
        u = ROCRAND_2POW32_INV + a;
        v = ROCRAND_2POW32_INV_2PI + b;
    }
    const float s = bm_sqrt(-2.0f * bm_log(u, vl), vs);
    float sn;
    if (vc == 0) sn = __ocml_native_sin_f32(v);
    else if (vc == 1) sn = __ocml_sin_f32(v);
    else { float c; sincosf(v, &sn, &c); }
    out[i] = sn * s;
}

}  // namespace

extern "C" int ds_philox_probe(unsigned long long seed, unsigned long long offset, float* out, int n, int variant, void* stream) {
    if (!out || n < 1 || variant < 0 || variant >= 96) return DS_E_ARG;
    hipLaunchKernelGGL(philox_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, offset, out, n, variant);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_philox_randn(const unsigned long long* seeds, unsigned long long offset, float* out, int batch, long long n,
                               long long threads_total, void* stream) {
    if (!seeds || !out || batch < 1 || n < 1 || threads_total < 256 || threads_total % 256) return DS_E_ARG;
    const long long total = (long long)batch * n;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(philox_randn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, seeds, offset, out, batch, n, threads_total);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_philox_randint(const unsigned long long* seeds, unsigned long long offset, unsigned int range, int* out, int batch,
                                 void* stream) {
    if (!seeds || !out || batch < 1 || range < 1) return DS_E_ARG;
    hipLaunchKernelGGL(philox_randint_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seeds, offset, range, out, batch);
    DS_CHECK_LAUNCH();
    return DS_OK;
}
