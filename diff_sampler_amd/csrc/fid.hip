// FID moment accumulation on the fp64 matrix pipe (diff-solvers-main/fid.py:62-71):
//     mu    += features.sum(0)                    [dim]        fp64
//     sigma += features.T @ features              [dim][dim]   fp64
// for one batch of detector features [rows][dim] (fp32 as the detector emits them, widened to fp64 exactly as the reference's
// `.to(torch.float64)` does, or already fp64), accumulated IN PLACE, one launch per batch.
//
// Work per batch: 2 * rows * dim^2 fp64 FLOP (rows = 64, dim = 2048: 0.54 GFLOP) on v_mfma_f64_16x16x4_f64 (78.6 TFLOP/s chip peak),
// plus one read-modify-write of sigma (2 * dim^2 * 8 B = 64 MiB at dim = 2048: ~11 us at 6 TB/s) -- HBM-bound up to ~100 rows per
// batch, matrix-bound above.  The features themselves (rows * dim * 4 B = 0.5 MB) stay in L2.
//
// Tile: one 256-thread workgroup per 64 x 64 block of sigma, four waves of 32 x 32 = 2 x 2 MFMA blocks (32 accumulator VGPRs), operands
// straight from global memory: for sigma[i][j] = sum_r F[r][i] * F[r][j] both MFMA operands are rows of F -- lane l of a k step holds
// F[r0 + (l >> 4)][i0 + (l & 15)] (A) and F[r0 + (l >> 4)][j0 + (l & 15)] (B): 64-byte contiguous segments, no transposition anywhere.
// The workgroup's sigma block is requested BEFORE the row loop, so the read half of the read-modify-write lies under the MFMAs.
// sigma[i][j] and sigma[j][i] add the same products in the same order: the update is bitwise symmetric.
// Column blocks with blockIdx.x == 0 also add their 64 column sums to mu (rows in order, fp64).
#include "ds_common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <bool F64>
__device__ __forceinline__ double feat_at(const void* f, size_t idx) {
    if constexpr (F64) return reinterpret_cast<const double*>(f)[idx];
    else return (double)reinterpret_cast<const float*>(f)[idx];
}

template <bool F64>
__global__ void __launch_bounds__(256) fid_moments_kernel(const void* __restrict__ feat, int ld, int rows, int dim,
                                                         double* __restrict__ mu, double* __restrict__ sigma) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.y * 64 + (wave >> 1) * 32;          // this wave's 32 rows of sigma (feature index i)
    const int j0 = blockIdx.x * 64 + (wave & 1) * 32;           // ... and 32 columns (feature index j)
    const int lc = lane & 15, lk = lane >> 4;

    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg (cdna_hip_programming.md section 3)
    f64x4 old[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + m * 16 + lk + 4 * r, j = j0 + n * 16 + lc;
                old[m][n][r] = (i < dim && j < dim) ? sigma[(size_t)i * dim + j] : 0.0;
            }

    f64x4 acc[2][2] = {};
    const bool ci[2] = {i0 + lc < dim, i0 + 16 + lc < dim}, cj[2] = {j0 + lc < dim, j0 + 16 + lc < dim};
    auto k_step = [&](int r0, bool ragged) {                    // four feature rows: one MFMA k step
        const int r = r0 + lk;
        const bool rv = !ragged || r < rows;
        const size_t base = (size_t)(rv ? r : 0) * ld;
        double a[2], b[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = (rv && ci[m]) ? feat_at<F64>(feat, base + i0 + m * 16 + lc) : 0.0;
#pragma unroll
        for (int n = 0; n < 2; ++n) b[n] = (rv && cj[n]) ? feat_at<F64>(feat, base + j0 + n * 16 + lc) : 0.0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    };
    int r0 = 0;
    for (; r0 + 16 <= rows; r0 += 16) {                         // 16 rows per trip: the 16 loads of a trip are independent of its MFMAs
#pragma unroll
        for (int u = 0; u < 4; ++u) k_step(r0 + 4 * u, false);
    }
    for (; r0 < rows; r0 += 4) k_step(r0, true);

#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + m * 16 + lk + 4 * r, j = j0 + n * 16 + lc;
                if (i < dim && j < dim) sigma[(size_t)i * dim + j] = old[m][n][r] + acc[m][n][r];
            }

    if (blockIdx.x == 0 && wave == 0) {                         // column sums of this block's 64 features
        const int c = blockIdx.y * 64 + lane;
        if (c < dim) {
            double s = 0.0;
            for (int r = 0; r < rows; ++r) s += feat_at<F64>(feat, (size_t)r * ld + c);
            mu[c] += s;
        }
    }
}

}  // namespace

extern "C" int ds_fid_moments(const void* features, int features_f64, int ld, int rows, int dim, double* mu, double* sigma, void* stream) {
    (void)hipGetLastError();
    if (!features || !mu || !sigma || rows < 0 || dim <= 0 || ld < dim) return DS_E_ARG;
    if (features_f64 != 0 && features_f64 != 1) return DS_E_ARG;
    if (rows == 0) return DS_OK;
    const unsigned t = (unsigned)((dim + 63) / 64);
    if (features_f64) hipLaunchKernelGGL(fid_moments_kernel<true>, dim3(t, t), dim3(256), 0, (hipStream_t)stream, features, ld, rows, dim, mu, sigma);
    else hipLaunchKernelGGL(fid_moments_kernel<false>, dim3(t, t), dim3(256), 0, (hipStream_t)stream, features, ld, rows, dim, mu, sigma);
    DS_CHECK_LAUNCH();
    return DS_OK;
}
