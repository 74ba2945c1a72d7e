// FID moment accumulation on the fp64 matrix pipe (diff-solvers-main/fid.py:62-71):
//     mu    += features.sum(0)                    [dim]        fp64
//     sigma += features.T @ features              [dim][dim]   fp64
// for one batch of detector features [rows][dim] (fp32 as the detector emits them, widened to fp64 exactly as the reference's
// `.to(torch.float64)` does, or already fp64), accumulated IN PLACE, one launch per batch.
//
// Work per batch: 2 * rows * dim^2 fp64 FLOP (rows = 64, dim = 2048: 0.54 GFLOP; 250 rows: 2.1 GFLOP) on v_mfma_f64_16x16x4_f64
// (78.6 TFLOP/s chip peak = 64 cycles per instruction and SIMD), plus one read-modify-write of sigma (2 * dim^2 * 8 B = 64 MiB at
// dim = 2048: ~11 us at 6 TB/s): matrix-bound from a few dozen rows per batch on.  The features (rows * dim * 4 B) stay in L2.
//
// Tile: one 256-thread workgroup per 128 x 128 block of sigma (dim = 2048: 256 workgroups = one per CU), four waves of 64 x 64 =
// 4 x 4 MFMA blocks (128 accumulator VGPRs).  For sigma[i][j] = sum_r F[r][i] * F[r][j] both MFMA operands are ROWS of F -- lane l of a
// k step holds F[r0 + (l >> 4)][i0 + (l & 15)] (A) and F[r0 + (l >> 4)][j0 + (l & 15)] (B): no transposition anywhere.  The rows are
// staged 16 at a time as fp64 in LDS ([16][128 + 16] doubles per operand; the 16-double pad puts consecutive rows 32 banks apart, so the
// 2 rows x 16 doubles a ds_read_b64 lane group touches are conflict free), coalesced 4-byte loads, widened once per element; the next 16
// rows are requested into registers before the 64 MFMAs of the current ones.  Per 16 rows a wave issues 32 ds_read_b64 for 64 MFMAs
// (4 096 matrix cycles): the loop is matrix-bound by construction.  // sigma[i][j] and sigma[j][i] add the same products in the same order: the update is bitwise symmetric.
// Workgroups of block column 0 also add the column sums of their 128 features to mu (rows in order, fp64, from the staged tile).
#include <type_traits>

#include "ds_common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int FT = 128;                 // features per tile side
constexpr int FR = 16;                  // feature rows per staged chunk
constexpr int FP = FT + 16;             // LDS row pitch in doubles

template <bool F64>
__device__ __forceinline__ double feat_at(const void* f, size_t idx) {
    if constexpr (F64) return reinterpret_cast<const double*>(f)[idx];
    else return (double)reinterpret_cast<const float*>(f)[idx];
}

template <bool F64>
__global__ void __launch_bounds__(256) fid_moments_kernel(const void* __restrict__ feat, int ld, int rows, int dim,
                                                         double* __restrict__ mu, double* __restrict__ sigma) {
    __shared__ double tile[2][FR][FP];                          // [operand: 0 = i side, 1 = j side][row][feature]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bi = blockIdx.y * FT, bj = blockIdx.x * FT;
    const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;       // this wave's 64 x 64 corner inside the tile
    const int lc = lane & 15, lk = lane >> 4;
    const int sc = tid & (FT - 1), sr = tid >> 7;                // staging: feature column, first row (rows sr, sr + 2, ...)

    // Requests are UNCONDITIONAL loads from clamped (always valid) addresses, zeroed by a select when they are staged: a predicated load
    // compiles to a branch with a full vmcnt(0) behind each of the 16 loads of a chunk (first version: 14 -> 22 TFLOP/s only).  The raw
    // values stay in registers as loaded (fp32 features: widened when they are written to LDS).
    typedef typename std::conditional<F64, double, float>::type raw_t;
    raw_t pre[2][FR / 2];
    const raw_t* fsrc = reinterpret_cast<const raw_t*>(feat);
    const int ci_ = min(bi + sc, dim - 1), cj_ = min(bj + sc, dim - 1);
    const bool vi = bi + sc < dim, vj = bj + sc < dim;
    auto request = [&](int r0) {                                // 16 rows x 128 features per operand -> registers
#pragma unroll
        for (int u = 0; u < FR / 2; ++u) {
            const size_t ro = (size_t)min(r0 + sr + 2 * u, rows - 1) * ld;
            pre[0][u] = fsrc[ro + ci_];
            pre[1][u] = fsrc[ro + cj_];                          // (diagonal workgroups: the same addresses again -- L1 hits, no branch)
        }
    };
    auto staged = [&](int o, int u, int r0) -> double {         // the value of row r0 + sr + 2u as staged: zero outside [rows) x [dim)
        const bool ok = (r0 + sr + 2 * u < rows) && (o ? vj : vi);
        return ok ? (double)pre[o][u] : 0.0;
    };
    f64x4 acc[4][4] = {};
    double colsum = 0.0;
    const double* ta = &tile[0][0][0];
    const double* tb = &tile[1][0][0];
    request(0);
    for (int r0 = 0; r0 < rows; r0 += FR) {
        __syncthreads();                                        // the previous chunk's reads are done
        DS_RACE_SKEW(tid >> 6);
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int u = 0; u < FR / 2; ++u) tile[o][sr + 2 * u][sc] = staged(o, u, r0);
        __syncthreads();
        if (r0 + FR < rows) request(r0 + FR);                    // in flight under this chunk's MFMAs
        if (blockIdx.x == 0 && tid < FT) {                      // mu: column sums of the i-side tile, rows in order
#pragma unroll
            for (int k = 0; k < FR; ++k) colsum += tile[0][k][tid];
        }
#pragma unroll
        for (int ks = 0; ks < FR / 4; ++ks) {
            double a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = ta[(ks * 4 + lk) * FP + wi + m * 16 + lc];
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = tb[(ks * 4 + lk) * FP + wj + n * 16 + lc];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }

    // read-modify-write of the wave's 64 x 64 block, one 16-row band at a time (16 loads in flight per lane)
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg (cdna_hip_programming.md section 3)
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double old[4][4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = bi + wi + m * 16 + lk + 4 * r, j = bj + wj + n * 16 + lc;
                old[n][r] = (i < dim && j < dim) ? sigma[(size_t)i * dim + j] : 0.0;
            }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = bi + wi + m * 16 + lk + 4 * r, j = bj + wj + n * 16 + lc;
                if (i < dim && j < dim) sigma[(size_t)i * dim + j] = old[n][r] + acc[m][n][r];
            }
    }
    if (blockIdx.x == 0 && tid < FT && bi + tid < dim) mu[bi + tid] += colsum;
}

}  // namespace

extern "C" int ds_fid_moments(const void* features, int features_f64, int ld, int rows, int dim, double* mu, double* sigma, void* stream) {
    (void)hipGetLastError();
    if (!features || !mu || !sigma || rows < 0 || dim <= 0 || ld < dim) return DS_E_ARG;
    if (features_f64 != 0 && features_f64 != 1) return DS_E_ARG;
    if (rows == 0) return DS_OK;
    const unsigned t = (unsigned)((dim + FT - 1) / FT);
    if (features_f64) hipLaunchKernelGGL(fid_moments_kernel<true>, dim3(t, t), dim3(256), 0, (hipStream_t)stream, features, ld, rows, dim, mu, sigma);
    else hipLaunchKernelGGL(fid_moments_kernel<false>, dim3(t, t), dim3(256), 0, (hipStream_t)stream, features, ld, rows, dim, mu, sigma);
    DS_CHECK_LAUNCH();
    return DS_OK;
}
