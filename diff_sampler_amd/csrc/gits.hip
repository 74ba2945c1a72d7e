// GITS schedule search, device side (gits-main/gits_utils.py:108-132, 237-255).
//
// The reference evaluates, for all N(N-1)/2 pairs (i, j) of a teacher trajectory, an Euler jump
// x_next = x_i + (t_j - t_i) d_i and a full-tensor metric on it -- ~1 830 x several elementwise/reduction passes over
// [B, C, H, W] for N = 61.  For the default metric ('dev': norm of the component of (c - x_next) perpendicular to the
// start->end chord bc) the jump is affine in (t_j - t_i), so every pair cost follows in closed form from SIX inner
// products per trajectory point and sample:
//     P = (c - x_i).bc   Q = d_i.bc   R = |c - x_i|^2   S = (c - x_i).d_i   T = |d_i|^2   N = |bc|^2
//     dev(i, j) = sqrt( R - 2 D S + D^2 T - (P - D Q)^2 / N ),  D = t_j - t_i
// ds_traj_moments computes them in one pass over the trajectory (fp64 accumulation: the difference of squares
// cancels).  'l1' / 'l2' have no such form (l1) or need all pairwise Grams (l2); ds_traj_pair_cost evaluates them
// directly, one block per (pair, sample), with the trajectory resident in the 256 MB Infinity Cache.
#include "ds_common.h"

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;
}

// grid = (n_pts, batch); out[(i*batch + b)*6 + {P,Q,R,S,T,N}]
__global__ void __launch_bounds__(256) traj_moments_kernel(const float* __restrict__ traj, const float* __restrict__ eps, int n_pts,
                                                           int batch, int per, double* __restrict__ out) {
    __shared__ double sh[4];
    const int i = blockIdx.x, b = blockIdx.y;
    const float* x = traj + ((size_t)i * batch + b) * per;
    const float* s0 = traj + (size_t)b * per;
    const float* c = traj + ((size_t)(n_pts - 1) * batch + b) * per;
    const float* d = (eps && i < n_pts - 1) ? eps + ((size_t)i * batch + b) * per : nullptr;
    double P = 0, Q = 0, R = 0, S = 0, T = 0, N = 0;
    for (int e = threadIdx.x; e < per; e += blockDim.x) {
        const double cv = c[e], bc = cv - (double)s0[e], cx = cv - (double)x[e];
        const double dv = d ? (double)d[e] : 0.0;
        P += cx * bc; Q += dv * bc; R += cx * cx; S += cx * dv; T += dv * dv; N += bc * bc;
    }
    double* o = out + ((size_t)i * batch + b) * 6;
    P = block_sum(P, sh); Q = block_sum(Q, sh); R = block_sum(R, sh); S = block_sum(S, sh); T = block_sum(T, sh); N = block_sum(N, sh);
    if (threadIdx.x == 0) { o[0] = P; o[1] = Q; o[2] = R; o[3] = S; o[4] = T; o[5] = N; }
}

// grid = (n_pts * n_pts, batch); cost[i*n_pts + j] += | x_i + (t_j - t_i) d_i - x_j |_p   (p = 1 or 2), i < j
__global__ void __launch_bounds__(256) traj_pair_cost_kernel(const float* __restrict__ traj, const float* __restrict__ eps,
                                                             const float* __restrict__ t, int n_pts, int batch, int per, int p_norm,
                                                             double* __restrict__ cost) {
    __shared__ double sh[4];
    const int i = blockIdx.x / n_pts, j = blockIdx.x - i * n_pts, b = blockIdx.y;
    if (j <= i || i >= n_pts - 1) return;
    const float* xi = traj + ((size_t)i * batch + b) * per;
    const float* xj = traj + ((size_t)j * batch + b) * per;
    const float* di = eps + ((size_t)i * batch + b) * per;
    const float dt = t[j] - t[i];
    double acc = 0.0;
    for (int e = threadIdx.x; e < per; e += blockDim.x) {
        const float r = (xi[e] + dt * di[e]) - xj[e];          // fp32 like the reference's x_next and difference
        acc += (p_norm == 1) ? fabs((double)r) : (double)r * (double)r;
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(&cost[(size_t)i * n_pts + j], (p_norm == 1) ? acc : sqrt(acc));
}

}  // namespace

extern "C" int ds_traj_moments(const float* traj, const float* eps, int n_pts, int batch, int per, double* out, void* stream) {
    (void)hipGetLastError();
    if (!traj || !out || n_pts < 2 || batch <= 0 || per <= 0) return DS_E_ARG;
    hipLaunchKernelGGL(traj_moments_kernel, dim3(n_pts, batch), dim3(256), 0, (hipStream_t)stream, traj, eps, n_pts, batch, per, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_traj_pair_cost(const float* traj, const float* eps, const float* t_steps, int n_pts, int batch, int per, int p_norm,
                                 double* cost, void* stream) {
    (void)hipGetLastError();
    if (!traj || !eps || !t_steps || !cost || n_pts < 2 || batch <= 0 || per <= 0 || (p_norm != 1 && p_norm != 2)) return DS_E_ARG;
    hipLaunchKernelGGL(traj_pair_cost_kernel, dim3(n_pts * n_pts, batch), dim3(256), 0, (hipStream_t)stream, traj, eps, t_steps, n_pts,
                       batch, per, p_norm, cost);
    DS_CHECK_LAUNCH();
    return DS_OK;
}
