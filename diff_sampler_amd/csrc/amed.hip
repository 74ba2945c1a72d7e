// AMED-Solver device side: the predictor MLP (amed-solver-main/training/networks.py:121-155) and the per-sample
// coefficient rows of the two stages of every AMED sampler (amed-solver-main/solvers_amed.py:123-148, 216-243,
// 320-382, 432-451, 560-612; DPM-Solver++ updates with `scale`: amed-solver-main/solver_utils.py:102-160).
// Everything here is per-sample scalar work (r, scale_dir, scale_time are [B] tensors in the reference), so it runs
// as one tiny kernel per stage instead of ~40 ATen launches on [B,1,1,1] tensors.
#include "ds_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float silu_exact(float v) { return v / (1.0f + expf(-v)); }

// One block (128 threads) per sample.
__global__ void __launch_bounds__(128) amed_predict_kernel(const ds_amed_predictor p, const float* __restrict__ bott, float t_cur,
                                                           float t_next, float* __restrict__ out) {
    __shared__ float s_h[256];
    __shared__ float s_feat[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* z = bott + (size_t)b * p.in_dim;
    // bottleneck encoder: hidden = silu(W0 z + b0)
    for (int o = tid; o < p.hidden; o += blockDim.x) {
        float acc = p.enc0_b[o];
        const float* w = p.enc0_w + (size_t)o * p.in_dim;
        for (int k = 0; k < p.in_dim; ++k) acc += w[k] * z[k];
        s_h[o] = silu_exact(acc);
    }
    // time embeddings of t_cur and t_next: PositionalEmbedding(nc, endpoint=True), [cos|sin] -> [sin|cos], Linear, SiLU
    if (tid < 2 * p.nc) {
        const int which = tid / p.nc, o = tid - which * p.nc;
        const float t = which ? t_next : t_cur;
        const int half = p.nc / 2;
        float acc = p.map0_b[o];
        for (int k = 0; k < p.nc; ++k) {
            const int i = (k < half) ? k : k - half;
            const float freq = powf(1.0f / 10000.0f, (float)i / (float)(half - 1));
            const float ang = t * freq;
            const float e = (k < half) ? sinf(ang) : cosf(ang);      // after the sin/cos swap
            acc += p.map0_w[o * p.nc + k] * e;
        }
        s_feat[p.out_dim + which * p.nc + o] = silu_exact(acc);
    }
    __syncthreads();
    if (tid < p.out_dim) {
        float acc = p.enc1_b[tid];
        const float* w = p.enc1_w + (size_t)tid * p.hidden;
        for (int k = 0; k < p.hidden; ++k) acc += w[k] * s_h[k];
        s_feat[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {
        const int nf = p.out_dim + 2 * p.nc;
        float r = p.fc_r_b[0];
        for (int k = 0; k < nf; ++k) r += p.fc_r_w[k] * s_feat[k];
        r = sigmoidf_(r);
        float sd = 1.0f, st = 1.0f;
        if (p.fc_sd_w) {
            float v = p.fc_sd_b[0];
            for (int k = 0; k < nf; ++k) v += p.fc_sd_w[k] * s_feat[k];
            sd = sigmoidf_(v) / (1.0f / (2.0f * p.scale_dir)) + (1.0f - p.scale_dir);
        }
        if (p.fc_st_w) {
            float v = p.fc_st_b[0];
            for (int k = 0; k < nf; ++k) v += p.fc_st_w[k] * s_feat[k];
            st = sigmoidf_(v) / (1.0f / (2.0f * p.scale_time)) + (1.0f - p.scale_time);
        }
        const float tm = powf(t_next, r) * powf(t_cur, 1.0f - r);        // solvers_amed.py:139
        float* o = out + (size_t)b * 4;
        o[0] = r; o[1] = sd; o[2] = st; o[3] = tm;
    }
}

// DPM-Solver++ coefficients for one sample (device version of solver_utils.dpmpp_coeffs).
__device__ void dpmpp_coeffs_dev(const float* th, int nh, float tn, int order, int px0, float scale, float& cx, float (&cm)[3]) {
    const float t0 = th[nh - 1];
    const float lam_n = -logf(tn), lam0 = -logf(t0);
    const float h = lam_n - lam0;
    const float phi1 = px0 ? expm1f(-h) : expm1f(h);
    cx = px0 ? tn / t0 : 1.0f;
    const float tf = px0 ? 1.0f : tn;
    cm[0] = cm[1] = cm[2] = 0.f;
    if (order == 1) { cm[0] = -scale * tf * phi1; return; }
    const float lam1 = -logf(th[nh - 2]);
    const float r0 = (lam0 - lam1) / h;
    if (order == 2) {
        cm[0] = -scale * tf * (phi1 + 0.5f * phi1 / r0);
        cm[1] = scale * tf * (0.5f * phi1 / r0);
        return;
    }
    const float lam2 = -logf(th[nh - 3]);
    const float r1 = (lam1 - lam2) / h;
    const float phi2 = px0 ? phi1 / h + 1.0f : phi1 / h - 1.0f;
    const float phi3 = phi2 / h - 0.5f;
    const float g = r0 / (r0 + r1), q = 1.0f / (r0 + r1);
    const float s2 = px0 ? 1.0f : -1.0f;
    const float A = s2 * phi2 * (1.f + g) - phi3 * q;
    const float Bq = -s2 * phi2 * g + phi3 * q;
    cm[0] = scale * tf * (-phi1 + A / r0);
    cm[1] = scale * tf * (-A / r0 + Bq / r1);
    cm[2] = scale * tf * (-Bq / r1);
}

__device__ __forceinline__ void ab_weights(int order, float (&w)[4]) {
    w[0] = w[1] = w[2] = w[3] = 0.f;
    if (order == 1) { w[0] = 1.f; }
    else if (order == 2) { w[0] = 1.5f; w[1] = -0.5f; }
    else if (order == 3) { w[0] = 23.f / 12.f; w[1] = -16.f / 12.f; w[2] = 5.f / 12.f; }
    else { w[0] = 55.f / 24.f; w[1] = -59.f / 24.f; w[2] = 37.f / 24.f; w[3] = -9.f / 24.f; }
}

// One thread per sample: writes coefficient row(s) for ds_solver_update (slots: 0 cx, 1 cm, 2..4 ch, 5 t, 6 sigma).
__global__ void amed_coefs_kernel(const ds_amed_coef_args a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.n) return;
    const float* pr = a.pred + (size_t)b * 4;
    const float r = pr[0], sd = pr[1], st = pr[2], tm = pr[3];
    const float t = a.t_cur, tn = a.t_next;
    float* c = a.coefs + (size_t)b * 8;
    float cx = 1.f, cm = 0.f, ch[3] = {0.f, 0.f, 0.f}, tdiv = t, sig = t;
    if (a.stage == 1) {
        // first evaluation was at the scalar (t_cur, sigma = t_cur); the step goes to the per-sample t_mid
        if (a.mode == DS_AMED_IPNDM) {
            float w[4]; ab_weights(a.order, w);
            cm = (tm - t) * w[0]; ch[0] = (tm - t) * w[1]; ch[1] = (tm - t) * w[2]; ch[2] = (tm - t) * w[3];
        } else if (a.mode == DS_AMED_DPMPP) {
            float* th = a.thist + (size_t)b * 4;          // [n, t_oldest.. t_newest] -> th[0] = count
            int nh = (int)th[0];
            // append t_cur (solvers_amed.py:581)
            if (nh == 3) { th[1] = th[2]; th[2] = th[3]; th[3] = t; } else { th[1 + nh] = t; nh += 1; th[0] = (float)nh; }
            float cmv[3];
            dpmpp_coeffs_dev(th + 1, nh, tm, a.order, a.predict_x0, 1.0f, cx, cmv);
            cm = cmv[0]; ch[0] = cmv[1]; ch[1] = cmv[2];
            tdiv = t;
        } else {
            cm = tm - t;                                   // Euler to t_mid (amed / euler / dpm_2)
        }
        if (a.sigma2) a.sigma2[b] = st * tm;               // sigma of the second evaluation (solvers_amed.py:143)
    } else {
        tdiv = tm; sig = st * tm;
        if (a.mode == DS_AMED_AMED) {
            cm = sd * (tn - t);                            // x' = x + scale_dir (t'-t) d_mid          (:145)
        } else if (a.mode == DS_AMED_EULER) {
            cm = sd * (tn - tm);                           // x' = x~ + scale_dir (t'-t_mid) d_mid      (:243)
        } else if (a.mode == DS_AMED_DPM2) {
            cm = sd * (tn - t) * (1.0f / (2.0f * r));      // (:451)
            ch[0] = sd * (tn - t) * (1.0f - 1.0f / (2.0f * r));
        } else if (a.mode == DS_AMED_IPNDM) {
            float w[4]; ab_weights(a.order, w);
            const float s = sd * (tn - tm);
            cm = s * w[0]; ch[0] = s * w[1]; ch[1] = s * w[2]; ch[2] = s * w[3];
        } else {                                           // DPMPP second update, history gets t_mid (:597)
            float* th = a.thist + (size_t)b * 4;
            int nh = (int)th[0];
            if (nh == 3) { th[1] = th[2]; th[2] = th[3]; th[3] = tm; } else { th[1 + nh] = tm; nh += 1; th[0] = (float)nh; }
            float cmv[3];
            dpmpp_coeffs_dev(th + 1, nh, tn, a.order, a.predict_x0, sd, cx, cmv);
            cm = cmv[0]; ch[0] = cmv[1]; ch[1] = cmv[2];
        }
    }
    c[0] = cx; c[1] = cm; c[2] = ch[0]; c[3] = ch[1]; c[4] = ch[2]; c[5] = tdiv; c[6] = sig; c[7] = 0.f;
}

}  // namespace

extern "C" int ds_amed_predict(const ds_amed_predictor* p, const float* bottleneck_mean, int n, float t_cur, float t_next, float* out,
                               void* stream) {
    (void)hipGetLastError();
    if (!p || !bottleneck_mean || !out || n <= 0) return DS_E_ARG;
    if (p->hidden > 256 || p->out_dim + 2 * p->nc > 64 || p->nc < 4) return DS_E_SHAPE;
    hipLaunchKernelGGL(amed_predict_kernel, dim3(n), dim3(128), 0, (hipStream_t)stream, *p, bottleneck_mean, t_cur, t_next, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

extern "C" int ds_amed_coefs(const ds_amed_coef_args* a, void* stream) {
    (void)hipGetLastError();
    if (!a || !a->pred || !a->coefs || a->n <= 0) return DS_E_ARG;
    if (a->mode == DS_AMED_DPMPP && !a->thist) return DS_E_ARG;
    hipLaunchKernelGGL(amed_coefs_kernel, dim3((a->n + 63) / 64), dim3(64), 0, (hipStream_t)stream, *a);
    DS_CHECK_LAUNCH();
    return DS_OK;
}
