// 3x3 convolution with AT MOST FOUR output channels -- the network heads (gfx950; round 4).
//   EDM SongUNet / DhariwalUNet: out_conv / `out` 3x3, C -> 3 (networks_edm.py:366-368, 515-516), fused with the GroupNorm + SiLU in front of it;
//   SD-1.5: `out` = GroupNorm, SiLU, conv 320 -> 4 (ldm/modules/diffusionmodules/openaimodel.py:677-681).
// The matrix kernels pad the output to a 64- or 128-column tile: 3 useful columns of 64 on the LDS-halo kernel (715 us per evaluation of the
// CIFAR-10 net at 256 images, 0.9 % of the sampler), 4 of 128 on the generic kernel in the fp16 engines (0.7 - 0.9 ms per evaluation: 3 - 4 % of
// SD-1.5 fp16 and ImageNet-64 fp16).  With <= 4 outputs the layer is 2 x 9 x C x 4 FLOP per pixel on 4 C bytes: VALU / memory work, not a
// matrix product.
// Layout of the work: a workgroup (256 threads) owns 256 consecutive pixels; a wave takes them eight at a time: lane = (pixel g = lane >> 3,
// channel quad j = lane & 7), so a load instruction reads 8 pixels x 128 contiguous bytes (32 channels).  For every 32-channel step s: the
// lane's {mu, A, B} quads of the fused input normalisation (if any), then the nine taps: one 16-byte load (clamped to the lane's own pixel
// outside the image and zeroed after the normalisation -- no predicated load: those compile to a branch and a full wait each), normalise
// + SiLU, four 16-byte weight vectors from LDS ([tap][step][channel of the quad][j][4 outputs]: 128 contiguous bytes per instruction and
// pixel group, the eight groups read the same addresses), 16 FMAs.  The eight channel-quad lanes of a pixel are summed at the end (two
// quad_perm adds + one row_half_mirror add) and lane j = 0 stores: channel-planar (NCHW, ds_conv_args.out_nchw) or rows.
// Round 5, UPD instantiations (ds_conv_args.update): the SOLVER UPDATE runs in this epilogue -- north_star's "fused back-to-back with the
// solver's scaled-AXPY / multistep linear-combination update" taken literally.  The lane that holds a pixel's <= 4 outputs F applies
// ds_upd_element (csrc/ds_common.h: the arithmetic of ds_solver_update, one definition for both forms => equal bits) per channel: EDM
// preconditioning D = c_skip x + c_out F, d = (x - D) / t, x' = cx xb + cm m + sum ch[k] hist[k], and stores x' and the history entry m next
// to F.  No update launch, no re-read of F; the per-sample coefficient rows of the AMED solvers work unchanged.
// Arithmetic: exact fp32 products, fp32 accumulation (order: channel-quad lane, then 32-channel steps, then taps) -- the same class as the fp32
// MFMA kernels it replaces; tests compare it with them (ds_conv_tune.mode = 8 switches it off).
#include "igemm_common.h"

namespace igemm {
namespace {

template <int CTRL>
__device__ __forceinline__ float thin_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// NORM: 0 = plain input, 1 = fused {mu, A, B} normalisation, 2 = normalisation + SiLU (compile time: a runtime branch per tap inside the
// loop split it into nine scheduling regions and the allocator spilled 144 registers)
template <int NORM, bool UPD>
__global__ void __launch_bounds__(256, 4) conv3x3_thin_kernel(const KParams p, const ds_update_args u) {
    extern __shared__ __attribute__((aligned(16))) float smem_thin[];
    f32x4* wl = reinterpret_cast<f32x4*>(smem_thin);              // [9][S][4][8] vectors of 4 outputs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = p.c0, S = C / 32;
    // ---- weights -> LDS.  Source rows are the packed [cout_pad][K] matrix, K = (step * 9 + tap) * 32 + cc (ops.pack_conv_weight); rows
    // >= cout are zero (row padding), so four rows can always be read.
    for (int i = tid; i < 9 * S * 32; i += 256) {
        const int j = i & 7, q = (i >> 3) & 3, rest = i >> 5;
        const int tap = rest / S, step = rest - tap * S;
        const float* b = p.b + (size_t)(step * 9 + tap) * 32 + j * 4 + q;
        wl[i] = f32x4{b[0], b[(size_t)p.ldb], b[2 * (size_t)p.ldb], b[3 * (size_t)p.ldb]};
    }
    __syncthreads();
    // a wave takes R rounds of eight pixels, R = 8 unless the layer is small (round 6: launch_conv3x3_thin, p.mtiles): its pixels lie in one image
    // (HW % 64 == 0).  The arithmetic of a pixel does not depend on R.
    const int R = p.mtiles;
    const int mb = blockIdx.x * (32 * R) + wave * (8 * R);
    if (mb >= p.M) return;
    const int img = mb / p.HW;
    const int g = lane >> 3, j = lane & 7;
    const float* coef = NORM ? p.norm + (size_t)img * 3 * C + j * 4 : nullptr;
    const float scale = p.scale;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (p.colbias) {
#pragma unroll
        for (int n = 0; n < 4; ++n) bias[n] = n < p.N ? p.colbias[n] : 0.f;
    }
#pragma unroll 1
    for (int round = 0; round < R; ++round) {
        const int m = mb + round * 8 + g;
        const int pim = m - img * p.HW;
        const int y = pim / p.W, x = pim - y * p.W;
        const float* src[9]; float keep[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
            src[t] = p.a0 + (size_t)(ok ? (img * p.H + yy) * p.W + xx : m) * p.lda0 + j * 4;
            keep[t] = ok ? 1.f : 0.f;
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int s = 0; s < S; ++s) {
            f32x4 mu = {0.f, 0.f, 0.f, 0.f}, ga = mu, be = mu;
            if constexpr (NORM) {
                mu = *reinterpret_cast<const f32x4*>(coef + s * 32);
                ga = *reinterpret_cast<const f32x4*>(coef + C + s * 32);
                be = *reinterpret_cast<const f32x4*>(coef + 2 * C + s * 32);
            }
            // taps in groups of TG loads in flight: 9 without the normalisation; 3 with it (nine interleaved SiLU quads do not fit 128 VGPRs)
            constexpr int TG = NORM ? 3 : 9;
#pragma unroll
            for (int t0 = 0; t0 < 9; t0 += TG) {
                f32x4 v[TG];
#pragma unroll
                for (int t = 0; t < TG; ++t) v[t] = *reinterpret_cast<const f32x4*>(src[t0 + t] + s * 32);
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    f32x4 a = v[t];
                    if constexpr (NORM) {
                        a = (a - mu) * ga + be;
                        if constexpr (NORM == 2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[e] = ds_silu(a[e]);
                        }
                    }
                    a *= keep[t0 + t];
                    const f32x4* wp = wl + (((t0 + t) * S + s) * 4) * 8 + j;
                    acc += a[0] * wp[0] + a[1] * wp[8] + a[2] * wp[16] + a[3] * wp[24];
                }
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float r = acc[n];
            r = thin_dpp_add<0xB1>(r);           // quad_perm [1, 0, 3, 2]
            r = thin_dpp_add<0x4E>(r);           // quad_perm [2, 3, 0, 1]
            r = thin_dpp_add<0x141>(r);          // row_half_mirror: the other quad of the 8-lane group
            acc[n] = r;
        }
        if (j == 0) {
            DsUpdCoefs k{};
            float cskip = 0.f, cout_ = 0.f;
            if constexpr (UPD) {
                k = ds_upd_load_coefs(u, img);
                cskip = ds_c_skip(k.sig, u.sigma_data); cout_ = ds_c_out(k.sig, u.sigma_data);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (n < p.N) {
                    const float o = (acc[n] + bias[n]) * scale;
                    if (p.out_planar) p.out[((size_t)img * p.N + n) * p.HW + pim] = o;
                    else p.out[(size_t)m * p.ldo + n] = o;
                    if constexpr (UPD) {                       // the solver update on this element: NCHW operands at the element's own offset
                        const size_t off = ((size_t)img * p.N + n) * p.HW + pim;
                        const float x = u.xe[off];
                        const float xb = u.xb != u.xe ? u.xb[off] : x;
                        float mm, xo;
                        ds_upd_element(k, cskip, cout_, true, u.store_d != 0, x, xb, o, u.hist[0] != nullptr, u.hist[0] ? u.hist[0][off] : 0.f,
                                       u.hist[1] != nullptr, u.hist[1] ? u.hist[1][off] : 0.f, u.hist[2] != nullptr, u.hist[2] ? u.hist[2][off] : 0.f, mm, xo);
                        if (u.m_out) u.m_out[off] = mm;
                        if (u.x_out) u.x_out[off] = xo;
                    }
                }
            }
        }
    }
}

}  // namespace

bool conv3x3_thin_applicable(const KParams& p) {
    if (p.taps != 9 || p.stride > 1 || p.N < 1 || p.N > 4) return false;
    if (p.c0 <= 0 || p.c0 % 32 || p.c1 || p.ec0 || p.ec1) return false;
    if (p.HW != p.H * p.W || p.HW % 64 || p.M % 64) return false;
    if (p.res || p.cbias || p.rowbias || p.stats || p.act != DS_ACT_NONE || p.splits > 1) return false;
    if (p.norm && p.norm_act != DS_ACT_NONE && p.norm_act != DS_ACT_SILU) return false;
    if (p.nrows_b < 4) return false;
    if (9 * p.c0 * 16 > 150 * 1024) return false;                 // the weight vectors must fit in LDS (C <= 1 056)
    return true;
}

// the fused update is on when the call carries a struct with at least one output (ds_conv_args.update)
static bool thin_update_on(const KParams& p) { return p.upd && (p.upd->x_out || p.upd->m_out); }

int launch_conv3x3_thin(KParams& p, hipStream_t stream) {
    const int smem = 9 * p.c0 * 16;
    // pixels per workgroup: 256 (four waves x eight rounds of eight), fewer rounds while the launch would leave CUs without a workgroup -- at 8
    // images a 32x32 head was 32 workgroups of 140 us each (round 6, tools/time_plan_ops.py: 146 us of a 6.2 ms evaluation; 280 us at 256 images)
    int rounds = 8;
    while (rounds > 1 && (p.M + 32 * rounds - 1) / (32 * rounds) < 512) rounds >>= 1;
    p.mtiles = rounds;
    const unsigned blocks = (unsigned)((p.M + 32 * rounds - 1) / (32 * rounds));
    const bool upd = thin_update_on(p);
    ds_update_args u{};
    if (upd) {
        u = *p.upd;                                                // read at call time (include/ds_engine.h)
        if (!p.out_planar || !u.xe || !u.xb || !u.raw || u.afs || u.f_ld != 0) return DS_E_ARG;
        if (u.c != p.N || (long long)u.n * u.h * u.w != (long long)p.M || u.h * u.w != p.HW) return DS_E_ARG;
        if (u.coefs && u.coef_rows != 1 && u.coef_rows != u.n) return DS_E_ARG;
    }
#define DST_LAUNCH(NORM_, UPD_)                                                                                    \
    do {                                                                                                           \
        DS_ENSURE_DYN_LDS((&conv3x3_thin_kernel<NORM_, UPD_>), 160 * 1024);                                        \
        hipLaunchKernelGGL((conv3x3_thin_kernel<NORM_, UPD_>), dim3(blocks), dim3(256), smem, stream, p, u);       \
    } while (0)
    if (p.norm && p.norm_act == DS_ACT_SILU) { if (upd) DST_LAUNCH(2, true); else DST_LAUNCH(2, false); }
    else if (p.norm) { if (upd) DST_LAUNCH(1, true); else DST_LAUNCH(1, false); }
    else { if (upd) DST_LAUNCH(0, true); else DST_LAUNCH(0, false); }
#undef DST_LAUNCH
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace igemm
