/*
 * ds_engine.h -- C ABI of libdsamd.so, the MI355X (gfx950) engine behind the diff-sampler hot path.
 *
 * The reference (zju-pi/diff-sampler) is pure Python/PyTorch and has no FFI layer: the
 * boundary it exposes is the Python surface of diff-solvers-main/{solvers.py, solver_utils.py, sample.py}.  This
 * header is the native boundary UNDER that surface: every entry point replaces one group of ATen op sequences the
 * reference issues (cited per function as file:line under /root/reference/diff-solvers-main unless noted), takes
 * raw device pointers + sizes + a hipStream_t (passed as void*), returns 0 on success or a non-zero code
 * (positive = hipError_t, negative = argument error, see DS_E_*), never allocates, never synchronises.
 * The Python mirrors of the reference functions (diff_sampler_amd/solvers.py etc.) bind it with ctypes;
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Tensor layouts
 *   user tensors (latents x, denoised D, history)  : NCHW fp32, exactly as the reference passes them
 *   denoiser activations (internal)                : NHWC fp32, "rows" = pixels (n*H*W + h*W + w), ld = floats/row
 *   conv / linear weights                          : packed [Cout_pad128][K] fp32; 3x3: K index = (slab*9 + tap)*32 + c
 *                                                    with slab = 32-channel block of the (concatenated) input; 1x1: K = c
 *   network output F (EDM nets)                    : channel-planar NCHW (ds_conv_args.out_nchw), read in place by
 *                                                    ds_solver_update
 */
#ifndef DS_ENGINE_H
#define DS_ENGINE_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: DS_API marks the entry points declared here as the ONLY exported symbols (the kernels'
 * C++ launch helpers stay internal to libdsamd.so; tests/test_abi_cpu.py compares the dynamic symbol table with this header). */
#define DS_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

#define DS_OK 0
#define DS_E_ARG (-1)      /* invalid argument combination            */
#define DS_E_ALIGN (-2)    /* pointer / leading dimension not aligned */
#define DS_E_SHAPE (-3)    /* unsupported shape (e.g. K % 32 != 0)    */

#define DS_ACT_NONE 0
#define DS_ACT_SILU 1
#define DS_ACT_GEGLU 2  /* ds_conv2d_nhwc, taps == 1 only: the GEGLU gate of ldm/modules/attention.py:45-52 fused into the
                          projection's epilogue.  The weight rows must be packed so that every 64-row block holds 32 value
                          rows followed by their 32 gate rows; out gets cout / 2 columns: value * gelu(gate). */

#define DS_RESAMPLE_NONE 0
#define DS_RESAMPLE_DOWN 1  /* 2x2 box filter, stride 2  (networks_edm.py:77 with resample_filter [1,1]) */
#define DS_RESAMPLE_UP 2    /* nearest neighbour x2      (networks_edm.py:75 with resample_filter [1,1]) */

DS_API int ds_version(void);      /* ABI version; a host must check it before passing argument structs.  4 (round 6): ds_norm_args.stats0 / stats1 / tune_variant and ds_attn_args.variant
                               * appended (the pass that computes its own GroupNorm statistics; struct size changed).  3 (round 5): ds_conv_args.update appended (the head-fused
                               * solver update; struct size changed), ds_build_experiments() added, ds_conv_args.norm_coefs also accepted with in_f16.
                               * 2 (round 4): ds_conv_args.tune / ds_update_args.variant appended (struct sizes changed), ds_fid_moments added, the
                               * process-global ds_debug_* setters removed. */
DS_API const char* ds_error_string(int code);
DS_API int ds_build_experiments(void);   /* build flags.  Bit 0: built with DS_BUILD_EXPERIMENTS=1 -- the library also holds the kernel variants kept as A/B records
                                          * (conv3x3_f16dmah, conv3x3_halo2 modes 0 / 1: reachable through ds_conv_args.tune only, never chosen by default);
                                          * clear: the product kernels only -- ds_conv_f16_supported() answers 0 and tune.f16dma_nw = 4 / tune.variant = 3 are
                                          * ignored for 3x3 layers.  Bit 1 (round 6): a DS_RACE_STRESS build -- tests only: the same kernels with ~2 us delays in the
                                          * waves that produce shared LDS contents (csrc/ds_common.h); never shipped as libdsamd.so. */

/* ---------------------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32; exact fp32).
 * Replaces F.conv2d 3x3 pad 1 / 1x1 (networks_edm.py:79), Linear (networks_edm.py:32-34), torch.cat of the decoder
 * skip (networks_edm.py:353: two sources are read in place), the bias add (:81), the embedding add (:167), the
 * residual add and skip_scale (:170-171, :177-178).
 *
 *   out[m, co] = act( ( sum_{tap,c} X(m, tap, c) * W[co, k(tap, c)] + bias[co] + cbias[img(m), co] + res[m, co] )
 *                     * out_scale )
 *   X(m, tap, c): zero-padded 3x3 neighbourhood of pixel m in the channel-concatenation [x0 | x1].
 * Constraints: c0 % 32 == 0, c1 % 32 == 0, all leading dimensions % 4 == 0, pointers 16-byte aligned,
 * W has ceil(cout/128)*128 rows (zero padded).
 */
struct ds_update_args;            /* defined below ("Solver side"); ds_conv_args.update points to one */

typedef struct ds_conv_tune {
    /* 1 = generic gather kernel instead of the LDS-halo / LDS-DMA kernels (both exact fp32: cross-check), 128 / 256 = forced M tile of
     * the halo kernel, 2 = halo kernel with register-staged weights, 4 = no 64-column tail tiles, 6 = no 8-wave DMA kernel for 1x1 /
     * Linear layers, 8 = no thin-output kernel (conv3x3_thin.hip, kernel id 2570) for 3x3 layers with cout <= 4: the matrix kernels as for
     * any other layer.  (Any non-zero mode or variant keeps the matrix kernels.) */
    int mode;
    /* Kernel variant of the LDS-halo 3x3 convolution.  Low five bits: 0 = default, 1 = software-pipelined tap loop of the 128-column
     * tiles, 3 = second-generation kernel (conv3x3_halo2.hip) where it applies, 6 / 7 = 256 x 256 tiles forced (tests at small sizes)
     * / switched off, other values = timing ablations compiled only with -DDS_CONV_ABLATIONS (wrong results on purpose; + 0x10000:
     * ablations of the 256 x 256 tile).  Bit 8 (256): the 256 x 256 tile's plain kernel instead of its default (scalar-addressed weight
     * DMA + non-temporal epilogue); bit 9 (512): multi-image tiles read their GroupNorm coefficient planes from global memory instead of
     * LDS; bit 11 (2048): the four-wave 128 x 128 tile also where the default is eight half-size waves; bit 12 (4096): the 128-column
     * tiles without the scalar-addressed weight DMA / non-temporal epilogue; bit 13 (8192): no 256 x 192 tiles for the 192-multiples
     * (ADM channel counts); bit 14 (16384): force them regardless of the tile count (tests). */
    int variant;
    int splits;        /* > 0: split-K factor of a convolution that has a workspace (clamped to what the layer allows; 1 = never split).  The
                        * fp16-activation 3x3 kernel (in_f16) splits on its own where its widest tiling covers at most half of the CUs */
    int f16dma_nb;     /* fp16-activation kernels: column-tile width 64 * nb, nb = 1..4 */
    int f16dma_nw;     /* fp16-activation GEMM: 4 / 8 = 128- / 256-row variant; fp16-activation 3x3: 4 = the four-wave half-slab kernel on
                        * 128-pixel tiles (two workgroups per CU, kernel id 2569), 8 = the eight-wave kernel on 256-pixel tiles (2566) */
    /* fp16-activation kernels, benchmarks only (results are WRONG when bits 0 - 5 are set): bit 0: no weight DMA after the prologue,
     * bit 1: no halo DMA after the first slab, bit 2: no epilogue, bit 4: no per-tap barrier, bit 5: no LDS fragment reads; bit 10
     * (results stay correct): fp16 residual rows requested one group ahead instead of early (profiles/r3_gemm_f16dma_epilogue.txt). */
    int ablate;
} ds_conv_tune;

typedef struct ds_conv_args {
    const float* x0; const float* x1;      /* sources; x1 may be NULL when c1 == 0                               */
    int c0, c1;                            /* channels taken from each source                                   */
    int ld0, ld1;                          /* floats per pixel row in each source                               */
    int n, h, w;                           /* images, height, width (same for input and output)                 */
    int taps;                              /* 9 (3x3, pad 1) or 1 (1x1 / linear)                                */
    const float* wgt;                      /* packed weights [cout_pad][taps*(c0+c1)]                           */
    int cout;
    const float* bias;                     /* [cout] or NULL                                                    */
    const float* cbias; int cbias_ld;      /* per-image channel bias [n or 1][cbias_ld] or NULL                 */
    int cbias_rows;                        /* 1 = broadcast over images, else n                                 */
    const float* res; int res_ld;          /* residual [M][res_ld] or NULL                                      */
    float out_scale;
    int act;                               /* DS_ACT_*                                                          */
    float* out; int out_ld;
    /* Fused input normalisation (taps == 9 only): the kernel reads in = act_in((x - mu) * A + B) instead of x, with
     * per-(image, channel) coefficient planes [n][3][c0+c1] = {mu, A, B} written by ds_gn_stats (coefs output).
     * Zero padding stays zero (it pads the NORMALISED tensor, networks_edm.py:160,167).  NULL = raw input.
     * Requires ds_conv3x3_halo_supported(h, w); otherwise DS_E_SHAPE. */
    const float* norm_coefs; int norm_act;
    /* Extra 1x1 sources appended along K (the skip projection fused into conv1, networks_edm.py:170): after the
     * 9*(c0+c1) columns the weight rows carry ec0+ec1 more columns that multiply [e0 | e1] at the output pixel.
     * ec0 == 0 = none.  Same constraints as c0/c1 (multiples of 32, ld % 4 == 0). */
    const float* e0; const float* e1; int ec0, ec1; int eld0, eld1;
    /* Output stride of a 3x3 convolution: 0 or 1 = dense; 2 = the LDM `Downsample` convolution (stride 2, pad 1,
     * ldm/modules/diffusionmodules/openaimodel.py:146-148): h, w are then the OUTPUT size and the input is 2h x 2w.
     * stride 2 excludes norm_coefs and the e0/e1 extras.  With in_f16 (fp16 rows in, `wgt` in the [slab64][tap][64] fp16 packing of the
     * stride-1 kernel) it also excludes `res`; availability: ds_conv_f16dma_stride2_supported(). */
    int stride;
    /* Optional split-K scratch (floats): layers whose output has too few tiles to fill the 256 CUs (small batch, 8x8 /
     * 16x16 stages) split the K loop over up to 64 workgroups per tile, each writing a raw partial tile here; a second
     * launch sums them in a fixed order (deterministic) and applies the epilogue.  NULL / 0 = never split.  The launcher
     * uses at most min(workspace_floats, 64 * M * cout) floats; contents are scratch. */
    float* workspace; long long workspace_floats;
    /* 1: write the output channel-planar, out[(img * cout + co) * h * w + pixel] (NCHW), instead of NHWC rows; used for the
     * 3-channel network output F so that the solver update reads 12 B per pixel, not a row padded to 16 B.  Only for
     * cout < 64 (the scalar epilogue); out_ld is ignored. */
    int out_nchw;
    /* Optional: column statistics of the OUTPUT for the consumer's GroupNorm, produced by the epilogue while the tile is
     * still in registers (replaces a ds_gn_stats pass over the tensor): stats_out[(rb * 2 + k) * cout + co] = sum (k = 0) /
     * sum of squares (k = 1) of out[rb * 64 .. rb * 64 + 63][co]; ceil(n*h*w / 64) * 2 * cout floats, consumed by
     * ds_gn_finalize.  Requires cout % 64 == 0 and the aligned (float4) epilogue; NULL = not needed. */
    float* stats_out;
    /* 1: REDUCED-PRECISION OPERANDS, the reference's `use_fp16` / autocast mode (networks_edm.py:486, sample.py:296): `wgt` holds fp16
     * weights packed for 64-channel slabs ([cout_pad][K] halfs, K = (slab64 * 9 + tap) * 64 + c, then the 1x1 extra columns in
     * 64-channel blocks), the kernel rounds the (normalised, activated) input to fp16 while it stages it and multiplies on
     * v_mfma_f32_32x32x16_f16 with fp32 accumulation; inputs, bias / residual / output tensors stay fp32.  taps == 9 only, and only
     * where ds_conv_f16_supported() says so -- otherwise DS_E_SHAPE (there is no silent fp32 fallback). */
    int wgt_f16;
    /* wgt_f16 == 2: SPLIT-fp16 OPERANDS, fp32 emulated on the fp16 matrix pipe.  Every operand x is represented as hi + lo with
     * hi = fp16(x), lo = fp16(x - hi); a product is hi*hi + hi*lo + lo*hi with fp32 accumulation (the dropped lo*lo term and the
     * rounding of lo are 2**-22 relative: fp32 class, far inside the engine's fp32 tolerances -- which is the condition for using
     * it).  `wgt` then holds [cout_pad][K] PAIRS: per 32-channel slab and tap 32 hi halfs followed by 32 lo halfs
     * (K order (slab32 * 9 + tap) * 64, then the 1x1 extra columns in 32-channel blocks), of the weights MULTIPLIED by 2**wgt_shift
     * (keeps hi and lo in fp16's normal range); the epilogue multiplies the accumulators by 2**-wgt_shift.  Available where
     * ds_conv_split_supported() says so, else DS_E_SHAPE. */
    int wgt_shift;
    /* 1 (with wgt_f16 == 1; taps == 9, or taps == 1 = csrc/gemm_f16dma.hip with the plain-K fp16 weights of the Linear layers): THE INPUT IS fp16 -- x0 (and e0) point to fp16 NHWC tensors [M][ld0] (ld0 / eld0 in halfs,
     * multiples of 8; c0 % 64 == 0, c1 == ec1 == 0), already normalised / activated by ds_norm_act(out_f16) -- the reference's storage
     * type in this mode (networks_edm.py:486 runs the U-Net body on x.to(float16)).  The convolution is then a pure matrix kernel
     * (csrc/conv3x3_f16dma.hip: both operands staged by LDS-DMA, 256-pixel x 64/128/192/256-channel tiles).
     * Round 5, taps == 9 and stride 1 only: with norm_coefs != NULL the sources are the RAW fp16 tensors and the kernel applies
     * in = act((x - mu) * A + B) to the halo in LDS (norm_act NONE / SILU; same arithmetic and rounding as ds_norm_act(out_f16), so
     * the results equal the two-launch form bit for bit); x1 / e1 (c1, ec1 multiples of 64, ld1 / eld1 in halfs) are then allowed --
     * the channel concatenation [x0 | x1] is not materialised; the planes are [n][3][c0 + c1].  Without norm_coefs, c1 == ec1 == 0.
     * Bias is fp32; residual / output are fp32 unless res_f16 / out_f16.  Availability: ds_conv_f16dma_supported(); otherwise DS_E_SHAPE. */
    int in_f16;
    /* 1: the OUTPUT is written as fp16 NHWC rows [M][out_ld halfs] (rounded to nearest even from the fp32 epilogue value; the GroupNorm
     * column sums of stats_out are those of the ROUNDED values, i.e. of the stored tensor).  The reference's fp16 mode keeps every
     * activation of the U-Net body in fp16 (networks_edm.py:486, :165-179; torch.autocast in the latent-diffusion path, sample.py:296):
     * conv0 outputs, block outputs (the residual stream) and projection outputs.  Requires in_f16, cout % 64 == 0, out_ld % 8 == 0, no out_nchw. */
    int out_f16;
    /* 1 (with in_f16): `res` is an fp16 tensor [M][res_ld halfs] (res_ld % 8 == 0, 16-byte aligned) -- the fp16 residual stream; added in fp32. */
    int res_f16;
    /* Per-call kernel selection overrides for benchmarks, A/B measurements and tests; all zero = the library's own choice (what the
     * engines pass).  They travel with the call (and with a ds_plan entry): the library keeps NO process-wide selection state, so two
     * threads / streams can run layers under different overrides at the same time (tests/test_hip_kernels.py: two-thread test). */
    ds_conv_tune tune;
    /* Round 5 (ABI version 3): the solver update FUSED into the network head.  Non-NULL with a non-NULL x_out or m_out inside: the layer must
     * be a network head that runs on conv3x3_thin_kernel (taps == 9, cout <= 4, out_nchw; ds_conv_kernel_id() == 2570) -- else DS_E_ARG --
     * and its epilogue then applies ds_solver_update's arithmetic to every output element while it still holds it in a register:
     * F = this layer's output (raw network output: update->raw must be 1, f_ld 0, afs 0; update->f is ignored), xe / xb / hist / coefs /
     * hcoefs / sigma_data / store_d / m_out / x_out / n, c, h, w as in ds_update_args (n, c, h, w must equal this layer's n, cout, h, w).
     * `out` still receives F.  One definition of the arithmetic serves both forms (csrc/ds_common.h: ds_upd_element), so the results equal
     * those of ds_conv2d_nhwc followed by ds_solver_update bit for bit.  The struct is READ AT CALL TIME: a ds_plan entry keeps the
     * pointer, so a host re-fills the same struct before each run (x_out == m_out == NULL = no fusion for this run).
     * Replaces the separate update launch of the linear solvers (solvers.py:76-81, :156-168, :245-258, :344-352, :574-585). */
    const struct ds_update_args* update;
} ds_conv_args;

DS_API int ds_conv2d_nhwc(const ds_conv_args* a, void* stream);

/* Which kernel ds_conv2d_nhwc dispatches this call to: 0 = generic gather kernel (igemm_f32_kernel<0>), 128 / 256 =
 * LDS-halo kernel with that M tile (conv3x3_halo_kernel<2> / <4>), 2561 = 8-wave LDS-DMA 1x1 / Linear kernel
 * (gemm_dma8_kernel), 2562 = fp16-operand halo kernel (conv3x3_halo2_kernel<W, 1>), 2560 = the same kernel on fp32 operands (tune.variant 3), 2563 = split-fp16 (fp32-emulated) halo kernel
 * (conv3x3_halo2_kernel<W, 2>), 2564 = fp16-operand 1x1 / Linear kernel (gemm_f16_kernel), 2565 = LDS-halo kernel <4> with
 * 256-pixel x 256-channel tiles (64 x 128 per wave; channel counts that are multiples of 256 on 16-, 32- and 64-column images), 1284 =
 * LDS-halo kernel with 128-pixel tiles on eight waves of 64 x 32 (layers with at most one tile per CU), 2570 = the thin-output 3x3 kernel
 * of the network heads (conv3x3_thin_kernel: cout <= 4, one fp32 source, no residual / per-image bias / statistics), 2571 = the stride-2
 * 3x3 convolution on fp16 rows (gemm_f16dma_kernel<.., GATHER>), 2573 (round 6) = the 1x1 / Linear on at most four rows (gemv_rows_kernel: the
 * embedding path's one-row projections; one fp32 source, bias / scale / SiLU only, cout >= 64; tune.mode != 0 keeps the matrix kernels).  Used by
 * bench.py to attribute time per kernel. */
DS_API int ds_conv_kernel_id(const ds_conv_args* a);

/* 1 if a 3x3 convolution on h x w images runs on the LDS-halo kernel (needed for norm_coefs), else 0. */
DS_API int ds_conv3x3_halo_supported(int h, int w);

/* fp16-operand 3x3 convolution (ds_conv_args.wgt_f16): returns 0 = not available for this geometry, 1 = available, 2 = available
 * and the fused input normalisation (norm_coefs) too.  n, h, w: images and size; cin = c0 + c1 and ecin = ec0 + ec1 with every
 * source a multiple of 64 channels. */
DS_API int ds_conv_f16_supported(int n, int h, int w, int c0, int c1, int ec0, int ec1);
/* 1 when a 3x3 layer of this geometry runs on the fp16-activation kernel (ds_conv_args.in_f16). */
DS_API int ds_conv_f16dma_supported(int n, int h, int w, int c0, int ec0, int cout);
/* 1 when a 1x1 / Linear layer [rows][k] -> [rows][cout] runs on the fp16-activation GEMM (in_f16 with taps == 1, csrc/gemm_f16dma.hip). */
DS_API int ds_gemm_f16dma_supported(long long rows, int k, int cout);
/* 1 when the latent-diffusion `Downsample` (3x3, stride 2, pad 1; ldm/modules/diffusionmodules/openaimodel.py:146-148) with n images of
 * OUTPUT size h x w (input 2h x 2w), c0 input and cout output channels runs on the fp16-activation GEMM in its gather form
 * (ds_conv_args.in_f16 = 1, wgt_f16 = 1, taps = 9, stride = 2; csrc/gemm_f16dma.hip).  Replaces, in fp16 mode, an fp32 copy of the
 * fp16 residual stream + the generic fp32 kernel. */
DS_API int ds_conv_f16dma_stride2_supported(int n, int h, int w, int c0, int cout);

/* 1x1 convolution / Linear with fp16 operands (wgt_f16 == 1 and taps == 1: `wgt` = [cout_pad][K] halfs in plain K order; the fp32
 * input rows are rounded to fp16 while they are staged): 1 if rows % 256 == 0 and every source is a multiple of 64 channels. */
DS_API int ds_gemm_f16_supported(long long rows, int c0, int c1);

/* The same for the split-fp16 (fp32-emulated) operands of wgt_f16 == 2; every source a multiple of 32 channels. */
DS_API int ds_conv_split_supported(int n, int h, int w, int c0, int c1, int ec0, int ec1);

/* Batched per-seed latent generator: out[b][i], i < n, = the tensor `torch.randn([n], generator=g_b, device=<this GPU>)` of a generator
 * with `g_b.manual_seed(seeds[b])` whose Philox offset is `offset` (0 for a fresh generator) -- bit for bit, for a whole batch of
 * seeds in one launch.  Replaces B generator constructions + B launches per batch (diff-solvers-main/sample.py:22-36,
 * StackedRandomGenerator.randn).  `threads_total` = 256 * min(CUs * (maxThreadsPerCU / 256), ceil(n / 256)), ATen's execution
 * policy for n elements; afterwards each generator's offset has advanced by ((n - 1) / (threads_total * 4) + 1) * 4. */
DS_API int ds_philox_randn(const unsigned long long* seeds, unsigned long long offset, float* out, int batch, long long n,
                    long long threads_total, void* stream);

/* Diagnostic, stateless (tools/probe_rng.py): element i = first Box-Muller output of Philox (seed, subsequence i, offset) with a
 * selectable build of log / sqrt / sin (variant in [0, 96)), to identify which one the installed torch's randn was compiled with. */
DS_API int ds_philox_probe(unsigned long long seed, unsigned long long offset, float* out, int n, int variant, void* stream);

/* out[b] = `torch.randint(range, size=[], generator=g_b)` at Philox offset `offset` (sample.py:283: class labels); range < 2**32;
 * the offset then advances by 4. */
DS_API int ds_philox_randint(const unsigned long long* seeds, unsigned long long offset, unsigned int range, int* out, int batch, void* stream);


/* Batched C[z] = act(alpha * A[z] * B[z]^T + rowbias + colbias) on the same MFMA core ("NT": both operands have k
 * contiguous).  Used for attention: S = Q K^T / sqrt(C) and O = P V (networks_edm.py:108, :176) and the transposed
 * V projection.  z = zb * heads + zh;  X_z = X + zb*x_bstride + zh*x_hstride. Constraints: k % 32 == 0. */
typedef struct ds_gemm_args {
    const float* a; int lda; long long a_bstride, a_hstride;   /* A[z]: [m][lda] */
    const float* b; int ldb; long long b_bstride, b_hstride;   /* B[z]: [n][ldb] */
    float* c; int ldc; long long c_bstride, c_hstride;         /* C[z]: [m][ldc] */
    int m, n, k;
    int batch, heads;
    float alpha;
    const float* rowbias;                  /* [m] or NULL  */
    const float* colbias;                  /* [n] or NULL  */
    int act;
} ds_gemm_args;

DS_API int ds_gemm_nt_batched(const ds_gemm_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GroupNorm statistics + fused normalise / affine / SiLU / resample pass (networks_edm.py:88-98, :160, :165-167).
 * ds_gn_stats: per (image, group) mean and 1/sqrt(var + eps) over the channel-concatenation [x0 | x1].
 * ds_norm_act: y = resample( act( (x - mean) * rstd * gamma * (1 + scale) + beta * (1 + scale) + shift ) ),
 *              any of {mean/rstd, gamma/beta, scale/shift} may be NULL (identity), so the same kernel is the raw
 *              resampler of the skip path.  Output is a single NHWC tensor with ld = out_ld.
 */
typedef struct ds_norm_args {
    const float* x0; const float* x1; int c0, c1; int ld0, ld1;
    int n, h, w;                           /* input geometry                                                    */
    int groups; float eps;
    float* mean; float* rstd;              /* [n][groups]  (outputs of ds_gn_stats, inputs of ds_norm_act)      */
    const float* gamma; const float* beta; /* [c0+c1]                                                           */
    const float* scale; const float* shift; int ss_ld; int ss_rows;  /* adaptive scale/shift [ss_rows][ss_ld]   */
    int act; int resample;
    float* out; int out_ld;
    float* coefs;                          /* ds_gn_stats only, optional: [n][3][c0+c1] planes {mu, A, B} with
                                              A = rstd*gamma*(1+scale), B = beta*(1+scale)+shift, for ds_conv2d_nhwc's
                                              fused input normalisation */
    /* ds_gn_stats only, optional scratch for small batches (n < 256): the statistics of one image are then gathered by
     * up to DS_GN_MAX_CHUNKS workgroups and a second small launch adds their partial sums in a fixed order.
     * partial: [n][DS_GN_MAX_CHUNKS][128] doubles (scratch).  counters: reserved, may be NULL.  partial == NULL = one
     * workgroup per image. */
    double* partial; int* counters;
    /* ds_norm_act only.  out_f16 = 1: `out` is an fp16 NHWC tensor [rows][out_ld] (out_ld in halfs, % 4 == 0) -- the activated tensor as
     * the reference's fp16 mode stores it (networks_edm.py:486), read by ds_conv2d_nhwc(in_f16).  raw_out (optional, with out_f16):
     * a second fp16 tensor [rows][raw_ld] that receives the UN-normalised (but resampled, concatenated) input -- the operand of a
     * block's 1x1 skip projection (networks_edm.py:170) when that projection is fused into conv1 as extra K columns.
     * With out_f16, a non-NULL `coefs` is an INPUT: the {mu, A, B} planes [n][3][c0+c1] a preceding ds_gn_finalize / ds_gn_stats wrote
     * (mean / rstd / gamma / beta / scale / shift are then ignored). */
    int out_f16; void* raw_out; int raw_ld;
    /* ds_norm_act / ds_gn_stats.  in_f16 bit 0: x0 is an fp16 tensor [rows][ld0 halfs]; bit 1: x1 is (ld1 in halfs) -- tensors written
     * with ds_conv_args.out_f16 (conv0 outputs, the fp16 residual stream).  Values are widened to fp32 before any arithmetic. */
    int in_f16;
    /* ds_norm_act only, ABI 4 (round 6).  stats0 (and stats1 for a second source) non-NULL: the per-(64-row block, channel) sums {sum, sum of
     * squares} [ceil(n h w / 64)][2][c] the producing convolutions left behind (ds_conv_args.stats_out) -- the pass then computes the GroupNorm
     * statistics of every image ITSELF (what ds_gn_finalize does in a launch of its own) and applies them: `coefs` must be NULL, gamma / beta /
     * scale / shift / groups / eps are read as ds_gn_finalize reads them, mean / rstd are not written.  Only for the 16-byte fp16 form of the
     * pass (fp16 rows in and out, no resampling, channel counts multiples of 8) on images of 64 ... 1 024 pixels (h * w % 64 == 0);
     * anything else returns DS_E_SHAPE.  (The engines do NOT use this form by default: measured, it saves the ~6 us launch and pays about as
     * much in every workgroup's own reduction -- profiles/r6_norm_pass_ab.txt; plan.FOLD_FINALIZE / DS_FOLD_GN_FINALIZE=1 switches it on.)
     * tune_variant (benchmarks / tests): bit 0 keeps the 8-byte kernel of rounds 3 - 5 where the 16-byte one would be taken; bits 1 / 2 change how
     * many workgroups share an image in the self-finalising form (0: column sums <= rows / 1 per workgroup; bit 1: <= rows / 4; bit 2: no cap). */
    const float* stats0; const float* stats1;
    int tune_variant;
} ds_norm_args;

#define DS_GN_MAX_CHUNKS 32
DS_API int ds_gn_stats(const ds_norm_args* a, void* stream);

/* GroupNorm statistics from the per-(64-row block, channel) sums the producing convolutions left behind
 * (ds_conv_args.stats_out) instead of a pass over the activations: same outputs as ds_gn_stats (mean / rstd per (image,
 * group) and, optionally, the {mu, A, B} coefficient planes) for the channel concatenation [source 0 | source 1].
 * Requires h * w % 64 == 0 (an image is a whole number of 64-row blocks). */
typedef struct ds_gn_finalize_args {
    const float* stats0; const float* stats1;   /* partial sums of the two sources (stats1 NULL when c1 == 0)          */
    int c0, c1;
    int n, hw, groups; float eps;
    const float* gamma; const float* beta; const float* scale; const float* shift; int ss_ld; int ss_rows;
    float* mean; float* rstd; float* coefs;     /* as in ds_norm_args                                                   */
} ds_gn_finalize_args;
DS_API int ds_gn_finalize(const ds_gn_finalize_args* a, void* stream);
DS_API int ds_norm_act(const ds_norm_args* a, void* stream);

/* Row softmax, in place or out of place: y[r, :] = softmax(x[r, :cols]) (networks_edm.py:108). */
DS_API int ds_softmax_rows(const float* x, float* y, long long rows, int cols, int ld, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused attention  out[b, i, h*d : (h+1)*d] = sum_j softmax_j(scale * q[b,i,h,:] . k[b,j,h,:]) v[b,j,h,:]
 * (networks_edm.py:98-110 + :171-176; ldm/modules/attention.py:168-194).  q/k/v/out are token-major matrices
 * [rows][ld] whose head h occupies columns h*d ... (h+1)*d - 1 (so NHWC activations and packed q|k|v projections are
 * consumed in place); image b starts at b * *_bs floats.  skv may be any length (cross-attention: 77); sq any.
 * d must satisfy ds_attention_supported(d).  One launch, online softmax, scores never written to HBM.
 */
typedef struct ds_attn_args {
    const float* q; const float* k; const float* v; float* out;
    int ldq, ldk, ldv, ldo;
    long long q_bs, k_bs, v_bs, o_bs;
    int batch, heads, sq, skv, d;
    float scale;
    int out_f16;     /* ds_attention_f16 only: `out` is an fp16 tensor (ldo / o_bs in halfs): the operand of the output projection in fp16 mode */
    int in_f16;      /* ds_attention_f16 only: bit 0: q is an fp16 tensor (ldq / q_bs in halfs, multiples of 8), bit 1: k and v are (ldk, k_bs
                        multiples of 8; ldv, v_bs of 4) -- the fp16 tensors the reference's qkv projection emits in its fp16 mode
                        (networks_edm.py:171-173; attention.py:168-176 under autocast); `scale` then multiplies the fp32 scores */
    int variant;     /* ABI 4.  ds_attention_f16: 0 = the library's choice (the kernel with one 32-query block per wave,
                        rounds 2 - 6); 1 = the same, explicitly; 2 = two query blocks per wave, skewed by half a phase (round 6 experiment, measured
                        5 - 8 % slower and therefore never chosen; head sizes <= 64, else DS_E_SHAPE) -- for benchmarks and tests: results do not
                        depend on it.  ds_attention (fp32), head sizes that are multiples of 128: 0 = the library's choice -- the channel-split block
                        (32 queries, four waves x d / 4 channels) while the query-split grid has fewer than 1 024 waves, i.e. small batches of the
                        single 256-wide head; 1 = query split, 2 = channel split.  The two agree to fp32 rounding (the scores are summed in a
                        different order), not bit for bit; other head sizes ignore the field */
} ds_attn_args;

DS_API int ds_attention(const ds_attn_args* a, void* stream);
DS_API int ds_attention_supported(int d);

/* The same operation with fp16 operands on the fp16 matrix pipe -- the attention of the reference's fp16 / autocast mode
 * (networks_edm.py:98-110 with use_fp16; ldm/modules/attention.py:168-194 under autocast, sample.py:296): q / k / v / out stay fp32
 * in memory, are rounded to fp16 (nearest even) while staged; scores, softmax statistics and accumulators are fp32; the softmax
 * weights are rounded to fp16 before the P V product as the reference casts them.  Head sizes: ds_attention_f16_supported(d)
 * (d % 8 == 0, d <= 160); otherwise DS_E_SHAPE -- the caller picks ds_attention, there is no silent fallback. */
DS_API int ds_attention_f16(const ds_attn_args* a, void* stream);
DS_API int ds_attention_f16_supported(int d);

/* LayerNorm over the last dimension (ldm/modules/attention.py:206-208): y[r, :] = (x[r, :] - mean) / sqrt(var + eps)
 * * gamma + beta, cols % 4 == 0, cols <= 2048. */
DS_API int ds_layernorm_rows(const float* x, int ldx, const float* gamma, const float* beta, float eps, float* y, int ldy,
                      long long rows, int cols, void* stream);
/* The same with an fp16 output tensor y16[rows][ldy halfs] (rounded to nearest even): in fp16 / autocast mode the LayerNorm output is only
 * the operand of the following projection (torch.autocast casts nn.Linear inputs to fp16). */
DS_API int ds_layernorm_rows_f16(const float* x, int ldx, const float* gamma, const float* beta, float eps, void* y16, int ldy,
                          long long rows, int cols, void* stream);
/* ... and with an fp16 input tensor x16[rows][ldx halfs] as well (a tensor of the fp16 residual stream; statistics and the affine map
 * in fp32 on the widened values). */
DS_API int ds_layernorm_rows_f16io(const void* x16, int ldx, const float* gamma, const float* beta, float eps, void* y16, int ldy,
                            long long rows, int cols, void* stream);

/* GEGLU gate (ldm/modules/attention.py:45-52): y[r, c] = x[r, c] * gelu(x[r, inner + c]) with the exact (erf) GELU. */
DS_API int ds_geglu(const float* x, int ldx, float* y, int ldy, long long rows, int inner, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Noise embedding front end (networks_edm.py:185-198, :314-315, :488-491).
 * sigma: [bs] device.  out[b, :] = PositionalEmbedding(ln(sigma_b)/4) with the precomputed freqs table
 * [nch/2]; swap=1 gives [sin | cos] (SongUNet), swap=0 [cos | sin] (DhariwalUNet).
 * swap bit 1 (value 2): `sigma` already holds the embedding argument (c_noise) -- the LDM `timestep_embedding`
 * (ldm/modules/diffusionmodules/util.py:151-171) fed by CFGPrecond's c_noise = M * sigma_inv(sigma) - 1.
 */
DS_API int ds_noise_embed(const float* sigma, int bs, const float* freqs, int nch, int swap, float* out, int out_ld, void* stream);

/* First-layer input: im2col of c_in(sigma) * x for the 3x3 stem conv, written as [n*h*w][kpad] rows with
 * k = tap*c + ch (zero padded to kpad, a multiple of 32), so that the stem runs as a 1x1 on the MFMA kernel.
 * x: NCHW [n][c][h][w]; sigma: [n] or [1] (sigma_rows).  Fuses networks_edm.py:490,493 (c_in * x). */
DS_API int ds_stem_im2col(const float* x, const float* sigma, int sigma_rows, float sigma_data, int n, int c, int h, int w,
                   float* out, int kpad, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Solver side (solvers.py / solver_utils.py).  All tensors NCHW fp32 with `per` = C*H*W elements per sample.
 *
 * ds_solver_update: one fused pass replacing 4-8 ATen elementwise launches per step (solvers.py:76-81, :156-168,
 * :245-258, :344-352, :574-585; solver_utils.py:102-163 after thresholding):
 *     D      = raw ? c_skip(sig) * xe + c_out(sig) * F_nhwc   (EDMPrecond epilogue, networks_edm.py:495)
 *                  : F                                        (already a denoised NCHW tensor)
 *     d      = afs ? xe / sqrt(1 + t^2) : (xe - D) / t         (solvers.py:77 / :80)
 *     m      = store_d ? d : D                                 (what the multistep history keeps)
 *     x_out  = cx * xb + cm * m + sum_k ch[k] * hist[k]
 * xe = the point the network was evaluated at, xb = the base point of the update (they differ in 2-stage solvers).
 * Per-sample coefficient arrays (AMED) are used when coef_rows == n: each of cx, cm, ch[k], t, sig is then read from
 * coefs[sample*8 + slot]; with coef_rows == 1 the same 8 floats are shared (device-resident: hipGraph replay reads
 * the row a ds_table_select node copied in); coefs == NULL takes the 8 floats by value from hcoefs.
 * Slots: 0 cx, 1 cm, 2..4 ch[0..2], 5 t (divisor of d), 6 sigma (precond), 7 unused.
 */
typedef struct ds_update_args {
    const float* xe; const float* xb;      /* NCHW                                                               */
    const float* f;                        /* raw: network output F -- NHWC [n*h*w][f_ld] rows, or channel-planar NCHW
                                              when f_ld == 0 (ds_conv_args.out_nchw); else NCHW denoised D      */
    int raw; int f_ld;
    const float* hist[3];                  /* NCHW history tensors or NULL                                       */
    const float* coefs; int coef_rows;     /* device [coef_rows][8]; NULL -> use hcoefs (host scalars by value)  */
    float hcoefs[8];
    int afs;
    float sigma_data;
    float* m_out;                          /* optional: write m (d or D) here, NCHW                              */
    int store_d;
    float* x_out;
    int n, c, h, w;
    int variant;                           /* ds_dpmpp_x0_step only: 0 = the library's choice, 1 = always the LDS kernel (tests compare the two) */
} ds_update_args;

DS_API int ds_solver_update(const ds_update_args* a, void* stream);

/* One DPM-Solver++ step in data-prediction form in ONE launch (solvers.py:674-702, solver_utils.py:77-86, :102-163): per sample
 *   D  = as in ds_solver_update (EDM preconditioning of the raw planar network output, a given denoised tensor, or the AFS direction)
 *   m0 = clamp(D, -s, s) / s,  s = max(quantile_p(|D|), 1)      (dynamic thresholding; the same radix select as ds_dynamic_threshold)
 *   x' = hcoefs[0] * xb + hcoefs[1] * m0 + hcoefs[2] * hist[0] + hcoefs[3] * hist[1]
 * m_out receives m0 (the solver's history entry), x_out receives x'.  Replaces the three launches D pass -> threshold -> combination.
 * c * h * w <= ~38 000 elements per sample (LDS); f_ld must be 0 (channel-planar F).
 * Samples of 3x32x32, 3x64x64, 4x64x64 (and 3x16x16) values with 16-B aligned tensors and at most two history tensors run on the
 * register-resident kernel (every operand touched once: 3-4 R + 2 W passes, HBM-bound from a few thousand images per launch);
 * ds_dpmpp_x0_step_in_registers(c*h*w) tells which, ds_update_args.variant = 1 forces the LDS kernel.
 * Both kernels return the exact order statistics, so their results are bit-identical. */
DS_API int ds_dpmpp_x0_step(const ds_update_args* a, float p, void* stream);
DS_API int ds_dpmpp_x0_step_in_registers(long long per_sample);

/* dst[0..row_floats) = table[(*step) * row_floats ...]; then optionally (*step)++ when advance != 0.  The only
 * per-step state of a captured sampler step: every kernel of the step reads its scalars (sigma, coefficients) from
 * `dst`, so one hipGraph replays for all steps. */
DS_API int ds_table_select(const float* table, int row_floats, int* step, int advance, float* dst, void* stream);

/* x0 <- clamp(x0, -s, s) / s with s = max(quantile_{0.995}(|x0|) per sample, 1): solver_utils.py:77-86, exact
 * torch.quantile semantics (linear interpolation between the two order statistics around p*(n-1)). */
DS_API int ds_dynamic_threshold(const float* x0, float* out, int n, int per, float p, void* stream);

/* CFGPrecond epilogue (networks_edm.py:668-690): D = x - sigma * F with, when `doubled`, the classifier-free
 * combination F = F_uncond + guidance * (F_cond - F_uncond) of the two halves of a 2n-image evaluation
 * (rows [0, n*h*w) = unconditional, [n*h*w, 2n*h*w) = conditional).  x / out NCHW [n][c][h][w]; f NHWC rows of f_ld
 * floats; sigma [sigma_rows] (1 = shared). */
DS_API int ds_cfg_denoise(const float* x, const float* f, int f_ld, const float* sigma, int sigma_rows, float guidance, int doubled,
                   int n, int c, int h, int w, float* out, void* stream);

/* y = a * x (latents * t_steps[0], solvers.py:68). */
DS_API int ds_scale(const float* x, float a, float* y, long long count, void* stream);

/* images uint8 NHWC <- clip(x * 127.5 + 128, 0, 255), x NCHW (sample.py:311). */
DS_API int ds_quantize_u8_nhwc(const float* x, uint8_t* out, int n, int c, int h, int w, void* stream);

/* dst[0..count) = value. */
DS_API int ds_fill(float* dst, float value, long long count, void* stream);

/* dst[r, 0:cols] = src[r, 0:cols] for r < rows (strided 2-D copy; pads/gathers label and sigma rows). */
DS_API int ds_copy_rows(const float* src, int src_ld, float* dst, int dst_ld, long long rows, int cols, void* stream);

/* Channel mean of an NHWC tensor: out[n][h*w] = mean_c x[n, hw, c] (AMED bottleneck tap, solvers_amed.py:24-28). */
DS_API int ds_channel_mean(const float* x, int ld, int c, long long rows, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * AMED-Solver (amed-solver-main/solvers_amed.py, amed-solver-main/training/networks.py:56-155).
 *
 * ds_amed_predict: the AMED_predictor MLP.  bottleneck_mean: [n][in_dim] channel mean of the U-Net bottleneck (zeros
 * under AFS); out[n][4] = {r, scale_dir, scale_time, t_mid = t_next^r * t_cur^(1-r)} per sample.
 * Weight pointers are row-major [out][in] fp32 device arrays of the predictor's state_dict; fc_sd_* / fc_st_* may be
 * NULL (scale_dir = 0 / scale_time = 0 at training time: the head does not exist and the factor is 1).
 */
typedef struct ds_amed_predictor {
    const float* map0_w; const float* map0_b;      /* map_layer0: [nc][nc], [nc]                  */
    const float* enc0_w; const float* enc0_b;      /* enc_layer0: [hidden][in_dim], [hidden]      */
    const float* enc1_w; const float* enc1_b;      /* enc_layer1: [out_dim][hidden], [out_dim]    */
    const float* fc_r_w; const float* fc_r_b;      /* [1][out_dim + 2 nc], [1]                    */
    const float* fc_sd_w; const float* fc_sd_b;
    const float* fc_st_w; const float* fc_st_b;
    int nc, in_dim, hidden, out_dim;
    float scale_dir, scale_time;
} ds_amed_predictor;

DS_API int ds_amed_predict(const ds_amed_predictor* p, const float* bottleneck_mean, int n, float t_cur, float t_next, float* out,
                    void* stream);

#define DS_AMED_AMED 0     /* amed_sampler           solvers_amed.py:69-159  */
#define DS_AMED_EULER 1    /* euler plugin           :163-257                */
#define DS_AMED_IPNDM 2    /* iPNDM plugin           :262-396                */
#define DS_AMED_DPM2 3     /* DPM-Solver-2 plugin    :400-494                */
#define DS_AMED_DPMPP 4    /* DPM-Solver++ plugin    :498-631                */

/* Per-sample coefficient rows ([n][8], the layout ds_solver_update reads) of stage 1 (step from t_cur to the learned
 * t_mid) or stage 2 (step to t_next after the evaluation at scale_time * t_mid).  `order` = the multistep order of this
 * stage (host bookkeeping).  thist: [n][4] floats {count, t_oldest, .., t_newest}, the per-sample DPM-Solver++ time
 * history (t_mid differs per sample); updated in place.  sigma2 (stage 1 only): [n] <- scale_time * t_mid. */
typedef struct ds_amed_coef_args {
    const float* pred;          /* [n][4] from ds_amed_predict */
    float t_cur, t_next;
    int mode, stage, order, predict_x0;
    float* thist;
    float* coefs;
    float* sigma2;
    int n;
} ds_amed_coef_args;

DS_API int ds_amed_coefs(const ds_amed_coef_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GITS schedule search (gits-main/gits_utils.py:108-132 cost matrix, :237-255 cal_deviation).
 * traj: [n_pts][batch][per] fp32 teacher trajectory (return_inters), eps: [n_pts-1][batch][per] its directions d_i
 * (return_eps; may be NULL for ds_traj_moments).
 *
 * ds_traj_moments: out[(i*batch + b)*6 + k], fp64, k = {P, Q, R, S, T, N} with b0 = traj[0], c = traj[n_pts-1]:
 *   P = (c - x_i).(c - b0)  Q = d_i.(c - b0)  R = |c - x_i|^2  S = (c - x_i).d_i  T = |d_i|^2  N = |c - b0|^2.
 *   The 'dev' cost of every Euler jump i -> j follows in closed form (see csrc/gits.hip).
 * ds_traj_pair_cost: cost[i*n_pts + j] += sum_b | x_i + (t_j - t_i) d_i - x_j |_p for all i < j (p_norm 1 or 2;
 *   cost must be zeroed by the caller; t_steps is a device array [n_pts]).
 */
DS_API int ds_traj_moments(const float* traj, const float* eps, int n_pts, int batch, int per, double* out, void* stream);
DS_API int ds_traj_pair_cost(const float* traj, const float* eps, const float* t_steps, int n_pts, int batch, int per, int p_norm,
                      double* cost, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * FID moments (diff-solvers-main/fid.py:62-71): one batch of detector features [rows][ld] (fp32 when features_f64 == 0 -- widened to
 * fp64 exactly, the reference's `.to(torch.float64)` -- or fp64 when 1; the first `dim` columns of each row) accumulated IN PLACE
 * into the running sums  mu[dim] += features.sum(0)  and  sigma[dim][dim] += features^T features  (row-major, fp64), on the fp64
 * matrix pipe (v_mfma_f64_16x16x4_f64), one launch per batch; any rows >= 0, any dim >= 1.  The update is bitwise symmetric.
 * The SUM all-reduce of mu / sigma over ranks (fid.py:74-75) and the finalisation (fid.py:76-78) stay with the caller
 * (diff_sampler_amd/fid.py: torch.distributed over RCCL). */
DS_API int ds_fid_moments(const void* features, int features_f64, int ld, int rows, int dim, double* mu, double* sigma, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Native launch plans (SURVEY.md section 8b "what a C-ABI engine should export": ds_unet_forward / ds_graph_capture_step).
 *
 * One network evaluation -- EDMPrecond.forward -> SongUNet / DhariwalUNet.forward (networks_edm.py:482-496, :312-355, :427-453) or
 * CFGPrecond.forward -> UNetModel.forward (networks_edm.py:668-690, openaimodel.py:710-742) -- is a fixed sequence of the launches
 * above over fixed workspaces (178 for the CIFAR-10 net, 361 for SD-1.5).  A ds_plan owns a COPY of every launch's argument
 * struct, so a host in any language records the sequence once (ds_plan_add, in launch order) and then runs the whole forward with
 * one call: ds_plan_run walks the list in C on `stream`; ds_plan_graph_capture records it into a hipGraph on a non-default
 * stream (north_star: "each NFE step a hipGraph-captured sequence") and ds_plan_graph_launch replays it.  The device pointers
 * inside the argument structs stay the caller's (the plan never allocates device memory); they must stay valid while the plan
 * lives.  The inputs of an evaluation (x, sigma, labels / context) are written into the plan's input buffers by the caller
 * before the run (ds_copy_rows / ds_fill on the same stream), the output is read from its output buffer after it.
 * The entry points with scalar arguments get argument structs here so that they can be recorded. */
typedef struct ds_layernorm_args { const float* x; int ldx; const float* gamma; const float* beta; float eps; float* y; int ldy;
                                   long long rows; int cols; } ds_layernorm_args;                       /* ds_layernorm_rows */
typedef struct ds_geglu_args { const float* x; int ldx; float* y; int ldy; long long rows; int inner; } ds_geglu_args;   /* ds_geglu */
typedef struct ds_noise_embed_args { const float* sigma; int bs; const float* freqs; int nch; int swap; float* out; int out_ld;
                                   } ds_noise_embed_args;                                              /* ds_noise_embed */
typedef struct ds_stem_im2col_args { const float* x; const float* sigma; int sigma_rows; float sigma_data; int n, c, h, w;
                                     float* out; int kpad; } ds_stem_im2col_args;                      /* ds_stem_im2col */

enum { DS_OP_CONV2D = 1,        /* ds_conv_args        -> ds_conv2d_nhwc      */
       DS_OP_GEMM = 2,          /* ds_gemm_args        -> ds_gemm_nt_batched  */
       DS_OP_GN_STATS = 3,      /* ds_norm_args        -> ds_gn_stats         */
       DS_OP_NORM_ACT = 4,      /* ds_norm_args        -> ds_norm_act         */
       DS_OP_GN_FINALIZE = 5,   /* ds_gn_finalize_args -> ds_gn_finalize      */
       DS_OP_ATTENTION = 6,     /* ds_attn_args        -> ds_attention        */
       DS_OP_ATTENTION_F16 = 7, /* ds_attn_args        -> ds_attention_f16    */
       DS_OP_LAYERNORM = 8,     /* ds_layernorm_args   -> ds_layernorm_rows   */
       DS_OP_GEGLU = 9,         /* ds_geglu_args       -> ds_geglu            */
       DS_OP_NOISE_EMBED = 10,  /* ds_noise_embed_args -> ds_noise_embed      */
       DS_OP_STEM_IM2COL = 11,  /* ds_stem_im2col_args -> ds_stem_im2col      */
       DS_OP_LAYERNORM_F16 = 12, /* ds_layernorm_args  -> ds_layernorm_rows_f16 (y = fp16 rows) */
       DS_OP_LAYERNORM_F16IO = 13 /* ds_layernorm_args -> ds_layernorm_rows_f16io (x and y = fp16 rows) */ };

typedef struct ds_plan ds_plan;
DS_API int ds_plan_create(ds_plan** out);
/* Appends one launch; `args` (the struct named above for `op`, `args_bytes` = its sizeof, checked) is copied.  DS_E_ARG otherwise. */
DS_API int ds_plan_add(ds_plan* plan, int op, const void* args, unsigned long long args_bytes);
DS_API int ds_plan_size(const ds_plan* plan);
/* Issues every launch of the plan on `stream`, in order.  Returns 0 or the first failing launch's code; ds_plan_last_failed then
 * gives its index (-1 when the last run succeeded). */
DS_API int ds_plan_run(ds_plan* plan, void* stream);
DS_API int ds_plan_last_failed(const ds_plan* plan);
/* Captures one run of the plan on `stream` (not the legacy default stream) into a hipGraph and instantiates it; a plan holds at most
 * one graph (a second capture replaces it).  ds_plan_graph_launch replays it on `stream`. */
DS_API int ds_plan_graph_capture(ds_plan* plan, void* stream);
DS_API int ds_plan_graph_launch(ds_plan* plan, void* stream);
DS_API void ds_plan_destroy(ds_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* DS_ENGINE_H */
