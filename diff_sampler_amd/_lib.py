"""ctypes binding of csrc/libdsamd.so (the C ABI declared in include/ds_engine.h).

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.  PyTorch is used by
callers only for device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DS_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libdsamd.so')      # DS_LIB_PATH: an alternative build (A/B runs of tools/)

c_float_p = C.POINTER(C.c_float)
vp = C.c_void_p

DS_ACT_NONE, DS_ACT_SILU, DS_ACT_GEGLU = 0, 1, 2
DS_RESAMPLE_NONE, DS_RESAMPLE_DOWN, DS_RESAMPLE_UP = 0, 1, 2
DS_GN_MAX_CHUNKS = 32


class ConvTune(C.Structure):
    """ds_conv_tune: per-call kernel selection overrides (include/ds_engine.h); all zero = the library's own choice."""
    _fields_ = [('mode', C.c_int), ('variant', C.c_int), ('splits', C.c_int), ('f16dma_nb', C.c_int), ('f16dma_nw', C.c_int),
                ('ablate', C.c_int)]


_tune_state = threading.local()


class tuning:
    """``with _lib.tuning(variant=3, mode=256): ...`` -- every ConvArgs CONSTRUCTED on this thread inside the block carries these
    ds_conv_tune overrides (benchmarks, A/B runs, tests).  The state is this thread's default for argument construction on the Python
    side; libdsamd.so itself is stateless: the overrides travel inside each call's (or plan entry's) argument struct."""

    def __init__(self, **kw):
        bad = set(kw) - {f for f, _ in ConvTune._fields_}
        if bad:
            raise TypeError(f'unknown ds_conv_tune fields {sorted(bad)}')
        self.kw = kw

    def __enter__(self):
        self.prev = getattr(_tune_state, 'kw', None)
        _tune_state.kw = dict(self.prev or {}, **self.kw)
        return self

    def __exit__(self, *exc):
        _tune_state.kw = self.prev
        return False


def current_tuning():
    kw = dict(_ENV_TUNE)
    kw.update(getattr(_tune_state, 'kw', None) or {})
    return kw


# benchmarks / A-B runs of whole programs: DS_CONV_VARIANT / DS_CONV (= tune.mode) / DS_CONV_ABLATE (= tune.ablate, e.g. 4096: the staged
# epilogue of the fp16-activation kernels) in the environment become the default overrides
_ENV_TUNE = {k: int(os.environ[e]) for k, e in (('variant', 'DS_CONV_VARIANT'), ('mode', 'DS_CONV'), ('ablate', 'DS_CONV_ABLATE')) if os.environ.get(e)}


_ENV_ATTN_VARIANT = int(os.environ.get('DS_ATTN_VARIANT', '0') or 0)


class ConvArgs(C.Structure):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for f, v in current_tuning().items():
            setattr(self.tune, f, int(v))

    _fields_ = [('x0', vp), ('x1', vp), ('c0', C.c_int), ('c1', C.c_int), ('ld0', C.c_int), ('ld1', C.c_int),
                ('n', C.c_int), ('h', C.c_int), ('w', C.c_int), ('taps', C.c_int), ('wgt', vp), ('cout', C.c_int),
                ('bias', vp), ('cbias', vp), ('cbias_ld', C.c_int), ('cbias_rows', C.c_int), ('res', vp),
                ('res_ld', C.c_int), ('out_scale', C.c_float), ('act', C.c_int), ('out', vp), ('out_ld', C.c_int),
                ('norm_coefs', vp), ('norm_act', C.c_int),
                ('e0', vp), ('e1', vp), ('ec0', C.c_int), ('ec1', C.c_int), ('eld0', C.c_int), ('eld1', C.c_int),
                ('stride', C.c_int), ('workspace', vp), ('workspace_floats', C.c_longlong),
                ('out_nchw', C.c_int), ('stats_out', vp), ('wgt_f16', C.c_int), ('wgt_shift', C.c_int), ('in_f16', C.c_int), ('out_f16', C.c_int), ('res_f16', C.c_int), ('tune', ConvTune),
                ('update', vp)]          # const ds_update_args*: the solver update fused into the network head (ABI 3)


class GemmArgs(C.Structure):
    _fields_ = [('a', vp), ('lda', C.c_int), ('a_bstride', C.c_longlong), ('a_hstride', C.c_longlong),
                ('b', vp), ('ldb', C.c_int), ('b_bstride', C.c_longlong), ('b_hstride', C.c_longlong),
                ('c', vp), ('ldc', C.c_int), ('c_bstride', C.c_longlong), ('c_hstride', C.c_longlong),
                ('m', C.c_int), ('n', C.c_int), ('k', C.c_int), ('batch', C.c_int), ('heads', C.c_int),
                ('alpha', C.c_float), ('rowbias', vp), ('colbias', vp), ('act', C.c_int)]


class NormArgs(C.Structure):
    _fields_ = [('x0', vp), ('x1', vp), ('c0', C.c_int), ('c1', C.c_int), ('ld0', C.c_int), ('ld1', C.c_int),
                ('n', C.c_int), ('h', C.c_int), ('w', C.c_int), ('groups', C.c_int), ('eps', C.c_float),
                ('mean', vp), ('rstd', vp), ('gamma', vp), ('beta', vp), ('scale', vp), ('shift', vp),
                ('ss_ld', C.c_int), ('ss_rows', C.c_int), ('act', C.c_int), ('resample', C.c_int), ('out', vp),
                ('out_ld', C.c_int), ('coefs', vp), ('partial', vp), ('counters', vp), ('out_f16', C.c_int), ('raw_out', vp), ('raw_ld', C.c_int), ('in_f16', C.c_int),
                ('stats0', vp), ('stats1', vp), ('tune_variant', C.c_int)]          # ABI 4: the pass that computes its own GroupNorm statistics


class AttnArgs(C.Structure):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if _ENV_ATTN_VARIANT and 'variant' not in k and len(a) < 21:
            self.variant = _ENV_ATTN_VARIANT          # DS_ATTN_VARIANT=1 / 2 (A/B runs of whole programs): ds_attn_args.variant of every call built here

    _fields_ = [('q', vp), ('k', vp), ('v', vp), ('out', vp), ('ldq', C.c_int), ('ldk', C.c_int), ('ldv', C.c_int),
                ('ldo', C.c_int), ('q_bs', C.c_longlong), ('k_bs', C.c_longlong), ('v_bs', C.c_longlong),
                ('o_bs', C.c_longlong), ('batch', C.c_int), ('heads', C.c_int), ('sq', C.c_int), ('skv', C.c_int),
                ('d', C.c_int), ('scale', C.c_float), ('out_f16', C.c_int), ('in_f16', C.c_int), ('variant', C.c_int)]      # variant: ABI 4


class GnFinalizeArgs(C.Structure):
    _fields_ = [('stats0', vp), ('stats1', vp), ('c0', C.c_int), ('c1', C.c_int), ('n', C.c_int), ('hw', C.c_int),
                ('groups', C.c_int), ('eps', C.c_float), ('gamma', vp), ('beta', vp), ('scale', vp), ('shift', vp),
                ('ss_ld', C.c_int), ('ss_rows', C.c_int), ('mean', vp), ('rstd', vp), ('coefs', vp)]


class UpdateArgs(C.Structure):
    _fields_ = [('xe', vp), ('xb', vp), ('f', vp), ('raw', C.c_int), ('f_ld', C.c_int), ('hist', vp * 3),
                ('coefs', vp), ('coef_rows', C.c_int), ('hcoefs', C.c_float * 8), ('afs', C.c_int),
                ('sigma_data', C.c_float), ('m_out', vp), ('store_d', C.c_int), ('x_out', vp),
                ('n', C.c_int), ('c', C.c_int), ('h', C.c_int), ('w', C.c_int), ('variant', C.c_int)]


class AmedPredictor(C.Structure):
    _fields_ = [('map0_w', vp), ('map0_b', vp), ('enc0_w', vp), ('enc0_b', vp), ('enc1_w', vp), ('enc1_b', vp),
                ('fc_r_w', vp), ('fc_r_b', vp), ('fc_sd_w', vp), ('fc_sd_b', vp), ('fc_st_w', vp), ('fc_st_b', vp),
                ('nc', C.c_int), ('in_dim', C.c_int), ('hidden', C.c_int), ('out_dim', C.c_int),
                ('scale_dir', C.c_float), ('scale_time', C.c_float)]


class AmedCoefArgs(C.Structure):
    _fields_ = [('pred', vp), ('t_cur', C.c_float), ('t_next', C.c_float), ('mode', C.c_int), ('stage', C.c_int),
                ('order', C.c_int), ('predict_x0', C.c_int), ('thist', vp), ('coefs', vp), ('sigma2', vp), ('n', C.c_int)]


class LayerNormArgs(C.Structure):
    _fields_ = [('x', vp), ('ldx', C.c_int), ('gamma', vp), ('beta', vp), ('eps', C.c_float), ('y', vp), ('ldy', C.c_int),
                ('rows', C.c_longlong), ('cols', C.c_int)]


class GegluArgs(C.Structure):
    _fields_ = [('x', vp), ('ldx', C.c_int), ('y', vp), ('ldy', C.c_int), ('rows', C.c_longlong), ('inner', C.c_int)]


class NoiseEmbedArgs(C.Structure):
    _fields_ = [('sigma', vp), ('bs', C.c_int), ('freqs', vp), ('nch', C.c_int), ('swap', C.c_int), ('out', vp), ('out_ld', C.c_int)]


class StemIm2colArgs(C.Structure):
    _fields_ = [('x', vp), ('sigma', vp), ('sigma_rows', C.c_int), ('sigma_data', C.c_float), ('n', C.c_int), ('c', C.c_int),
                ('h', C.c_int), ('w', C.c_int), ('out', vp), ('kpad', C.c_int)]


# ds_plan_add op codes (include/ds_engine.h)
DS_OP_CONV2D, DS_OP_GEMM, DS_OP_GN_STATS, DS_OP_NORM_ACT, DS_OP_GN_FINALIZE, DS_OP_ATTENTION, DS_OP_ATTENTION_F16, DS_OP_LAYERNORM, \
    DS_OP_GEGLU, DS_OP_NOISE_EMBED, DS_OP_STEM_IM2COL, DS_OP_LAYERNORM_F16, DS_OP_LAYERNORM_F16IO = range(1, 14)

_SIGNATURES = {
    'ds_version': (C.c_int, []),
    'ds_build_experiments': (C.c_int, []),
    'ds_error_string': (C.c_char_p, [C.c_int]),
    'ds_conv2d_nhwc': (C.c_int, [C.POINTER(ConvArgs), vp]),
    'ds_conv_kernel_id': (C.c_int, [C.POINTER(ConvArgs)]),
    'ds_conv3x3_halo_supported': (C.c_int, [C.c_int, C.c_int]),
    'ds_conv_f16_supported': (C.c_int, [C.c_int] * 7),
    'ds_conv_f16dma_supported': (C.c_int, [C.c_int] * 6),
    'ds_gemm_f16dma_supported': (C.c_int, [C.c_longlong, C.c_int, C.c_int]),
    'ds_conv_f16dma_stride2_supported': (C.c_int, [C.c_int] * 5),
    'ds_conv_split_supported': (C.c_int, [C.c_int] * 7),
    'ds_gemm_f16_supported': (C.c_int, [C.c_longlong, C.c_int, C.c_int]),
    'ds_gemm_nt_batched': (C.c_int, [C.POINTER(GemmArgs), vp]),
    'ds_gn_stats': (C.c_int, [C.POINTER(NormArgs), vp]),
    'ds_norm_act': (C.c_int, [C.POINTER(NormArgs), vp]),
    'ds_gn_finalize': (C.c_int, [C.POINTER(GnFinalizeArgs), vp]),
    'ds_softmax_rows': (C.c_int, [vp, vp, C.c_longlong, C.c_int, C.c_int, vp]),
    'ds_attention': (C.c_int, [C.POINTER(AttnArgs), vp]),
    'ds_attention_supported': (C.c_int, [C.c_int]),
    'ds_attention_f16': (C.c_int, [C.POINTER(AttnArgs), vp]),
    'ds_attention_f16_supported': (C.c_int, [C.c_int]),
    'ds_layernorm_rows': (C.c_int, [vp, C.c_int, vp, vp, C.c_float, vp, C.c_int, C.c_longlong, C.c_int, vp]),
    'ds_layernorm_rows_f16': (C.c_int, [vp, C.c_int, vp, vp, C.c_float, vp, C.c_int, C.c_longlong, C.c_int, vp]),
    'ds_layernorm_rows_f16io': (C.c_int, [vp, C.c_int, vp, vp, C.c_float, vp, C.c_int, C.c_longlong, C.c_int, vp]),
    'ds_geglu': (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_longlong, C.c_int, vp]),
    'ds_cfg_denoise': (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'ds_noise_embed': (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, vp]),
    'ds_stem_im2col': (C.c_int, [vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    'ds_solver_update': (C.c_int, [C.POINTER(UpdateArgs), vp]),
    'ds_dpmpp_x0_step': (C.c_int, [C.POINTER(UpdateArgs), C.c_float, vp]),
    'ds_dpmpp_x0_step_in_registers': (C.c_int, [C.c_longlong]),
    'ds_table_select': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp]),
    'ds_dynamic_threshold': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    'ds_scale': (C.c_int, [vp, C.c_float, vp, C.c_longlong, vp]),
    'ds_quantize_u8_nhwc': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'ds_fill': (C.c_int, [vp, C.c_float, C.c_longlong, vp]),
    'ds_copy_rows': (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_longlong, C.c_int, vp]),
    'ds_amed_predict': (C.c_int, [C.POINTER(AmedPredictor), vp, C.c_int, C.c_float, C.c_float, vp, vp]),
    'ds_amed_coefs': (C.c_int, [C.POINTER(AmedCoefArgs), vp]),
    'ds_fid_moments': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    'ds_traj_moments': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'ds_traj_pair_cost': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'ds_philox_randn': (C.c_int, [vp, C.c_ulonglong, vp, C.c_int, C.c_longlong, C.c_longlong, vp]),
    'ds_philox_probe': (C.c_int, [C.c_ulonglong, C.c_ulonglong, vp, C.c_int, C.c_int, vp]),
    'ds_philox_randint': (C.c_int, [vp, C.c_ulonglong, C.c_uint, vp, C.c_int, vp]),
    'ds_channel_mean': (C.c_int, [vp, C.c_int, C.c_int, C.c_longlong, vp, vp]),
    'ds_plan_create': (C.c_int, [C.POINTER(vp)]),
    'ds_plan_add': (C.c_int, [vp, C.c_int, vp, C.c_ulonglong]),
    'ds_plan_size': (C.c_int, [vp]),
    'ds_plan_run': (C.c_int, [vp, vp]),
    'ds_plan_last_failed': (C.c_int, [vp]),
    'ds_plan_graph_capture': (C.c_int, [vp, vp]),
    'ds_plan_graph_launch': (C.c_int, [vp, vp]),
    'ds_plan_destroy': (None, [vp]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None


class DsError(RuntimeError):
    pass


def load():
    """Load libdsamd.so (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; importing it FIRST makes libdsamd.so bind to that same runtime instance
    # (same SONAME), so that torch's stream handles and allocations are valid in our launches.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DsError(f'{LIB_PATH} is missing: build it with `python diff_sampler_amd/build.py` '
                      f'(or __graft_entry__.build()).  The HIP engine has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.ds_version() != 4:
        raise DsError(f'{LIB_PATH} reports ABI version {lib.ds_version()}, this binding is written for 4: rebuild it (python diff_sampler_amd/build.py)')
    _lib = lib
    return lib


def check(code, what=''):
    if code != 0:
        msg = load().ds_error_string(code)
        raise DsError(f'{what or "libdsamd call"} failed with code {code}: {msg.decode() if msg else "?"}')


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_HIP = [None]


def capture_id():
    """Id of the stream capture the current stream is recording into (hipStreamGetCaptureInfo), or 0 when it is not capturing.  Scopes
    per-capture caches (ldm_engine: the context K / V projections recorded once per hipGraph) to ONE capture: a later capture -- by anyone,
    on the same static tensors -- never matches an earlier capture's key."""
    if _HIP[0] is None:
        _HIP[0] = C.CDLL('libamdhip64.so')
        _HIP[0].hipStreamGetCaptureInfo.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_ulonglong)]
        _HIP[0].hipStreamGetCaptureInfo.restype = C.c_int
    status, cid = C.c_int(0), C.c_ulonglong(0)
    rc = _HIP[0].hipStreamGetCaptureInfo(stream_ptr(), C.byref(status), C.byref(cid))
    if rc != 0 or status.value != 1:          # hipStreamCaptureStatusActive == 1
        return 0
    return int(cid.value) or 1
