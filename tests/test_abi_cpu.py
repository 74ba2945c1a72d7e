"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/ds_engine.h declares."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_library_builds_and_exports_header_symbols():
    from diff_sampler_amd import build, _lib
    build.build_lib(verbose=False)
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'ds_engine.h')).read()
    declared = set(re.findall(r'^DS_API\s+(?:int|long long|const char\*|void)\s+(ds_\w+)\s*\(', header, flags=re.M))
    assert declared, 'no declarations parsed'
    assert not re.findall(r'^(?:int|long long|const char\*|void)\s+ds_\w+\s*\(', header, flags=re.M), 'an entry point without DS_API'
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in ds_engine.h but not exported'
    assert set(_lib.EXPORTS) == declared
    # the ABI surface IS the header: the library is built with -fvisibility=hidden, so its dynamic symbol table must hold exactly the
    # DS_API functions as code symbols -- no C++ launch helper (igemm::...) leaks out
    import subprocess
    nm = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    text = {ln.split()[2] for ln in nm.splitlines() if len(ln.split()) == 3 and ln.split()[1] in 'TtWw'}
    assert text == declared, (sorted(text - declared), sorted(declared - text))
    assert lib.ds_build_experiments() in (0, 1)
    assert lib.ds_version() >= 1
    assert lib.ds_error_string(-3) == b'unsupported shape'


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from diff_sampler_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    try:
        _lib.load()
    except _lib.DsError as e:
        assert 'no CPU fallback' in str(e)
    else:
        raise AssertionError('expected DsError')


def test_native_plan_records_every_launch_and_checks_argument_sizes():
    """ds_plan_* (include/ds_engine.h "Native launch plans"): host-side bookkeeping only -- no kernel runs without a GPU."""
    import ctypes as C
    import diff_sampler_amd.arch as arch
    from diff_sampler_amd import _lib
    from diff_sampler_amd.engine import UNetEngine
    lib = _lib.load()
    spec = arch.edm_precond_spec(**dict(arch.NAMED_CONFIGS['tiny_song']))
    P = UNetEngine(spec, arch.init_params(spec, seed=1), device='cpu').plan(4, 1)
    h = P.native()
    assert lib.ds_plan_size(h) == len(P.ops) > 20
    assert P.native().value == h.value                       # cached
    assert lib.ds_plan_last_failed(h) == -1
    a = _lib.ConvArgs()
    assert lib.ds_plan_add(h, _lib.DS_OP_CONV2D, C.byref(a), C.sizeof(a) - 4) != 0      # wrong struct size
    assert lib.ds_plan_add(h, 99, C.byref(a), C.sizeof(a)) != 0                         # unknown op code
    assert lib.ds_plan_add(None, _lib.DS_OP_CONV2D, C.byref(a), C.sizeof(a)) != 0
    assert lib.ds_plan_graph_launch(h, None) != 0                                       # nothing captured
    assert lib.ds_plan_size(h) == len(P.ops)
    P.close()
    assert P._native is None


def test_fp16_activation_gemm_epilogue_keeps_the_kernel_argument_out_of_private_memory():
    """Round-4 regression guard (no GPU needed: hipcc cross-compiles).  The fused epilogue of the fp16-activation kernels
    (igemm_common.h: epi_process) once fetched `p.acc_scale` with a <4 x float> load straddling neighbouring KParams fields, which kept a
    28-byte slice of the kernel argument in PRIVATE memory -- re-read from scratch in every pass of every tile -- and its GEGLU branch
    pushed the 192-column tiles into register spills.  Every instantiation of gemm_f16dma_kernel with column tiles of at most 192 channels
    must need no scratch at all (the 256-column tile keeps its known accumulator spills in the epilogue)."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src = os.path.join(ROOT, 'diff_sampler_amd', 'csrc', 'gemm_f16dma.hip')
    r = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', src, '-o', os.devnull, '-Rpass-analysis=kernel-resource-usage'],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    found = re.findall(r'Function Name: (\S*gemm_f16dma_kernelILi(\d)ELi(\d)E\S*).*?ScratchSize \[bytes/lane\]: (\d+)', r.stderr, flags=re.S)
    assert len(found) >= 7, r.stderr[-1500:]
    for name, nb, nw, scratch in found:
        if int(nb) <= 3:
            assert int(scratch) == 0, (name, scratch)
