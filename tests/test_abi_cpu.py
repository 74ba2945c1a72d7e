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
    declared = set(re.findall(r'^(?:int|long long|const char\*)\s+(ds_\w+)\s*\(', header, flags=re.M))
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in ds_engine.h but not exported'
    assert set(_lib.EXPORTS) == declared
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
