"""Build csrc/*.hip into csrc/libdsamd.so for gfx950 (explicit hipcc, in-tree so the .so travels with the repo)."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libdsamd.so')
ARCH = 'gfx950'
# extra device-compiler flags (experiments: DS_HIPCC_FLAGS="-mllvm -amdgpu-mfma-vgpr-form=0")
EXTRA_FLAGS = os.environ.get('DS_HIPCC_FLAGS', '').split()
# DS_BUILD_EXPERIMENTS=1: also build the kernel variants that are kept only as A/B records (docs/HISTORY.md) -- never chosen by the engines:
# conv3x3_f16dmah.hip (four-wave half-slab fp16 convolution, measured 4 - 19 % slower), conv3x3_halo2_kernel<., 0 / 1> (the hand-scheduled fp32
# twin and the fp32-activation fp16 kernel that conv3x3_f16dma superseded).  The default library holds the product kernels only;
# ds_build_experiments() tells a host (and the tests of those variants) which build it loaded.
EXPERIMENTS = os.environ.get('DS_BUILD_EXPERIMENTS', '0') == '1'
EXPERIMENT_SOURCES = ('conv3x3_f16dmah.hip',)
# -fvisibility=hidden: only the DS_API entry points of include/ds_engine.h are exported (tests/test_abi_cpu.py checks the symbol table)
BASE_FLAGS = ['-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden'] + (['-DDS_BUILD_EXPERIMENTS=1'] if EXPERIMENTS else [])
STAMP = os.path.join(CSRC, '.build_flags')


def sources():
    return sorted(s for s in glob.glob(os.path.join(CSRC, '*.hip')) if EXPERIMENTS or os.path.basename(s) not in EXPERIMENT_SOURCES)


def _flags_stamp():
    return ' '.join(BASE_FLAGS + EXTRA_FLAGS)


def _stamp_matches():
    try:
        with open(STAMP) as fh:
            return fh.read() == _flags_stamp()
    except OSError:
        return False


def needs_build():
    if not os.path.exists(LIB) or not _stamp_matches():
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def headers():
    return sorted(glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h')))


def _code_only(text):
    """C / C++ source with comments and whitespace runs removed: what the compiler sees, roughly (string literals containing comment
    markers do not occur in csrc/)."""
    import re
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    return re.sub(r'\s+', ' ', text).strip()


def source_sha256(tu):
    """Hash of the CODE a translation unit of csrc/ is compiled from: its .hip source and every header of csrc/ and include/ (a superset
    of what it includes), comments and whitespace stripped -- a comment edit in a header must not invalidate a measurement.  bench.py ties
    profiler-derived numbers (profiles/*pmc*.json) to the kernel source they were measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, tu)] + headers():
        h.update(os.path.basename(f).encode() + b'\0')
        with open(f, 'r', encoding='utf-8') as fh:
            h.update(_code_only(fh.read()).encode())
    return h.hexdigest()


def source_hashes():
    return {os.path.basename(s): source_sha256(os.path.basename(s)) for s in sources()}


def build_lib(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: libdsamd.so cannot be built on this machine')
    from concurrent.futures import ThreadPoolExecutor

    newest_header = max([os.path.getmtime(h) for h in headers()] or [0.0])
    if not _stamp_matches():
        force = True                                            # objects compiled under other flags (visibility, experiments) are stale

    def compile_one(src):
        obj = src[:-4] + '.o'
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header):
            return obj                                          # this translation unit is up to date
        cmd = [hipcc, f'--offload-arch={ARCH}'] + BASE_FLAGS + EXTRA_FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    # one hipcc per translation unit, in parallel (the two convolution files dominate: ~2-3 min each)
    with ThreadPoolExecutor(max_workers=max(1, min(len(sources()), os.cpu_count() or 1))) as pool:
        objs = list(pool.map(compile_one, sources()))
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-fvisibility=hidden', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, 'w') as fh:
        fh.write(_flags_stamp())
    return LIB


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
