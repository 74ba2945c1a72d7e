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


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: libdsamd.so cannot be built on this machine')
    from concurrent.futures import ThreadPoolExecutor

    def compile_one(src):
        obj = src[:-4] + '.o'
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC'] + EXTRA_FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    # one hipcc per translation unit, in parallel (the two convolution files dominate: ~2-3 min each)
    with ThreadPoolExecutor(max_workers=max(1, min(len(sources()), os.cpu_count() or 1))) as pool:
        objs = list(pool.map(compile_one, sources()))
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
