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


# Test-only VARIANT builds of the same sources (round 6): DS_RACE_STRESS libraries whose LDS-producing waves are delayed ~2 us
# (csrc/ds_common.h: DS_RACE_SKEW) -- two complementary wave masks.  They live next to libdsamd.so as libdsamd_<tag>.so (objects in
# csrc/build_<tag>/), are loaded only through DS_LIB_PATH (tests/test_hip_race_stress.py) and are never what the engines run.
# 'stress_g5' = stress_a with the round-5 race RE-INTRODUCED (conv3x3_f16dma.hip without its prologue barrier, docs/HISTORY.md G.5): the
# test suite asserts that the kernel tests FAIL against it -- the detector detects.  It recompiles that one translation unit and links the
# other objects of stress_a.
VARIANTS = {'stress_a': ['-DDS_RACE_STRESS=0x21'], 'stress_b': ['-DDS_RACE_STRESS=0xDE'],
            'stress_g5': ['-DDS_RACE_STRESS=0x21', '-DDS_TEST_DROP_G5_BARRIER=1'],
            'timeline': ['-DDS_TIMELINE=1']}        # diagnostics (tools/timeline_gemm.py): phase stamps per workgroup, never what the engines run
TEST_VARIANTS = ('stress_a', 'stress_b', 'stress_g5')      # what __graft_entry__.build() compiles next to the product ('timeline' is built on demand by its tool)
VARIANT_BASE = {'stress_g5': ('stress_a', ('conv3x3_f16dma.hip',))}        # tag -> (variant whose objects it shares, the translation units it compiles itself)


def variant_lib(tag):
    return os.path.join(CSRC, f'libdsamd_{tag}.so') if tag else LIB


def _variant_paths(tag):
    """(library, object directory, stamp file, extra flags) of a build: tag '' = the product library."""
    if not tag:
        return LIB, CSRC, STAMP, []
    odir = os.path.join(CSRC, f'build_{tag}')
    return variant_lib(tag), odir, os.path.join(odir, '.build_flags'), list(VARIANTS[tag])


def _stamp_ok(stamp, text):
    try:
        with open(stamp) as fh:
            return fh.read() == text
    except OSError:
        return False


def variant_needs_build(tag):
    lib, odir, stamp, extra = _variant_paths(tag)
    if not os.path.exists(lib) or not _stamp_ok(stamp, _flags_stamp() + ' ' + ' '.join(extra)):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in sources() + headers())


def build_libs(tags=('',), force=False, verbose=True):
    """Compile every translation unit of every requested build in ONE pool (one hipcc per (build, translation unit); the two convolution
    files dominate: ~2-3 min each), then link each library."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    todo = [t for t in tags if force or (needs_build() if not t else variant_needs_build(t))]
    for t in list(todo):                                        # a variant that shares objects needs its base built in the same pass
        b = VARIANT_BASE.get(t, (None,))[0]
        if b is not None and b not in todo:
            todo.insert(todo.index(t), b)
    if not todo:
        return [variant_lib(t) for t in tags]
    if not os.path.exists(hipcc):
        raise RuntimeError('hipcc not found: libdsamd.so cannot be built on this machine')
    from concurrent.futures import ThreadPoolExecutor

    newest_header = max([os.path.getmtime(h) for h in headers()] or [0.0])
    jobs, links = [], []
    for tag in todo:
        lib, odir, stamp, extra = _variant_paths(tag)
        os.makedirs(odir, exist_ok=True)
        text = _flags_stamp() + ((' ' + ' '.join(extra)) if tag else '')
        stale = force or not _stamp_ok(stamp, text)             # objects compiled under other flags (visibility, experiments) are stale
        objs = []
        base, own = VARIANT_BASE.get(tag, (None, ()))
        for src in sources():
            if base is not None and os.path.basename(src) not in own:
                objs.append(os.path.join(_variant_paths(base)[1], os.path.basename(src)[:-4] + '.o'))       # compiled by the base variant's jobs
                continue
            obj = os.path.join(odir, os.path.basename(src)[:-4] + '.o')
            objs.append(obj)
            if stale or not os.path.exists(obj) or os.path.getmtime(obj) <= max(os.path.getmtime(src), newest_header):
                jobs.append((os.path.getsize(src), [hipcc, f'--offload-arch={ARCH}'] + BASE_FLAGS + EXTRA_FLAGS + extra + ['-c', src, '-o', obj]))
        links.append((lib, objs, stamp, text))

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    jobs.sort(key=lambda j: -j[0])                              # the long compilations first
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs) or 1, os.cpu_count() or 1))) as pool:
        list(pool.map(run, [j[1] for j in jobs]))
    for lib, objs, stamp, text in links:
        run([hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-fvisibility=hidden', '-o', lib] + objs)
        with open(stamp, 'w') as fh:
            fh.write(text)
    return [variant_lib(t) for t in tags]


def build_lib(force=False, verbose=True):
    return build_libs(('',), force=force, verbose=verbose)[0]


if __name__ == '__main__':
    build_libs(('',) + (tuple(VARIANTS) if '--variants' in sys.argv else ()), force='--force' in sys.argv)
