#!/usr/bin/env python3
"""Per-kernel register / scratch use of a csrc translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kernel_resources.py gemm_f16dma conv3x3_f16dma [--spills-only]
Used when an epilogue or a pipeline changes: a kernel that keeps a slice of its argument or a live range in private memory shows up here
(ScratchSize > 0) long before it shows up as a slow layer (docs/HISTORY.md section E.13)."""
import os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'diff_sampler_amd', 'csrc')


def resources(tu):
    with tempfile.TemporaryDirectory() as d:
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(HERE, '..', 'include'), '-I' + CSRC, '-c',
               '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-o', os.path.join(d, 'x.o'), os.path.join(CSRC, tu + '.hip')]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark: (?:\s*)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', line)
        if not m:
            continue
        k, v = m.groups()
        if k == 'Function Name':
            cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip() or v}
            out.append(cur)
        elif cur is not None:
            cur[k.split(' [')[0]] = int(v)
    return out


if __name__ == '__main__':
    spills_only = '--spills-only' in sys.argv
    for tu in [a for a in sys.argv[1:] if not a.startswith('--')]:
        for r in resources(tu):
            if spills_only and not r.get('ScratchSize'):
                continue
            name = re.sub(r'\(igemm::KParams\)|igemm::\(anonymous namespace\)::|void ', '', r['name'])
            print(f"{name:58s} VGPRs {r.get('VGPRs', 0):3d}  AGPRs {r.get('AGPRs', 0):3d}  scratch {r.get('ScratchSize', 0):4d} B/lane  spilled {r.get('VGPRs Spill', 0):3d}  "
                  f"occupancy {r.get('Occupancy', 0)}")
