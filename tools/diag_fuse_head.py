"""Diagnostic of the head-fused solver update (round 5): for a few samplers on the tiny SongUNet, is each form deterministic run to run, and
where along the trajectory do the fused and the two-launch form differ?  (It found the last-bit disagreement of c_skip / c_out at small sigma
between two kernels that inlined the same source line under different contraction choices -- docs/HISTORY.md G.7.)
    python tools/diag_fuse_head.py        # on a GPU box"""
import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from diff_sampler_amd import solvers
from diff_sampler_amd.engine import EDMDenoiser
from oracle import cases
G = '/root/repo/tests/golden'
z = np.load(os.path.join(G, 'sampler_tiny_song.npz'))
net = EDMDenoiser.from_config('tiny_song', seed=int(z['seed']))
lat = torch.from_numpy(z['latents']).cuda()
def run(fn, mode, **kw):
    solvers.FUSE_HEAD = mode
    r = getattr(solvers, fn)(net, lat, return_inters=True, **kw)
    torch.cuda.synchronize()
    return r
for tag, fn, kind, rho, n, extra in cases.SAMPLER_CASES:
    if tag not in ('euler', 'heun', 'dpm2', 'ipndm4', 'dpmpp2m_eps'):
        continue
    ts = torch.from_numpy(z[f'{tag}_t']).cuda()
    kw = dict(num_steps=n, t_steps=ts, **extra)
    a0, a1 = run(fn, False, **kw), run(fn, False, **kw)
    b0, b1 = run(fn, True, **kw), run(fn, True, **kw)
    d = (a0 - b0).abs().flatten(1).max(1).values.tolist()
    print(tag, 'unfused repeat equal:', torch.equal(a0, a1), 'fused repeat equal:', torch.equal(b0, b1), 'fused vs unfused per step max|d|:', ['%.2e' % v for v in d], flush=True)
