"""Diagnostic: engine.fuse_norm16 'auto' / True against False on ONE ImageNet-64 fp16 net at the benchmark batch (same inputs, per-sample sigma):
equal bits expected wherever the tiles agree; reports which images / how many elements differ, per repetition."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd.engine import EDMDenoiser
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
net = EDMDenoiser.from_config('imagenet64', seed=62, use_fp16=True)
g = torch.Generator().manual_seed(3)
x = (torch.randn(B, 3, 64, 64, generator=g) * 0.7).cuda()
sig = torch.full((B,), 0.7); sig[::7] = 1.3
lab = torch.eye(1000)[torch.randint(1000, (B,), generator=g)].cuda()
outs = {}
for mode in (False, 'auto', True, False, 'auto'):
    net.engine.fuse_norm16 = mode
    o = net(x, sig.cuda(), class_labels=lab).clone()
    torch.cuda.synchronize()
    key = str(mode)
    if key in outs:
        print(key, 'repeat equal:', torch.equal(o, outs[key]))
    outs.setdefault(key, o)
base = outs['False']
for k, o in outs.items():
    d = (o - base).abs()
    per = d.flatten(1).max(1).values
    bad = (per > 0).nonzero().flatten().tolist()
    print(k, 'finite', bool(torch.isfinite(o).all()), 'max|d| vs pass plan', float(d.max()), 'images differing', len(bad), bad[:16], 'elements', int((d > 0).sum()))
