"""Diagnostic: where does the batched Philox generator differ from torch's per-seed generators on this GPU?"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd.sample import StackedRandomGenerator  # noqa: E402

dev = torch.device('cuda')
rep = {}
for shape in [(5,), (1024,), (3, 32, 32), (4, 64, 64)]:
    seeds = [0, 1, 12345]
    ours = StackedRandomGenerator(dev, seeds).randn([len(seeds), *shape], device=dev).reshape(len(seeds), -1).cpu()
    ref = torch.stack([torch.randn(shape, generator=torch.Generator(dev).manual_seed(s), device=dev) for s in seeds]).reshape(len(seeds), -1).cpu()
    ne = (ours != ref)
    ob, rb = ours.view(torch.int32), ref.view(torch.int32)
    ulp = (ob.long() - rb.long()).abs()
    idx = ne.nonzero()[:6].tolist()
    # is ref a permutation / shift of ours?  look up ref[0, 0..3] in ours[0]
    where = [(ours[0] == ref[0, k]).nonzero().flatten().tolist()[:3] for k in range(min(4, ref.shape[1]))]
    rep[str(shape)] = dict(n=int(ref.shape[1]), mismatches=int(ne.sum()), max_ulp=int(ulp.max()), first=[(i, float(ours[i[0], i[1]]), float(ref[i[0], i[1]]), int(ulp[i[0], i[1]])) for i in idx],
                           ours_head=ours[0, :6].tolist(), ref_head=ref[0, :6].tolist(), ref_found_in_ours_at=where)
prop = torch.cuda.get_device_properties(dev)
rep['device'] = dict(cus=prop.multi_processor_count, max_threads=prop.max_threads_per_multi_processor, name=prop.name)
g = torch.Generator(dev).manual_seed(0)
torch.randn(3072, generator=g, device=dev)
rep['offset_after_3072'] = int(g.get_offset())
g = torch.Generator(dev).manual_seed(0)
torch.randint(1000, size=[], generator=g, device=dev)
rep['offset_after_randint'] = int(g.get_offset())
import ctypes as C
from diff_sampler_amd import _lib
lib = _lib.load()
n = 4096
ref = torch.randn(n, generator=torch.Generator(dev).manual_seed(0), device=dev)
out = torch.empty(n, device=dev)
match = {}
for variant in range(96):
    lib.ds_philox_probe(0, 0, C.c_void_p(out.data_ptr()), n, variant, _lib.stream_ptr())
    torch.cuda.synchronize()
    mism = int((out != ref).sum())
    match[variant] = mism
rep['probe_mismatches_by_variant'] = match
rep['probe_best'] = sorted(match.items(), key=lambda kv: kv[1])[:6]
print(json.dumps(rep, indent=1))
