"""Error measures of the full-size parity tests, and the log of what they observed.

A sampler trajectory starts at sigma_max = 80 (values of scale ~300) and ends on an image of scale ~3: dividing |difference| by the maximum
over the WHOLE trajectory leaves the final image unconstrained (0.16 absolute on a scale-3 image at 5e-4).  `per_step_rel` therefore
normalises every step by that step's own golden maximum; the tests bound every step and, explicitly, the final image.

Every GPU parity test records what it measured through `record`; the file lands in gpurun_out/ (merged back by gpurun) and the kept copy is
profiles/r6_parity.json (round 5: r5_parity.json)."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, 'gpurun_out', 'r6_parity.json')


def rel(a, b):
    """max |a - b| over max |b|."""
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def per_step_rel(traj, gold):
    """[steps] list: max |traj[i] - gold[i]| / max |gold[i]| -- each step against its own scale."""
    assert tuple(traj.shape) == tuple(gold.shape), (traj.shape, gold.shape)
    return [rel(traj[i], gold[i]) for i in range(gold.shape[0])]


def step_scales(gold):
    return [float(torch.as_tensor(gold[i]).abs().max()) for i in range(gold.shape[0])]


def record(key, **values):
    """Merge {key: values} into gpurun_out/r6_parity.json (best effort: a read-only tree must not fail a parity test)."""
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        data = {}
        if os.path.exists(LOG):
            with open(LOG) as f:
                data = json.load(f)
        data.setdefault(key, {}).update(values)
        with open(LOG, 'w') as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass
