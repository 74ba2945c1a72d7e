"""GPU: the one-launch batched latent generator is BIT-IDENTICAL to the reference's per-seed generators
(diff-solvers-main/sample.py:22-36: torch.Generator(device).manual_seed(seed % 2**32) -> randn / randint), integer claim:
torch.equal, no tolerance."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


class ReferenceStack:
    """The reference class, verbatim semantics: one torch.Generator per seed."""

    def __init__(self, device, seeds):
        self.generators = [torch.Generator(device).manual_seed(int(seed) % (1 << 32)) for seed in seeds]

    def randn(self, size, **kwargs):
        return torch.stack([torch.randn(size[1:], generator=gen, **kwargs) for gen in self.generators])

    def randint(self, *args, size, **kwargs):
        return torch.stack([torch.randint(*args, size=size[1:], generator=gen, **kwargs) for gen in self.generators])


@pytest.mark.parametrize('shape', [(3, 32, 32), (3, 64, 64), (4, 64, 64), (77, 768), (5,), (3, 16, 16)])
def test_batched_randn_is_bit_identical_to_per_seed_generators(shape):
    from diff_sampler_amd.sample import StackedRandomGenerator
    dev = torch.device('cuda')
    n_seeds = 1000 if len(shape) == 3 and shape[1] == 32 else 37
    seeds = list(range(0, n_seeds - 3)) + [2 ** 32 - 1, 2 ** 32 + 5, 123456789012]
    ours = StackedRandomGenerator(dev, seeds).randn([len(seeds), *shape], device=dev)
    ref = ReferenceStack(dev, seeds).randn([len(seeds), *shape], device=dev)
    assert ours.dtype == ref.dtype and ours.shape == ref.shape
    assert torch.equal(ours, ref)


def test_call_sequence_latents_then_labels_then_more_noise():
    """sample.py draws latents, then class labels, and (ms_coco) further noise from the SAME generators: offsets must chain."""
    from diff_sampler_amd.sample import StackedRandomGenerator
    dev = torch.device('cuda')
    seeds = [0, 1, 7, 99, 1003, 49999]
    a, b = StackedRandomGenerator(dev, seeds), ReferenceStack(dev, seeds)
    assert torch.equal(a.randn([6, 3, 64, 64], device=dev), b.randn([6, 3, 64, 64], device=dev))
    la, lb = a.randint(1000, size=[6], device=dev), b.randint(1000, size=[6], device=dev)
    assert la.dtype == lb.dtype and torch.equal(la, lb)
    assert torch.equal(a.randn([6, 77, 768], device=dev), b.randn([6, 77, 768], device=dev))
    assert torch.equal(a.randint(10, size=[6], device=dev), b.randint(10, size=[6], device=dev))
    # a call the fast path does not cover (float64) falls back to real generators fast-forwarded to the same offset
    assert torch.equal(a.randn([6, 8], device=dev, dtype=torch.float64), b.randn([6, 8], device=dev, dtype=torch.float64))
    assert torch.equal(a.randn([6, 3, 8, 8], device=dev), b.randn([6, 3, 8, 8], device=dev))


def test_large_tensor_uses_several_values_per_thread():
    """Above 256 * CUs * 8 elements ATen's kernel walks a grid-stride loop and consumes all four normals of a Philox block."""
    from diff_sampler_amd.sample import StackedRandomGenerator
    dev = torch.device('cuda')
    prop = torch.cuda.get_device_properties(dev)
    n = 256 * prop.multi_processor_count * (prop.max_threads_per_multi_processor // 256) * 5 + 123
    seeds = [3, 4]
    a, b = StackedRandomGenerator(dev, seeds), ReferenceStack(dev, seeds)
    assert torch.equal(a.randn([2, n], device=dev), b.randn([2, n], device=dev))
    assert torch.equal(a.randn([2, 100], device=dev), b.randn([2, 100], device=dev))       # offsets advanced identically


def test_fast_path_is_self_checked_and_falls_back_on_mismatch(monkeypatch):
    """First use per device compares the batched kernel with torch's own generators (offset 0 and a non-zero offset); a build where
    they differ by one ulp must fall back to real per-seed generators instead of silently changing the latents."""
    from diff_sampler_amd import ops
    from diff_sampler_amd.sample import StackedRandomGenerator
    dev = torch.device('cuda')
    StackedRandomGenerator._fast_ok.clear()
    assert StackedRandomGenerator.fast_path_verified(dev) is True            # this build: bit-identical
    seeds = [5, 6, 7]
    ref = ReferenceStack(dev, seeds).randn([3, 3, 32, 32], device=dev)
    # a "different torch build": the kernel's values are off by one ulp
    real = ops.philox_randn

    def off_by_an_ulp(seeds_, offset, out, n):
        step = real(seeds_, offset, out, n)
        out.copy_(torch.nextafter(out, torch.full_like(out, 10.0)))
        return step
    monkeypatch.setattr(ops, 'philox_randn', off_by_an_ulp)
    StackedRandomGenerator._fast_ok.clear()
    assert StackedRandomGenerator.fast_path_verified(dev) is False
    got = StackedRandomGenerator(dev, seeds).randn([3, 3, 32, 32], device=dev)  # per-seed generators: still the reference's numbers
    assert torch.equal(got, ref)
    monkeypatch.undo()
    StackedRandomGenerator._fast_ok.clear()
    assert StackedRandomGenerator.fast_path_verified(dev) is True
