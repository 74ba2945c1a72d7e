"""Golden of the latent-diffusion oracle in its reduced-precision variant (oracle.ldm_net.operands_f16) at full SD-1.5 size.

    python oracle/gen_f16_golden.py        # writes tests/golden/ldm_sd15_f16ops.npz  (about 3 minutes of CPU)
    python oracle/gen_f16_golden.py --traj # writes tests/golden/ldm_sd15_traj_b2_f16ops.npz (round 5; about 15 minutes of CPU): the fp16-stream
                                           # oracle's whole config-5 TRAJECTORY for the two latents of ldm_sd15_traj_b2.npz (which the REAL
                                           # reference produced, oracle/gen_golden.py --part full5b), with the layer lists of the plan at the
                                           # BENCHMARK batch (16 latents = 32 U-Net images, bench.py --config sd15 --dtype fp16) and the oracle's own
                                           # per-step distance from the real reference's fp32 trajectory (the noise floor of the mode)

Inputs = those of tests/golden/ldm_sd15.npz (which the REAL reference produced, oracle/gen_golden.py --part ldm); the set of layers
whose multiplicands are rounded to fp16 (and of layers whose output is stored in fp16) is read off the product's own launch plan (built on the CPU: no kernel runs) through
tests/_f16_names.ldm_prefixes / ldm_stored_prefixes and stored in the file, so that the GPU test can assert the engine still routes exactly those layers to
the fp16-operand kernels before it compares numbers.  The fp32 mode of the same oracle is pinned to the real reference by
tests/test_oracle_golden.py; this variant differs from it only by the rounding hook."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd.ldm_engine import LDMUNetEngine
    from oracle import ldm_net
    import _f16_names
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'ldm_sd15.npz'))
    kw = dict(la.NAMED_LDM_CONFIGS['sd15'])
    spec = la.ldm_unet_spec(**kw)
    params = la.init_ldm_params(spec, seed=int(z['seed']))
    eng = LDMUNetEngine(spec, params, device='cpu', use_fp16=True)
    plan = eng.plan(2, 2, 77)                       # one latent, CFG-doubled, per-sample sigma rows (the golden passes sigma as a vector)
    pre = sorted(_f16_names.ldm_prefixes(plan))
    sto = sorted(_f16_names.ldm_stored_prefixes(plan))       # ... and the layers whose output is stored as an fp16 tensor (rounded there)
    net = ldm_net.OracleCFG(params, kw, la.alphas_cumprod(spec), guidance_rate=7.5)
    x, cond, uncond, sigma = (torch.from_numpy(z[k]) for k in ('x', 'cond', 'uncond', 'sigma'))
    s, st = set(pre), set(sto)
    with torch.no_grad(), ldm_net.operands_f16(lambda name: name in s, stored=lambda name: name in st):
        out = net(x, sigma, condition=cond, unconditional_condition=uncond)
        eps = torch.cat(net.last_eps)                # [2, 4, 64, 64]: unconditional, conditional noise predictions of that evaluation
    with torch.no_grad():
        out32 = net(x, sigma, condition=cond, unconditional_condition=uncond)
    ref = torch.from_numpy(z['out_vec'])
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    print('f16-operand oracle vs real-reference fp32 golden:', rel(out, ref), ' fp32 oracle vs golden:', rel(out32, ref))
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'ldm_sd15_f16ops.npz'), out_f16ops=out.numpy(), eps_f16ops=eps.numpy(), f16_layers=np.array(pre), f16_stored=np.array(sto),
                        rel_vs_fp32_golden=rel(out, ref))


def traj():
    import diff_sampler_amd.ldm_arch as la
    from diff_sampler_amd.ldm_engine import LDMUNetEngine
    from oracle import ldm_net, solvers_ref
    import _f16_names
    from _parity import per_step_rel
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'ldm_sd15_traj_b2.npz'))
    kw = dict(la.NAMED_LDM_CONFIGS['sd15'])
    spec = la.ldm_unet_spec(**kw)
    params = la.init_ldm_params(spec, seed=int(z['seed']))
    eng = LDMUNetEngine(spec, params, device='cpu', use_fp16=True)
    lists = {}
    for n in (32, 4):                                 # the benchmark batch (16 latents under guidance), and this golden's own two latents
        plan = eng.plan(n, 1, 77)                     # the sampler passes sigma as a scalar: one embedding row
        lists[n] = (sorted(_f16_names.ldm_prefixes(plan)), sorted(_f16_names.ldm_stored_prefixes(plan)))
    assert lists[32] == lists[4], 'the fp16 layer set must not depend on the batch between 4 and 32 U-Net images'
    pre, sto = lists[32]
    net = ldm_net.OracleCFG(params, kw, la.alphas_cumprod(spec), guidance_rate=7.5)
    lat, cond, uncond = (torch.from_numpy(z[k]) for k in ('latents', 'cond', 'uncond'))
    t_min, t_max = net.sigma_inv(torch.tensor(net.sigma_min)), net.sigma_inv(torch.tensor(net.sigma_max))
    sched = net.sigma(t_max + torch.arange(6) / 5 * (t_min - t_max))          # 'discrete', rho = 1 (solver_utils.py:49-53)
    s, st = set(pre), set(sto)
    with torch.no_grad(), ldm_net.operands_f16(lambda name: name in s, stored=lambda name: name in st):
        # the FIRST evaluation of the trajectory on its own: the raw U-Net outputs (unconditional, conditional) before the guidance
        # combination amplifies their differences -- the sharp per-evaluation pin at the benchmark batch
        out0 = net(lat * sched[0], sched[0], condition=cond, unconditional_condition=uncond)
        eps0 = torch.cat(net.last_eps)                # [4, 4, 64, 64]: two unconditional, then two conditional noise predictions
        tr = solvers_ref.sample('dpm_pp', net, lat, sched, condition=cond, unconditional_condition=uncond, want_inters=True,
                                max_order=2, predict_x0=False, lower_order_final=True, num_steps=6)
    gold = torch.from_numpy(z['traj'])
    noise = per_step_rel(tr, gold)
    print('fp16-stream oracle vs real-reference fp32 trajectory, per step:', noise, flush=True)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'ldm_sd15_traj_b2_f16ops.npz'), traj_f16ops=tr.numpy(), eps0_f16ops=eps0.numpy(), out0_f16ops=out0.numpy(), f16_layers=np.array(pre),
                        f16_stored=np.array(sto), per_step_rel_vs_fp32_golden=np.array(noise), plan_images=np.array([32, 4]))


if __name__ == '__main__':
    traj() if '--traj' in sys.argv else main()
