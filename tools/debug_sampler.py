import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_sampler_amd import solvers
from diff_sampler_amd.engine import EDMDenoiser
z = np.load('tests/golden/sampler_tiny_song.npz')
net = EDMDenoiser.from_config('tiny_song', seed=int(z['seed']))
lat = torch.from_numpy(z['latents']).cuda()
for tag, fn, kw in [('euler', solvers.euler_sampler, {}), ('heun', solvers.heun_sampler, {}), ('ipndm4', solvers.ipndm_sampler, dict(max_order=4))]:
    ts = torch.from_numpy(z[f'{tag}_t']).cuda()
    inters = fn(net, lat, num_steps=len(ts), t_steps=ts, return_inters=True, **kw).cpu()
    gold = torch.from_numpy(z[f'{tag}_inters'])
    print(tag, [f'{float((inters[i]-gold[i]).abs().max()/gold[i].abs().max()):.2e}' for i in range(len(gold))])
# denoiser at the schedule sigmas vs oracle
import diff_sampler_amd.arch as arch
from oracle.edm_net import edm_denoise
kw = dict(arch.NAMED_CONFIGS['tiny_song']); spec = arch.edm_precond_spec(**kw); params = arch.init_params(spec, seed=int(z['seed']))
gold = torch.from_numpy(z['euler_inters']); ts = z['euler_t']
for i in range(len(ts)-1):
    x = gold[i]
    with torch.no_grad(): ref = edm_denoise(params, kw, x, torch.tensor(float(ts[i])))
    got_f = net(x.cuda(), float(ts[i])).cpu()
    got_t = net(x.cuda(), torch.tensor(float(ts[i]))).cpu()
    print(i, float(ts[i]), 'float-sigma err', float((got_f-ref).abs().max()/ref.abs().max()), 'tensor-sigma err', float((got_t-ref).abs().max()/ref.abs().max()))
