import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
from ts_trajgen_amd import synth
from ts_trajgen_amd.search import Searcher
from ts_trajgen_amd import search as _s
_pe = _s.plan_engine
def _dbg(*a, **k):
    r = _pe(*a, **k); print('plan', a[0] / 2**30, r, flush=True); return r
_s.plan_engine = _dbg
net, _ = bench.build_network('bj')
od = synth.make_od_pairs(net, 131072, seed=2, workers=16, default_len='hops')
model, cfg = bench.build_model_cpu(net)
torch.cuda.set_device(0)
model = bench.to_device(model, cfg, 'cuda:0')
s = Searcher('cuda:0', net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution)
w = {k: v[:40000] for k, v in od.items()}
s.generate(model, w['start'], w['start_time'], w['des'], w['default_len'], max_step=5000, return_device=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    r = s.generate(model, od['start'], od['start_time'], od['des'], od['default_len'], max_step=5000, return_device=True,
                   slots=24576, pool_pages=932893, ramp_steps=1500, streams=2)
    torch.cuda.synchronize(); dt = time.time() - t0
    print('explicit sizes: %.2f s = %.0f trajectories/s' % (dt, 131072 / dt), {k: s.last_stats[k] for k in ('engines', 'supersteps', 'page_waits', 'restarts', 'pool_pages')}, flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    r = s.generate(model, od['start'], od['start_time'], od['des'], od['default_len'], max_step=5000, return_device=True,
                   slots=24576, pool_pages=933731, ramp_steps=1500, streams=2)
    torch.cuda.synchronize(); dt = time.time() - t0
    print('explicit with the pool of the plan: %.2f s = %.0f trajectories/s' % (dt, 131072 / dt), {k: s.last_stats[k] for k in ('engines', 'supersteps', 'page_waits', 'restarts', 'pool_pages')}, flush=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    r = s.generate(model, od['start'], od['start_time'], od['des'], od['default_len'], max_step=5000, return_device=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    print('no sizes given: %.2f s = %.0f trajectories/s' % (dt, 131072 / dt), {k: s.last_stats[k] for k in ('engines', 'supersteps', 'page_waits', 'restarts', 'pool_pages')}, flush=True)
