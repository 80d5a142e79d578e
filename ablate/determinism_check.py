"""Do repeated runs of one large batch give bit-identical results, per recurrence form, and where do the forms differ?
usage: python ablate/determinism_check.py [n_od] [runs]   (diagnostic; BJ-scale workload sized like bench.py)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench

n_od = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
from ts_trajgen_amd import synth
net, _ = bench.build_network('bj')
od = synth.make_od_pairs(net, n_od, seed=2, workers=16)
model, cfg = bench.build_model_cpu(net)
import torch
dev = 'cuda:0'
model = bench.to_device(model, cfg, dev)
from ts_trajgen_amd.engine import AStarEngine
from ts_trajgen_amd.search import Searcher
s = Searcher(dev, net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution)
tables = model.tables()
fixed = AStarEngine.fixed_bytes_per_slot(s.graph(), 5000, 1)
page = AStarEngine.bytes_per_page(tables)
pps = (1 + 5000 + 63) // 64
free_b, _t = torch.cuda.mem_get_info(torch.device(dev))
budget = int(free_b * 0.93) - (2 * 256 + 8) * 4 * n_od
slots = 24576
pool = min(slots * pps, int((budget - slots * fixed) // page))
keys = ('rid', 'time', 'len', 'status', 'steps')
out = {}
for fmt in (1, 0):
    s._drop_engines()
    s.state_format = fmt
    for r in range(runs):
        t0 = time.time()
        res = s.generate(model, od['start'], od['start_time'], od['des'], od['default_len'], max_step=5000, slots=slots,
                         lmax=256, return_device=True, pool_pages=pool, ramp_steps=1500)
        torch.cuda.synchronize()
        st = s.last_stats
        print('fmt %d run %d: %.1f s, waits %d restarts %d' % (fmt, r, time.time() - t0, st['page_waits'], st['restarts']), flush=True)
        out[(fmt, r)] = {k: res[k].clone() for k in keys}


def diff(a, b):
    same = torch.ones(n_od, dtype=torch.bool, device=a['len'].device)
    for k in ('rid', 'time'):
        same &= (a[k] == b[k]).all(dim=1)
    for k in ('len', 'status', 'steps'):
        same &= a[k] == b[k]
    return torch.nonzero(~same).flatten().cpu().tolist()


for fmt in (1, 0):
    for r in range(1, runs):
        print('fmt %d: run 0 vs run %d differ at' % (fmt, r), diff(out[(fmt, 0)], out[(fmt, r)]))
print('split vs fp32 (run 0) differ at', diff(out[(1, 0)], out[(0, 0)]))

# the OD pairs on which the two forms differ: alone (8 slots, worst-case pool), per form, against the big batch
bad = diff(out[(1, 0)], out[(0, 0)])
for fmt in (1, 0):
    s._drop_engines()
    s.state_format = fmt
    for i in bad[:4]:
        sl = slice(i, i + 1)
        r1 = s.generate(model, od['start'][sl], od['start_time'][sl], od['des'][sl], od['default_len'][sl], max_step=5000,
                        slots=8, lmax=256, return_device=True, poplog_cap=5008)
        big = out[(fmt, 0)]
        same = all(torch.equal(r1[k][0], big[k][i]) for k in keys)
        n1, nb = int(r1['len'][0]), int(big['len'][i])
        print('fmt %d OD %d alone == in the batch: %s (len %d / %d, steps %d / %d, status %d / %d)' % (
            fmt, i, same, n1, nb, int(r1['steps'][0]), int(big['steps'][i]), int(r1['status'][0]), int(big['status'][i])))
        if not same:
            print('  alone:', r1['rid'][0, :n1].tolist())
            print('  batch:', big['rid'][i, :nb].tolist())
