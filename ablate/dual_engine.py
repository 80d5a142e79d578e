"""Do independent engines on separate HIP streams overlap?  The BJ-scale bench workload dealt over E Searchers with
S searches in flight and 1/E of the page pool each, run from E host threads.  (Latency-bound layer 0 / frontier
launches of one engine run under the matrix-core / bandwidth-bound launches of another.)
usage: python ablate/dual_engine.py n_od engines slots_per_engine"""
import faulthandler, os, sys, threading, time
faulthandler.dump_traceback_later(170, exit=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from ts_trajgen_amd import synth
from ts_trajgen_amd.engine import AStarEngine
from ts_trajgen_amd.search import Searcher

n_od, E, SLOTS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
net, _ = bench.build_network('bj')
od = synth.make_od_pairs(net, n_od, seed=2, workers=16, default_len='hops')
model, cfg = bench.build_model_cpu(net)
dev = 'cuda:0'
torch.cuda.set_device(0)
model = bench.to_device(model, cfg, dev)
tables = model.tables()
searchers = [Searcher(dev, net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution) for _ in range(E)]
fixed = AStarEngine.fixed_bytes_per_slot(searchers[0].graph(), 5000, 1)
page = AStarEngine.bytes_per_page(tables)
pps = (1 + 5000 + 63) // 64
free_b, _t = torch.cuda.mem_get_info(torch.device(dev))
budget = int(free_b * 0.93) - (2 * 256 + 8) * 4 * n_od
pool = min(SLOTS * pps, int((budget / E - SLOTS * fixed) // page))
print('engines %d x %d searches, %d pool pages each' % (E, SLOTS, pool), flush=True)


def run(s, o, out, key):
    torch.cuda.set_device(0)
    res = s.generate(model, o['start'], o['start_time'], o['des'], o['default_len'], max_step=5000, slots=SLOTS,
                     lmax=256, return_device=True, pool_pages=pool, ramp_steps=1500 if pool < SLOTS * pps else 0)
    out[key] = dict(s.last_stats)


parts = [{k: v[e::E] for k, v in od.items()} for e in range(E)]
for rep in range(2):
    out = {}
    th = [threading.Thread(target=run, args=(searchers[e], parts[e], out, e)) for e in range(E)]
    torch.cuda.synchronize()
    t0 = time.time()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.time() - t0
    print('%.2f s = %.0f trajectories/s (supersteps %s, page waits %s)'
          % (dt, n_od / dt, [out[e]['supersteps'] for e in range(E)], [out[e]['page_waits'] for e in range(E)]), flush=True)
