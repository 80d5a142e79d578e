"""BASELINE.json configs[3] timing: two-stage generation at Porto scale (10 x 10 regions), N OD pairs on one GPU.
usage: python tools/two_stage_timing.py [n_od] [max_step]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ts_trajgen_amd import synth  # noqa: E402

n_od = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
max_step = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
net = synth.porto_scale_network()
od = synth.make_od_pairs(net, n_od, seed=7, workers=16)
reg = synth.region_tables(net, tiles=10)

import torch  # noqa: E402
from tests.test_gpu_configs import _region_model  # noqa: E402
from tests.test_gpu_fullsize import _model  # noqa: E402
from ts_trajgen_amd.search import DoubleLayerSearcher  # noqa: E402

road_model, cfg = _model(net)
region_model, rcfg, rdf = _region_model(net, road_model, reg)
s = DoubleLayerSearcher('cuda:0', net.adjacent_list, net.rid_gps, net.road_length, reg['region_adjacent_list'],
                        reg['region_dist'], reg['region_transfer_freq'], reg['rid2region'], reg['region2rid'],
                        net.road_time_distribution, reg['region_time_distribution'])
w = slice(0, 256)
s.generate(region_model, road_model, od['start'][w], od['start_time'][w], od['des'][w], od['default_len'][w],
           max_step=max_step, seeds=list(range(256)))
torch.cuda.synchronize()
t0 = time.time()
locs, tims, flags = s.generate(region_model, road_model, od['start'], od['start_time'], od['des'], od['default_len'],
                               max_step=max_step, seeds=list(range(n_od)))
torch.cuda.synchronize()
dt = time.time() - t0
reached = sum(int(l[-1] == d) for l, d in zip(locs, od['des']))
print('two-stage: %d OD pairs in %.2f s = %.0f trajectories/s; reached %d (%.1f%%); rounds %d; road expansions %d; '
      'mean length %.1f; inside the engine calls %.0f%% of the wall time'
      % (n_od, dt, n_od / dt, reached, 100.0 * reached / n_od, s.last_stats['rounds'],
         s.last_stats['road_expansions'], np.mean([len(l) for l in locs]), 100.0 * s.last_stats['gpu_busy_frac']))
