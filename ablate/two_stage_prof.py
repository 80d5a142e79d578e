"""Where bench.two_stage_record spends its time (cProfile of the timed call)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ts_trajgen_amd import synth
net = synth.porto_scale_network()
od = synth.make_od_pairs(net, 16384, seed=7, workers=16)
torch.cuda.set_device(0)
pr = cProfile.Profile()
pr.enable()
r = bench.two_stage_record(net, od, 'cuda:0', 5000)
pr.disable()
print(r)
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
