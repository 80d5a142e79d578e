"""Time the GAT road encoder alone at BJ scale (kernel trace with rocprofv3 --kernel-trace --stats -- python3 ablate/gat_time.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
net, _ = bench.build_network('bj')
model, cfg = bench.build_model_cpu(net)
model = bench.to_device(model, cfg, 'cuda:0')
fh = model.function_h
x = fh.input_features()
apt = fh.Apt.to(x.device)
for reps in (1, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fh.gat_encoder((x, apt))
    ev[1].record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('reps %d: %.3f ms per call (events), host issue %.3f ms per call' % (reps, ev[0].elapsed_time(ev[1]) / reps, (t1 - t0) * 1e3 / reps))
