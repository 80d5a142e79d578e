"""Probe: 1 KB row gathers from pools of growing size (does address-translation reach limit random gathers?)."""
import torch, time
dev = 'cuda:0'
rows_out = 1 << 21   # 2M rows x 1 KB = 2 GB gathered
out = torch.empty((rows_out, 256), dtype=torch.float32, device=dev)
for gb in (1, 4, 16, 64, 128, 240):
    n = gb * (1 << 30) // 1024
    try:
        x = torch.empty((n, 256), dtype=torch.float32, device=dev)
    except RuntimeError as e:
        print(gb, 'alloc failed'); continue
    res = []
    for mode in ('random', 'page64'):
        if mode == 'random':
            idx = torch.randint(0, n, (rows_out,), device=dev)
        else:   # random 64-row pages, rows within a page consecutive (64 KB runs)
            pg = torch.randint(0, n // 64, (rows_out // 64,), device=dev)
            idx = (pg.unsqueeze(1) * 64 + torch.arange(64, device=dev).unsqueeze(0)).reshape(-1)
        torch.index_select(x, 0, idx, out=out)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            torch.index_select(x, 0, idx, out=out)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 5
        res.append('%s %.2f TB/s read' % (mode, rows_out * 1024 / ms / 1e9))
    # dependent pointer chase, one lane: latency
    print('pool %3d GB: %s' % (gb, ', '.join(res)), flush=True)
    del x
    torch.cuda.empty_cache()
