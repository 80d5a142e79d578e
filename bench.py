"""Benchmark of the hot path: batched road-level A* trajectory generation on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the whole path (frontier -> LSTM cell on MFMA -> candidate scoring, repeated
until every search ends) over one batch of synthetic OD pairs per GPU, inputs resident in HBM.
Workload at N = 1: BASELINE.json configs[1] (Porto-scale synthetic road graph, N = 11 095 segments,
E = 25 000 turn edges, 64k OD pairs, BJ model dims, random-init weights, max_step 5000).  For N > 1
every rank gets its own 64k OD pairs (weak scaling; OD pairs shard with no data-path collective) and one
RCCL all-gather collects the padded trajectories inside the timed region.
Prints ONE JSON line on rank 0 (metric: generated trajectories/sec) with `roofline` for the dominant
kernel (durations from HIP events around every launch of sampled supersteps, on the launch stream) and
`cpu_baseline` (the CPU oracle, i.e. a port of the reference algorithm, on a bounded OD sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

BJ_DIMS = {   # our_model_generate.py:41-59
    'function_g': {'road_emb_size': 256, 'time_emb_size': 50, 'hidden_size': 256, 'dropout_p': 0.6,
                   'lstm_layer_num': 2, 'pretrain_road_rep': None, 'dis_weight': 0.5, 'device': None},
    'function_h': {'embed_dim': 256, 'gps_emb_dim': 10, 'num_of_heads': 5, 'concat': False, 'device': None,
                   'distance_mode': 'l2'},
    'dis_weight': 0.45,
}
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense f16/bf16 matrix peak (MI355X_MICROARCH.md, matrix cores table)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--config', default='porto', choices=['grid', 'porto', 'bj', 'stress'])
    ap.add_argument('--ods', type=int, default=65536, help='OD pairs per GPU per step')
    ap.add_argument('--warmup-ods', type=int, default=4096, help='OD pairs per GPU in a warm-up step')
    ap.add_argument('--max-step', type=int, default=5000)
    ap.add_argument('--slots', type=int, default=0, help='searches in flight per GPU (0 = auto)')
    ap.add_argument('--lmax', type=int, default=256)
    ap.add_argument('--oversub', type=float, default=2.2,
                    help='searches in flight per worst-case reservation of the state-page pool (1 = reserve the worst case)')
    ap.add_argument('--ramp', type=int, default=1500, help='supersteps over which the first searches are started')
    ap.add_argument('--profile-every', type=int, default=8, help='event-bracket every k-th chunk of supersteps')
    ap.add_argument('--lstm', default='split', choices=['split', 'fp32'],
                    help='recurrent matmul: split-f16 three-pass matrix-core form (fp32-grade products) or plain fp32 MFMA')
    ap.add_argument('--cpu-seconds', type=float, default=20.0, help='CPU baseline budget (0 = skip)')
    return ap.parse_args()


def build_network(name):
    from ts_trajgen_amd import synth
    if name == 'grid':
        return synth.street_grid_network(32), 'synthetic 32x32 street grid, N=1984'
    if name == 'porto':
        return synth.porto_scale_network(), 'Porto-scale synthetic road graph N=11095 E=25000'
    if name == 'stress':
        return synth.stress_network(), 'stress synthetic city N=500000 E=1250000, GAT hidden 128, max trajectory 256'
    return synth.bj_scale_network(), 'BJ-scale synthetic road graph N=40306 E=100000'


def build_model(net, device, gat_dim=None):
    import copy
    import torch
    from ts_trajgen_amd.generator import GeneratorV4
    cfg = copy.deepcopy(BJ_DIMS)
    if gat_dim:
        cfg['function_h']['embed_dim'] = gat_dim
    cfg['function_g']['device'] = device
    cfg['function_h']['device'] = device
    torch.manual_seed(0)
    model = GeneratorV4(config=cfg, data_feature=net.data_feature(torch))
    model = model.to(device)
    model.function_h.node_features = model.function_h.node_features.to(device)
    model.train(False)
    return model, cfg


def pmc_traffic(kernel, lstm, rows_per_launch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json, written by
    tools/gpu_profile.sh + tools/pmc_summary.py: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
    same bench at 8192 OD pairs, corrected as MI355X_MICROARCH.md prescribes).  Counters cannot be read from
    inside the timed run: the figure of that pass is scaled from its work items per launch to this run's (every
    kernel's traffic is per work item); None when absent."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if not os.path.exists(path):
        return None, None
    with open(path) as f:
        js = json.load(f)
    names = {'frontier_kernel': 'frontier_kernel', 'score_frontier_kernel': 'score_frontier_kernel<4, true>',
             'lstm_layer_kernel[l0]': 'lstm_layer0_split_cand_kernel' if lstm == 'split' else 'lstm_layer_kernel<true>',
             'lstm_layer_kernel[l1]': 'lstm_layer_split_kernel<false>' if lstm == 'split' else 'lstm_layer_kernel<false>'}
    e = js.get(names.get(kernel, kernel))
    if not e:
        return None, None
    pass_rows = js.get('_work_items_per_launch')
    scale = rows_per_launch / pass_rows if pass_rows else 1.0
    note = ', scaled x%.2f from %.0f work items per launch in the PMC pass' % (scale, pass_rows) if pass_rows else ''
    if 'hbm_bytes_corrected' in e:
        return e['hbm_bytes_corrected'] * scale, ('profiles/pmc_traffic.json: 2*FETCH_SIZE + WRITE_SIZE per launch '
                                                  '(gfx950 correction)' + note)
    return e['hbm_bytes_raw'] * scale, ('profiles/pmc_traffic.json: FETCH_SIZE + WRITE_SIZE per launch, raw (narrow '
                                        'accesses: uncalibrated)' + note)


def gat_encoder_timing(model, net, cfg, reps=5):
    """One-off kernel group (1): the GAT road-graph encoder.  HIP-event time of the whole tsg_gat_forward
    (projection GEMM, skip GEMM, node scores, edge max, aggregation) on the launch stream, against its
    algorithmic HBM bytes: 4N*FIN + 4N*NH*F + E(4NH*F + 8NH + 8) + 4N*F."""
    import torch
    fh = model.function_h
    x = fh.input_features()
    apt = fh.Apt.to(x.device)
    fh.gat_encoder((x, apt))
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fh.gat_encoder((x, apt))
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    n, fin = x.shape
    nh, f = cfg['function_h']['num_of_heads'], cfg['function_h']['embed_dim']
    e = int(apt.shape[1])
    byts = 4.0 * n * fin + 4.0 * n * nh * f + e * (4.0 * nh * f + 8 * nh + 8) + 4.0 * n * f
    return {'ms': ms, 'nodes': int(n), 'edges': e, 'algorithmic_bytes': byts, 'bound': 'hbm',
            'achieved': byts / (ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
            'frac': byts / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}


def measured_peaks(device):
    """what this very box reaches on a plain copy and on a dense f16 matrix product (torch kernels:
    a device-to-device copy of 1 GiB, an 8192^3 f16 matmul through hipBLASLt), next to the vendor peaks."""
    import torch
    out = {}
    try:
        n = 1 << 28
        src = torch.empty(n, dtype=torch.float32, device=device).fill_(1.0)
        dst = torch.empty_like(src)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        dst.copy_(src)
        ev[0].record()
        for _ in range(10):
            dst.copy_(src)
        ev[1].record()
        torch.cuda.synchronize()
        out['hbm_copy_GBs'] = 10 * 2.0 * n * 4 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9
        del src, dst
        a = torch.randn((8192, 8192), dtype=torch.float16, device=device)
        b = torch.randn((8192, 8192), dtype=torch.float16, device=device)
        c = a @ b
        ev[0].record()
        for _ in range(10):
            c = a @ b
        ev[1].record()
        torch.cuda.synchronize()
        out['f16_gemm_TFLOPs'] = 10 * 2.0 * 8192 ** 3 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e12
        del a, b, c
    except RuntimeError as e:      # e.g. no room left beside the page pool: the vendor peaks stand alone
        out['error'] = str(e)[:120]
    return out


def cpu_baseline(net, model, cfg, od, budget_s, max_step):
    """The CPU oracle (port of search.py:240-424 + generator_v4.py:56-72) on the first ODs of the workload."""
    import torch
    from oracle import search as osr
    from oracle.model import OracleGenerator
    g_state = {k: v.detach().cpu() for k, v in model.function_g.state_dict().items()}
    h_state = {k: v.detach().cpu() for k, v in model.function_h.state_dict().items()}
    ei = np.stack([net.adj_mx.row, net.adj_mx.col]).astype(np.int64)
    om = OracleGenerator(g_state, h_state, float(model.w1.detach().cpu()[0]), net.node_features, ei,
                         cfg['function_h']['num_of_heads'], cfg['function_h']['embed_dim'],
                         cfg['function_g']['lstm_layer_num'])
    om.setup_node_emb()
    rtd = net.road_time_distribution.astype(int)
    t0 = time.time()
    done = 0
    steps = 0
    while done < len(od['des']) and (time.time() - t0 < budget_s or done < 1):
        st = osr.SearchStats()
        osr.astar_search(om, net.adjacent_list, net.rid_gps, net.road_length, rtd, [int(od['start'][done])],
                         [int(od['start_time'][done])], int(od['des'][done]), int(od['default_len'][done]),
                         max_step, stats=st)
        steps += st.steps
        done += 1
    dt = time.time() - t0
    return {'value': done / dt, 'unit': 'trajectories/s', 'cores': int(torch.get_num_threads()), 'kind': 'port',
            'sample': 'first %d OD pairs of the same workload (%d expansions) in %.1f s, torch-CPU fp32'
                      % (done, steps, dt),
            'expansions_per_s': steps / dt}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    from ts_trajgen_amd import synth
    net, net_name = build_network(args.config)
    # OD pairs for all ranks (seeded), sharded round-robin by index; BFS default_len before the GPU is touched
    n_total = args.ods * world
    od = synth.make_od_pairs(net, n_total, seed=2, workers=max(1, min(16, (os.cpu_count() or 1) // world)),
                             default_len='lattice' if args.config == 'stress' else 'hops', shard=(rank, world))
    od_warm = {k: v[:args.warmup_ods] for k, v in od.items()}

    import torch
    import torch.distributed as dist
    device = 'cuda:%d' % local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device(device))
    from ts_trajgen_amd.search import Searcher
    model, cfg = build_model(net, device, gat_dim=128 if args.config == 'stress' else None)
    searcher = Searcher(device, net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution)
    searcher.state_format = 1 if args.lstm == 'split' else 0
    tables = model.tables()
    # searches in flight and the state-page pool they share: HBM holds `slots` fixed workspaces plus a pool that
    # would cover the worst case of slots / oversub searches (a search holds only the pages it has filled)
    from ts_trajgen_amd.engine import AStarEngine
    fixed = AStarEngine.fixed_bytes_per_slot(searcher.graph(), args.max_step, 1)
    page = AStarEngine.bytes_per_page(tables)
    pps = (1 + args.max_step + 63) // 64
    free_b, _total = torch.cuda.mem_get_info(torch.device(device))
    budget = int(free_b * 0.93) - (2 * args.lmax + 8) * 4 * args.ods
    slots = args.slots
    if slots <= 0:
        slots = int(budget // (fixed + pps * page / args.oversub))
        slots = min(slots, 32768, args.ods)
        # LSTM grid: 8 workgroups per 128 searches, 768 resident at a time -> whole waves of workgroups at 12 288
        slots = slots // 12288 * 12288 if slots >= 12288 else max(128, slots // 1024 * 1024 if slots >= 1024 else slots // 128 * 128)
    pool_pages = min(slots * pps, int((budget - slots * fixed) // page))
    ramp = args.ramp if pool_pages < slots * pps else 0

    def one_step(o, profile_every=0):
        res = searcher.generate(model, o['start'], o['start_time'], o['des'], o['default_len'],
                                max_step=args.max_step, slots=slots, lmax=args.lmax, return_device=True,
                                profile_every=profile_every, pool_pages=pool_pages, ramp_steps=ramp)
        if world > 1:   # the single exchange step of the path: collect the padded trajectories (RCCL over xGMI)
            packed = torch.cat([res['rid'], res['time'], res['len'].unsqueeze(1), res['status'].unsqueeze(1)], dim=1)
            gathered = torch.empty((world * packed.shape[0], packed.shape[1]), dtype=packed.dtype,
                                   device=packed.device)
            dist.all_gather_into_tensor(gathered, packed)
            res['gathered'] = gathered
        return res

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(od_warm)
    sync()
    t0 = time.time()
    stats = []
    for _ in range(args.steps):
        res = one_step(od, profile_every=args.profile_every)
        stats.append(dict(searcher.last_stats))
    sync()
    dt = time.time() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_done = args.steps * args.ods * world
    status = res['status'].cpu().numpy()
    exp_total = sum(s['expansions'] for s in stats)
    if world > 1:
        t = torch.tensor([exp_total], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        exp_all = float(t.item())
    else:
        exp_all = float(exp_total)

    if rank == 0:
        hp, nl = tables.hp, tables.n_layers
        ms = np.sum([s['prof_ms'] for s in stats], axis=0)
        launches = np.sum([s['prof_launches'] for s in stats], axis=0)
        prof_items = sum(s['prof_items'] for s in stats)
        evals = sum(s['model_evals'] for s in stats)
        chain_sum = sum(s['chain_sum'] for s in stats)
        cand_sum = sum(s['cand_sum'] for s in stats)
        items = sum(s['items'] for s in stats)
        names = ['frontier_kernel', 'score_frontier_kernel'] + ['lstm_layer_kernel[l%d]' % l for l in range(nl)]
        kern = {}
        rows_per_launch = prof_items / max(1, launches[0])
        avg_L = chain_sum / max(1, evals)
        avg_C = cand_sum / max(1, evals)
        scored_frac = evals / max(1, items)
        for i, nm in enumerate(names):
            if launches[i] == 0:
                continue
            avg_ms = ms[i] / launches[i]
            k = {'avg_ms': avg_ms, 'launches': int(launches[i]), 'share': float(ms[i] / ms[:2 + nl].sum())}
            if nm.startswith('lstm'):
                l = i - 2
                K = hp if l == 0 else 2 * hp
                flops = rows_per_launch * 2.0 * 4 * hp * K
                # split form: three f16 matrix-core passes per algorithmic product -> a third of the dense f16 peak
                peak = PEAK_F16_MFMA_TFLOPS / 3.0 if args.lstm == 'split' else PEAK_F32_MFMA_TFLOPS
                k.update({'bound': 'mfma', 'achieved': flops / (avg_ms * 1e-3) / 1e12, 'peak': peak,
                          'unit': 'TFLOP/s'})
            elif nm.startswith('score'):
                # own + ancestor top-layer h rows, gathered out_fc rows (+bias), factored H tables,
                # lon/lat, CSR row, item record, cand_lp write
                per = 4 * hp * (1 + avg_L) + avg_C * (4 * (hp + 1) + 80 + 16 + 8) + 40 + 16 + 32
                byts = rows_per_launch * scored_frac * per
                k.update({'bound': 'hbm', 'achieved': byts / (avg_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                          'unit': 'GB/s'})
            else:
                per = 40 * avg_C + 32 + 32 + 128      # SURVEY 8(d) frontier bytes + item + slot state
                byts = rows_per_launch * per
                k.update({'bound': 'hbm', 'achieved': byts / (avg_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                          'unit': 'GB/s'})
            k['frac'] = k['achieved'] / k['peak']
            kern[nm] = k
        dom = max(kern, key=lambda n: kern[n]['share']) if kern else None
        roofline = None
        if dom:
            d = kern[dom]
            traffic, traffic_src = pmc_traffic(dom, args.lstm, rows_per_launch)
            roofline = {'kernel': dom, 'bound': d['bound'], 'achieved': d['achieved'], 'peak': d['peak'],
                        'unit': d['unit'], 'frac': d['frac'], 'traffic': traffic, 'traffic_source': traffic_src,
                        'avg_launch_ms': d['avg_ms'], 'rows_per_launch': rows_per_launch}
        gat = gat_encoder_timing(model, net, cfg)
        peaks = measured_peaks(device)
        for k in list(kern.values()) + ([roofline] if roofline else []):
            ref = peaks.get('hbm_copy_GBs') if k['bound'] == 'hbm' else (
                peaks['f16_gemm_TFLOPs'] / 3.0 if args.lstm == 'split' and 'f16_gemm_TFLOPs' in peaks else None)
            k['measured_peak'] = ref      # this box: device-to-device copy / a third of an 8192^3 f16 matmul
            k['frac_of_measured'] = k['achieved'] / ref if ref else None
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            cpu = cpu_baseline(net, model, cfg, od, args.cpu_seconds, args.max_step)
        out = {
            'metric': 'generated trajectories/sec (road-level A*)', 'value': n_done / dt, 'unit': 'trajectories/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt * 1e3 / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s, %d OD pairs per GPU, BJ model dims (hidden 256, 2-layer LSTM, GAT 5x256), '
                                   'random-init weights, max_step %d' % (net_name, args.ods, args.max_step),
                       'od_pairs_per_gpu': args.ods, 'slots_in_flight': slots, 'state_pool_pages': pool_pages,
                       'state_pool_gb': pool_pages * page / 1e9, 'lstm': args.lstm, 'parallelism': 'od-shard x%d' % world},
            'expansions_per_s': exp_all / dt, 'expansions_per_od': exp_all / n_done,
            'reached_frac': float((status & 0xff == 1).mean()), 'fallback_frac': float((status & 0xff == 2).mean()),
            'avg_prefix_len_scored': avg_L, 'avg_candidates_scored': avg_C,
            'supersteps': int(sum(s['supersteps'] for s in stats)), 'work_items': int(items),
            'page_waits': int(sum(s['page_waits'] for s in stats)), 'restarts': int(sum(s['restarts'] for s in stats)),
            'min_free_pages': int(min(s['min_free_pages'] for s in stats)),
            'roofline': roofline, 'kernels': kern, 'measured_peaks': peaks, 'gat_encoder': gat, 'cpu_baseline': cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
