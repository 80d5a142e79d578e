"""Benchmark of the hot path: batched road-level A* trajectory generation on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the whole path (frontier -> LSTM cell on MFMA -> candidate scoring, repeated
until every search ends) over one batch of synthetic OD pairs per GPU, inputs resident in HBM.
Default workload: BASELINE.json configs[2], the configuration the north-star target is quoted on: the
BJ-scale synthetic road graph (N = 40 306 segments, E = 100 000 turn edges), BJ model dims, random-init
weights, max_step 5000, 1 048 576 OD pairs over 8 GPUs = 131 072 OD pairs per GPU per step, dealt
od_id % world.  Every rank runs its share (weak scaling: at N = 8 the job IS configs[2] as written, at
N = 1 it is one rank's share of it; OD pairs shard with no data-path collective) and one RCCL all-gather
collects the padded trajectories inside the timed region.  `--total-ods T` fixes the whole job instead
(strong scaling); `--config porto` is BASELINE.json configs[1] (65 536 OD pairs per GPU).
Prints ONE JSON line on rank 0 (metric: generated trajectories/sec) with `roofline` for the dominant
kernel (durations from HIP events around every launch of sampled supersteps, on the launch stream),
`cpu_baseline` (the CPU oracle, i.e. a port of the reference algorithm, on a fixed OD sample, one search per
host core, run before the GPU is touched) and `two_stage` (BASELINE.json configs[3] on the same GPU).
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
    ap.add_argument('--config', default='bj', choices=['grid', 'porto', 'bj', 'stress'])
    ap.add_argument('--ods', type=int, default=0,
                    help='OD pairs per GPU per step (0 = the config\'s own: bj 131072 = 1M / 8, others 65536)')
    ap.add_argument('--total-ods', type=int, default=0,
                    help='OD pairs of the whole job per step, dealt od_id %% world (strong scaling); overrides --ods')
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
    ap.add_argument('--cpu-seconds', type=float, default=10.0,
                    help='CPU baseline: length of a timed pass in seconds (0 = skip)')
    ap.add_argument('--cpu-passes', type=int, default=3, help='CPU baseline: timed passes over the sample (the median counts)')
    ap.add_argument('--cpu-ods', type=int, default=32, help='CPU baseline: the fixed sample = first k OD pairs')
    ap.add_argument('--engines', type=int, default=1,
                    help='independent engines side by side in the timed step (Searcher.generate streams=E); the '
                         'default line keeps one, so that its per-kernel times are those of undisturbed launches')
    ap.add_argument('--two-engines', type=int, default=1,
                    help='N = 1: also time one step of the same workload on two engines (sub-record; 0 = skip)')
    ap.add_argument('--fp32-step', type=int, default=1,
                    help='N = 1, --lstm split: also time one step of the same workload with the strict-fp32 recurrence '
                         '(state_format 0) and compare its trajectories with the split-f16 step (sub-record; 0 = skip)')
    ap.add_argument('--two-stage-ods', type=int, default=16384,
                    help='OD pairs of the two-stage sub-record (configs[3], N = 1 only; 0 = skip)')
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


def build_model_cpu(net, gat_dim=None):
    """The generator with the reference constructors' random init under torch.manual_seed(0), still on the CPU
    (nothing here touches the GPU: the CPU baseline forks its workers from this state)."""
    import copy
    import torch
    from ts_trajgen_amd.generator import GeneratorV4
    cfg = copy.deepcopy(BJ_DIMS)
    if gat_dim:
        cfg['function_h']['embed_dim'] = gat_dim
    cfg['function_g']['device'] = 'cpu'
    cfg['function_h']['device'] = 'cpu'
    torch.manual_seed(0)
    model = GeneratorV4(config=cfg, data_feature=net.data_feature(torch))
    model.train(False)
    return model, cfg


def to_device(model, cfg, device):
    cfg['function_g']['device'] = device
    cfg['function_h']['device'] = device
    model = model.to(device)
    model.function_h.device = device
    model.function_h.gat_encoder.device = device
    model.function_h.Apt = model.function_h.Apt.to(device)
    model.function_h.node_features = model.function_h.node_features.to(device)
    model.train(False)
    return model


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
    names = {'frontier_kernel': 'frontier_kernel',
             'score_frontier_kernel': 'score_frontier_kernel<4, %s, true>' % ('true' if lstm == 'split' else 'false'),
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
    """One-off kernel group (1): the GAT road-graph encoder.  HIP-event time of the whole tsg_gat_forward on the
    launch stream against its algorithmic HBM bytes: 4N*FIN + 4N*NH*F + E(4NH*F + 8NH + 8) + 4N*F.
    That figure prices the reference's formulation (projection written, projected rows gathered) and stays the
    unit of work; with head averaging and FIN <= 64 the library runs the layer in the input space
    (tsg_gat.hip: weights prepare, node scores, edge max, aggregation of the FIN-wide rows, one product with
    bias + ELU) and moves far fewer bytes, reported beside it as `moved_bytes`."""
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
    out = {'ms': ms, 'nodes': int(n), 'edges': e, 'algorithmic_bytes': byts, 'bound': 'hbm',
           'achieved': byts / (ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
           'frac': byts / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
    if not cfg['function_h']['concat'] and fin <= 64 and fin != f and n >= (nh + 1) * fin:
        kw = (nh + 1) * fin
        # x read for the scores and the own row, 2 score planes written and gathered per edge, FIN-wide rows
        # gathered per edge, the [N][(NH+1)FIN] operand written and read once, the stacked weights, the output
        moved = 8.0 * n * fin + 8.0 * n * nh + e * (8.0 * nh + 4.0 * fin + 4) + 8.0 * n * kw + 4.0 * kw * f + 4.0 * n * f
        out.update({'form': 'input-space (projection not materialised)', 'moved_bytes': moved,
                    'moved_GBs': moved / (ms * 1e-3) / 1e9,
                    # `frac` prices SURVEY 8(d)'s unit of work (the materialised form's bytes) and is NOT an HBM
                    # efficiency: this form moves `moved_bytes`; against those the encoder runs at frac_of_moved
                    'frac_of_moved': moved / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS})
    return out


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


_CPU = {}


def _cpu_worker_init():
    import torch
    torch.set_num_threads(1)
    # one warm-up search per worker (lazy initialisation, first-touch of the forked pages)
    from oracle import search as osr
    o = _CPU
    osr.astar_search(o['om'], o['net'].adjacent_list, o['net'].rid_gps, o['net'].road_length, o['rtd'],
                     [int(o['od']['start'][0])], [int(o['od']['start_time'][0])], int(o['od']['des'][0]),
                     int(o['od']['default_len'][0]), 30)


def _cpu_worker_window(arg):
    """One worker, one pass: searches of the fixed sample (worker k starts at OD pair k and strides by the number of
    workers, wrapping around) until the pass's deadline; a search cut short by the deadline counts the expansions it
    made.  Returns (expansions, searches finished)."""
    from oracle import search as osr
    k, n_workers, t_end = arg
    o = _CPU
    n = o['n_sample']
    steps = finished = 0
    i = k % n
    while time.time() < t_end:
        st = osr.SearchStats()
        st.deadline = t_end
        osr.astar_search(o['om'], o['net'].adjacent_list, o['net'].rid_gps, o['net'].road_length, o['rtd'],
                         [int(o['od']['start'][i])], [int(o['od']['start_time'][i])], int(o['od']['des'][i]),
                         int(o['od']['default_len'][i]), o['max_step'], stats=st)
        steps += st.steps
        finished += int(time.time() < t_end)
        i = (i + n_workers) % n
    return steps, finished


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(net, model, cfg, od, window_s, max_step, n_sample, passes=3):
    """The CPU oracle (port of search.py:240-424 + generator_v4.py:56-72: the reference's algorithm, one OD pair at a
    time, the LSTM re-run over the whole prefix at every expansion) on a FIXED sample: the first `n_sample` OD pairs
    of the workload, same max_step, one search per host core (forked workers, torch at one thread each: the
    reference's batch-1 model call does not use more), after one warm-up search per worker.  Runs before this
    process touches the GPU (the workers are forks).  SURVEY 8(d): one warm-up, then `passes` timed passes, the MEDIAN
    counts.  A search of this workload takes a core 15-25 s, so a pass is a fixed window of `window_s` seconds in which
    every core runs searches of the sample and the EXPANSIONS completed are counted (the data-independent unit;
    a search cut short by the end of the window counts what it did): the bench finishes in minutes whatever the
    host.  main() turns the median expansions/s into trajectories/s with the workload's mean expansions per OD pair
    (the GPU run of the same searches counts the same expansions: `step` of search.py:420)."""
    import multiprocessing as mp
    import torch
    from oracle.model import OracleGenerator
    workers = max(1, min(int(os.environ.get('OMP_NUM_THREADS', '0') or 0) or 16, os.cpu_count() or 1, 16))
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(1)      # the parent stays single-threaded until the workers are forked
    g_state = {k: v.detach().cpu() for k, v in model.function_g.state_dict().items()}
    h_state = {k: v.detach().cpu() for k, v in model.function_h.state_dict().items()}
    ei = np.stack([net.adj_mx.row, net.adj_mx.col]).astype(np.int64)
    om = OracleGenerator(g_state, h_state, float(model.w1.detach().cpu()[0]), net.node_features, ei,
                         cfg['function_h']['num_of_heads'], cfg['function_h']['embed_dim'],
                         cfg['function_g']['lstm_layer_num'])
    om.setup_node_emb()
    n_sample = min(n_sample, len(od['des']))
    _CPU.update({'om': om, 'net': net, 'rtd': net.road_time_distribution.astype(int), 'od': od,
                 'max_step': max_step, 'n_sample': n_sample})
    runs = []
    with mp.get_context('fork').Pool(workers, initializer=_cpu_worker_init) as pool:
        pool.map(_cpu_noop, range(workers * 4))          # every worker is up and warm before the clock starts
        for _ in range(max(1, passes)):
            t0 = time.time()
            res = pool.map(_cpu_worker_window, [(k, workers, t0 + window_s) for k in range(workers)], chunksize=1)
            dt = time.time() - t0
            runs.append((sum(r[0] for r in res) / dt, int(sum(r[0] for r in res)), int(sum(r[1] for r in res)), dt))
    torch.set_num_threads(prev_threads)
    _CPU.clear()
    runs.sort()
    rate, steps, finished, dt = runs[len(runs) // 2]          # the median pass
    return {'value': None, 'unit': 'trajectories/s', 'cores': workers, 'kind': 'port',
            'expansions_per_s': rate,
            'sample': 'median of %d passes of %.0f s: in a pass each of %d cores (forked workers, torch-CPU fp32 at 1 '
                      'thread each, 1 warm-up search per worker) runs searches of the fixed sample (first %d OD pairs '
                      'of the same workload, max_step %d) until the window ends; the median pass made %d expansions '
                      'in %.1f s (%d searches ran to their end)'
                      % (len(runs), window_s, workers, n_sample, max_step, steps, dt, finished),
            'passes': [{'expansions_per_s': r[0], 'expansions': r[1], 'searches_finished': r[2], 'seconds': r[3]}
                       for r in runs],
            'cpu_model': cpu_model_name(), 'host_cpus': os.cpu_count(), 'torch_threads_per_worker': 1}


def _cpu_noop(_):
    time.sleep(0.01)
    return 0


def classify_differences(searcher, model, od, idx, max_step, lmax, formats=(1, 0), tol=1e-5):
    """OD pairs whose trajectories differ between the two recurrence forms: run them again under both with the pop
    log on and find the first pop at which the two searches part.  'near_ties': the priorities of that pop agree
    within `tol` relative (the documented exception of the north star: two candidates nearly equally likely, the
    last bits decide); 'unexplained': anything else."""
    import torch
    runs = []
    keep = searcher.state_format
    for fmt in formats:
        searcher._drop_engines()
        searcher.state_format = fmt
        runs.append(searcher.generate(model, od['start'][idx], od['start_time'][idx], od['des'][idx],
                                      od['default_len'][idx], max_step=max_step, slots=min(1024, max(8, len(idx))),
                                      lmax=lmax, poplog_cap=max_step + 8))
    searcher._drop_engines()
    searcher.state_format = keep
    a, b = runs
    near = unexplained = vanished = 0
    for i in range(len(idx)):
        if all(torch.equal(a[k][i], b[k][i]) for k in ('rid', 'time', 'len', 'status', 'steps')):
            vanished += 1          # cannot happen (results do not depend on the batch); counted to be sure
            continue
        # (a pair may differ in its step count alone: two pops changed places on the way to the same trajectory)
        ka, kb = int(a['poplog_n'][i]), int(b['poplog_n'][i])
        ra, rb = a['poplog_rid'][i, :ka].numpy(), b['poplog_rid'][i, :kb].numpy()
        m = min(ka, kb)
        diff = np.flatnonzero(ra[:m] != rb[:m])
        if len(diff) == 0:
            unexplained += 1
            continue
        j = int(diff[0])
        pa, pb = float(a['poplog_prio'][i, j]), float(b['poplog_prio'][i, j])
        if abs(pa - pb) < tol * max(1.0, abs(pb)):
            near += 1
        else:
            unexplained += 1
    return {'near_ties': near, 'unexplained': unexplained, 'same_on_rerun': vanished}


def two_stage_record(net, od, device, max_step):
    """BASELINE.json configs[3]: two-stage generation (region-level A*, then road-level A* leg by leg) on the
    Porto-scale graph with a 10 x 10 region partition, region model 128/32/128; whole-job trajectories/s."""
    import torch
    from ts_trajgen_amd import dataset as ds
    from ts_trajgen_amd import synth
    from ts_trajgen_amd.generator import GeneratorV4
    from ts_trajgen_amd.search import DoubleLayerSearcher
    n_od = len(od['des'])
    free_gb = torch.cuda.mem_get_info(torch.device(device))[0] / 1e9
    reg = synth.region_tables(net, tiles=10)
    road, cfg = build_model_cpu(net)
    road = to_device(road, cfg, device)
    road.function_h._setup_node_emb()
    feats = ds.region_features_from_road_emb(road.function_h.node_emb_feature[:net.num_roads], reg['rid2region'],
                                             reg['num_regions'])
    feats = torch.cat([feats, torch.zeros((1, feats.shape[1]), device=feats.device)], dim=0)
    r = reg['num_regions']
    rcfg = ds._gen_config((128, 32, 128, 128, 10, 5), device, True)          # our_model_generate.py:62-83
    rdf = {'road_num': r + 1, 'time_size': 2881, 'road_pad': r, 'time_pad': 2880, 'adj_mx': reg['region_adj_mx'],
           'node_features': feats.cpu(), 'img_width': 52, 'img_height': 43}
    torch.manual_seed(1)
    region = GeneratorV4(config=rcfg, data_feature=rdf).to(device)
    region.function_h.node_features = region.function_h.node_features.to(device)
    region.train(False)
    s = DoubleLayerSearcher(device, net.adjacent_list, net.rid_gps, net.road_length, reg['region_adjacent_list'],
                            reg['region_dist'], reg['region_transfer_freq'], reg['rid2region'], reg['region2rid'],
                            net.road_time_distribution, reg['region_time_distribution'])
    w = slice(0, 256)
    s.generate(region, road, od['start'][w], od['start_time'][w], od['des'][w], od['default_len'][w],
               max_step=max_step, seeds=list(range(256)))
    torch.cuda.synchronize()
    t0 = time.time()
    locs, tims, flags = s.generate(region, road, od['start'], od['start_time'], od['des'], od['default_len'],
                                   max_step=max_step, seeds=list(range(n_od)))
    torch.cuda.synchronize()
    dt = time.time() - t0
    reached = sum(int(l[-1] == d) for l, d in zip(locs, od['des']))
    return {'value': n_od / dt, 'unit': 'trajectories/s', 'od_pairs': n_od, 'seconds': dt,
            'workload': 'two-stage: Porto-scale graph N=%d, %d regions (10 x 10 tiling), region model 128/32/128, '
                        'per-OD RandomState(od_id), max_step %d' % (net.num_roads, r, max_step),
            'reached_frac': reached / n_od, 'rounds': s.last_stats['rounds'],
            'road_expansions_per_od': s.last_stats['road_expansions'] / n_od,
            'gpu_busy_frac': s.last_stats.get('gpu_busy_frac'), 'free_hbm_gb_at_start': free_gb}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    from ts_trajgen_amd import synth
    net, net_name = build_network(args.config)
    scaling = 'weak'
    if args.total_ods > 0:
        n_total, scaling = args.total_ods, 'strong'
    else:
        per_gpu = args.ods if args.ods > 0 else (131072 if args.config == 'bj' else 256 if args.config == 'grid' else 65536)
        n_total = per_gpu * world
    # OD pairs for all ranks (seeded), sharded round-robin by index; BFS default_len before the GPU is touched
    # (tsg_host_hop_distance: `setup_workers` pthreads; 131 072 OD pairs at BJ scale take 36 s on 2 threads of the
    # build container, 9 s on 8)
    setup_workers = max(1, min(16, (os.cpu_count() or 1) // world))
    t_setup = time.time()
    od = synth.make_od_pairs(net, n_total, seed=2, workers=setup_workers,
                             default_len='lattice' if args.config == 'stress' else 'hops', shard=(rank, world))
    t_setup = time.time() - t_setup
    n_local = len(od['des'])
    od_warm = {k: v[:args.warmup_ods] for k, v in od.items()}
    net2 = od2 = None
    if world == 1 and args.two_stage_ods > 0:      # configs[3] sub-record: its OD pairs, also before the GPU is touched
        net2 = synth.porto_scale_network()
        od2 = synth.make_od_pairs(net2, args.two_stage_ods, seed=7, workers=max(1, min(16, os.cpu_count() or 1)))

    model, cfg = build_model_cpu(net, gat_dim=128 if args.config == 'stress' else None)
    cpu = None
    if world == 1 and args.cpu_seconds > 0:       # forks its workers: must come before the GPU is touched
        cpu = cpu_baseline(net, model, cfg, od, args.cpu_seconds, args.max_step, args.cpu_ods, args.cpu_passes)

    import torch
    import torch.distributed as dist
    # rehearsal hooks (a one-GPU box): TSG_BENCH_DEVICE puts every rank on that device, TSG_BENCH_BACKEND=gloo
    # replaces RCCL, which refuses two ranks on one GPU; the driver's multi-GPU runs set neither
    dev_index = int(os.environ.get('TSG_BENCH_DEVICE', local_rank))
    backend = os.environ.get('TSG_BENCH_BACKEND', 'nccl')
    device = 'cuda:%d' % dev_index
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
        # first contact with the communicator BEFORE the free memory is read and the page pool sized: RCCL creates its
        # channels and workspaces at the first collective of each kind
        warm_in = torch.zeros(1 << 16, dtype=torch.int32, device=device)
        warm_out = torch.zeros(world << 16, dtype=torch.int32, device=device)
        dist.all_reduce(warm_in)
        dist.all_gather_into_tensor(warm_out, warm_in)
        torch.cuda.synchronize()
        del warm_in, warm_out
    from ts_trajgen_amd.search import Searcher
    model = to_device(model, cfg, device)
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
    budget = int(free_b * 0.93) - (2 * args.lmax + 8) * 4 * n_local
    rccl_headroom = 0
    if world > 1:
        # the exchange step's buffers (rows trimmed to the longest trajectory; priced here at half of lmax, far above
        # what the synthetic graphs produce) and an explicit reserve for RCCL's own channels and workspaces, which
        # it allocates at the first collective, after the page pool has been sized
        from ts_trajgen_amd.distributed import gather_plan_bytes
        rccl_headroom = 4 << 30
        budget -= gather_plan_bytes(n_total, world, args.lmax, width=args.lmax // 2) + rccl_headroom
    slots = args.slots
    if slots <= 0:
        slots = int(budget // (fixed + pps * page / args.oversub))
        slots = min(slots, 32768, n_local)
        # LSTM grid: 8 workgroups per 128 searches, 768 resident at a time -> whole waves of workgroups at 12 288
        slots = slots // 12288 * 12288 if slots >= 12288 else max(128, slots // 1024 * 1024 if slots >= 1024 else slots // 128 * 128)
    pool_pages = min(slots * pps, int((budget - slots * fixed) // page))
    ramp = args.ramp if pool_pages < slots * pps else 0

    def one_step(o, profile_every=0):
        res = searcher.generate(model, o['start'], o['start_time'], o['des'], o['default_len'],
                                max_step=args.max_step, slots=slots, lmax=args.lmax, return_device=True,
                                profile_every=profile_every, pool_pages=pool_pages, ramp_steps=ramp,
                                streams=args.engines)
        if world > 1:   # the single exchange step of the path: collect the padded trajectories (RCCL over xGMI)
            from ts_trajgen_amd.distributed import gather_results
            n_call = n_total if o is od else len(o['des']) * world
            res['gathered'] = gather_results(res, n_call, args.lmax, rank, world)
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
    n_done = args.steps * n_total
    status = res['status'].cpu().numpy()
    exp_total = sum(s['expansions'] for s in stats)
    tot = torch.tensor([exp_total, sum(s['model_evals'] for s in stats), sum(s['items'] for s in stats)],
                       dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tot)
    exp_all, evals_all, items_all = [float(x) for x in tot.tolist()]

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
                mfma = {'achieved': flops / (avg_ms * 1e-3) / 1e12, 'peak': peak, 'unit': 'TFLOP/s'}
                mfma['frac'] = mfma['achieved'] / peak
                # algorithmic HBM bytes per row: previous h and c of the layer in, new h and c out (4 rows of
                # 4 Hp bytes); layer 0 also gathers the pre-multiplied input rows xproj[rid] and tproj[tcode]
                # (4 * 4 Hp bytes each), layers above read the new h of the layer below (4 Hp)
                per = 4 * 4 * hp + (2 * 16 * hp if l == 0 else 4 * hp)
                hbm = {'achieved': rows_per_launch * per / (avg_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s'}
                hbm['frac'] = hbm['achieved'] / PEAK_HBM_GBS
                binding = hbm if hbm['frac'] > mfma['frac'] else mfma      # the roofline the kernel is nearer to
                k.update({'bound': 'hbm' if binding is hbm else 'mfma', 'achieved': binding['achieved'],
                          'peak': binding['peak'], 'unit': binding['unit'], 'mfma': mfma, 'hbm': hbm,
                          'algorithmic_bytes_per_row': per, 'algorithmic_flop_per_row': 2.0 * 4 * hp * K})
            elif nm.startswith('score'):
                # own + ancestor top-layer h rows, gathered out_fc rows (+bias), factored H tables,
                # lon/lat, CSR row, item record, cand_lp write
                per = 4 * hp * (1 + avg_L) + avg_C * (4 * (hp + 1) + 80 + 16 + 8) + 40 + 16 + 32
                byts = rows_per_launch * scored_frac * per
                k.update({'bound': 'hbm', 'achieved': byts / (avg_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                          'unit': 'GB/s', 'algorithmic_bytes_per_scored_row': per})
            else:
                per = 40 * avg_C + 32 + 32 + 128      # SURVEY 8(d) frontier bytes + item + slot state
                byts = rows_per_launch * per
                k.update({'bound': 'hbm', 'achieved': byts / (avg_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS,
                          'unit': 'GB/s', 'note': 'latency-bound: chains of dependent 8-32 byte accesses'})
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
        dual = None
        if world == 1 and args.two_engines and args.engines == 1 and n_local >= 8:
            # the same step on two engines side by side (Searcher.generate streams=2: half the searches in flight and
            # half the page pool each, two host threads, two streams): whole-job throughput of that configuration,
            # and its results against the single-engine step's
            searcher._drop_engines()
            searcher.generate(model, od_warm['start'], od_warm['start_time'], od_warm['des'], od_warm['default_len'],
                              max_step=args.max_step, slots=slots, lmax=args.lmax, return_device=True,
                              pool_pages=pool_pages, ramp_steps=ramp, streams=2)
            torch.cuda.synchronize()
            t1 = time.time()
            res2 = searcher.generate(model, od['start'], od['start_time'], od['des'], od['default_len'],
                                     max_step=args.max_step, slots=slots, lmax=args.lmax, return_device=True,
                                     pool_pages=pool_pages, ramp_steps=ramp, streams=2)
            torch.cuda.synchronize()
            dt2 = time.time() - t1
            st2 = searcher.last_stats
            dual = {'value': n_local / dt2, 'unit': 'trajectories/s', 'engines': 2, 'seconds': dt2,
                    'searches_in_flight_per_engine': slots // 2 // 128 * 128, 'supersteps': st2['supersteps'],
                    'page_waits': st2['page_waits'], 'restarts': st2['restarts'],
                    'queues_apart': st2.get('queues_apart'),      # the two engines' streams passed the overlap test
                    'identical_to_single_engine': bool(all(torch.equal(res[k], res2[k])
                                                           for k in ('rid', 'time', 'len', 'status', 'steps'))),
                    'note': 'same workload and sizes as the timed step; latency-bound launches of one engine run '
                            'under the matrix-core / bandwidth-bound launches of the other (bench.py --engines 2 makes '
                            'it the timed step)'}
            del res2
        fp32 = None
        if world == 1 and args.fp32_step and args.lstm == 'split' and args.engines == 1 and n_local >= 8:
            # the reference's recurrence is fp32 (function_g_fc.py:169); the timed step carries its operands as two
            # f16 numbers (22 bits).  The same step on the plain fp32 matrix-core path (state_format 0), with the
            # same sizes: whole-job throughput, per-kernel times, and its trajectories against the split-f16 step's
            searcher._drop_engines()
            searcher.state_format = 0
            kw32 = dict(max_step=args.max_step, slots=slots, lmax=args.lmax, return_device=True,
                        pool_pages=pool_pages, ramp_steps=ramp, streams=1)
            searcher.generate(model, od_warm['start'], od_warm['start_time'], od_warm['des'], od_warm['default_len'], **kw32)
            torch.cuda.synchronize()
            t1 = time.time()
            res32 = searcher.generate(model, od['start'], od['start_time'], od['des'], od['default_len'],
                                      profile_every=args.profile_every, **kw32)
            torch.cuda.synchronize()
            dt32 = time.time() - t1
            st32 = dict(searcher.last_stats)
            same = torch.ones(n_local, dtype=torch.bool, device=res['len'].device)
            for k in ('rid', 'time'):
                same &= (res[k] == res32[k]).all(dim=1)
            for k in ('len', 'status', 'steps'):
                same &= res[k] == res32[k]
            differ = torch.nonzero(~same).flatten().cpu().numpy()
            del res32
            searcher.state_format = 1
            fp32 = {'value': n_local / dt32, 'unit': 'trajectories/s', 'seconds': dt32,
                    'recurrence': 'fp32 matrix cores (v_mfma_f32_32x32x2_f32), hidden states stored as fp32',
                    'vs_split_f16': (n_local / dt32) / (n_done / dt), 'supersteps': st32['supersteps'],
                    'kernel_avg_us': {nm: (1e3 * st32['prof_ms'][i] / st32['prof_launches'][i]
                                           if st32['prof_launches'][i] else None) for i, nm in enumerate(names)},
                    'identical_to_split': int(n_local - len(differ)), 'od_pairs': int(n_local),
                    'differing': int(len(differ))}
            # layer 1 against the fp32 matrix-core peak (MI355X_MICROARCH.md): rows x 4hp x 2hp x 2 per launch
            t_l1 = fp32['kernel_avg_us'].get('lstm_layer_kernel[l1]')
            if t_l1 and st32['prof_launches'][0]:
                rows32 = st32['prof_items'] / st32['prof_launches'][0]
                tf = rows32 * 2.0 * 4 * hp * 2 * hp / (t_l1 * 1e-6) / 1e12
                fp32['layer1_mfma'] = {'achieved': tf, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                       'frac': tf / PEAK_F32_MFMA_TFLOPS}
            if 0 < len(differ) <= 4096:
                fp32.update(classify_differences(searcher, model, od, differ, args.max_step, args.lmax))
        two = None
        if world == 1 and args.two_stage_ods > 0:
            del res
            searcher._drop_engines()          # give the page pool back before the two-stage engines are sized
            # no empty_cache(): the two-stage engines take their workspaces out of torch's cached blocks.  Memory
            # handed back to the driver and allocated again costs 0.3 s per 42 GB tensor on this box (4 s in all).
            if os.environ.get('TSG_PROFILE_TWO_STAGE'):      # diagnostic: where the sub-record's time goes
                import cProfile
                import pstats
                pr = cProfile.Profile()
                pr.enable()
                two = two_stage_record(net2, od2, device, args.max_step)
                pr.disable()
                pstats.Stats(pr, stream=sys.stderr).sort_stats('cumulative').print_stats(40)
            else:
                two = two_stage_record(net2, od2, device, args.max_step)
        recurrence = ('split-f16 x3: every fp32 operand as f16 major + f16 minor (22 significant bits), three f16 '
                      'matrix-core passes, fp32 accumulation; hidden states stored in that form, cell states fp32'
                      if args.lstm == 'split' else 'fp32 matrix cores (v_mfma_f32_32x32x2_f32)')
        if cpu is not None:      # expansions/s -> trajectories/s with this workload's mean expansions per OD pair
            per_od = exp_all / n_done
            cpu['value'] = cpu['expansions_per_s'] / per_od
            cpu['sample'] += '; trajectories/s = expansions/s / %.0f expansions per OD pair of this workload' % per_od
        out = {
            'metric': 'generated trajectories/sec (road-level A*)', 'value': n_done / dt, 'unit': 'trajectories/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt * 1e3 / args.steps,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': 'f32' if args.lstm != 'split' else 'f32 (recurrence operands: split-f16, 22 bits; fp32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': '%s, %d OD pairs per step over %d GPU(s) dealt od_id %% world (%d per GPU), BJ model '
                                   'dims (hidden 256, 2-layer LSTM, GAT 5x256), random-init weights, max_step %d'
                                   % (net_name, n_total, world, n_local, args.max_step),
                       'baseline_config': {'bj': 'configs[2]', 'porto': 'configs[1]', 'grid': 'configs[0]',
                                           'stress': 'configs[4]'}[args.config],
                       'od_pairs_per_step': n_total, 'od_pairs_per_gpu': n_local, 'slots_in_flight': slots,
                       'warmup_od_pairs': int(len(od_warm['des'])),      # a warm-up step is this many OD pairs per GPU, not a full step
                       'engines': args.engines, 'queues_apart': stats[-1].get('queues_apart'),
                       'state_pool_pages': pool_pages, 'state_pool_gb': pool_pages * page / 1e9, 'lstm': args.lstm,
                       'recurrence': recurrence, 'parallelism': 'od-shard x%d' % world},
            'expansions_per_s': exp_all / dt, 'expansions_per_od': exp_all / n_done,
            # `expansions` counts every `step` of search.py:420, including the ones the frontier kernel closes without
            # a model call (single-successor, gated-out and futile expansions: identical results); what ran the
            # recurrence is work_items, what ran the whole model (recurrence + scoring) is model_evals
            'work_items_per_s': items_all / dt, 'model_evals_per_s': evals_all / dt,
            'reached_frac': float((status & 0xff == 1).mean()), 'fallback_frac': float((status & 0xff == 2).mean()),
            'avg_prefix_len_scored': avg_L, 'avg_candidates_scored': avg_C,
            'supersteps': int(sum(s['supersteps'] for s in stats)), 'work_items': int(items),
            'page_waits': int(sum(s['page_waits'] for s in stats)), 'restarts': int(sum(s['restarts'] for s in stats)),
            'min_free_pages': int(min(s['min_free_pages'] for s in stats)),
            'roofline': roofline, 'kernels': kern, 'measured_peaks': peaks, 'gat_encoder': gat, 'cpu_baseline': cpu, 'two_engines': dual,
            'fp32_recurrence': fp32, 'two_stage': two,
            'setup': {'od_pairs_s': t_setup, 'workers': setup_workers, 'rccl_headroom_bytes': rccl_headroom,
                      'note': 'hop distances (default_len) of this rank\'s OD pairs before the GPU is touched'},
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
