"""Generation driver: the OD loop of our_model_generate.py:394-437 on the batched device engine.

    python -m ts_trajgen_amd.generate --local true --dataset_name Porto_Taxi --device cuda:0

Same inputs (dataset files, `./save/<dataset>/*.pt` checkpoints), same output CSV
(`traj_id,"r1,r2,...","t1,t2,..."`), same counters printed at the end (`fail cnt`, `region astar fail cnt`)
and the same fail logs.  What differs from the reference: all OD pairs of the test file are searched at
once (DoubleLayerSearcher.generate), the border-road order of OD i is drawn from `RandomState(seed + i)`
instead of the global numpy stream, and under `torchrun` the ODs are dealt round-robin to the ranks and
gathered with one all-gather (distributed.py).
"""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist

from . import dataset as ds
from . import distributed as tsg_dist
from .generator import GeneratorV4
from .search import DoubleLayerSearcher, Searcher


def str2bool(s):
    """utils/parser.py: argparse-friendly bool."""
    if isinstance(s, bool):
        return s
    if s.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if s.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('bool value expected.')


def build_generator(config, data_feature, device, g_state=None, h_state=None):
    """our_model_generate.py:394-399: construct, load the two sub-module checkpoints, eval mode."""
    model = GeneratorV4(config=config, data_feature=data_feature).to(device)
    if g_state is not None:
        model.function_g.load_state_dict(g_state)
    if h_state is not None:
        model.function_h.load_state_dict(h_state)
    model.train(False)
    return model


def _load_state(path, device):
    return torch.load(path, map_location=device, weights_only=True)


def _pack(locs, tims, flags, lmax, device):
    n = len(locs)
    rid = np.zeros((n, lmax), dtype=np.int32)
    tim = np.zeros((n, lmax), dtype=np.int32)
    ln = np.zeros(n, dtype=np.int32)
    for i, (a, b) in enumerate(zip(locs, tims)):
        ln[i] = len(a)
        rid[i, :len(a)] = a
        tim[i, :len(b)] = b
    t = lambda x: torch.from_numpy(x).to(device)
    return {'rid': t(rid), 'time': t(tim), 'len': t(ln), 'status': t(np.asarray(flags, dtype=np.int32)),
            'steps': torch.zeros(n, dtype=torch.int32, device=device)}


def run(data, road_generator, region_generator, device, start, code, des, default_len, max_step=5000,
        seed=0, mode='double', slots=None):
    """Search every OD pair; returns (locs, tims, is_astar, searcher).  Under an initialised process group
    each rank searches od_id % world == rank and every rank returns the full, ordered result."""
    s = data['searcher']
    searcher = DoubleLayerSearcher(device=device, adjacent_list=s['adjacent_list'],
                                   road_center_gps=s['road_center_gps'], road_length=s['road_length'],
                                   region_adjacent_list=s['region_adjacent_list'], region_dist=s['region_dist'],
                                   region_transfer_freq=s['region_transfer_freq'], rid2region=s['rid2region'],
                                   road_time_distribution=s['road_time_distribution'],
                                   region_time_distribution=s['region_time_distribution'],
                                   region2rid=s['region2rid'])
    n = len(des)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    idx = tsg_dist.shard_indices(n, rank, world)
    with torch.no_grad():
        if mode == 'double':
            locs, tims, flags = searcher.generate(region_generator, road_generator, start[idx], code[idx], des[idx],
                                                  default_len[idx], max_step=max_step, seeds=seed + idx,
                                                  slots=slots)
        else:                                                    # road-level A* only (Searcher.astar_search)
            res = searcher.road_search.generate(road_generator, start[idx], code[idx], des[idx], default_len[idx],
                                                max_step=max_step, slots=slots,
                                                lmax=max(256, int(default_len.max()) + 1))
            ln = res['len'].tolist()
            locs = [res['rid'][i, :ln[i]].tolist() for i in range(len(idx))]
            tims = [res['time'][i, :ln[i]].tolist() for i in range(len(idx))]
            flags = [1] * len(idx)
    if world > 1:
        lmax_local = torch.tensor([max([len(a) for a in locs] + [1])], dtype=torch.int64,
                                  device=device if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(lmax_local, op=dist.ReduceOp.MAX)       # row width only; the data moves in one all-gather
        lmax = int(lmax_local.item())
        pk = _pack(locs, tims, flags, lmax, device if dist.get_backend() == 'nccl' else 'cpu')
        full = tsg_dist.gather_results(tsg_dist.pack_rows(pk, lmax), n, lmax, rank, world)
        ln = full['len'].tolist()
        locs = [full['rid'][i, :ln[i]].tolist() for i in range(n)]
        tims = [full['time'][i, :ln[i]].tolist() for i in range(n)]
        flags = full['status'].tolist()
        logs = [None] * world
        dist.all_gather_object(logs, (sorted(searcher.fail_find_log), sorted(searcher.region_fail_find_log)))
        for a, b in logs:
            searcher.fail_find_log.update(tuple(x) for x in a)
            searcher.region_fail_find_log.update(tuple(x) for x in b)
    return locs, tims, flags, searcher


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--local', type=str2bool, default=False)
    parser.add_argument('--dataset_name', type=str, default='BJ_Taxi')
    parser.add_argument('--device', type=str, default='cuda:0')
    parser.add_argument('--data_root', type=str, default=None, help="overrides --local's choice of root")
    parser.add_argument('--save_dir', type=str, default='./save')
    parser.add_argument('--max_step', type=int, default=5000)
    parser.add_argument('--seed', type=int, default=0)
    parser.add_argument('--mode', choices=('double', 'road'), default='double')
    parser.add_argument('--random_init', type=str2bool, default=False,
                        help='no checkpoints: keep the constructors\' random weights (smoke runs)')
    args = parser.parse_args(argv)
    data_root = args.data_root or ('./data/' if args.local else '/mnt/data/jwj/')
    device = args.device
    if 'RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1 and not dist.is_initialized():
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        if device.startswith('cuda'):
            device = 'cuda:%d' % local_rank
            torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl' if device.startswith('cuda') else 'gloo')
    rank = dist.get_rank() if dist.is_initialized() else 0

    data = ds.load_dataset(data_root, args.dataset_name, device=device)
    gen_config, region_gen_config = ds.gen_configs(args.dataset_name, device, data['meta'])
    save = os.path.join(args.save_dir, args.dataset_name)
    states = [None] * 4
    if not args.random_init:
        states = [_load_state(os.path.join(save, f), device) for f in
                  ('function_g_fc.pt', 'gat_fc.pt', 'region_function_g_fc.pt', 'region_gat_fc.pt')]
    else:
        torch.manual_seed(args.seed)
    road_generator = build_generator(gen_config, data['data_feature'], device, states[0], states[1])
    region_generator = build_generator(region_gen_config, data['region_data_feature'], device, states[2], states[3])

    ids, start, code, des, default_len = ds.read_test_csv(data['paths']['test_csv'])
    locs, tims, flags, searcher = run(data, road_generator, region_generator, device, start, code, des, default_len,
                                      max_step=args.max_step, seed=args.seed, mode=args.mode)
    if rank == 0:
        ds.write_generate_csv(data['paths']['generate_csv'], ids, locs, tims)
        fail_cnt = sum(1 for loc, d in zip(locs, des) if loc[-1] != int(d))
        region_astar_fail_cnt = sum(1 for f in flags if f == 0)
        print('fail cnt ', fail_cnt)
        print('region astar fail cnt ', region_astar_fail_cnt)
        searcher.save_fail_log()
    if dist.is_initialized():
        dist.barrier()
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
