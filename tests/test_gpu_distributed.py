"""The N > 1 path with REAL searches on the GPU box (one MI355X): OD sharding + the exchange step of
ts-trajgen_amd/distributed.py around the device engine.

* world size 1 under the `nccl` backend on cuda:0: the RCCL code path itself (communicator set-up, the two
  all-gathers of gather_results on device tensors) on hardware; results equal the plain call's.
* world size 2 under `gloo`, both ranks on cuda:0 (RCCL refuses two ranks on one GPU): every rank runs its own
  shard od_id % 2 on an engine of its own and receives all trajectories in OD order; equal to the single-process
  results (golden G5: 64 searches with dead ends, unreachable destinations, budget exhaustion).
The CPU test (tests/test_distributed.py) covers the same plumbing with a stand-in for the local search.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests._fixtures import GoldenRoadWorld, build_generator, load_json

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LMAX = 64
KEYS = ('rid', 'time', 'len', 'status', 'steps')


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _world_and_cases():
    road = GoldenRoadWorld()
    cases = load_json('g5_astar.json')['cases']
    args = ([c['start'] for c in cases], [c['start_time'] for c in cases], [c['des'] for c in cases],
            [c['default_len'] for c in cases], np.array([c['max_step'] for c in cases], dtype=np.int32))
    return road, cases, args


def _local_generate(searcher, model):
    return lambda st, tm, de, dl, ms: searcher.generate(model, st, tm, de, dl, max_step=ms, slots=16, lmax=LMAX,
                                                        return_device=True)


def _same(got, ref):
    w = got['width']
    assert w == int(ref['len'].max().clamp(max=LMAX))
    for k in KEYS:
        want = ref[k][:, :w] if ref[k].dim() == 2 else ref[k]
        assert torch.equal(got[k].cpu(), want.cpu()), k
    assert int(ref['rid'][:, w:].abs().sum()) == 0


def test_generate_sharded_over_rccl_world_size_1():
    from ts_trajgen_amd.distributed import generate_sharded
    from ts_trajgen_amd.search import Searcher
    road, cases, args = _world_and_cases()
    model = build_generator(road, DEV)
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    ref = _local_generate(s, model)(*args)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % _free_port(), rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        got = generate_sharded(_local_generate(s, model), *args, lmax=LMAX)
        torch.cuda.synchronize()
        assert got['rid'].is_cuda
        _same(got, ref)
    finally:
        dist.destroy_process_group()


def _rank_main(rank, world, port, q):
    try:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        from ts_trajgen_amd.distributed import generate_sharded
        from ts_trajgen_amd.search import Searcher
        road, cases, args = _world_and_cases()
        model = build_generator(road, DEV)
        s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
        got = generate_sharded(_local_generate(s, model), *args, lmax=LMAX)
        n_local = s.last_stats['n_od']
        out = {k: got[k].cpu().numpy() for k in KEYS}
        out['width'] = got['width']
        q.put((rank, n_local, out, None))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as ex:      # reported to the parent instead of a silent hang
        q.put((rank, -1, None, repr(ex)))


def test_two_ranks_on_one_gpu_match_single_process():
    from ts_trajgen_amd.search import Searcher
    road, cases, args = _world_and_cases()
    model = build_generator(road, DEV)
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    ref = {k: v.cpu() for k, v in _local_generate(s, model)(*args).items()}
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
    for rank, n_local, out, err in got:
        assert err is None, 'rank %d: %s' % (rank, err)
        assert n_local == len(range(rank, len(cases), 2))      # each rank searched only its own shard
        res = {k: torch.from_numpy(out[k]) for k in KEYS}
        res['width'] = out['width']
        _same(res, ref)
