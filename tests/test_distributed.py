"""World-size-2 gloo test of the OD sharding + single all-gather (runs on CPU; the local search is a
deterministic stand-in so that only the distributed plumbing is under test)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

LMAX = 8


def fake_local_generate(start, tim, des, dlen, max_step):
    n = len(des)
    rid = torch.zeros((n, LMAX), dtype=torch.int32)
    t = torch.zeros((n, LMAX), dtype=torch.int32)
    ln = torch.zeros(n, dtype=torch.int32)
    for i in range(n):
        k = 1 + int(des[i]) % (LMAX - 1)
        rid[i, :k] = torch.arange(k, dtype=torch.int32) + int(start[i])
        t[i, :k] = int(tim[i])
        ln[i] = k
    return {'rid': rid, 'time': t, 'len': ln, 'status': torch.as_tensor(np.asarray(dlen) % 3, dtype=torch.int32),
            'steps': torch.as_tensor(np.asarray(max_step), dtype=torch.int32)}


def _worker(rank, world, port, n, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ts_trajgen_amd.distributed import generate_sharded
    rng = np.random.RandomState(0)
    start, tim, des, dlen = (rng.randint(0, 1000, n) for _ in range(4))
    ms = rng.randint(1, 50, n)
    res = generate_sharded(fake_local_generate, start, tim, des, dlen, ms, lmax=LMAX)
    ref = fake_local_generate(start, tim, des, dlen, ms)
    w = res['width']
    ok = w == int(ref['len'].max()) and all(
        torch.equal(res[k].cpu(), ref[k][:, :w] if ref[k].dim() == 2 else ref[k]) for k in ref)
    ok = ok and all(int(ref[k][:, w:].abs().sum()) == 0 for k in ('rid', 'time'))      # nothing beyond the trimmed width
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_generate_matches_single_process():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 37, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(got) == [(0, True), (1, True)]


def test_shard_indices_partition():
    from ts_trajgen_amd.distributed import shard_indices
    for n in (0, 1, 7, 64):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(np.concatenate(parts).tolist()) == list(range(n))


def test_encode_time():
    from ts_trajgen_amd.utils import encode_time
    assert encode_time('2018-10-01T08:30:00Z') == 8 * 60 + 30           # Monday
    assert encode_time('2018-10-06T08:30:59Z') == 8 * 60 + 30 + 1440    # Saturday
    assert encode_time('2018-10-07T23:59:00Z') == 2879                   # Sunday
