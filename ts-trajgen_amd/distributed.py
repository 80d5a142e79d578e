"""OD-sharded multi-GPU generation: one process per GPU, weights and graph replicated, OD pairs dealt
round-robin by index (od_id % world), and ONE all-gather (RCCL over xGMI with the `nccl` backend, `gloo`
on CPU for tests) of the fixed-size padded result rows.  The searches never exchange anything else:
the reference loops OD pairs independently (our_model_generate.py:417-426).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n, rank, world):
    """Indices of the OD pairs owned by `rank` (round-robin evens out per-OD difficulty)."""
    return np.arange(rank, n, world, dtype=np.int64)


def pack_rows(res, lmax):
    """[n_local, 2*lmax + 3] int32: rid row, time row, len, status, steps."""
    return torch.cat([res['rid'], res['time'], res['len'].unsqueeze(1), res['status'].unsqueeze(1),
                      res['steps'].unsqueeze(1)], dim=1).to(torch.int32).contiguous()


def gather_results(local_packed, n_total, lmax, rank, world, group=None):
    """All-gather the per-rank packed rows (padded to the largest shard) and restore OD order."""
    per = (n_total + world - 1) // world
    width = 2 * lmax + 3
    buf = torch.zeros((per, width), dtype=torch.int32, device=local_packed.device)
    buf[:local_packed.shape[0]] = local_packed
    flat = torch.empty((world * per, width), dtype=torch.int32, device=local_packed.device)
    dist.all_gather_into_tensor(flat, buf, group=group)
    out = flat.view(world, per, width)
    full = torch.zeros((n_total, width), dtype=torch.int32, device=local_packed.device)
    for r in range(world):
        idx = torch.from_numpy(shard_indices(n_total, r, world)).to(local_packed.device)
        full[idx] = out[r, :idx.numel()]
    return {'rid': full[:, :lmax], 'time': full[:, lmax:2 * lmax], 'len': full[:, 2 * lmax],
            'status': full[:, 2 * lmax + 1], 'steps': full[:, 2 * lmax + 2]}


def generate_sharded(local_generate, start_rids, start_times, des_rids, default_lens, max_step, lmax=256,
                     group=None):
    """Run `local_generate(start, time, des, default_len, max_step) -> result dict` on this rank's shard of
    the OD list and return the gathered results for ALL OD pairs (identical on every rank).

    `local_generate` is normally `lambda *a: searcher.generate(model, *a, lmax=lmax, return_device=True)`.
    Without an initialised process group this is just the local call."""
    n = len(des_rids)
    if not (dist.is_available() and dist.is_initialized()):
        return local_generate(start_rids, start_times, des_rids, default_lens, max_step)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    idx = shard_indices(n, rank, world)
    ms = np.broadcast_to(np.asarray(max_step, dtype=np.int32), (n,))
    res = local_generate(np.asarray(start_rids)[idx], np.asarray(start_times)[idx], np.asarray(des_rids)[idx],
                         np.asarray(default_lens)[idx], ms[idx])
    return gather_results(pack_rows(res, lmax), n, lmax, rank, world, group)
