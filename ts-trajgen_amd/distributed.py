"""OD-sharded multi-GPU generation: one process per GPU, weights and graph replicated, OD pairs dealt
round-robin by index (od_id % world), and ONE exchange step at the end (RCCL over xGMI with the `nccl` backend,
`gloo` on CPU for tests): an all-gather of three words per OD pair, then an all-gather of the trajectory rows trimmed
to the longest trajectory of the job.  The searches never exchange anything else:
the reference loops OD pairs independently (our_model_generate.py:417-426).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n, rank, world):
    """Indices of the OD pairs owned by `rank` (round-robin evens out per-OD difficulty)."""
    return np.arange(rank, n, world, dtype=np.int64)


def pack_rows(res, lmax):
    """[n_local, 2*lmax + 3] int32: rid row, time row, len, status, steps (kept for callers of the old one-shot
    exchange; gather_results takes the result dict itself)."""
    return torch.cat([res['rid'], res['time'], res['len'].unsqueeze(1), res['status'].unsqueeze(1),
                      res['steps'].unsqueeze(1)], dim=1).to(torch.int32).contiguous()


def gather_plan_bytes(n_total, world, lmax, width=None):
    """Device bytes gather_results needs at its peak for `n_total` OD pairs (trimmed row width `width`, lmax when
    unknown): the rank's own trimmed rows, the gathered block and the result in OD order, plus the three meta
    columns twice."""
    per = (n_total + world - 1) // world
    w = 2 * int(width if width is not None else lmax)
    return 4 * (per * w + 2 * world * per * w + 2 * world * per * 3 + per * 3)


def gather_results(res, n_total, lmax, rank, world, group=None):
    """The single exchange step of the path: every rank receives the trajectories of ALL OD pairs, in OD order.

    `res`: this rank's result dict (rid / time [n_local, lmax], len / status / steps [n_local]) for the OD pairs
    rank, rank + world, ...  Two collectives: first the three per-OD words (len, status, steps) of every rank, which
    also tell every rank the longest trajectory of the whole job; then the rid / time rows TRIMMED to that length
    (the mean trajectory has ~25 points, the padded row lmax = 256: a quarter or less of the bytes and of the
    buffers).  Returns rid / time [n_total, width] with width = the longest trajectory (<= lmax; columns beyond
    it would be all zero), len / status / steps [n_total].  A packed [n_local, 2*lmax+3] tensor (pack_rows) is
    accepted in place of the dict."""
    if torch.is_tensor(res):
        res = {'rid': res[:, :lmax], 'time': res[:, lmax:2 * lmax], 'len': res[:, 2 * lmax],
               'status': res[:, 2 * lmax + 1], 'steps': res[:, 2 * lmax + 2]}
    dev = res['rid'].device
    n_local = int(res['len'].shape[0])
    per = (n_total + world - 1) // world
    meta = torch.zeros((per, 3), dtype=torch.int32, device=dev)
    meta[:n_local, 0] = res['len']
    meta[:n_local, 1] = res['status']
    meta[:n_local, 2] = res['steps']
    meta_all = torch.empty((world * per, 3), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(meta_all, meta, group=group)
    width = int(meta_all[:, 0].max().clamp(min=1, max=lmax)) if n_total else 1
    rows = torch.zeros((per, 2, width), dtype=torch.int32, device=dev)
    rows[:n_local, 0] = res['rid'][:, :width]
    rows[:n_local, 1] = res['time'][:, :width]
    rows_all = torch.empty((world * per, 2, width), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(rows_all, rows, group=group)
    del rows
    # OD pair i sits at [i % world][i // world]: one transposing copy puts the job back into OD order
    full = rows_all.view(world, per, 2, width).transpose(0, 1).reshape(world * per, 2, width)[:n_total]
    del rows_all
    m = meta_all.view(world, per, 3).transpose(0, 1).reshape(world * per, 3)[:n_total]
    return {'rid': full[:, 0], 'time': full[:, 1], 'len': m[:, 0], 'status': m[:, 1], 'steps': m[:, 2],
            'width': width}


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
    return gather_results(res, n, lmax, rank, world, group)
