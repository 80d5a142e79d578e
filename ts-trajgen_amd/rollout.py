"""Batched stochastic rollouts on the device (tsg_rollout_run): the loops of Searcher.road_random_sample
(search.py:614-716) and DoubleLayerSearcher.region_random_sample (search.py:1163-1256) for many
(prefix, destination, default_len) rows at once -- the shape of the Monte-Carlo reward estimation in
rollout.py:42-48,86-92, which asks for (seq_len - 2) * rollout_times samples per generated trajectory.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


@torch.no_grad()
def sample_batch(tables, graph, prefixes, des, default_lens, uniforms=None, generator=None, return_probs=False,
                 max_state_bytes=8 << 30):
    """prefixes: list of (loc_list, tim_list) with >= 1 point each.  uniforms: optional (n, n_draws) floats in
    [0, 1) -- the k-th model evaluation of row b uses uniforms[b, k] (inverse-CDF draw, include/tsg.h);
    default: torch.rand on the device (`generator`).  Returns dict(loc, tim lists, status (n,) ndarray,
    n_drawn (n,) ndarray[, probs (n, n_draws, max_degree) tensor])."""
    dev = tables.device
    n = len(des)
    if n == 0:
        return {'loc': [], 'tim': [], 'status': np.zeros(0, np.int32), 'n_drawn': np.zeros(0, np.int32)}
    des_np = np.asarray(des, dtype=np.int32)
    dl_np = np.asarray(default_lens, dtype=np.int32)
    plen = np.array([len(p[0]) for p in prefixes], dtype=np.int32)
    if len(prefixes) != n or len(dl_np) != n or plen.min() < 1:
        raise ValueError('every row needs a prefix of at least one point')
    if any(len(p[0]) != len(p[1]) for p in prefixes):
        raise ValueError('prefix loc / tim lengths differ')
    max_step = np.maximum(20, dl_np)                                     # search.py:636
    lmax = int(max(plen.max(), max_step.max() + 1))
    lmax = (lmax + 7) // 8 * 8
    n_draws = int(max_step.max())
    nr = graph.num_roads
    rid = np.zeros((n, lmax), dtype=np.int32)
    tim = np.zeros((n, lmax), dtype=np.int32)
    for b, (pl, pt) in enumerate(prefixes):
        rid[b, :len(pl)] = pl
        tim[b, :len(pt)] = pt
    if rid.min() < 0 or rid.max() >= nr or des_np.min() < 0 or des_np.max() >= nr:
        raise IndexError('road id outside [0, %d)' % nr)
    if des_np.max() >= tables.n_nodes_h or nr > tables.c_model.n_roads:
        raise IndexError('road id outside the model tables')
    if tim.min() < 0 or tim.max() >= min(2880, tables.n_time):
        raise IndexError('time code outside [0, 2880)')
    start_code = tim[np.arange(n), plen - 1]
    if uniforms is not None:
        uniforms = torch.as_tensor(uniforms, dtype=torch.float32)
        if uniforms.dim() != 2 or uniforms.shape[0] != n:
            raise ValueError('uniforms must be (n, n_draws)')
        n_draws = int(uniforms.shape[1])
    hp, nl = tables.hp, tables.n_layers
    per_row = 2 * nl * lmax * hp * 4
    chunk = max(1, min(n, int(max_state_bytes // per_row)))
    L = _lib.lib()
    i32 = torch.int32
    out_loc, out_tim = [None] * n, [None] * n
    status = np.zeros(n, dtype=np.int32)
    drawn = np.zeros(n, dtype=np.int32)
    probs = torch.zeros((n, n_draws, graph.max_degree), dtype=torch.float32, device=dev) if return_probs else None
    stream = torch.cuda.current_stream(dev)
    states = [torch.zeros((chunk * lmax, hp), dtype=torch.float32, device=dev) for _ in range(2 * nl)]
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        m = hi - lo
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        d_rid, d_tim = t(rid[lo:hi]), t(tim[lo:hi])
        d_len, d_des, d_dl, d_sc = t(plen[lo:hi]), t(des_np[lo:hi]), t(dl_np[lo:hi]), t(start_code[lo:hi])
        d_ns = torch.zeros(m, dtype=i32, device=dev)
        d_nd = torch.zeros(m, dtype=i32, device=dev)
        d_st = torch.zeros(m, dtype=i32, device=dev)
        if uniforms is None:
            d_u = torch.rand((m, max(n_draws, 1)), dtype=torch.float32, device=dev, generator=generator)
        else:
            d_u = uniforms[lo:hi].contiguous().to(dev)
            if d_u.numel() == 0:
                d_u = torch.zeros((m, 1), dtype=torch.float32, device=dev)
        d_prev = torch.zeros(m, dtype=torch.int64, device=dev)
        d_out = torch.zeros(m, dtype=torch.int64, device=dev)
        d_r = torch.zeros(m, dtype=i32, device=dev)
        d_t = torch.zeros(m, dtype=i32, device=dev)
        r = _lib.Rollout()
        r.n_traj, r.lmax, r.n_draws = m, lmax, n_draws
        r.out_rid, r.out_time, r.len = _lib.ptr(d_rid), _lib.ptr(d_tim), _lib.ptr(d_len)
        r.n_state, r.n_drawn, r.status = _lib.ptr(d_ns), _lib.ptr(d_nd), _lib.ptr(d_st)
        r.des, r.default_len, r.start_code = _lib.ptr(d_des), _lib.ptr(d_dl), _lib.ptr(d_sc)
        r.uniform = _lib.ptr(d_u)
        d_p = probs[lo:hi] if return_probs else None
        r.prob_log = _lib.ptr(d_p)
        for l in range(nl):
            r.h[l] = states[2 * l].data_ptr()
            r.c[l] = states[2 * l + 1].data_ptr()
        r.state_bytes = states[0].numel() * 4
        r.prev_idx, r.out_idx, r.rid, r.tcode = _lib.ptr(d_prev), _lib.ptr(d_out), _lib.ptr(d_r), _lib.ptr(d_t)
        n_steps = int(plen[lo:hi].max() + max_step[lo:hi].max() + 1)
        _lib.check(L.tsg_rollout_run(C.byref(tables.c_model), C.byref(graph.c_graph), C.byref(r), n_steps,
                                     C.c_void_p(stream.cuda_stream)), 'tsg_rollout_run')
        h_rid, h_tim = d_rid.cpu().numpy(), d_tim.cpu().numpy()
        h_len, h_st = d_len.cpu().numpy(), d_st.cpu().numpy()
        if (h_st == 0).any() or (h_st & _lib.STATUS_OVERFLOW).any():
            raise RuntimeError('rollout did not finish inside its step / draw / row budget')
        for b in range(m):
            out_loc[lo + b] = h_rid[b, :h_len[b]].tolist()
            out_tim[lo + b] = h_tim[b, :h_len[b]].tolist()
        status[lo:hi] = h_st
        drawn[lo:hi] = d_nd.cpu().numpy()
    res = {'loc': out_loc, 'tim': out_tim, 'status': status, 'n_drawn': drawn}
    if return_probs:
        res['probs'] = probs
    return res
