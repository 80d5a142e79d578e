"""Host side of the batched A* engine: device tables (model + graph), workspaces, the superstep loop.

Everything numeric happens in libtsg.so; this module only lays tensors out the way include/tsg.h
documents (gate-interleaved, hidden padded to a multiple of 64), sizes the workspaces and polls the
completion counter.  PyTorch is used for device memory, streams and one-off layout shuffles.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib

DEFAULT_SPEED = 8.334           # search.py:11


def _ceil(a, b):
    return (a + b - 1) // b * b


def gate_permutation(hidden, hp):
    """Index tensor `src` of length 4*hp: interleaved column c takes torch gate-major column src[c]
    (or -1 = zero padding).  Layout: c = (u // 32) * 128 + gate * 32 + u % 32."""
    c = np.arange(4 * hp)
    tile, rem = c // 128, c % 128
    gate, u = rem // 32, tile * 32 + rem % 32
    return np.where(u < hidden, gate * hidden + u, -1)


def _permute_cols(w_t, hidden, hp):
    """w_t: [K, 4*hidden] (torch gate-major columns) -> [K, 4*hp] interleaved, zero padded."""
    src = gate_permutation(hidden, hp)
    out = torch.zeros((w_t.shape[0], 4 * hp), dtype=torch.float32, device=w_t.device)
    valid = torch.from_numpy(np.flatnonzero(src >= 0)).to(w_t.device)
    out[:, valid] = w_t[:, torch.from_numpy(src[src >= 0]).to(w_t.device)]
    return out


def _unit_major_cols(w_t, hidden, hp):
    """w_t: [K, 4*hidden] (torch gate-major columns) -> [K, 4*hp] with the four gates of a unit side by side
    (column 4*u + gate), zero padded: the layout of tsg_model.xproj / tproj, whose rows the recurrence reads as one
    16-byte piece per unit."""
    out = torch.zeros((w_t.shape[0], hp, 4), dtype=torch.float32, device=w_t.device)
    out[:, :hidden, :] = w_t.reshape(w_t.shape[0], 4, hidden).permute(0, 2, 1)
    return out.reshape(w_t.shape[0], 4 * hp)


def heap_words(n):
    """8-byte words of a slot's queue stretch that holds n entries: the blocked heap of tsg_search.hip for the first
    twelve levels (entries 1..4095: three levels of the binary heap per 64-byte block, blocks numbered like an 8-ary
    heap, one block of offset in front and one spare block behind), the entries beyond in plain heap order behind
    the blocks; rounded to whole 128-byte lines.  tsg_heap_room (tsg_common.h) is its inverse."""
    n = max(int(n), 1)
    if n > 4095:
        return _ceil(8 * (585 + 2) + 8 + (n - 4095), 16)
    d = n.bit_length() - 1
    lvl = d // 3
    p8 = 1 << (3 * lvl)
    blocks = (p8 - 1) // 7 + ((n - p8 + 1) if d == 3 * lvl else p8)
    return _ceil(8 * (blocks + 2), 16)


def _pad_rows(w, rows):
    if w.shape[0] == rows:
        return w.contiguous()
    out = torch.zeros((rows,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
    out[:w.shape[0]] = w
    return out


def gemm(a, bt, bias=None):
    """C = A . Bt (+ bias) through tsg_gemm_f32 (fp32 MFMA)."""
    L = _lib.lib()
    a = a.contiguous().float()
    bt = bt.contiguous().float()
    m, k = a.shape
    n = bt.shape[1]
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    if bias is not None:
        bias = bias.contiguous().float()
    with torch.cuda.device(a.device):
        _lib.check(L.tsg_gemm_f32(_lib.ptr(a), k, _lib.ptr(bt), n, _lib.ptr(bias), _lib.ptr(out), n, m, n, k,
                                  _lib.current_stream()), 'tsg_gemm_f32')
    return out


def split_f16_image(w_t):
    """[K][4Hp] fp32 weights -> the split-f16 LDS image of tsg_model.whh0_s / wl_s:
    major = f16(w), minor = f16((w - major) * 2^11) (values below the f16 normal range go entirely to the minor
    part), tiled [4Hp/128][K/32][major|minor][128 cols][32 k]."""
    K, N = w_t.shape
    if K % 32 or N % 128:
        raise ValueError('split image needs K % 32 == 0 and N % 128 == 0')
    if float(w_t.abs().max()) >= 6.0e4:
        raise ValueError('LSTM weight outside the f16 range: use state_format=0')
    tiny = w_t.abs() < 2.0 ** -14
    major = torch.where(tiny, torch.zeros_like(w_t), w_t).to(torch.float16)
    minor = ((w_t - major.float()) * 2048.0).to(torch.float16)

    def tile(x):
        return x.t().reshape(N // 128, 128, K // 32, 32).permute(0, 2, 1, 3)

    return torch.stack([tile(major), tile(minor)], dim=2).contiguous()


class ModelTables(object):
    """Device tables of one GeneratorV4 (generator_v4.py:16-29) in the layout of `tsg_model`."""

    def __init__(self, generator):
        g, h = generator.function_g, generator.function_h
        dev = generator.w1.device
        if dev.type != 'cuda':
            raise RuntimeError('ModelTables needs the generator on a HIP device (no CPU fallback)')
        self.device = dev
        with torch.no_grad():
            H = g.hidden_size
            hp = _ceil(H, 64)
            er = g.road_emb_size
            nl = g.lstm_layer_num
            if nl > _lib.TSG_MAX_LAYERS:
                raise ValueError('at most %d LSTM layers' % _lib.TSG_MAX_LAYERS)
            sd = {k: v.detach().float() for k, v in g.state_dict().items()}
            w_ih0 = sd['lstm.weight_ih_l0']                                  # (4H, Er+Et)
            self.xproj = gemm(sd['road_emb.weight'], _unit_major_cols(w_ih0[:, :er].t().contiguous(), H, hp))
            b0 = _unit_major_cols((sd['lstm.bias_ih_l0'] + sd['lstm.bias_hh_l0']).unsqueeze(0), H, hp).squeeze(0)
            self.tproj = gemm(sd['time_emb.weight'], _unit_major_cols(w_ih0[:, er:].t().contiguous(), H, hp), b0)
            self.whh0_t = _pad_rows(_permute_cols(sd['lstm.weight_hh_l0'].t().contiguous(), H, hp), hp)
            self.wl_t = [None] * _lib.TSG_MAX_LAYERS
            self.bl = [None] * _lib.TSG_MAX_LAYERS
            for l in range(1, nl):
                top = _pad_rows(_permute_cols(sd['lstm.weight_ih_l%d' % l].t().contiguous(), H, hp), hp)
                bot = _pad_rows(_permute_cols(sd['lstm.weight_hh_l%d' % l].t().contiguous(), H, hp), hp)
                self.wl_t[l] = torch.cat([top, bot], dim=0).contiguous()
                self.bl[l] = _permute_cols((sd['lstm.bias_ih_l%d' % l] + sd['lstm.bias_hh_l%d' % l]).unsqueeze(0),
                                           H, hp).squeeze(0).contiguous()
            self.whh0_s = split_f16_image(self.whh0_t)
            self.wl_s = [None] * _lib.TSG_MAX_LAYERS
            for l in range(1, nl):
                self.wl_s[l] = split_f16_image(self.wl_t[l])
            ow = sd['out_fc.weight']                                        # (road_num, H)
            self.out_w = torch.zeros((ow.shape[0], hp), dtype=torch.float32, device=dev)
            self.out_w[:, :H] = ow
            self.out_b = sd['out_fc.bias'].contiguous()
            # function H: cached GAT node embedding, then the factored emb_fc (distance_gat_fc.py:129-131)
            h._setup_node_emb()
            emb = h.node_emb_feature
            F = emb.shape[1]
            we = h.emb_fc.weight.detach().float()                           # (10, 2F)
            self.pfrom = gemm(emb, we[:, :F].t().contiguous())
            self.pto = gemm(emb, we[:, F:].t().contiguous(), h.emb_fc.bias.detach().float())
            self.hparams = torch.cat([h.dis_fc.weight.detach().float().reshape(-1),
                                      h.dis_fc.bias.detach().float().reshape(-1),
                                      h.out_fc.weight.detach().float().reshape(-1),
                                      h.out_fc.bias.detach().float().reshape(-1),
                                      generator.w1.detach().float().reshape(-1)]).contiguous()
            self.hidden, self.hp, self.n_layers = H, hp, nl
            self.n_roads_g = ow.shape[0]
            self.n_nodes_h = emb.shape[0]
            self.n_time = sd['time_emb.weight'].shape[0]
        m = _lib.Model()
        m.n_roads = min(self.n_roads_g, self.n_nodes_h)
        m.n_time, m.hp, m.n_layers = self.n_time, hp, nl
        m.xproj, m.tproj, m.whh0_t = _lib.ptr(self.xproj), _lib.ptr(self.tproj), _lib.ptr(self.whh0_t)
        for l in range(1, nl):
            m.wl_t[l] = self.wl_t[l].data_ptr()
            m.bl[l] = self.bl[l].data_ptr()
            m.wl_s[l] = self.wl_s[l].data_ptr()
        m.whh0_s = _lib.ptr(self.whh0_s)
        m.out_w, m.out_b = _lib.ptr(self.out_w), _lib.ptr(self.out_b)
        m.pfrom, m.pto, m.hparams = _lib.ptr(self.pfrom), _lib.ptr(self.pto), _lib.ptr(self.hparams)
        self.c_model = m

    @torch.no_grad()
    def predict(self, trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask=None, logits=False):
        """GeneratorV4.predict for (B, L) prefixes and (B, C) candidates; returns (p, g, h), each (B, C); with
        `logits` also function G's and function H's candidate scores before their softmax.

        With `trace_mask` (1 = real, 0 = padding; function_g_fc.py:162-180) the recurrence runs over each row's
        real positions only and the query is the last real hidden state, as the reference's packed LSTM does.
        The attention, however, runs over EVERY position of the re-padded LSTM output: the reference assigns its
        masked_fill to `mask`, not to the weights (function_g_fc.py:33), so the padded positions (all-zero rows
        out of pad_packed_sequence) keep logit 0 and their share exp(0 - max) of the softmax.  That is what is
        reproduced here: a row of length n in a batch padded to T attends over its n states plus T - n zero rows.
        As in the reference, the mask must be as wide as the longest row (pad_packed_sequence pads to the longest
        row of the batch and masked_fill then needs equal shapes)."""
        L = _lib.lib()
        dev = self.device
        B, T = trace_loc.shape
        Cn = candidate_set.shape[1]
        if Cn > _lib.TSG_MAX_DEGREE:
            raise ValueError('at most %d candidates per row' % _lib.TSG_MAX_DEGREE)
        loc = trace_loc.to(dev).to(torch.int32).contiguous()
        tim = trace_time.to(dev).to(torch.int32).contiguous()
        if trace_mask is not None:
            mk = trace_mask.to(dev) != 0
            lens = mk.sum(dim=1).to(torch.int64)
            if int(lens.min()) < 1:
                raise RuntimeError('every row of a masked batch needs at least one real position '
                                   '(pack_padded_sequence, function_g_fc.py:165)')
            if int(lens.max()) != T:
                raise RuntimeError('trace_mask must be as wide as the longest row: the reference re-pads the LSTM '
                                   'output to the longest row and masked_fill needs equal shapes '
                                   '(function_g_fc.py:33,167)')
            pos0 = torch.arange(T, device=dev).unsqueeze(0)
            if not bool((mk == (pos0 < lens.unsqueeze(1))).all()):
                raise ValueError('trace_mask must be 1 on a prefix of every row (right padding)')
            loc = torch.where(mk, loc, torch.zeros_like(loc))        # padding ids are never run through the model
            tim = torch.where(mk, tim, torch.zeros_like(tim))
        else:
            lens = torch.full((B,), T, dtype=torch.int64, device=dev)
        if int(loc.min()) < 0 or int(loc.max()) >= self.xproj.shape[0]:
            raise IndexError('road id outside the road embedding table')
        if int(tim.min()) < 0 or int(tim.max()) >= self.n_time:
            raise IndexError('time code outside the time embedding table')
        cand = candidate_set.to(dev).to(torch.int32).contiguous()
        if int(cand.min()) < 0 or int(cand.max()) >= self.c_model.n_roads:
            raise IndexError('candidate id outside the model tables')
        d32 = des.to(dev).to(torch.int32).reshape(B).contiguous()
        if int(d32.min()) < 0 or int(d32.max()) >= self.n_nodes_h:
            raise IndexError('destination id outside the node embedding table')
        dis = candidate_dis.to(dev).float().contiguous()
        hp, nl = self.hp, self.n_layers
        # state rows b * T + t; one more all-zero row at the end stands for every padded position
        hs = [torch.zeros((B * T + 1, hp), dtype=torch.float32, device=dev) for _ in range(nl)]
        cs = [torch.zeros((B * T + 1, hp), dtype=torch.float32, device=dev) for _ in range(nl)]
        zero_row = B * T
        base = torch.arange(B, dtype=torch.int64, device=dev) * T
        stream = _lib.current_stream()
        with torch.cuda.device(dev):
            for t in range(T):
                live = lens > t                     # rows whose position t is real
                out_idx = torch.where(live, base + t, torch.full_like(base, -1)).contiguous()
                prev_idx = (base + t - 1).contiguous() if t > 0 else torch.full_like(base, -1)
                rid_t = loc[:, t].contiguous()
                tc_t = tim[:, t].contiguous()
                for l in range(nl):
                    s = _lib.LstmStep()
                    s.layer, s.n_rows_host, s.n_rows_dev = l, B, None
                    s.prev_idx, s.out_idx = _lib.ptr(prev_idx), _lib.ptr(out_idx)
                    s.rid, s.tcode = _lib.ptr(rid_t), _lib.ptr(tc_t)
                    s.h_below = _lib.ptr(hs[l - 1]) if l > 0 else None
                    s.h, s.c = _lib.ptr(hs[l]), _lib.ptr(cs[l])
                    s.state_rows = B * T + 1
                    _lib.check(L.tsg_lstm_step_batched(C.byref(self.c_model), C.byref(s), stream),
                               'tsg_lstm_step_batched')
            # chain of row b: T - len zero rows (the padded positions), then its real states, the last one = query
            pos = torch.arange(T, dtype=torch.int64, device=dev).unsqueeze(0).expand(B, T)
            pad = T - lens
            real = pos >= pad.unsqueeze(1)
            chain = torch.where(real, base.unsqueeze(1) + pos - pad.unsqueeze(1),
                                torch.full_like(pos, zero_row)).reshape(-1).contiguous()
            chain_ptr = (torch.arange(B + 1, dtype=torch.int32, device=dev) * T).contiguous()
            cand_ptr = (torch.arange(B + 1, dtype=torch.int32, device=dev) * Cn).contiguous()
            prob = torch.empty((B, Cn), dtype=torch.float32, device=dev)
            pg = torch.empty_like(prob)
            ph = torch.empty_like(prob)
            lg = torch.empty_like(prob) if logits else None
            lh = torch.empty_like(prob) if logits else None
            sb = _lib.ScoreBatch()
            sb.n_rows = B
            sb.chain_ptr, sb.chain, sb.h_top = _lib.ptr(chain_ptr), _lib.ptr(chain), _lib.ptr(hs[nl - 1])
            sb.cand_ptr, sb.cand, sb.dis, sb.des = _lib.ptr(cand_ptr), _lib.ptr(cand), _lib.ptr(dis), _lib.ptr(d32)
            sb.prob, sb.prob_g, sb.prob_h = _lib.ptr(prob), _lib.ptr(pg), _lib.ptr(ph)
            sb.logit_g, sb.logit_h = _lib.ptr(lg), _lib.ptr(lh)
            _lib.check(L.tsg_score_candidates(C.byref(self.c_model), C.byref(sb), stream), 'tsg_score_candidates')
        if logits:
            return prob, pg, ph, lg, lh
        return prob, pg, ph

    @torch.no_grad()
    def h_pair_scores(self, from_node, to_node, distance):
        """DistanceGatFC.predict_cache (distance_gat_fc.py:118-134): function H's score of (from, to, distance)
        triples, before any softmax; 1-D tensors of equal length."""
        L = _lib.lib()
        dev = self.device
        f = from_node.to(dev).to(torch.int32).reshape(-1).contiguous()
        t = to_node.to(dev).to(torch.int32).reshape(-1).contiguous()
        d = distance.to(dev).float().reshape(-1).contiguous()
        n = int(f.numel())
        if int(t.numel()) != n or int(d.numel()) != n:
            raise ValueError('from_node, to_node and distance must have the same length')
        out = torch.empty(n, dtype=torch.float32, device=dev)
        if n == 0:
            return out
        if int(f.min()) < 0 or int(f.max()) >= self.c_model.n_roads or int(t.min()) < 0 or int(t.max()) >= self.n_nodes_h:
            raise IndexError('node id outside the node embedding table')
        with torch.cuda.device(dev):
            _lib.check(L.tsg_h_pair_scores(C.byref(self.c_model), n, _lib.ptr(f), _lib.ptr(t), _lib.ptr(d),
                                           _lib.ptr(out), _lib.current_stream()), 'tsg_h_pair_scores')
        return out


class GraphTables(object):
    """Device substrate of a road network in the layout of `tsg_graph`."""

    def __init__(self, device, adjacent_list, road_center_gps, road_length, road_time_distribution,
                 num_roads=None, rid_label=None):
        from .synth import csr_from_adjacent_list
        self.device = torch.device(device)
        if num_roads is None:
            num_roads = int(road_time_distribution.shape[1])
        n = self.num_roads = int(num_roads)
        indptr, indices = csr_from_adjacent_list(adjacent_list, n)
        if len(indices) and (indices.min() < 0 or indices.max() >= n):
            raise ValueError('adjacent_list refers to a road id outside [0, %d)' % n)
        deg = np.diff(indptr)
        self.max_degree = int(max(1, deg.max() if len(deg) else 1))
        if self.max_degree > _lib.TSG_MAX_DEGREE:
            raise ValueError('out-degree %d exceeds the kernel limit %d' % (self.max_degree, _lib.TSG_MAX_DEGREE))
        lonlat = np.zeros((n, 2), dtype=np.float64)
        for r in range(n):
            gps = road_center_gps.get(str(r)) if road_center_gps is not None else None
            if gps is not None:
                lonlat[r, 0], lonlat[r, 1] = gps[0], gps[1]
        # search.py:75 (astype(int)) and :376-381: cost = rtd if rtd != 1 else int(round(length / 8.334))
        rtd = np.asarray(road_time_distribution).astype(int)[:, :n]
        default = np.zeros(n, dtype=np.int64)
        for r in range(n):
            ln = road_length.get(str(r)) if road_length is not None else None
            default[r] = int(round(ln / DEFAULT_SPEED)) if ln is not None else 1
        cost = np.where(rtd != 1, rtd, default[None, :]).astype(np.int32)
        self.indptr_np, self.indices_np = indptr, indices
        dev = self.device
        self.indptr = torch.from_numpy(indptr).to(dev)
        self.indices = torch.from_numpy(indices if len(indices) else np.zeros(1, np.int32)).to(dev)
        f = 1.0 / 298.257223563
        u = np.arctan((1.0 - f) * np.tan(lonlat[:, 1] * (np.pi / 180.0)))
        geo4 = np.stack([lonlat[:, 0], lonlat[:, 1], np.sin(u), np.cos(u)], axis=1)
        self.geo = torch.from_numpy(np.ascontiguousarray(geo4)).to(dev)
        self.cost = torch.from_numpy(np.ascontiguousarray(cost)).to(dev)
        self.label = None
        if rid_label is not None:
            self.set_labels(rid_label)
        self._build()

    @classmethod
    def for_regions(cls, device, region_adjacent_list, region_dist, region_time_distribution, num_regions=None):
        """Region-level graph of DoubleLayerSearcher.astar_search_in_region (search.py:936-1117):
        successors = keys of region_adjacent_list[str(r)] in dict order (:992), candidate distance =
        region_dist[c][des] / 100 (:1049), per-candidate travel time where the region's time distribution
        is unknown (:1073-1078), regions whose key holds no successor are closed without a step (:993-995)."""
        self = cls.__new__(cls)
        self.device = torch.device(device)
        region_dist = np.asarray(region_dist, dtype=np.float64)
        n = self.num_roads = int(num_regions if num_regions is not None else region_dist.shape[0])
        if region_dist.shape[0] < n or region_dist.shape[1] < n:
            raise ValueError('region_dist smaller than the number of regions')
        indptr = np.zeros(n + 1, dtype=np.int32)
        lists = []
        empty = np.zeros(n, dtype=np.uint8)
        for r in range(n):
            d = region_adjacent_list.get(str(r))
            keys = [int(k) for k in d.keys()] if d is not None else []
            if d is not None and len(keys) == 0:
                empty[r] = 1
            lists.append(keys)
            indptr[r + 1] = indptr[r] + len(keys)
        indices = np.array([c for l in lists for c in l], dtype=np.int32)
        if len(indices) and (indices.min() < 0 or indices.max() >= n):
            raise ValueError('region_adjacent_list refers to a region outside [0, %d)' % n)
        self.max_degree = int(max(1, np.diff(indptr).max() if n else 1))
        if self.max_degree > _lib.TSG_MAX_DEGREE:
            raise ValueError('region out-degree %d exceeds the kernel limit %d' % (self.max_degree, _lib.TSG_MAX_DEGREE))
        rtd = np.asarray(region_time_distribution).astype(int)[:, :n]
        cost = np.where(rtd != 1, rtd, -1).astype(np.int32)          # -1: use the per-pair cost
        dist = np.ascontiguousarray(region_dist[:n, :n])
        pair = np.rint(dist / DEFAULT_SPEED).astype(np.int32)        # int(round(.)): half-to-even like Python
        dev = self.device
        self.indptr_np, self.indices_np = indptr, indices
        self.indptr = torch.from_numpy(indptr).to(dev)
        self.indices = torch.from_numpy(indices if len(indices) else np.zeros(1, np.int32)).to(dev)
        self.geo = None
        self.cost = torch.from_numpy(np.ascontiguousarray(cost)).to(dev)
        self.label = None
        self.dist_table = torch.from_numpy(dist).to(dev)
        self.pair_cost = torch.from_numpy(np.ascontiguousarray(pair)).to(dev)
        self.empty_key = torch.from_numpy(empty).to(dev)
        self._build()
        return self

    def set_labels(self, rid_label):
        lab = np.asarray(rid_label, dtype=np.int32)
        if lab.shape != (self.num_roads,):
            raise ValueError('rid_label must have one entry per road')
        self.label = torch.from_numpy(lab).to(self.device)
        self._build()

    def _build(self):
        g = _lib.Graph()
        g.n_roads, g.max_degree = self.num_roads, self.max_degree
        g.indptr, g.indices, g.geo = _lib.ptr(self.indptr), _lib.ptr(self.indices), _lib.ptr(self.geo)
        g.cost, g.label = _lib.ptr(self.cost), _lib.ptr(self.label)
        g.dist_table = _lib.ptr(getattr(self, 'dist_table', None))
        g.pair_cost = _lib.ptr(getattr(self, 'pair_cost', None))
        g.empty_key = _lib.ptr(getattr(self, 'empty_key', None))
        self.c_graph = g


class AStarEngine(object):
    """Workspaces + superstep loop of the batched A* (tsg_engine_run).

    The recurrent state of expanded nodes lives in pages of 64 rows drawn from one pool (see __init__ and
    include/tsg.h); `streams` is accepted for compatibility and ignored (one engine per device)."""

    PAGE_ROWS = 64      # TSG_PAGE_ROWS

    def __init__(self, tables, graph, slots, max_step, max_prefix=1, lmax=256, streams=1, prune=True, use_graph=True,
                 state_format=1, max_pops=None, pool_pages=None, admit_reserve=None, ramp_steps=0, stream=None):
        """pool_pages: state pages (64 rows each) shared by all searches; None reserves the worst case
        (slots * ceil(exp_cap / 64): no search ever waits).  A smaller pool lets more searches fit in HBM (a
        search holds only the pages it has filled); see include/tsg.h, tsg_engine."""
        self.tables, self.graph = tables, graph
        dev = self.device = tables.device
        n = graph.num_roads
        self.slots = B = int(slots)   # one engine per device (sub-engines on side streams bought nothing: DESIGN.md)
        self.max_step, self.max_prefix, self.lmax = int(max_step), int(max_prefix), int(lmax)
        self.exp_cap = self.max_prefix + self.max_step
        if max_pops is None:
            max_pops = int(os.environ.get('TSG_MAX_POPS', '4'))
        self.max_pops = int(max_pops)
        self.node_cap = int(min(n, 1 + self.max_step * graph.max_degree)) + 1
        # queue entries per search: one push per successor per expansion at most, + the start node
        self.heap_entries = 2 + self.max_step * graph.max_degree
        self.heap_cap = heap_words(self.heap_entries)       # words of the slot's stretch (blocked heap)
        if self.node_cap >= (1 << 20):
            raise ValueError('node_cap %d exceeds 2^20-1' % self.node_cap)
        hp, nl = tables.hp, tables.n_layers
        i32, f32 = torch.int32, torch.float32
        self.pages_per_slot = pps = (self.exp_cap + self.PAGE_ROWS - 1) // self.PAGE_ROWS
        self.n_pages = int(pool_pages) if pool_pages is not None else B * pps
        if self.n_pages < pps:
            raise ValueError('the page pool must hold at least one whole search (%d pages)' % pps)
        self.n_pages = min(self.n_pages, B * pps)
        self.admit_reserve = int(admit_reserve) if admit_reserve is not None else (
            0 if self.n_pages == B * pps else max(pps, self.n_pages // 12))
        self.ramp_steps = int(ramp_steps)
        rows = self.n_pages * self.PAGE_ROWS
        self.slot_state = torch.zeros((B, 40), dtype=i32, device=dev)
        self.heap = torch.empty((B, self.heap_cap), dtype=torch.int64, device=dev)
        self.table = torch.zeros((B, n), dtype=torch.int64, device=dev)
        self.nodes = torch.empty((B, self.node_cap, 8), dtype=i32, device=dev)
        self.exp_meta = torch.empty((B, self.exp_cap, 4), dtype=i32, device=dev)
        self.up_levels = 64     # TSG_UP_LEVELS: ancestors listed per expanded node
        w1 = float(tables.hparams[41])
        self.prune_spread = (abs(1.0 - w1) + abs(w1)) * (1.0 + 1e-6) if prune else 0.0
        self.exp_up = torch.empty((rows, self.up_levels), dtype=i32, device=dev)
        self.h = [torch.empty((rows, hp), dtype=f32, device=dev) for _ in range(nl)]
        self.c = [torch.empty((rows, hp), dtype=f32, device=dev) for _ in range(nl)]
        self.slot_pages = torch.empty((B, pps), dtype=i32, device=dev)
        self.page_ring = torch.empty(self.n_pages, dtype=i32, device=dev)
        self.retry = None       # sized per OD batch in run()
        self.page_flags = torch.zeros((2, 256, 32), dtype=i32, device=dev)
        nreg = (B + 4095) // 4096                   # TSG_REGION_SLOTS: the work list is kept per region of slots
        self.items = torch.zeros((nreg * 4096, 12), dtype=i32, device=dev)
        self.item_counts = torch.zeros((2, nreg, 32), dtype=i32, device=dev)
        self.cand_lp = torch.empty((B, graph.max_degree), dtype=f32, device=dev)
        self.cand_id = torch.zeros((B, graph.max_degree), dtype=i32, device=dev)
        self.row_meta = torch.empty((rows, 2), dtype=i32, device=dev)
        self.counters = torch.zeros((2, 16), dtype=i32, device=dev)   # row 0: page control, row 1: OD queue / totals
        e = _lib.Engine()
        e.slots, e.exp_cap, e.node_cap, e.heap_cap = B, self.exp_cap, self.node_cap, self.heap_cap
        for name in ('slot_state', 'heap', 'table', 'nodes', 'exp_meta', 'exp_up', 'items', 'item_counts', 'cand_lp', 'cand_id',
                     'row_meta', 'slot_pages',
                     'page_ring', 'page_flags'):
            t = getattr(self, name)
            setattr(e, name, t.data_ptr())
            setattr(e, name + '_bytes', t.numel() * t.element_size())
        for l in range(nl):
            e.h[l] = self.h[l].data_ptr()
            e.c[l] = self.c[l].data_ptr()
        e.state_bytes = rows * hp * 4
        e.n_pages, e.pages_per_slot = self.n_pages, pps
        e.admit_reserve, e.ramp_steps = self.admit_reserve, self.ramp_steps
        e.up_levels = self.up_levels
        e.prune_spread = self.prune_spread
        e.state_format = int(state_format)
        e.max_pops = self.max_pops
        e.counters = self.counters[0].data_ptr()
        e.counters_bytes = 64
        e.shared_counters = self.counters[1].data_ptr()
        self.c_engines = [e]
        # `stream`: the stream this engine launches on (a Searcher that runs several engines side by side keeps one
        # stream per engine for its whole life: which hardware queue a new stream lands on is the runtime's choice,
        # and two engines whose streams share a queue do not overlap)
        self.side_streams = [stream if stream is not None else torch.cuda.Stream(device=dev)]
        self.use_graph = bool(use_graph) and os.environ.get('TSG_NO_GRAPH', '') != '1'   # env: diagnostics only
        self.state_format = int(state_format)
        self.split_calls = False        # tests: drive the supersteps through tsg_frontier_expand_select + tsg_score_frontier
        self._pinned = [torch.zeros((2, 16), dtype=i32).pin_memory() for _ in range(2)]
        self.last_stats = {}

    def close(self):
        """Give the workspaces back now (the object may sit in reference cycles, and a Searcher that replaces its
        engine needs the memory before the garbage collector would get to it)."""
        for name, val in list(self.__dict__.items()):
            if torch.is_tensor(val) or (isinstance(val, list) and val and all(torch.is_tensor(v) for v in val)):
                setattr(self, name, None)
        self.c_engines = []

    @staticmethod
    def fixed_bytes_per_slot(graph, max_step, max_prefix=1):
        """Per-search workspaces other than state pages: queue, node records, rid table, expansion records."""
        exp_cap = max_prefix + max_step
        node_cap = min(graph.num_roads, 1 + max_step * graph.max_degree) + 1
        heap_cap = heap_words(2 + max_step * graph.max_degree)
        pps = (exp_cap + 63) // 64
        return heap_cap * 8 + graph.num_roads * 8 + node_cap * 32 + exp_cap * 16 + pps * 4 + 40 * 4 + 48 \
            + graph.max_degree * 8

    @staticmethod
    def bytes_per_page(tables):
        """One state page: 64 rows of (h, c) per layer + the 64-entry ancestor list of each row."""
        return 64 * (2 * tables.n_layers * tables.hp * 4 + 64 * 4 + 8)

    @staticmethod
    def bytes_per_slot(tables, graph, max_step, max_prefix=1):
        """Worst-case reservation: the fixed workspaces plus ceil(exp_cap / 64) state pages."""
        pps = (max_prefix + max_step + 63) // 64
        return AStarEngine.fixed_bytes_per_slot(graph, max_step, max_prefix) + pps * AStarEngine.bytes_per_page(tables)

    @torch.no_grad()
    def reconstruct_paths(self, slot, node, last_rid=None, last_tcode=None, lmax=None):
        """Trajectories out of the expansion records (tsg_reconstruct_paths): request i = (search slot, index of
        one of its expanded nodes) [+ one more point (last_rid[i], last_tcode[i]) behind it].  Returns device
        tensors (rid [n, lmax], time [n, lmax], len [n]).  Valid for searches in flight and for finished slots
        whose records have not been reused by a later OD pair."""
        L = _lib.lib()
        dev = self.device
        lmax = int(lmax or self.lmax)

        def dv(a):
            return None if a is None else torch.as_tensor(np.asarray(a, dtype=np.int32)).to(dev)
        slot_t, node_t, lr_t, lt_t = dv(slot), dv(node), dv(last_rid), dv(last_tcode)
        n = int(slot_t.numel())
        if int(node_t.numel()) != n or (lr_t is not None and int(lr_t.numel()) != n) or (
                lt_t is not None and int(lt_t.numel()) != n):
            raise ValueError('slot / node / last_rid / last_tcode must have one entry per request')
        rid = torch.zeros((n, lmax), dtype=torch.int32, device=dev)
        tim = torch.zeros((n, lmax), dtype=torch.int32, device=dev)
        ln = torch.zeros(n, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.tsg_reconstruct_paths(C.byref(self.c_engines[0]), n, _lib.ptr(slot_t), _lib.ptr(node_t),
                                               _lib.ptr(lr_t), _lib.ptr(lt_t), lmax, _lib.ptr(rid), _lib.ptr(tim),
                                               _lib.ptr(ln), _lib.current_stream()), 'tsg_reconstruct_paths')
        return rid, tim, ln

    def run(self, prefix_ptr, prefix_rid, prefix_time, des, default_len, max_step, allow_label=None,
            poplog_cap=0, chunk=32, return_device=False, profile_every=0):
        """Run all OD pairs; returns dict(rid [n, lmax], time, len, status, steps[, poplog_*])."""
        L = _lib.lib()
        dev = self.device
        i32 = torch.int32

        def dv(a):
            if torch.is_tensor(a):
                return a.to(dev).to(i32).contiguous()
            return torch.from_numpy(np.array(a, dtype=np.int32, copy=True)).to(dev)
        prefix_ptr, prefix_rid, prefix_time = dv(prefix_ptr), dv(prefix_rid), dv(prefix_time)
        des, default_len, max_step_t = dv(des), dv(default_len), dv(max_step)
        n_od = int(des.numel())
        if n_od == 0:
            z = torch.zeros((0,), dtype=i32)
            return {'rid': torch.zeros((0, self.lmax), dtype=i32), 'time': torch.zeros((0, self.lmax), dtype=i32),
                    'len': z, 'status': z, 'steps': z}
        # host-side validation of everything the kernels index with
        n = self.graph.num_roads
        plen = (prefix_ptr[1:] - prefix_ptr[:-1])
        if int(prefix_ptr.numel()) != n_od + 1 or int(plen.min()) < 1 or int(plen.max()) > self.max_prefix:
            raise ValueError('every OD needs a prefix of 1..%d elements' % self.max_prefix)
        if int(prefix_ptr[-1]) != int(prefix_rid.numel()) or prefix_rid.numel() != prefix_time.numel():
            raise ValueError('prefix_ptr does not match prefix arrays')
        if int(prefix_rid.min()) < 0 or int(prefix_rid.max()) >= n or int(des.min()) < 0 or int(des.max()) >= n:
            raise IndexError('road id outside [0, %d)' % n)
        if int(des.max()) >= self.tables.n_nodes_h or n > self.tables.c_model.n_roads:
            raise IndexError('road id outside the model tables')
        if int(prefix_time.min()) < 0 or int(prefix_time.max()) >= min(2880, self.tables.n_time):
            raise IndexError('time code outside [0, 2880)')
        if int(max_step_t.max()) > self.max_step or int(max_step_t.min()) < 0:
            raise ValueError('max_step exceeds the engine capacity %d' % self.max_step)
        od = _lib.OdBatch()
        od.n_od, od.lmax = n_od, self.lmax
        od.prefix_ptr, od.prefix_rid, od.prefix_time = _lib.ptr(prefix_ptr), _lib.ptr(prefix_rid), _lib.ptr(prefix_time)
        od.des, od.default_len, od.max_step = _lib.ptr(des), _lib.ptr(default_len), _lib.ptr(max_step_t)
        allow_t = None
        if allow_label is not None:
            if self.graph.label is None:
                raise ValueError('allow_label needs GraphTables.set_labels()')
            allow_t = dv(allow_label)
            od.allow_label = _lib.ptr(allow_t)
        out_rid = torch.zeros((n_od, self.lmax), dtype=i32, device=dev)
        out_time = torch.zeros((n_od, self.lmax), dtype=i32, device=dev)
        out_len = torch.zeros(n_od, dtype=i32, device=dev)
        out_status = torch.zeros(n_od, dtype=i32, device=dev)
        out_steps = torch.zeros(n_od, dtype=i32, device=dev)
        od.out_rid, od.out_time, od.out_len = _lib.ptr(out_rid), _lib.ptr(out_time), _lib.ptr(out_len)
        od.out_status, od.out_steps = _lib.ptr(out_status), _lib.ptr(out_steps)
        pl_rid = pl_prio = pl_n = None
        od.poplog_cap = int(poplog_cap)
        if poplog_cap > 0:
            pl_rid = torch.zeros((n_od, poplog_cap), dtype=i32, device=dev)
            pl_prio = torch.zeros((n_od, poplog_cap), dtype=torch.float32, device=dev)
            pl_n = torch.zeros(n_od, dtype=i32, device=dev)
            od.poplog_rid, od.poplog_prio, od.poplog_n = _lib.ptr(pl_rid), _lib.ptr(pl_prio), _lib.ptr(pl_n)
        m, g = self.tables.c_model, self.graph.c_graph
        self.retry = torch.empty(n_od + self.slots, dtype=i32, device=dev)
        self.c_engines[0].retry = self.retry.data_ptr()
        self.c_engines[0].retry_bytes = self.retry.numel() * 4
        eng = self.c_engines[0]
        main = torch.cuda.current_stream(dev)
        stream = self.side_streams[0]            # graph capture needs a stream of its own (not the legacy default)
        handle = C.c_void_p(stream.cuda_stream)
        supersteps = 0
        nk = 2 + _lib.TSG_MAX_LAYERS
        prof_ms = (C.c_float * nk)()
        prof_n = (C.c_int32 * nk)()
        prof_items = 0
        chunk_i = 0
        pending = []
        graph = None
        K = 1                                    # row of `counters` with the OD queue / totals
        cnt = torch.zeros((2, 16), dtype=torch.int32)
        with torch.cuda.device(dev):
            self.counters.zero_()
            stream.wait_stream(main)
            _lib.check(L.tsg_engine_reset(C.byref(eng), C.byref(g), handle), 'tsg_engine_reset')
            # every superstep each unfinished slot advances by >= 1 expansion / prefix element / finished OD
            per_search = self.max_step + self.max_prefix + 4    # + the write-out superstep and the idle one behind it
            if self.max_pops > 0:       # yields: at most heap_cap pops per search, max_pops of them per superstep
                per_search += self.heap_entries // self.max_pops + 1
            bound = (math.ceil(n_od / self.slots) + 1) * per_search + 8 + self.ramp_steps
            if self.n_pages < self.slots * self.pages_per_slot:     # searches may wait for pages or restart
                bound = 4 * bound + 8 * math.ceil(n_od * self.pages_per_slot / self.n_pages) * per_search
            try:
                while True:
                    if profile_every and chunk_i % profile_every == 0:
                        # sampled chunk: drain the stream, then time every kernel of the chunk with HIP events
                        stream.synchronize()
                        items0 = int(self.slot_state[:, 19].sum())
                        _lib.check(L.tsg_engine_run_profiled(C.byref(m), C.byref(g), C.byref(od), C.byref(eng),
                                                             supersteps, chunk, handle, prof_ms, prof_n),
                                   'tsg_engine_run_profiled')
                        cnt = self.counters.cpu()
                        pending = []
                        prof_items += int(self.slot_state[:, 19].sum()) - items0
                    else:
                        # two chunks stay in flight: the completion counter of chunk i-1 is inspected (from a pinned
                        # snapshot copied on the stream) only after chunk i has been enqueued, so the GPU never
                        # waits for the host round trip
                        if self.use_graph:
                            if graph is None:
                                gx = C.c_void_p()
                                _lib.check(L.tsg_engine_graph_create(C.byref(m), C.byref(g), C.byref(od), C.byref(eng),
                                                                     chunk, handle, C.byref(gx)),
                                           'tsg_engine_graph_create')
                                graph = gx
                            _lib.check(L.tsg_engine_graph_launch(graph, handle), 'tsg_engine_graph_launch')
                        elif self.split_calls:      # one superstep = the two individually launchable halves
                            for k in range(chunk):
                                _lib.check(L.tsg_frontier_expand_select(C.byref(g), C.byref(od), C.byref(eng),
                                                                        supersteps + k, handle),
                                           'tsg_frontier_expand_select')
                                _lib.check(L.tsg_score_frontier(C.byref(m), C.byref(g), C.byref(eng), supersteps + k,
                                                                handle), 'tsg_score_frontier')
                        else:
                            _lib.check(L.tsg_engine_run(C.byref(m), C.byref(g), C.byref(od), C.byref(eng), supersteps,
                                                        chunk, handle), 'tsg_engine_run')
                        snap = self._pinned[chunk_i % 2]
                        with torch.cuda.stream(stream):
                            snap.copy_(self.counters, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        pending.append((ev, snap))
                        if len(pending) >= 2:
                            ev0, snap0 = pending.pop(0)
                            ev0.synchronize()
                            cnt = snap0.clone()
                    supersteps += chunk
                    chunk_i += 1
                    if int(cnt[K, 3]) >= n_od:
                        break
                    if supersteps > bound:
                        raise RuntimeError('A* engine did not finish within %d supersteps (finished %d of %d)'
                                           % (bound, int(cnt[K, 3]), n_od))
            finally:
                stream.synchronize()
                main.wait_stream(stream)
                if graph is not None:
                    L.tsg_engine_graph_destroy(graph)
            cnt = self.counters.cpu()
        if int(cnt[K, 6]) != 0:
            raise RuntimeError('A* engine hit a workspace cap (node/heap/exp capacity)')
        st = self.slot_state.to(torch.int64).sum(dim=0).cpu()
        self.last_stats = {'supersteps': supersteps, 'expansions': int(cnt[K, 4]), 'model_evals': int(st[16]),
                           'n_od': n_od, 'items': int(st[19]),
                           'chain_sum': int(st[18]), 'cand_sum': int(st[17]), 'skipped_all_closed': int(st[20]),
                           'prof_ms': list(prof_ms), 'prof_launches': list(prof_n), 'prof_items': prof_items,
                           'page_waits': int(st[21]), 'restarts': int(cnt[0, 7]),
                           'pool_pages': self.n_pages, 'min_free_pages': int(cnt[0, 12])}
        res = {'rid': out_rid, 'time': out_time, 'len': out_len, 'status': out_status, 'steps': out_steps}
        if poplog_cap > 0:
            res.update({'poplog_rid': pl_rid, 'poplog_prio': pl_prio, 'poplog_n': pl_n})
        if not return_device:
            res = {k: v.cpu() for k, v in res.items()}
        return res
