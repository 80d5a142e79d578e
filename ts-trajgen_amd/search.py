"""Searcher / DoubleLayerSearcher drop-ins (reference: search.py:55-75, :240-612, :821-934).

Same constructor and `astar_search*` signatures and return values as the reference; plus the batched
entry `generate(...)` that runs many OD pairs at once on the device engine (the reference loops them
one at a time, our_model_generate.py:417-426).  All search state lives in HBM (csrc/tsg_search.hip).
"""
import json
import os

import numpy as np
import torch

from . import _lib
from .engine import AStarEngine, GraphTables

STATUS_UNTOUCHED, STATUS_REACHED, STATUS_FALLBACK = (_lib.STATUS_UNTOUCHED, _lib.STATUS_REACHED,
                                                     _lib.STATUS_FALLBACK)


def _auto_slots(tables, graph, max_step, max_prefix, n_od, device, budget_frac=0.6, cap=8192):
    per = AStarEngine.bytes_per_slot(tables, graph, max_step, max_prefix)
    free, _total = torch.cuda.mem_get_info(device)
    fit = max(1, int(free * budget_frac) // per)
    slots = min(cap, fit, max(n_od, 1))
    if slots >= 128:
        slots = slots // 128 * 128      # whole LSTM row tiles
    return int(slots)


class Searcher(object):
    def __init__(self, device, adjacent_list, road_center_gps=None, road_length=None, road_time_distribution=None):
        self.device = torch.device(device)
        self.adjacent_list = adjacent_list
        self.road_center_gps = road_center_gps
        self.road_length = road_length
        self.road_time_distribution = road_time_distribution.astype(int)
        self._graph = None
        self._engines = {}
        self._label_key = None

    # -- device substrate ------------------------------------------------------------------------
    def graph(self):
        if self._graph is None:
            self._graph = GraphTables(self.device, self.adjacent_list, self.road_center_gps, self.road_length,
                                      self.road_time_distribution)
        return self._graph

    def _engine(self, gen_model, slots, max_step, max_prefix, lmax, streams=1):
        tables = gen_model.tables()
        key = (id(tables), slots, max_step, max_prefix, lmax, streams)
        eng = self._engines.get(key)
        if eng is None:
            self._engines.clear()       # one engine's workspaces at a time
            eng = AStarEngine(tables, self.graph(), slots, max_step, max_prefix, lmax, streams)
            self._engines[key] = eng
        return eng

    # -- batched entry (new) ---------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, gen_model, start_rids, start_times, des_rids, default_lens, max_step=500, prefixes=None,
                 allow_label=None, slots=None, lmax=256, poplog_cap=0, return_device=False, profile_every=0,
                 streams=None):
        """Road-level A* for many OD pairs.

        start_rids/start_times/des_rids/default_lens: int arrays of length n (prefix = one element), or pass
        `prefixes` = list of (loc_list, tim_list) for multi-element prefixes (astar_search_in_region legs).
        max_step: int or per-OD array.  Returns dict with padded `rid`/`time` [n, lmax], `len`, `status`,
        `steps` (torch tensors)."""
        des = np.asarray(des_rids, dtype=np.int32)
        n_od = len(des)
        if prefixes is None:
            prefix_ptr = np.arange(n_od + 1, dtype=np.int32)
            prefix_rid = np.asarray(start_rids, dtype=np.int32)
            prefix_time = np.asarray(start_times, dtype=np.int32)
            max_prefix = 1
        else:
            lens = np.array([len(p[0]) for p in prefixes], dtype=np.int32)
            prefix_ptr = np.zeros(n_od + 1, dtype=np.int32)
            np.cumsum(lens, out=prefix_ptr[1:])
            prefix_rid = np.concatenate([np.asarray(p[0], dtype=np.int32) for p in prefixes])
            prefix_time = np.concatenate([np.asarray(p[1], dtype=np.int32) for p in prefixes])
            max_prefix = int(lens.max())
        ms = np.broadcast_to(np.asarray(max_step, dtype=np.int32), (n_od,)).copy()
        ms_cap = int(ms.max()) if n_od else 1
        tables = gen_model.tables()
        if slots is None:
            slots = _auto_slots(tables, self.graph(), ms_cap, max_prefix, n_od, self.device)
        if streams is None:
            streams = 1
        eng = self._engine(gen_model, int(slots), ms_cap, max_prefix, lmax, int(streams))
        dl = np.asarray(default_lens, dtype=np.int32)
        al = None if allow_label is None else np.broadcast_to(np.asarray(allow_label, dtype=np.int32), (n_od,))
        # a slot serves at most TSG_GEN_MAX searches per batch (rid-table generations): split huge batches
        per_run = max(1, eng.slots) * 3000
        parts, stats = [], []
        for lo in range(0, max(n_od, 1), per_run):
            hi = min(n_od, lo + per_run)
            sub_ptr = prefix_ptr[lo:hi + 1] - prefix_ptr[lo]
            p0, p1 = int(prefix_ptr[lo]), int(prefix_ptr[hi])
            parts.append(eng.run(sub_ptr, prefix_rid[p0:p1], prefix_time[p0:p1], des[lo:hi], dl[lo:hi], ms[lo:hi],
                                 allow_label=None if al is None else al[lo:hi], poplog_cap=poplog_cap,
                                 return_device=return_device, profile_every=profile_every))
            stats.append(eng.last_stats)
        if len(parts) == 1:
            res = parts[0]
            self.last_stats = stats[0]
        else:
            res = {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}
            self.last_stats = {k: (sum(s[k] for s in stats) if not isinstance(stats[0][k], list)
                                   else [sum(x) for x in zip(*[s[k] for s in stats])]) for k in stats[0]}
        return res

    # -- reference API (one OD per call) ---------------------------------------------------------
    def _single(self, gen_model, trace_loc, trace_tim, des, default_len, max_step, allow_label=None):
        lmax = max(256, len(trace_loc) + 8, int(default_len) + 1)
        while True:
            res = self.generate(gen_model, None, None, [int(des)], [int(default_len)], max_step,
                                prefixes=[(list(trace_loc), list(trace_tim))], allow_label=allow_label, slots=1,
                                lmax=lmax)
            n = int(res['len'][0])
            if n <= lmax:
                break
            lmax = n          # trajectory longer than the output row: rerun with a wide enough row
        return res['rid'][0, :n].tolist(), res['time'][0, :n].tolist()

    def astar_search(self, gen_model, trace_loc, trace_tim, des, default_len, max_step=500):
        return self._single(gen_model, trace_loc, trace_tim, des, default_len, max_step)

    def astar_search_in_region(self, gen_model, trace_loc, trace_tim, des, default_len, region_rid_set,
                               max_step=500):
        key = frozenset(region_rid_set)
        if self._label_key != key:
            lab = np.zeros(self.graph().num_roads, dtype=np.int32)
            members = np.fromiter((r for r in key if 0 <= r < len(lab)), dtype=np.int64)
            lab[members] = 1
            self.graph().set_labels(lab)
            self._label_key = key
        return self._single(gen_model, trace_loc, trace_tim, des, default_len, max_step, allow_label=[1])
