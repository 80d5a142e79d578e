"""Searcher / DoubleLayerSearcher drop-ins (reference: search.py:55-75, :240-612, :821-934).

Same constructor and `astar_search*` signatures and return values as the reference; plus the batched
entry `generate(...)` that runs many OD pairs at once on the device engine (the reference loops them
one at a time, our_model_generate.py:417-426).  All search state lives in HBM (csrc/tsg_search.hip).
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib
from .engine import AStarEngine, GraphTables

STATUS_UNTOUCHED, STATUS_REACHED, STATUS_FALLBACK = (_lib.STATUS_UNTOUCHED, _lib.STATUS_REACHED,
                                                     _lib.STATUS_FALLBACK)


def _auto_slots(tables, graph, max_step, max_prefix, n_od, device, budget_frac=0.6, cap=8192, held=0):
    per = AStarEngine.bytes_per_slot(tables, graph, max_step, max_prefix)
    free, _total = torch.cuda.mem_get_info(device)
    # what torch's caching allocator holds without using it is as available to the engine as the driver's free memory,
    # and so is what the Searcher's own engines hold (`held`): they are released before a new one is allocated
    free += max(0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)) + int(held)
    fit = max(1, int(free * budget_frac) // per)
    slots = min(cap, fit, max(n_od, 1))
    if slots >= 128:
        slots = slots // 128 * 128      # whole LSTM row tiles
    return int(slots)


def plan_engine(free_bytes, fixed_per_slot, page_bytes, pages_per_search, n_od, out_bytes_per_od, oversub=2.2,
                cap=24576, budget_frac=0.93):
    """Sizes for a large batch of road-level searches when the caller leaves them to us (what bench.py does by hand):
    `slots` searches in flight on a state-page pool that would cover the worst case of slots / oversub of them (a
    search holds only the pages it has filled), at most `cap` (the measured best at BJ scale),
    in whole multiples of 1 024.  Returns (slots, pool_pages, ramp_steps); ramp_steps > 0 when the pool is below the
    worst case, so that the first searches start spread over that many supersteps."""
    budget = int(free_bytes * budget_frac) - out_bytes_per_od * n_od
    per = fixed_per_slot + pages_per_search * page_bytes / oversub
    slots = int(max(budget, 0) // per)
    slots = min(slots, cap, max(n_od, 1))
    if slots >= 1024:
        slots = slots // 1024 * 1024
    elif slots >= 128:
        slots = slots // 128 * 128
    slots = max(slots, 1)
    pool = min(slots * pages_per_search, int(max(budget - slots * fixed_per_slot, 0) // page_bytes))
    pool = max(pool, pages_per_search)
    return slots, pool, (1500 if pool < slots * pages_per_search else 0)


def streams_apart(device, n, tries=16, spin_cycles=1_500_000, report=None):
    """n streams that run side by side with each other and with the default stream.  Streams are bound to a few
    hardware queues (four per process here) in an order the runtime chooses, and two streams on one queue do not
    overlap: of eight new torch streams on an MI355X box, four pairs shared a queue, two of them with the default
    stream (ablate/stream_queues.py).  Test: two spin kernels on two streams take the time of one when the
    streams sit on different queues (the minimum of three spins per pair: a busy host can only lengthen a spin).
    `report` (a dict) receives 'queues_apart': True when all n streams passed the test against each other and the
    default stream, False when the list had to be padded with untested streams (a pair of engines on streams that
    share a queue does not overlap: 15.5 k instead of 21.6 k trajectories/s at BJ scale), None when the test
    could not be run."""
    import time
    dev = torch.device(device)
    if report is None:
        report = {}
    report['queues_apart'] = None
    if n <= 0:
        return []
    if not hasattr(torch.cuda, '_sleep'):
        return [torch.cuda.Stream(device=dev) for _ in range(n)]
    with torch.cuda.device(dev):
        def spin(a, b=None):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            with torch.cuda.stream(a):
                torch.cuda._sleep(spin_cycles)
            if b is not None:
                with torch.cuda.stream(b):
                    torch.cuda._sleep(spin_cycles)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0
        base = torch.cuda.default_stream(dev)
        chosen, spare = [], []
        first = torch.cuda.Stream(device=dev)
        spin(first)
        one = min(spin(first) for _ in range(3))
        cand = first
        for _ in range(tries):
            if all(min(spin(cand, o) for _ in range(3)) < 1.5 * one for o in chosen + [base]):
                chosen.append(cand)
                if len(chosen) == n:
                    break
            else:
                spare.append(cand)
            cand = torch.cuda.Stream(device=dev)
        report['queues_apart'] = len(chosen) == n
        return (chosen + spare + [torch.cuda.Stream(device=dev) for _ in range(n)])[:n]


AUTO_PLAN_MIN_OD = 32768        # batches from this size on get plan_engine's sizes and two engines by default


class Searcher(object):
    # 1: hidden states as split-f16 words, recurrence on the f16 matrix cores in three passes (fp32-grade
    # products, include/tsg.h tsg_engine.state_format); 0: plain fp32 states on the fp32 matrix-core path
    state_format = 1
    # Memory policy of generate() when the caller gives no sizes: the share of the device memory that is free
    # (driver-free + torch's cached-but-unused blocks + what this Searcher's own engines hold) an engine may take.
    # Large plain batches (>= AUTO_PLAN_MIN_OD pairs) are planned against `plan_budget_frac`, everything else
    # against `budget_frac`.  A caller that allocates afterwards (a training loop, a second model) lowers them on
    # its Searcher, or passes slots / pool_pages itself, or sets auto_plan = False.
    budget_frac = 0.6
    plan_budget_frac = 0.93
    auto_plan = True

    def __init__(self, device, adjacent_list, road_center_gps=None, road_length=None, road_time_distribution=None):
        self.device = torch.device(device)
        self.adjacent_list = adjacent_list
        self.road_center_gps = road_center_gps
        self.road_length = road_length
        self.road_time_distribution = road_time_distribution.astype(int)
        self._graph = None
        self._engines = {}
        self._streams = []          # one per engine of a side-by-side run, kept for the Searcher's life
        self._stream_report = {}    # streams_apart's verdict on them ('queues_apart'), copied into last_stats
        self._label_key = None

    # -- device substrate ------------------------------------------------------------------------
    def graph(self):
        if self._graph is None:
            self._graph = GraphTables(self.device, self.adjacent_list, self.road_center_gps, self.road_length,
                                      self.road_time_distribution)
        return self._graph

    def _held_bytes(self):
        total = 0
        for val in self._engines.values():
            for eng in (val if isinstance(val, list) else [val]):
                for v in eng.__dict__.values():
                    for x in (v if isinstance(v, list) else [v]):
                        if torch.is_tensor(x) and x.is_cuda:
                            total += x.numel() * x.element_size()
        return total

    def _drop_engines(self):
        """Release the workspaces of every engine held (before a new set is allocated, or on request)."""
        for val in self._engines.values():
            for eng in (val if isinstance(val, list) else [val]):
                eng.close()
        self._engines.clear()

    def _engine(self, gen_model, slots, max_step, max_prefix, lmax, streams=1, pool_pages=None, admit_reserve=None,
                ramp_steps=0, exact=True):
        """The engine (device workspaces + captured graph) for a call.  exact=False (the caller left the number of
        slots to us): an engine that is already there and at least as large in every capacity serves the call, so
        that the rounds of a two-stage generation, whose batches shrink and whose prefixes grow, do not allocate a
        new set of workspaces each (capacities are limits only: they never change a result)."""
        tables = gen_model.tables()
        key = (id(tables), slots, max_step, max_prefix, lmax, self.state_format, pool_pages, admit_reserve, ramp_steps)
        eng = self._engines.get(key)
        if eng is None and not exact:
            for k, cand in self._engines.items():
                if (k[0] == key[0] and k[4] == lmax and k[5:] == key[5:] and cand.slots >= slots and
                        cand.max_step >= max_step and cand.max_prefix >= max_prefix):
                    return cand
        if eng is None:
            self._drop_engines()        # one engine's workspaces at a time
            if not self._streams:
                self._streams = streams_apart(self.device, 1, report=self._stream_report)
            eng = AStarEngine(tables, self.graph(), slots, max_step, max_prefix, lmax, streams,
                              state_format=self.state_format, pool_pages=pool_pages, admit_reserve=admit_reserve,
                              ramp_steps=ramp_steps, stream=self._streams[0])
            self._engines[key] = eng
        return eng

    # -- batched entry (new) ---------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, gen_model, start_rids, start_times, des_rids, default_lens, max_step=500, prefixes=None,
                 allow_label=None, slots=None, lmax=256, poplog_cap=0, return_device=False, profile_every=0,
                 streams=None, pool_pages=None, admit_reserve=None, ramp_steps=0):
        """Road-level A* for many OD pairs.

        start_rids/start_times/des_rids/default_lens: int arrays of length n (prefix = one element), or pass
        `prefixes` = list of (loc_list, tim_list) for multi-element prefixes (astar_search_in_region legs).
        max_step: int or per-OD array.  Returns dict with padded `rid`/`time` [n, lmax], `len`, `status`,
        `steps` (torch tensors).
        pool_pages / admit_reserve / ramp_steps: state-page pool of the engine (AStarEngine.__init__); the default
        reserves the worst case for every search in flight.
        streams: E > 1 deals the OD pairs over E engines that run side by side (`slots` and `pool_pages` are the
        totals over all of them; _generate_engines); the results do not depend on it."""
        des = np.asarray(des_rids, dtype=np.int32)
        n_od = len(des)
        if prefixes is None:
            prefix_ptr = np.arange(n_od + 1, dtype=np.int32)
            prefix_rid = np.asarray(start_rids, dtype=np.int32)
            prefix_time = np.asarray(start_times, dtype=np.int32)
            max_prefix = 1
        elif isinstance(prefixes, tuple) and len(prefixes) == 3 and isinstance(prefixes[0], np.ndarray):
            prefix_ptr = prefixes[0].astype(np.int32)           # CSR form: (ptr [n+1], rid, time)
            prefix_rid = prefixes[1].astype(np.int32)
            prefix_time = prefixes[2].astype(np.int32)
            # capacity in powers of two from 64: the prefixes of a two-stage generation grow from round to round, and
            # every new capacity is a new set of workspaces
            max_prefix = max(64, 1 << (int(np.diff(prefix_ptr).max()) - 1).bit_length())
        else:
            lens = np.array([len(p[0]) for p in prefixes], dtype=np.int32)
            prefix_ptr = np.zeros(n_od + 1, dtype=np.int32)
            np.cumsum(lens, out=prefix_ptr[1:])
            prefix_rid = np.concatenate([np.asarray(p[0], dtype=np.int32) for p in prefixes])
            prefix_time = np.concatenate([np.asarray(p[1], dtype=np.int32) for p in prefixes])
            max_prefix = (int(lens.max()) + 63) // 64 * 64
        ms = np.broadcast_to(np.asarray(max_step, dtype=np.int32), (n_od,)).copy()
        # every counted step closes a road that was open (search.py:279-288, :420), so a search never takes more than
        # num_roads steps: a larger budget cannot bind, and the capacity reserved per search follows the smaller one
        # (a region graph of 100 nodes under max_step 5000 would otherwise reserve 80 state pages per search)
        ms = np.minimum(ms, self.graph().num_roads + 1)
        ms_cap = int(ms.max()) if n_od else 1
        tables = gen_model.tables()
        auto = slots is None
        if auto and self.auto_plan and pool_pages is None and prefixes is None and n_od >= AUTO_PLAN_MIN_OD:
            # a large plain batch with nothing said about sizes: the searches in flight, the shared page pool and
            # the start ramp as the bench sizes them, two engines side by side unless the caller chose
            free, _total = torch.cuda.mem_get_info(self.device)
            free += max(0, torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device))
            free += self._held_bytes()      # what the engines of an earlier call hold is ours to use again
            free = free >> 30 << 30         # whole GiB: the same plan, and with it the same engines, call after call
            slots, pool_pages, ramp_steps = plan_engine(
                free, AStarEngine.fixed_bytes_per_slot(self.graph(), ms_cap, max_prefix),
                AStarEngine.bytes_per_page(tables), (max_prefix + ms_cap + 63) // 64, n_od, (2 * lmax + 8) * 4,
                budget_frac=self.plan_budget_frac)
            if streams is None:
                streams = 2
            auto = False
        elif auto:
            slots = _auto_slots(tables, self.graph(), ms_cap, max_prefix, n_od, self.device,
                                budget_frac=self.budget_frac, held=self._held_bytes())
        n_eng = max(1, int(streams or 1))
        dl = np.asarray(default_lens, dtype=np.int32)
        al = None if allow_label is None else np.broadcast_to(np.asarray(allow_label, dtype=np.int32), (n_od,))
        run_kw = dict(poplog_cap=poplog_cap, return_device=return_device, profile_every=profile_every)
        if n_eng > 1 and n_od >= 4 * n_eng and int(slots) >= 2 * n_eng:
            return self._generate_engines(gen_model, n_eng, int(slots), ms_cap, max_prefix, lmax, pool_pages,
                                          admit_reserve, ramp_steps, prefix_ptr, prefix_rid, prefix_time, des, dl, ms, al,
                                          run_kw)
        eng = self._engine(gen_model, int(slots), ms_cap, max_prefix, lmax, 1, pool_pages, admit_reserve,
                           ramp_steps, exact=not auto)
        res, self.last_stats = self._run_engine(eng, prefix_ptr, prefix_rid, prefix_time, des, dl, ms, al, run_kw)
        self.last_stats = dict(self.last_stats, engines=1, queues_apart=self._stream_report.get('queues_apart'))
        return res

    @staticmethod
    def _merge_stats(stats):
        if len(stats) == 1:
            return stats[0]
        lows = ('min_free_pages',)
        return {k: ([sum(x) for x in zip(*[s[k] for s in stats])] if isinstance(stats[0][k], list)
                    else min(s[k] for s in stats) if k in lows else sum(s[k] for s in stats)) for k in stats[0]}

    def _run_engine(self, eng, prefix_ptr, prefix_rid, prefix_time, des, dl, ms, al, run_kw):
        """All OD pairs of the arrays through one engine; returns (result dict, statistics)."""
        n_od = len(des)
        # a slot serves at most TSG_GEN_MAX searches per batch (rid-table generations): split huge batches
        per_run = max(1, eng.slots) * 3000
        parts, stats = [], []
        for lo in range(0, max(n_od, 1), per_run):
            hi = min(n_od, lo + per_run)
            sub_ptr = prefix_ptr[lo:hi + 1] - prefix_ptr[lo]
            p0, p1 = int(prefix_ptr[lo]), int(prefix_ptr[hi])
            parts.append(eng.run(sub_ptr, prefix_rid[p0:p1], prefix_time[p0:p1], des[lo:hi], dl[lo:hi], ms[lo:hi],
                                 allow_label=None if al is None else al[lo:hi], **run_kw))
            stats.append(eng.last_stats)
        if len(parts) == 1:
            return parts[0], stats[0]
        return {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}, self._merge_stats(stats)

    def _generate_engines(self, gen_model, n_eng, slots, ms_cap, max_prefix, lmax, pool_pages, admit_reserve, ramp_steps,
                          prefix_ptr, prefix_rid, prefix_time, des, dl, ms, al, run_kw):
        """`streams` = E > 1: the OD pairs are dealt round-robin over E independent engines (slots / E searches in
        flight and pool_pages / E state pages each), every engine driven by a host thread of its own on a stream of
        its own.  The GPU then has the launches of E supersteps to choose from: the latency-bound ones of one engine
        (frontier, layer 0 of the recurrence) run under the matrix-core- and bandwidth-bound ones of another.  OD
        pairs are independent, so the results are those of one engine (tests/test_gpu_fullsize.py), returned in
        the caller's order.  Measured at BJ scale, 24 576 searches in flight: 19.2k -> 21.2k trajectories/s with two
        engines; three or more lose to the smaller launches."""
        import threading
        tables = gen_model.tables()
        n_od = len(des)
        slots_each = max(1, slots // n_eng)
        if slots_each >= 128:
            slots_each = slots_each // 128 * 128
        pool_each = None if pool_pages is None else max(1, int(pool_pages) // n_eng)
        reserve_each = None if admit_reserve is None else max(0, int(admit_reserve) // n_eng)
        key = ('set', id(tables), n_eng, slots_each, ms_cap, max_prefix, lmax, self.state_format, pool_each,
               reserve_each, ramp_steps)
        engines = self._engines.get(key)
        if engines is None:
            self._drop_engines()
            if len(self._streams) < n_eng:        # no engine is left that holds one of the old ones
                self._streams = streams_apart(self.device, n_eng, report=self._stream_report)
            engines = [AStarEngine(tables, self.graph(), slots_each, ms_cap, max_prefix, lmax, 1,
                                   state_format=self.state_format, pool_pages=pool_each, admit_reserve=reserve_each,
                                   ramp_steps=ramp_steps, stream=self._streams[e]) for e in range(n_eng)]
            self._engines[key] = engines
        plen = np.diff(prefix_ptr).astype(np.int64)
        out, err = [None] * n_eng, [None] * n_eng
        idx = [np.arange(e, n_od, n_eng) for e in range(n_eng)]

        def work(e):
            try:
                with torch.cuda.device(self.device):
                    ii = idx[e]
                    ln = plen[ii]
                    ptr = np.zeros(len(ii) + 1, dtype=np.int32)
                    np.cumsum(ln, out=ptr[1:])
                    # positions of the chosen prefixes' elements in the flat arrays
                    pos = np.repeat(prefix_ptr[:-1][ii].astype(np.int64) - ptr[:-1], ln) + np.arange(int(ptr[-1]))
                    out[e] = self._run_engine(engines[e], ptr, prefix_rid[pos], prefix_time[pos], des[ii], dl[ii], ms[ii],
                                              None if al is None else al[ii], run_kw)
            except BaseException as ex:      # re-raised in the calling thread
                err[e] = ex

        threads = [threading.Thread(target=work, args=(e,)) for e in range(n_eng)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for ex in err:
            if ex is not None:
                raise ex
        res = {}
        for k in out[0][0]:
            first = out[0][0][k]
            full = torch.zeros((n_od,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
            for e in range(n_eng):
                full[torch.from_numpy(idx[e]).to(first.device)] = out[e][0][k]
            res[k] = full
        self.last_stats = self._merge_stats([o[1] for o in out])
        self.last_stats['supersteps'] = max(o[1]['supersteps'] for o in out)
        self.last_stats['engines'] = n_eng
        self.last_stats['queues_apart'] = self._stream_report.get('queues_apart')
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


    # -- stochastic rollouts (search.py:614-716) ---------------------------------------------------
    def random_sample_batch(self, gen_model, prefixes, des, default_lens, uniforms=None, generator=None,
                            return_probs=False):
        """Many road_random_sample rollouts at once (rollout.py:86-92 asks for (seq_len - 2) * rollout_times
        of them per trajectory).  See rollout.sample_batch for arguments and the draw rule."""
        from .rollout import sample_batch
        return sample_batch(gen_model.tables(), self.graph(), prefixes, des, default_lens, uniforms=uniforms,
                            generator=generator, return_probs=return_probs)

    def road_random_sample(self, gen_model, trace_loc, trace_tim, des, default_len):
        res = self.random_sample_batch(gen_model, [(list(trace_loc), list(trace_tim))], [int(des)],
                                       [int(default_len)])
        return res['loc'][0], res['tim'][0]


    # -- beam search (search.py:137-238) -----------------------------------------------------------
    def beam_search_batch(self, gen_model, prefixes, des, width=15, max_step=50, prob_threshold=1e-4, seeds=None,
                          randint=None):
        """Many beam searches at once; see beam.beam_search_batch.  Returns (locs, tims, errors)."""
        from .beam import beam_search_batch
        return beam_search_batch(gen_model.tables(), self.adjacent_list, self.road_center_gps, prefixes, des,
                                 width=width, max_step=max_step, prob_threshold=prob_threshold, seeds=seeds,
                                 randint=randint)

    def beam_search(self, gen_model, trace_loc, trace_tim, des, width=15, max_step=50, prob_threshold=1e-4):
        """Reference signature and return value; the final draw uses the global numpy RNG (search.py:237).  When
        every branch dies the reference indexes an empty list (search.py:230): IndexError here too."""
        locs, tims, err = self.beam_search_batch(gen_model, [(list(trace_loc), list(trace_tim))], [int(des)],
                                                 width=width, max_step=max_step, prob_threshold=prob_threshold)
        if err[0]:
            raise IndexError('list index out of range')
        return locs[0], tims[0]


def ordered_draw(rng, a, p):
    """RandomState.choice(a, len(a), p=p, replace=False) (search.py:900-902) without its per-call overhead
    (argument checks, np.unique): the same uniforms are taken from `rng` in the same order and the result is the
    same, element for element (tests/test_host.py compares the two).  Falls back to rng.choice for a generator
    without random_sample."""
    if not hasattr(rng, 'random_sample'):
        return rng.choice(a, len(a), p=p, replace=False)
    size = len(a)
    p = p.copy()
    found = []
    while len(found) < size:
        x = rng.random_sample(size - len(found))
        if found:
            p[found] = 0
        cdf = np.cumsum(p)
        cdf /= cdf[-1]
        seen = set(found)
        for v in cdf.searchsorted(x, side='right').tolist():      # first occurrences, in the order drawn
            if v not in seen:
                seen.add(v)
                found.append(v)
    return a[found]


def border_orders(seeds, leg_ptr, leg_pair, pair_ptr, pair_p):
    """The border-road orders of many OD pairs at once (search.py:897-902): OD pair i draws, from its own
    numpy.random.RandomState(seeds[i]), one order per leg leg_ptr[i] <= l < leg_ptr[i+1]; leg l permutes the list of
    pair leg_pair[l], whose (normalised) probabilities are pair_p[pair_ptr[q]:pair_ptr[q+1]].  Returns (out_ptr,
    out_idx): the order of leg l is out_idx[out_ptr[l]:out_ptr[l+1]] (indices into the pair's list), element for
    element what ordered_draw / RandomState.choice(replace=False, p=...) return (tests/test_host.py).  Runs in
    libtsg_host.so (plain C): seeding 10^5 generators and drawing in Python was the bottleneck of the two-stage path."""
    H = _lib.host_lib()
    seeds = np.asarray(seeds, dtype=np.int64)
    if len(seeds) and (int(seeds.min()) < 0 or int(seeds.max()) > 0xFFFFFFFF):
        raise ValueError('Seed must be between 0 and 2**32 - 1')      # as numpy.random.RandomState(seed)
    seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
    leg_ptr = np.ascontiguousarray(leg_ptr, dtype=np.int64)
    leg_pair = np.ascontiguousarray(leg_pair, dtype=np.int32)
    pair_ptr = np.ascontiguousarray(pair_ptr, dtype=np.int64)
    pair_p = np.ascontiguousarray(pair_p, dtype=np.float64)
    sizes = (pair_ptr[1:] - pair_ptr[:-1])[leg_pair] if len(leg_pair) else np.zeros(0, dtype=np.int64)
    out_ptr = np.zeros(len(leg_pair) + 1, dtype=np.int64)
    np.cumsum(sizes, out=out_ptr[1:])
    out_idx = np.zeros(max(1, int(out_ptr[-1])), dtype=np.int32)
    max_size = int((pair_ptr[1:] - pair_ptr[:-1]).max()) if len(pair_ptr) > 1 else 1

    def ptr(a):
        return a.ctypes.data_as(C.c_void_p)
    rc = H.tsg_host_border_orders(len(seeds), ptr(seeds), ptr(leg_ptr), ptr(leg_pair), ptr(pair_ptr), ptr(pair_p),
                                  ptr(out_ptr), ptr(out_idx), max(1, max_size))
    if rc == -2:      # numpy's choice(replace=False, p) raises the same way
        raise ValueError('a border-road probability vector has fewer positive entries than roads, or a negative / '
                         'NaN entry (numpy: "Fewer non-zero entries in p than size")')
    if rc != 0:
        raise RuntimeError('tsg_host_border_orders failed (%d)' % rc)
    return out_ptr, out_idx


class DoubleLayerSearcher(object):
    """Two-stage search (reference: search.py:821-934): region-level A* first (search.py:936-1117), then
    road-level A* inside each region of the region path towards a border road drawn from the transfer
    frequencies (:897-902), then the last leg to the destination (:925-931).

    `astar_search` keeps the reference's one-OD signature and return value; `generate` runs many OD pairs:
    one batched region search, then rounds of batched road legs (every unfinished OD submits its current
    attempt).  The border-road order of an OD is drawn from its own `RandomState(seed)` (the reference uses
    the global numpy RNG once per OD in sequence; seeding per OD is what makes a batch reproducible)."""

    def __init__(self, device, adjacent_list, road_center_gps, road_length,
                 region_adjacent_list, region_dist, region_transfer_freq,
                 rid2region, region2rid, road_time_distribution, region_time_distribution):
        self.device = torch.device(device)
        self.adjacent_list = adjacent_list
        self.road_center_gps = road_center_gps
        self.road_length = road_length
        self.region_adjacent_list = region_adjacent_list
        self.region_dist = region_dist
        self.region_transfer_freq = region_transfer_freq
        self.rid2region = rid2region
        self.region2rid = region2rid
        self.region_time_distribution = region_time_distribution.astype(int)
        self.road_search = Searcher(device, adjacent_list, road_center_gps, road_length, road_time_distribution)
        self.fail_find_log = set()
        self.region_fail_find_log = set()
        self._region_search = None
        self._labelled = False
        self.last_stats = {}

    # -- device substrate ------------------------------------------------------------------------
    def _regions(self):
        if self._region_search is None:
            s = Searcher.__new__(Searcher)
            s.device = self.device
            s._graph = GraphTables.for_regions(self.device, self.region_adjacent_list, self.region_dist,
                                               self.region_time_distribution, len(self.region2rid))
            s._engines = {}
            s._streams = []
            s._stream_report = {}
            s._label_key = None
            self._region_search = s
        return self._region_search

    def _label_roads(self):
        if not self._labelled:
            g = self.road_search.graph()
            lab = np.full(g.num_roads, -1, dtype=np.int32)
            for reg, rids in self.region2rid.items():           # region_rid_set = set(region2rid[str(region)])
                lab[np.asarray(rids, dtype=np.int64)] = int(reg)
            g.set_labels(lab)
            self.road_search._label_key = 'regions'
            self._labelled = True

    # -- region stage ----------------------------------------------------------------------------
    def _region_paths(self, region_model, start_regions, start_times, des_regions, max_step):
        n = len(des_regions)
        if n == 0:
            return [], []
        res = self._regions().generate(region_model, start_regions, start_times, des_regions,
                                       np.full(n, 10, dtype=np.int32),      # best_default_len = 10 (search.py:967)
                                       max_step=max_step,
                                       # a region path visits a region at most once (closed set)
                                       lmax=max(64, min(int(np.max(max_step)), self._regions().graph().num_roads) + 2))
        traces, flags = [], []
        r_rid, r_len, r_status = res['rid'].numpy(), res['len'].numpy(), res['status'].numpy()
        for i in range(n):
            if (int(r_status[i]) & 0xff) == STATUS_UNTOUCHED:
                raise AssertionError('region search produced no trace (search.py:1113)')
            k = int(r_len[i])
            tr = r_rid[i, :k].tolist()
            traces.append(tr)
            flags.append(1 if tr[-1] == int(des_regions[i]) else 0)         # search.py:1114-1117
        return traces, flags

    def astar_search_in_region(self, region_model, start_region, start_time, des_region, max_step=500):
        tr, fl = self._region_paths(region_model, [start_region], [start_time], [des_region],
                                    np.asarray([max_step], dtype=np.int32))
        return tr[0], fl[0]

    # -- stochastic region rollouts (search.py:1163-1256) ------------------------------------------
    def region_random_sample_batch(self, region_model, prefixes, des, default_lens, uniforms=None, generator=None,
                                   return_probs=False):
        return self._regions().random_sample_batch(region_model, prefixes, des, default_lens, uniforms=uniforms,
                                                   generator=generator, return_probs=return_probs)

    def region_random_sample(self, region_model, trace_loc, trace_tim, des, default_len):
        res = self.region_random_sample_batch(region_model, [(list(trace_loc), list(trace_tim))], [int(des)],
                                              [int(default_len)])
        return res['loc'][0], res['tim'][0]

    def road_random_sample(self, gen_model, trace_loc, trace_tim, des, default_len):
        return self.road_search.road_random_sample(gen_model, trace_loc, trace_tim, des, default_len)

    # -- batched two-stage generation ------------------------------------------------------------
    @torch.no_grad()
    def generate(self, region_model, road_model, start_rids, start_times, des_rids, default_lens, max_step=500,
                 seeds=None, rngs=None, slots=None, leg_lmax=None):
        """Returns (list of trace_loc, list of trace_tim, list of is_astar).
        leg_lmax: width of the output rows of a round of legs (default: 256 or more, by the longest prefix); legs
        whose trajectory is longer are run again with a wider row, so the value never changes a result."""
        import time
        n = len(des_rids)
        if n == 0:          # e.g. a rank whose shard of the OD list is empty: nothing to run, still a valid result
            self.last_stats = {'rounds': 0, 'road_expansions': 0}
            return [], [], []
        t_begin = time.time()
        t_engine = 0.0      # wall time inside the batched engine calls (region search + road legs)
        ms = np.broadcast_to(np.asarray(max_step, dtype=np.int32), (n,)).copy()
        if rngs is None:
            rngs = [None] * n      # RandomState(seed of OD i), made when OD i first draws (seeding costs 25 us)
        self._label_roads()
        sreg = [self.rid2region[str(int(r))] for r in start_rids]
        dreg = [self.rid2region[str(int(r))] for r in des_rids]
        t0 = time.time()
        region_traces, is_astar = self._region_paths(region_model, sreg, start_times, dreg, ms)
        t_engine += time.time() - t0
        for i in range(n):
            if is_astar[i] != 1:
                self.region_fail_find_log.add((sreg[i], dreg[i]))
        # per-OD state as arrays; the Python left per OD pair and round is the choice of its next target
        des_np = np.asarray(des_rids, dtype=np.int32)
        dl_np = np.asarray(default_lens, dtype=np.int32)
        cap = 256
        p_rid = np.zeros((n, cap), dtype=np.int32)      # the trajectory so far (= prefix of the next leg)
        p_tim = np.zeros((n, cap), dtype=np.int32)
        p_rid[:, 0] = np.asarray(start_rids, dtype=np.int32)
        p_tim[:, 0] = np.asarray(start_times, dtype=np.int32)
        plen = np.ones(n, dtype=np.int64)
        leg = np.zeros(n, dtype=np.int64)               # index into the region path
        nleg = np.array([len(rt) - 1 for rt in region_traces], dtype=np.int64)
        attempt = np.zeros(n, dtype=np.int64)
        olen = np.zeros(n, dtype=np.int64)
        order = [None] * n                              # border roads of the current region pair, in the drawn order
        done = np.zeros(n, dtype=bool)
        target = np.zeros(n, dtype=np.int32)
        allow = np.zeros(n, dtype=np.int32)
        rounds = expansions = 0
        lab = self.road_search.graph().label
        label_np = lab.cpu().numpy() if lab is not None else None
        pair_cache = {}

        vector = all(r is None for r in rngs)    # every OD pair draws from RandomState(seed): orders pre-drawn in C
        if vector:
            # all legs of all OD pairs: region pair -> (border rids, probabilities); one order per leg, drawn up front
            # (leg k of an OD pair takes the k-th draw of its generator whether or not the legs before it succeed)
            pair_id, pair_rids, pair_ps = {}, [], []
            leg_ptr = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(nleg, out=leg_ptr[1:])
            leg_pair = np.zeros(int(leg_ptr[-1]), dtype=np.int32)
            leg_from = np.zeros(int(leg_ptr[-1]), dtype=np.int32)
            for i, rt in enumerate(region_traces):
                o = int(leg_ptr[i])
                for k in range(len(rt) - 1):
                    key = (rt[k], rt[k + 1])
                    q = pair_id.get(key)
                    if q is None:
                        b = self.region_transfer_freq[str(rt[k])][str(rt[k + 1])]
                        freq = np.array(b['transfer_freq'])
                        q = pair_id[key] = len(pair_rids)
                        pair_rids.append(np.asarray(b['transfer_rid'], dtype=np.int32))
                        pair_ps.append(freq / np.sum(freq))
                    leg_pair[o + k] = q
                    leg_from[o + k] = rt[k]
            pair_ptr = np.zeros(len(pair_rids) + 1, dtype=np.int64)
            np.cumsum([len(r) for r in pair_rids], out=pair_ptr[1:])
            rid_cat = np.concatenate(pair_rids) if pair_rids else np.zeros(0, dtype=np.int32)
            p_cat = np.concatenate(pair_ps) if pair_ps else np.zeros(0, dtype=np.float64)
            sd = np.asarray(seeds, dtype=np.int64) if seeds is not None else np.arange(n, dtype=np.int64)
            out_ptr, out_idx = border_orders(sd, leg_ptr, leg_pair, pair_ptr, p_cat)
            order_rid = rid_cat[pair_ptr[leg_pair].repeat(np.diff(out_ptr)) + out_idx[:int(out_ptr[-1])]] \
                if len(leg_pair) else np.zeros(0, dtype=np.int32)
            last_reg = np.array([rt[-1] for rt in region_traces], dtype=np.int32)
            in_last = (label_np[des_np] == last_reg) if label_np is not None else np.zeros(n, dtype=bool)

        def set_targets(ids):
            """target / allow / olen of the OD pairs `ids` from the pre-drawn orders (vector form of set_target)"""
            if ids.size == 0:
                return
            il = leg[ids] < nleg[ids]
            gl = np.minimum(leg_ptr[ids] + leg[ids], max(0, len(leg_pair) - 1))
            if len(leg_pair):
                olen[ids] = np.where(il, out_ptr[gl + 1] - out_ptr[gl], 0)
                pos = np.minimum(out_ptr[gl] + np.minimum(attempt[ids], np.maximum(olen[ids] - 1, 0)),
                                 max(0, len(order_rid) - 1))
                target[ids] = np.where(il, order_rid[pos] if len(order_rid) else 0, des_np[ids])
                allow[ids] = np.where(il, leg_from[gl], np.where(in_last[ids], last_reg[ids], -1))
            else:
                target[ids] = des_np[ids]
                allow[ids] = np.where(in_last[ids], last_reg[ids], -1)

        def set_target(i):
            rt = region_traces[i]
            k = int(leg[i])
            if k < nleg[i]:
                if order[i] is None:
                    key = (rt[k], rt[k + 1])
                    ent = pair_cache.get(key)
                    if ent is None:
                        b = self.region_transfer_freq[str(rt[k])][str(rt[k + 1])]
                        freq = np.array(b['transfer_freq'])
                        ent = pair_cache[key] = (np.asarray(b['transfer_rid']), freq / np.sum(freq))
                    if rngs[i] is None:
                        rngs[i] = np.random.RandomState(int(seeds[i]) if seeds is not None else i)
                    order[i] = ordered_draw(rngs[i], ent[0], ent[1])                      # search.py:900-902
                    olen[i] = len(order[i])
                    attempt[i] = 0
                target[i] = order[i][attempt[i]]
                allow[i] = rt[k]
            else:
                des = int(des_np[i])
                in_last = label_np is not None and int(label_np[des]) == int(rt[-1])
                target[i] = des
                allow[i] = int(rt[-1]) if in_last else -1                                  # search.py:925-931

        if vector:
            set_targets(np.arange(n))
        else:
            for i in range(n):
                set_target(i)
        while True:
            act = np.flatnonzero(~done)
            m = int(act.size)
            if m == 0:
                break
            ln = plen[act]
            ptr = np.zeros(m + 1, dtype=np.int32)
            np.cumsum(ln, out=ptr[1:])
            keep = np.arange(cap)[None, :] < ln[:, None]
            lmax = max(256, int(ln.max()) + 64, int(dl_np[act].max()) + 1)
            lmax = (lmax + 255) // 256 * 256 if leg_lmax is None else int(leg_lmax)
            tgt = target[act]
            t0 = time.time()
            res = self.road_search.generate(road_model, None, None, tgt, dl_np[act], max_step=ms[act],
                                            prefixes=(ptr, p_rid[act][keep], p_tim[act][keep]), allow_label=allow[act],
                                            slots=slots, lmax=lmax)
            t_engine += time.time() - t0
            expansions += self.road_search.last_stats.get('expansions', 0)
            rounds += 1
            r_rid, r_time = res['rid'].numpy(), res['time'].numpy()
            r_len, r_status = res['len'].numpy().astype(np.int64), res['status'].numpy()
            over = ((r_status & _lib.STATUS_OVERFLOW) != 0) | (r_len > r_rid.shape[1])
            if over.any():
                # a leg longer than the output row (the reference has no length limit): run those legs again with a
                # row wide enough for the longest of them, as Searcher._single does; the other legs keep their result
                oi = np.flatnonzero(over)
                if (r_len[oi] <= 0).any():
                    raise RuntimeError('A* engine hit a workspace cap (node/heap/exp capacity)')
                wide = (int(r_len[oi].max()) + 255) // 256 * 256
                optr = np.zeros(len(oi) + 1, dtype=np.int32)
                np.cumsum(ln[oi], out=optr[1:])
                res2 = self.road_search.generate(road_model, None, None, tgt[oi], dl_np[act][oi], max_step=ms[act][oi],
                                                 prefixes=(optr, p_rid[act][oi][keep[oi]], p_tim[act][oi][keep[oi]]),
                                                 allow_label=allow[act][oi], slots=slots, lmax=wide)
                st2 = res2['status'].numpy()
                if ((st2 & _lib.STATUS_OVERFLOW) != 0).any():
                    raise RuntimeError('trajectory longer than the widened output row')
                grown_r = np.zeros((m, wide), dtype=np.int32)
                grown_t = np.zeros((m, wide), dtype=np.int32)
                grown_r[:, :r_rid.shape[1]] = r_rid
                grown_t[:, :r_rid.shape[1]] = r_time
                grown_r[oi] = res2['rid'].numpy()
                grown_t[oi] = res2['time'].numpy()
                r_rid, r_time = grown_r, grown_t
                r_len[oi] = res2['len'].numpy().astype(np.int64)
                r_status = r_status.copy()
                r_status[oi] = st2
            if r_rid.shape[1] > cap:
                grow = r_rid.shape[1] - cap
                p_rid = np.concatenate([p_rid, np.zeros((n, grow), dtype=np.int32)], axis=1)
                p_tim = np.concatenate([p_tim, np.zeros((n, grow), dtype=np.int32)], axis=1)
                cap = r_rid.shape[1]
            p_rid[act, :r_rid.shape[1]] = r_rid
            p_tim[act, :r_rid.shape[1]] = r_time
            plen[act] = r_len
            last = r_rid[np.arange(m), r_len - 1]
            in_leg = leg[act] < nleg[act]
            hit = last == tgt
            adv = act[in_leg & hit]                                      # search.py:912-914
            leg[adv] += 1
            if vector:
                attempt[adv] = 0
            else:
                for i in adv.tolist():
                    order[i] = None
            miss = in_leg & ~hit
            if miss.any():
                self.fail_find_log.update(zip(last[miss].tolist(), tgt[miss].tolist()))      # search.py:917
                mi = act[miss]
                attempt[mi] += 1
                done[mi[attempt[mi] >= olen[mi]]] = True                 # search.py:918-919
            fin = ~in_leg
            bad = fin & (last != des_np[act])
            if bad.any():
                self.fail_find_log.update(zip(last[bad].tolist(), des_np[act][bad].tolist()))   # search.py:932-933
            done[act[fin]] = True
            if vector:
                set_targets(act[~done[act]])
            else:
                for i in act[~done[act]].tolist():
                    set_target(i)
        loc = [p_rid[i, :plen[i]].tolist() for i in range(n)]
        tim = [p_tim[i, :plen[i]].tolist() for i in range(n)]
        # share of the wall time spent inside the engine calls (the device runs supersteps throughout them; the rest is
        # the host between rounds: targets, prefixes, logs)
        self.last_stats = {'rounds': rounds, 'road_expansions': expansions,
                           'gpu_busy_frac': t_engine / max(1e-9, time.time() - t_begin)}
        return loc, tim, is_astar

    def astar_search(self, region_model, road_model, start_rid, start_tim, des, default_len, max_step=500):
        class _GlobalRng(object):      # the reference draws from the global numpy RNG (search.py:900)
            choice = staticmethod(np.random.choice)
        loc, tim, fl = self.generate(region_model, road_model, [start_rid], [start_tim], [des], [default_len],
                                     max_step=max_step, rngs=[_GlobalRng()], slots=1)
        return loc[0], tim[0], fl[0]

    def save_fail_log(self):
        """search.py:1258-1269: ./log/find_error_log.json and ./log/region_find_error_log.json."""
        os.makedirs('./log', exist_ok=True)
        with open('./log/find_error_log.json', 'w') as f:
            json.dump([[int(a), int(b)] for a, b in self.fail_find_log], f)
        with open('./log/region_find_error_log.json', 'w') as f:
            json.dump([[int(a), int(b)] for a, b in self.region_fail_find_log], f)
