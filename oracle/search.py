"""CPU restatement of the A* searches of search.py.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows search.py:14-52 (SearchNode), :240-424 (Searcher.astar_search), :426-612
(astar_search_in_region), :859-934 (DoubleLayerSearcher.astar_search), :936-1117
(region-level A*), :614-716 (road_random_sample), :1163-1256 (region_random_sample) and :137-238
(beam_search).  ``queue.PriorityQueue`` is a locked ``heapq`` of ``(priority,
SearchNode)`` tuples; the oracle uses ``heapq`` directly with the same tuple shape so
that exact-priority ties fall to ``SearchNode.__lt__`` on the node's *current*
log_prob and then to heap position, as in the reference.
"""
import heapq
import time
import math

import numpy as np

from . import geodesic

DEFAULT_SPEED = 8.334          # search.py:11

REACHED, FALLBACK, UNTOUCHED = 1, 2, 0


class _Node(object):
    """search.py:14-52."""
    __slots__ = ('trace_loc', 'trace_tim', 'rid', 'date_time', 'log_prob')

    def __init__(self, trace_loc, trace_tim, rid, date_time, log_prob):
        self.trace_loc = list(trace_loc)
        self.trace_tim = list(trace_tim)
        self.rid = rid
        self.date_time = date_time
        self.log_prob = log_prob

    def __lt__(self, other):
        return self.log_prob < other.log_prob


class SearchStats(object):
    def __init__(self):
        self.steps = 0          # expansions counted by `step` (search.py:420)
        self.model_calls = 0
        self.status = UNTOUCHED
        self.pops = []          # (rid, priority) of every non-stale pop, in order
        # diagnostics for the parity tests (they observe the search, they never change it):
        self.near_tol = None    # when set: relaxations of an OPEN road whose old and new log_prob lie within
        self.near_relax = []    # near_tol * max(1, |log_prob|) are listed here as (pop index, road, new - old):
        #                         the `<` of search.py:403 is then decided by the last bits (a documented near-tie)
        self.keep_traces = False
        self.deadline = None    # bench.py's CPU baseline: stop at this time.time() (the expansions so far are counted)
        self.traces = []        # with keep_traces: the trajectory (trace_loc) of every non-stale pop


def travel_cost(rtd, road_length, hour, rid):
    """search.py:308-313 / 376-381: seconds to traverse `rid` when entered in `hour`."""
    if rtd[hour][rid] != 1:
        return int(rtd[hour][rid])
    return int(round(road_length[str(rid)] / DEFAULT_SPEED))


def astar_search(model, adjacent_list, road_center_gps, road_length, rtd, trace_loc, trace_tim, des,
                 default_len, max_step=500, region_rid_set=None, stats=None, dist_fn=None):
    """search.py:240-424; with ``region_rid_set`` it is search.py:426-612 (expansion gated at :487).

    ``model.predict(trace_loc, trace_tim, des, cand, dis)`` returns the (C,) f32 probabilities.
    ``rtd`` must already be ``astype(int)`` (search.py:75).
    """
    if stats is None:
        stats = SearchStats()
    if dist_fn is None:
        dist_fn = geodesic.candidate_dis
    close_set = set()
    open_set = []
    rid2node = {}
    trace_loc = list(trace_loc)
    trace_tim = list(trace_tim)
    des_gps = road_center_gps[str(des)]
    start_time = trace_tim[-1]
    weekday_off = 1440 if start_time >= 1440 else 0                     # :265-267
    start_node = _Node(trace_loc, trace_tim, trace_loc[-1], (start_time - weekday_off) * 60, 0)
    heapq.heappush(open_set, (start_node.log_prob, start_node))          # :271
    rid2node[trace_loc[-1]] = start_node
    step = 0
    best_trace = None
    best_score = 0
    while open_set and step < max_step:                                  # :279
        if stats.deadline is not None and time.time() > stats.deadline:
            break
        cost, now = heapq.heappop(open_set)
        now_rid = now.rid
        if now_rid in close_set:                                         # :284-287
            continue
        close_set.add(now_rid)
        rid2node.pop(now_rid, None)
        stats.pops.append((now_rid, float(cost)))
        if stats.keep_traces:
            stats.traces.append(list(now.trace_loc))
        if now_rid == des:                                               # :291-297
            best_trace = (now.trace_loc, now.trace_tim)
            stats.status = REACHED
            break
        gate = str(now_rid) in adjacent_list
        if region_rid_set is not None:
            gate = gate and (now_rid in region_rid_set)                  # :487
        if gate:
            cand = adjacent_list[str(now_rid)]
            if len(cand) == 1:                                           # :302-305
                cand_lp = [now.log_prob]
            else:
                dis = [dist_fn(des_gps, road_center_gps[str(c)]) for c in cand]          # :352-357
                prob = model.predict(now.trace_loc, now.trace_tim, des, cand,
                                     np.asarray(dis, dtype=np.float32))                  # :359-367
                stats.model_calls += 1
                lp = prob.log()                                          # :370 (fp32)
                lp += now.log_prob                                       # :373 (fp32 += python float)
                cand_lp = [lp[i].item() for i in range(len(cand))]       # :386
            hour = now.date_time // 3600                                 # :308 / :376
            cost_time = travel_cost(rtd, road_length, hour, now_rid)
            next_time = (now.date_time + cost_time) % 86400
            next_code = next_time // 60 + weekday_off
            new_tim = now.trace_tim + [next_code]
            for c, c_lp in zip(cand, cand_lp):
                c_score = np.exp(c_lp)
                if stats.near_tol is not None and c in rid2node and \
                        abs(c_lp - rid2node[c].log_prob) < stats.near_tol * max(1.0, abs(c_lp)):
                    stats.near_relax.append((len(stats.pops) - 1, c, c_lp - rid2node[c].log_prob))
                if c not in rid2node and c not in close_set:             # :388 / :319
                    node = _Node(now.trace_loc + [c], new_tim, c, next_time, c_lp)
                    heapq.heappush(open_set, (-c_lp, node))
                    rid2node[c] = node
                    if len(node.trace_loc) == default_len and c_score > best_score:      # :397-402
                        best_trace = (node.trace_loc, node.trace_tim)
                        best_score = c_score
                elif c in rid2node and rid2node[c].log_prob < c_lp:      # :403 / :334
                    node = rid2node[c]
                    node.log_prob = c_lp
                    node.trace_loc = now.trace_loc + [c]
                    node.trace_tim = list(new_tim)
                    node.date_time = next_time
                    heapq.heappush(open_set, (-c_lp, node))
                    if len(node.trace_loc) == default_len and c_score > best_score:      # :414-419
                        best_trace = (node.trace_loc, node.trace_tim)
                        best_score = c_score
        step += 1                                                        # :420
    stats.steps = step
    if best_trace is not None:                                           # :421-424
        if stats.status != REACHED:
            stats.status = FALLBACK
        return best_trace[0], best_trace[1]
    return trace_loc, trace_tim


def region_astar_search(region_model, region_adjacent_list, region_dist, region_time_distribution,
                        start_region, start_time, des_region, max_step=500, stats=None):
    """search.py:936-1117 (DoubleLayerSearcher.astar_search_in_region). Returns (trace, is_astar)."""
    if stats is None:
        stats = SearchStats()
    close_set = set()
    open_set = []
    region2node = {}
    weekday_off = 1440 if start_time >= 1440 else 0
    start_node = _Node([start_region], [start_time], start_region, (start_time - weekday_off) * 60, 0)
    heapq.heappush(open_set, (start_node.log_prob, start_node))
    region2node[start_region] = start_node
    step = 0
    best_default_len = 10                                                # :967
    best_trace = None
    best_score = 0
    while open_set and step < max_step:
        cost, now = heapq.heappop(open_set)
        now_region = now.rid
        if now_region in close_set:
            continue
        close_set.add(now_region)
        region2node.pop(now_region, None)
        stats.pops.append((now_region, float(cost)))
        if now_region == des_region:
            best_trace = (now.trace_loc, now.trace_tim)
            break
        if str(now_region) in region_adjacent_list:
            cand = [int(k) for k in region_adjacent_list[str(now_region)].keys()]        # :992 (eval of int keys)
            if len(cand) == 0:                                           # :993-995 (no step += 1)
                continue
            hour = now.date_time // 3600
            if len(cand) == 1:
                cand_lp = [now.log_prob]
            else:
                dis = [region_dist[c][des_region] / 100 for c in cand]   # :1049
                prob = region_model.predict(now.trace_loc, now.trace_tim, des_region, cand,
                                            np.asarray(dis, dtype=np.float32))
                stats.model_calls += 1
                lp = prob.log()
                lp += now.log_prob
                cand_lp = [lp[i].item() for i in range(len(cand))]
            for c, c_lp in zip(cand, cand_lp):
                c_score = np.exp(c_lp)
                if region_time_distribution[hour][now_region] != 1:      # :1073 / :1006
                    cost_time = int(region_time_distribution[hour][now_region])
                else:
                    cost_time = int(round(region_dist[now_region][c] / DEFAULT_SPEED))   # :1076
                next_time = (now.date_time + cost_time) % 86400
                next_code = next_time // 60 + weekday_off
                new_tim = now.trace_tim + [next_code]
                if c not in region2node and c not in close_set:
                    node = _Node(now.trace_loc + [c], new_tim, c, next_time, c_lp)
                    heapq.heappush(open_set, (-c_lp, node))
                    region2node[c] = node
                    if len(node.trace_loc) == best_default_len and c_score > best_score:
                        best_trace = (node.trace_loc, node.trace_tim)
                        best_score = c_score
                elif c in region2node and region2node[c].log_prob < c_lp:
                    node = region2node[c]
                    node.log_prob = c_lp
                    node.trace_loc = now.trace_loc + [c]
                    node.trace_tim = new_tim
                    node.date_time = next_time
                    heapq.heappush(open_set, (-c_lp, node))
                    if len(node.trace_loc) == best_default_len and c_score > best_score:
                        best_trace = (node.trace_loc, node.trace_tim)
                        best_score = c_score
        step += 1
    stats.steps = step
    assert best_trace is not None                                        # :1113
    if best_trace[0][-1] == des_region:
        return best_trace[0], 1
    return best_trace[0], 0


def double_layer_astar_search(region_model, road_model, tables, start_rid, start_tim, des, default_len,
                              max_step=500, rng=None, fail_log=None, region_fail_log=None, stats=None):
    """search.py:859-934.  ``tables`` is a dict with the DoubleLayerSearcher constructor arguments
    (search.py:826-828).  ``rng`` replaces the global ``np.random`` of :900 (per-OD RandomState)."""
    if rng is None:
        rng = np.random
    rid2region = tables['rid2region']
    start_region = rid2region[str(start_rid)]
    des_region = rid2region[str(des)]
    region_trace, is_astar = region_astar_search(region_model, tables['region_adjacent_list'],
                                                 tables['region_dist'], tables['region_time_distribution'],
                                                 start_region, start_tim, des_region, max_step)
    if is_astar != 1 and region_fail_log is not None:
        region_fail_log.add((start_region, des_region))
    now_region = region_trace[0]
    trace_loc = [start_rid]
    trace_tim = [start_tim]
    road_args = (tables['adjacent_list'], tables['road_center_gps'], tables['road_length'],
                 tables['road_time_distribution'])
    total = stats if stats is not None else SearchStats()
    for target_region in region_trace[1:]:
        border = tables['region_transfer_freq'][str(now_region)][str(target_region)]
        border_set = border['transfer_rid']
        freq = np.array(border['transfer_freq'])
        order = rng.choice(border_set, len(border_set), p=freq / np.sum(freq), replace=False)   # :900-902
        region_set = set(tables['region2rid'][str(now_region)])
        success = False
        for mid in order:
            st = SearchStats()
            trace_loc, trace_tim = astar_search(road_model, *road_args, trace_loc, trace_tim, int(mid),
                                                default_len, max_step, region_rid_set=region_set, stats=st)
            total.steps += st.steps
            total.model_calls += st.model_calls
            if trace_loc[-1] == mid:
                success = True
                break
            elif fail_log is not None:
                fail_log.add((trace_loc[-1], int(mid)))
        if not success:
            return trace_loc, trace_tim, is_astar
        now_region = target_region
    region_set = set(tables['region2rid'][str(now_region)])
    st = SearchStats()
    if des in region_set:                                                # :925
        trace_loc, trace_tim = astar_search(road_model, *road_args, trace_loc, trace_tim, des, default_len,
                                            max_step, region_rid_set=region_set, stats=st)
    else:
        trace_loc, trace_tim = astar_search(road_model, *road_args, trace_loc, trace_tim, des, default_len,
                                            max_step, stats=st)
    total.steps += st.steps
    total.model_calls += st.model_calls
    if trace_loc[-1] != des and fail_log is not None:
        fail_log.add((trace_loc[-1], des))
    return trace_loc, trace_tim, is_astar


# ---------------------------------------------------------------------------------------------------
# stochastic rollouts (search.py:614-716 and :1163-1256)
def multinomial_choice(prob):
    """search.py:686 / :1233: `prob.multinomial(1).item()` on torch's global CPU generator."""
    return int(prob.multinomial(1).item())


def inverse_cdf_choice(prob, u):
    """The draw rule of the device path (include/tsg.h, tsg_rollout): first i with u < p_0 + ... + p_i,
    accumulated in fp32 in candidate order; the last candidate when no prefix sum exceeds u."""
    cum = np.float32(0.0)
    for i, p in enumerate(np.asarray(prob, dtype=np.float32)):
        cum = np.float32(cum + p)
        if np.float32(u) < cum:
            return i
    return len(prob) - 1


def road_random_sample(model, adjacent_list, road_center_gps, road_length, rtd, trace_loc, trace_tim, des,
                       default_len, choose=None, dist_fn=None, log=None):
    """search.py:614-716.  ``choose(prob, k)`` picks the index of the k-th model evaluation (default: torch
    multinomial like the reference).  ``log`` collects (prob list, chosen index) per evaluation."""
    if choose is None:
        choose = lambda p, k: multinomial_choice(p)
    if dist_fn is None:
        dist_fn = geodesic.candidate_dis
    trace_loc = list(trace_loc)
    trace_tim = list(trace_tim)
    start_time = trace_tim[-1]
    weekday_off = 1440 if start_time >= 1440 else 0                      # :631-634
    step = len(trace_loc) - 1                                            # :635
    max_step = max(20, default_len)                                      # :636
    now_road = trace_loc[-1]
    now_date_time = (start_time - weekday_off) * 60                      # :638 (never advanced afterwards)
    des_gps = road_center_gps[str(des)]
    k = 0
    while now_road != des and step < max_step:                           # :640
        if str(now_road) not in adjacent_list:
            break                                                        # :708-710
        cand = adjacent_list[str(now_road)]
        if len(cand) == 0:
            break                                                        # :646-648
        if len(cand) == 1:
            next_road = cand[0]                                          # :650-665
        else:
            dis = [dist_fn(des_gps, road_center_gps[str(c)]) for c in cand]          # :667-672
            prob = model.predict(trace_loc, trace_tim, des, cand, dis)               # :680-683
            idx = choose(prob, k)                                                    # :686
            if log is not None:
                log.append(([float(x) for x in prob], int(idx)))
            k += 1
            next_road = cand[idx]
        cost = travel_cost(rtd, road_length, now_date_time // 3600, now_road)        # :656-660 / :698-702
        next_time = (now_date_time + cost) % 86400
        trace_tim.append(next_time // 60 + weekday_off)
        trace_loc.append(next_road)
        now_road = next_road
        step += 1                                                        # :711
    if trace_loc[-1] == des:
        return trace_loc, trace_tim                                      # :712-713
    return trace_loc[:default_len], trace_tim[:default_len]              # :714-716


def region_random_sample(region_model, region_adjacent_list, region_dist, region_time_distribution, trace_loc,
                         trace_tim, des, default_len, choose=None, log=None):
    """search.py:1163-1256."""
    if choose is None:
        choose = lambda p, k: multinomial_choice(p)
    trace_loc = list(trace_loc)
    trace_tim = list(trace_tim)
    start_time = trace_tim[-1]
    weekday_off = 1440 if start_time >= 1440 else 0
    step = len(trace_loc) - 1
    max_step = max(20, default_len)
    now_region = trace_loc[-1]
    now_date_time = (start_time - weekday_off) * 60                      # :1188 (never advanced afterwards)
    rtd = region_time_distribution
    k = 0
    while now_region != des and step < max_step:                         # :1189
        if str(now_region) not in region_adjacent_list:
            break                                                        # :1248-1250
        cand = [int(c) for c in region_adjacent_list[str(now_region)].keys()]        # :1194-1195
        if len(cand) == 0:
            break
        hour = now_date_time // 3600
        if len(cand) == 1:
            next_region = cand[0]
            who = next_region                                            # :1207 indexes the NEXT region
        else:
            dis = [region_dist[c][des] / 100 for c in cand]              # :1218-1220
            prob = region_model.predict(trace_loc, trace_tim, des, cand, dis)
            idx = choose(prob, k)                                        # :1233
            if log is not None:
                log.append(([float(x) for x in prob], int(idx)))
            k += 1
            next_region = cand[idx]
            who = now_region                                             # :1238 indexes the CURRENT region
        if rtd[hour][who] != 1:
            cost = int(rtd[hour][who])
        else:
            cost = int(round(region_dist[now_region][next_region] / DEFAULT_SPEED))  # :1210-1211 / :1241-1242
        next_time = (now_date_time + cost) % 86400
        trace_tim.append(next_time // 60 + weekday_off)
        trace_loc.append(next_region)
        now_region = next_region
        step += 1
    if trace_loc[-1] == des:
        return trace_loc, trace_tim
    return trace_loc[:default_len], trace_tim[:default_len]


def beam_search(model, adjacent_list, road_center_gps, trace_loc, trace_tim, des, width=15, max_step=50,
                prob_threshold=1e-4, randint=None, dist_fn=None, info=None):
    """search.py:137-238 (Searcher.beam_search).  Each layer expands every node of the beam (one model call per
    node that has successors, time advancing one MINUTE per hop, :199), children that are the destination go to
    end_beam, the others are sorted by log_prob (stable, descending) and cut to `width`; the loop stops early
    only when end_beam holds EXACTLY `width` nodes (:215); the result is drawn uniformly among the end nodes whose
    probability is within prob_threshold of the best (:227-238).  With no end node and an empty beam the
    reference indexes an empty list: IndexError, as here.

    ``randint(n)`` stands in for np.random.randint(n) (:237); ``info`` (dict) receives 'index_bound'."""
    if dist_fn is None:
        dist_fn = geodesic.candidate_dis
    if randint is None:
        randint = np.random.randint
    trace_loc = list(trace_loc)
    trace_tim = list(trace_tim)
    des_gps = road_center_gps[str(des)]
    start_time = trace_tim[-1]
    weekday_off = 1440 if start_time >= 1440 else 0                       # :163-165
    cur_beam = [_Node(trace_loc, trace_tim, trace_loc[-1], start_time - weekday_off, 0)]
    end_beam = []
    step = 0
    while step < max_step:                                               # :173
        next_beam = []
        for node in cur_beam:
            if str(node.rid) not in adjacent_list:                       # :178
                continue
            cand = adjacent_list[str(node.rid)]
            dis = [dist_fn(des_gps, road_center_gps[str(c)]) for c in cand]               # :181-185
            prob = model.predict(node.trace_loc, node.trace_tim, des, cand, np.asarray(dis, dtype=np.float32))
            lp = prob.log()                                              # :194
            lp += node.log_prob                                          # :196
            next_time = (node.date_time + 1) % 1440                      # :199
            new_tim = node.trace_tim + [next_time + weekday_off]
            for i, c in enumerate(cand):
                child = _Node(node.trace_loc + [c], new_tim, c, next_time, lp[i].item())
                (end_beam if c == des else next_beam).append(child)      # :207-212
        if len(end_beam) == width:                                       # :215
            break
        next_beam = sorted(next_beam, key=lambda x: x.log_prob, reverse=True)
        cur_beam = next_beam[:width]
        step += 1
    if len(end_beam) == 0:                                               # :223
        end_beam = cur_beam
    end_beam = sorted(end_beam, key=lambda x: x.log_prob, reverse=True)
    index = 1
    best_prob = np.exp(end_beam[0].log_prob)                             # IndexError when every branch died
    while index < len(end_beam):
        if best_prob - np.exp(end_beam[index].log_prob) > prob_threshold:
            break
        index += 1
    if info is not None:
        info['index_bound'] = index
    pick = int(randint(index))
    return end_beam[pick].trace_loc, end_beam[pick].trace_tim
