"""CPU restatement of the A* searches of search.py.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows search.py:14-52 (SearchNode), :240-424 (Searcher.astar_search), :426-612
(astar_search_in_region), :859-934 (DoubleLayerSearcher.astar_search) and :936-1117
(region-level A*).  ``queue.PriorityQueue`` is a locked ``heapq`` of ``(priority,
SearchNode)`` tuples; the oracle uses ``heapq`` directly with the same tuple shape so
that exact-priority ties fall to ``SearchNode.__lt__`` on the node's *current*
log_prob and then to heap position, as in the reference.
"""
import heapq
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
        cost, now = heapq.heappop(open_set)
        now_rid = now.rid
        if now_rid in close_set:                                         # :284-287
            continue
        close_set.add(now_rid)
        rid2node.pop(now_rid, None)
        stats.pops.append((now_rid, float(cost)))
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
