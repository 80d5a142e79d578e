"""Beam search on the device model (reference: Searcher.beam_search, search.py:137-238) for many OD pairs at once.

One layer of the reference = one model call per node of the beam over that node's WHOLE trace.  Here every node
owns a state row: the recurrent cell is advanced once from the parent's row (`tsg_lstm_step_batched`), the
candidates of all beam nodes of all OD pairs of a layer are scored in one launch (`tsg_score_candidates`:
attention over the rows of the node's trace, gathered out_fc rows, factored H), and the per-OD bookkeeping
(stable descending sort, cut to `width`, the `len(end_beam) == width` stop, the final draw among the end nodes
within `prob_threshold` of the best) follows the reference line by line on the host.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from . import geo


class _Beam(object):
    """Nodes of one OD pair: parallel lists (parent node, rid, time code, minute of day, log_prob, state row)."""

    def __init__(self):
        self.parent, self.rid, self.tcode, self.minute, self.lp, self.row = [], [], [], [], [], []

    def add(self, parent, rid, tcode, minute, lp):
        self.parent.append(parent)
        self.rid.append(rid)
        self.tcode.append(tcode)
        self.minute.append(minute)
        self.lp.append(lp)
        self.row.append(-1)
        return len(self.rid) - 1

    def path(self, i):
        loc, tim = [], []
        while i >= 0:
            loc.append(self.rid[i])
            tim.append(self.tcode[i])
            i = self.parent[i]
        return loc[::-1], tim[::-1]

    def rows(self, i):
        out = []
        while i >= 0:
            out.append(self.row[i])
            i = self.parent[i]
        return out[::-1]


@torch.no_grad()
def beam_search_batch(tables, adjacent_list, road_center_gps, prefixes, des, width=15, max_step=50,
                      prob_threshold=1e-4, seeds=None, randint=None):
    """prefixes: list of (loc_list, tim_list); des: list of destinations.  The final draw of OD pair b uses
    np.random.RandomState(seeds[b]).randint, or `randint` (one callable for all), or np.random.randint.
    Returns (locs, tims, errors): errors[b] is None or 'IndexError' (every branch died, search.py:230)."""
    L = _lib.lib()
    dev = tables.device
    n_od = len(des)
    hp, nl = tables.hp, tables.n_layers
    width, max_step = int(width), int(max_step)
    plen = [len(p[0]) for p in prefixes]
    if n_od == 0:
        return [], [], []
    if min(plen) < 1 or any(len(p[0]) != len(p[1]) for p in prefixes):
        raise ValueError('every OD pair needs a prefix of >= 1 point with matching loc / tim lengths')
    n_rows = sum(plen) + n_od * max(1, width) * max_step + n_od
    hs = [torch.empty((n_rows, hp), dtype=torch.float32, device=dev) for _ in range(nl)]
    cs = [torch.empty((n_rows, hp), dtype=torch.float32, device=dev) for _ in range(nl)]
    next_row = [0]
    stream_dev = torch.cuda.device(dev)

    def lstm_rows(prev_rows, rids, tcodes):
        """Advance the recurrent cell for a batch of new nodes; returns their state rows."""
        k = len(rids)
        out = np.arange(next_row[0], next_row[0] + k, dtype=np.int64)
        next_row[0] += k
        if next_row[0] > n_rows:
            raise RuntimeError('beam search state rows exhausted')
        prev_t = torch.from_numpy(np.asarray(prev_rows, dtype=np.int64)).to(dev)
        out_t = torch.from_numpy(out).to(dev)
        rid_t = torch.from_numpy(np.asarray(rids, dtype=np.int32)).to(dev)
        tc_t = torch.from_numpy(np.asarray(tcodes, dtype=np.int32)).to(dev)
        if int(rid_t.min()) < 0 or int(rid_t.max()) >= tables.xproj.shape[0]:
            raise IndexError('road id outside the road embedding table')
        if int(tc_t.min()) < 0 or int(tc_t.max()) >= tables.n_time:
            raise IndexError('time code outside the time embedding table')
        with stream_dev:
            for l in range(nl):
                s = _lib.LstmStep()
                s.layer, s.n_rows_host, s.n_rows_dev = l, k, None
                s.prev_idx, s.out_idx = _lib.ptr(prev_t), _lib.ptr(out_t)
                s.rid, s.tcode = _lib.ptr(rid_t), _lib.ptr(tc_t)
                s.h_below = _lib.ptr(hs[l - 1]) if l > 0 else None
                s.h, s.c = _lib.ptr(hs[l]), _lib.ptr(cs[l])
                s.state_rows = n_rows
                _lib.check(L.tsg_lstm_step_batched(C.byref(tables.c_model), C.byref(s), _lib.current_stream()),
                           'tsg_lstm_step_batched')
        return out

    # ---- the known prefix, one position per launch (all OD pairs together); the last point is the root node
    beams = [_Beam() for _ in range(n_od)]
    woff = []
    for b, (pl, pt) in enumerate(prefixes):
        off = 1440 if pt[-1] >= 1440 else 0                              # search.py:163-165
        woff.append(off)
        for k in range(len(pl)):
            beams[b].add(k - 1, int(pl[k]), int(pt[k]), int(pt[k]) - off if k == len(pl) - 1 else 0, 0.0)
    for k in range(max(plen) - 1):
        who = [b for b in range(n_od) if k < plen[b] - 1]
        rows = lstm_rows([beams[b].row[k - 1] if k > 0 else -1 for b in who],
                         [beams[b].rid[k] for b in who], [beams[b].tcode[k] for b in who])
        for b, r in zip(who, rows):
            beams[b].row[k] = int(r)
    cur = [[plen[b] - 1] for b in range(n_od)]
    end = [[] for _ in range(n_od)]
    active = [True] * n_od
    des = [int(d) for d in des]
    for _step in range(max_step):                                        # search.py:173
        work = [(b, i) for b in range(n_od) if active[b] for i in cur[b]
                if str(beams[b].rid[i]) in adjacent_list]                # :178
        if not any(active):
            break
        nxt = [[] for _ in range(n_od)]
        if work:
            rows = lstm_rows([beams[b].row[beams[b].parent[i]] if beams[b].parent[i] >= 0 else -1 for b, i in work],
                             [beams[b].rid[i] for b, i in work], [beams[b].tcode[i] for b, i in work])
            chain, chain_ptr, cand, cand_ptr, dis, des_rows = [], [0], [], [0], [], []
            for (b, i), r in zip(work, rows):
                beams[b].row[i] = int(r)
                chain.extend(beams[b].rows(i))
                chain_ptr.append(len(chain))
                cs_ = adjacent_list[str(beams[b].rid[i])]
                if len(cs_) > _lib.TSG_MAX_DEGREE:
                    raise ValueError('at most %d candidates per road' % _lib.TSG_MAX_DEGREE)
                cand.extend(int(c) for c in cs_)
                cand_ptr.append(len(cand))
                dis.extend(geo.candidate_dis(road_center_gps[str(des[b])],
                                             [road_center_gps[str(c)] for c in cs_]).astype(np.float32))
                des_rows.append(des[b])
            cand_np = np.asarray(cand, dtype=np.int32)
            if cand_np.min() < 0 or cand_np.max() >= tables.c_model.n_roads or max(des_rows) >= tables.n_nodes_h:
                raise IndexError('road id outside the model tables')
            t = {k: torch.from_numpy(np.asarray(v, dtype=dt)).to(dev) for k, v, dt in
                 (('chain', chain, np.int64), ('chain_ptr', chain_ptr, np.int32), ('cand', cand_np, np.int32),
                  ('cand_ptr', cand_ptr, np.int32), ('dis', dis, np.float32), ('des', des_rows, np.int32))}
            prob = torch.empty(len(cand), dtype=torch.float32, device=dev)
            sb = _lib.ScoreBatch()
            sb.n_rows = len(work)
            sb.chain_ptr, sb.chain, sb.h_top = _lib.ptr(t['chain_ptr']), _lib.ptr(t['chain']), _lib.ptr(hs[nl - 1])
            sb.cand_ptr, sb.cand, sb.dis, sb.des = (_lib.ptr(t['cand_ptr']), _lib.ptr(t['cand']), _lib.ptr(t['dis']),
                                                    _lib.ptr(t['des']))
            sb.prob, sb.prob_g, sb.prob_h = _lib.ptr(prob), None, None
            with stream_dev:
                _lib.check(L.tsg_score_candidates(C.byref(tables.c_model), C.byref(sb), _lib.current_stream()),
                           'tsg_score_candidates')
            parent_lp = torch.from_numpy(np.repeat(np.asarray([beams[b].lp[i] for b, i in work], dtype=np.float32),
                                                   np.diff(cand_ptr))).to(dev)
            lp = (torch.log(prob) + parent_lp).cpu().numpy()             # :194-196, fp32
            for w, (b, i) in enumerate(work):
                bm = beams[b]
                minute = (bm.minute[i] + 1) % 1440                       # :199: one minute per hop
                for k in range(cand_ptr[w], cand_ptr[w + 1]):
                    j = bm.add(i, cand[k], minute + woff[b], minute, float(lp[k]))
                    (end[b] if cand[k] == des[b] else nxt[b]).append(j)  # :207-212
        for b in range(n_od):
            if not active[b]:
                continue
            if len(end[b]) == width:                                     # :215 (exact equality, as written)
                active[b] = False
                continue
            order = sorted(nxt[b], key=lambda j: beams[b].lp[j], reverse=True)   # stable, :219
            cur[b] = order[:width]
    locs, tims, errors = [], [], []
    for b in range(n_od):
        bm = beams[b]
        eb = end[b] if end[b] else cur[b]                                # :223-224
        eb = sorted(eb, key=lambda j: bm.lp[j], reverse=True)
        if not eb:                                                       # the reference indexes an empty list (:230)
            locs.append(None)
            tims.append(None)
            errors.append('IndexError')
            continue
        index = 1
        best = np.exp(bm.lp[eb[0]])
        while index < len(eb):
            if best - np.exp(bm.lp[eb[index]]) > prob_threshold:
                break
            index += 1
        if seeds is not None:
            pick = int(np.random.RandomState(int(seeds[b])).randint(index))
        else:
            pick = int((randint or np.random.randint)(index))            # :237
        loc, tim = bm.path(eb[pick])
        locs.append(loc)
        tims.append(tim)
        errors.append(None)
    return locs, tims, errors
