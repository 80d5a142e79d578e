"""BJ model dims (hidden 256, 2-layer LSTM, GAT 5 x 256) against the REFERENCE's own output and the oracle, through
the C ABI on MI355X.

* G11 (tests/golden/g11_bj_dims_config0.npz): BASELINE.json configs[0] exactly as the reference ran it
  (search.py:240-424 driving generator_v4.py:56-72; 32 x 32 street grid, 256 OD pairs, max_step 5000, weights from
  the reference constructors under torch.manual_seed(0)): all 256 trajectories and full pop traces, for both forms
  of the recurrence (state_format 1 = split-f16 three-pass matrix-core form, 0 = plain fp32 matrix cores).  Every
  trajectory AND pop order is identical to the reference's or a proven near-tie (the first diverging pop within
  1e-5 in priority; the counts are printed and held to the recorded ones); every priority (= -sum log f, the
  f-score the north star bounds) up to such a divergence, all of them otherwise, is within 1e-4 relative of the
  reference's.
* f-scores at long prefixes: the probabilities of the first expansion behind prefixes of L = 1 .. 256 roads (the
  recurrence run over the whole prefix in the engine's own state format, the attention over all L rows incl. the
  64-row hops of the ancestor lists) against the oracle, g / h / p / log p within 1e-4 relative; the same through
  the explicit-row entry points (GeneratorV4.predict_parts: tsg_lstm_step_batched + tsg_score_candidates).
"""
import numpy as np
import pytest
import torch

from tests._fixtures import G11World, classify_traj, report_near_ties

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
RTOL = 1e-4          # north_star: f-scores within 1e-4 relative in fp32
PREFIX_LENS = (1, 2, 17, 64, 65, 200, 256)

# near-ties recorded on MI355X (deterministic kernels: the counts do not vary from run to run)
RECORDED_NEAR_TIES = {1: 14, 0: 15}
RECORDED_RELAX_TIES = {1: 2, 0: 2}


@pytest.fixture(scope='module')
def world():
    w = G11World()
    model, cfg = w.model(DEV)
    return w, model, cfg


@pytest.mark.parametrize('state_format', [1, 0])
def test_g11_reference_at_bj_dims(world, state_format):
    from ts_trajgen_amd.search import Searcher
    w, model, cfg = world
    od, net = w.od, w.net
    s = Searcher(DEV, net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution)
    s.state_format = state_format
    cap = int(w.z['npop'].max()) + 64
    res = s.generate(model, od['start'], od['start_time'], od['des'], od['default_len'], max_step=w.MAX_STEP,
                     slots=256, lmax=256, poplog_cap=cap)
    kinds, worst, n_cmp, where, order_only, relax_ties, all_rel = [], 0.0, 0, None, 0, [], []
    om_box = []

    def oracle():
        if not om_box:
            om_box.append(w.oracle_model(model, cfg))
        return om_box[0]
    for i in range(256):
        c = w.case(i)
        kind = classify_traj(res, i, c['loc'], c['tim'], c['pop_rid'], c['pop_prio'])
        k = int(res['poplog_n'][i])
        got_r = res['poplog_rid'][i, :k].tolist()
        got, want = res['poplog_prio'][i, :k].numpy().astype(np.float64), c['pop_prio'].astype(np.float64)
        # the common head of the two pop traces: same roads in the same order
        m = min(k, len(c['pop_rid']))
        same = np.asarray(got_r[:m]) == np.asarray(c['pop_rid'][:m])
        head = m if same.all() else int(np.argmin(same))
        if kind == 'exact' and head < max(k, len(c['pop_rid'])):
            # the same trajectory, but two pops changed places on the way: that, too, must be a proven near-tie
            assert head < m and abs(got[head] - want[head]) < 1e-5 * max(1.0, abs(want[head])), \
                'OD %d: pop order diverges at pop %d by %g (not a near-tie)' % (i, head, got[head] - want[head])
            kind = 'near-tie'
            order_only += 1
        kinds.append(kind)
        # Every priority up to the divergence (all of them for an exact search) within 1e-4 relative of the reference.
        # Numerical error is far smaller (~1e-7: see worst_clean below).  A deviation above 1e-5 has another cause:
        # the relaxation of an open road whose old and new log_prob agree to the last bits (search.py:403's `<`, a
        # near-tie): the road keeps the other parent, its own priority is the same, but everything expanded
        # behind it carries a different prefix through the recurrence.  Such a deviation must be PROVEN: the
        # oracle (pinned to this very golden) re-runs the search and must show a near-tie relaxation, made earlier,
        # on the path of the deviating pop; the pops of that search behind it are then not compared.
        rel = np.abs(got[:head] - want[:head]) / np.maximum(np.abs(want[:head]), 1.0)   # below 1 (the start node's 0): absolute
        if head and float(rel.max()) > RTOL:
            raise AssertionError('OD %d: priority off by %g relative at pop %d' % (i, rel.max(), int(rel.argmax())))
        big = np.flatnonzero(rel > 1e-5)
        if len(big):
            j = int(big[0])
            st = _oracle_search(w, oracle(), i)
            assert [p[0] for p in st.pops] == c['pop_rid']
            path = set(st.traces[j])
            proof = [r for r in st.near_relax if r[0] < j and r[1] in path]
            assert proof, 'OD %d: priority off by %g relative at pop %d and no near-tie relaxation on its path' % (
                i, rel[j], j)
            relax_ties.append((i, j, float(rel[j]), proof[0]))
            head, rel = j, rel[:j]
        n_cmp += head
        if head:
            all_rel.append(rel)
            if float(rel.max()) > worst:
                jj = int(rel.argmax())
                worst, where = float(rel.max()), (i, jj, float(want[jj]), float(got[jj]))
    print('[g11] state_format %d: worst relative priority error over %d of %d pops: %.3g (OD %d, pop %d: reference '
          '%.7g, here %.7g); %d of the near-ties below change the pop order only (same trajectory); near-tie '
          'relaxations that change the path behind a road (OD, pop, relative deviation there, (pop, road, gap) of '
          'the relaxation): %s' % ((state_format, n_cmp, int(w.z['npop'].sum()), worst) + where + (order_only, relax_ties)))
    q = np.quantile(np.concatenate(all_rel), [0.5, 0.99, 0.9999])
    print('[g11] relative priority error: median %.2g, 99 %% %.2g, 99.99 %% %.2g' % tuple(q))
    assert worst <= 1e-5 and len(relax_ties) <= RECORDED_RELAX_TIES[state_format]
    report_near_ties('G11 reference at BJ dims, state_format %d' % state_format, kinds, RECORDED_NEAR_TIES[state_format])
    lens = res['len'].numpy()
    assert (lens[[k == 'exact' for k in kinds]] == w.z['len'][[k == 'exact' for k in kinds]]).all()
    assert n_cmp >= 0.95 * int(w.z['npop'].sum())      # the near-ties leave nearly every pop compared


def _oracle_search(w, om, i):
    """OD pair i of G11 through the oracle, single-threaded torch (batch-1 model calls), with the diagnostics on."""
    from oracle import search as osr
    st = osr.SearchStats()
    st.near_tol, st.keep_traces = 1e-5, True
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        osr.astar_search(om, w.net.adjacent_list, w.net.rid_gps, w.net.road_length,
                         w.net.road_time_distribution.astype(int), [int(w.od['start'][i])],
                         [int(w.od['start_time'][i])], int(w.od['des'][i]), int(w.od['default_len'][i]), w.MAX_STEP,
                         stats=st)
    finally:
        torch.set_num_threads(threads)
    return st


def _prefix_cases(w, n_per_len=6, seed=11):
    """Random walks of L roads on the street grid (each hop a successor; roads may repeat: a prefix is not checked
    against the closed set, search.py:264-272) that end on a road with at least two successors, a destination
    elsewhere, time codes following the travel-cost rule loosely (any code in [0, 2880) is a valid input)."""
    rs = np.random.RandomState(seed)
    net = w.net
    n = net.num_roads
    cases = []
    for L in PREFIX_LENS:
        made = 0
        while made < n_per_len:
            cur = int(rs.randint(n))
            loc = [cur]
            ok = True
            for _ in range(L - 1):
                succ = net.adjacent_list.get(str(cur), [])
                if not succ:
                    ok = False
                    break
                cur = int(succ[rs.randint(len(succ))])
                loc.append(cur)
            if not ok or len(net.adjacent_list.get(str(cur), [])) < 2:
                continue
            t0 = int(rs.randint(2880 - 1440 * (L > 1)))
            tim = [min(2879, t0 + k // 2) for k in range(L)]
            des = int(rs.randint(n))
            if des in net.adjacent_list[str(cur)] or des == cur:
                continue
            cases.append({'loc': loc, 'tim': tim, 'des': des, 'cand': [int(c) for c in net.adjacent_list[str(cur)]]})
            made += 1
    return cases


def _oracle_parts(w, om, c):
    from oracle.geodesic import candidate_dis
    gps = w.net.rid_gps
    dis = np.array([candidate_dis(gps[str(c['des'])], gps[str(k)]) for k in c['cand']], dtype=np.float32)  # search.py:352-357
    g = om.predict_g(c['loc'], c['tim'], c['cand']).numpy()
    h = om.predict_h(c['cand'], c['des'], dis).numpy()
    p = om.predict(c['loc'], c['tim'], c['des'], c['cand'], dis).numpy()
    return dis, g, h, p


class _sharper(object):
    """Random-init weights give nearly uniform probabilities, which forgive errors in the attention-pooled state.
    Scaling function G's out_fc weights makes its softmax sharp (x100: g from 0.01 to 0.6), so the 1e-4 bound
    then really binds on the recurrence and the attention."""

    def __init__(self, model, scale):
        self.w, self.scale = model.function_g.out_fc.weight, float(scale)

    def __enter__(self):
        if self.scale != 1.0:
            with torch.no_grad():
                self.keep = self.w.detach().clone()
                self.w.mul_(self.scale)

    def __exit__(self, *exc):
        if self.scale != 1.0:
            with torch.no_grad():
                self.w.copy_(self.keep)


@pytest.mark.parametrize('sharp', [1.0, 100.0])
def test_f_scores_at_long_prefixes_engine_both_formats(world, sharp):
    """The engine's own path at BJ dims: prefixes of L = 1 .. 256 roads are run through the recurrence in the engine's
    state format (split-f16 words or fp32), the first expansion attends over all L rows; its log-probabilities are
    read from the slot's scored-candidate row (max_step 1: nothing follows) and held to the oracle."""
    from ts_trajgen_amd.search import Searcher
    w, model, cfg = world
    with _sharper(model, sharp):
        _engine_f_scores(w, model, cfg, sharp)


def _engine_f_scores(w, model, cfg, sharp):
    from ts_trajgen_amd.search import Searcher
    net = w.net
    om = w.oracle_model(model, cfg)
    cases = _prefix_cases(w)
    want = [_oracle_parts(w, om, c) for c in cases]
    spread = max(float(x[3].max() / x[3].min()) for x in want)
    worst = {}
    for fmt in (1, 0):
        s = Searcher(DEV, net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution)
        s.state_format = fmt
        n = len(cases)
        res = s.generate(model, None, None, [c['des'] for c in cases], [len(c['loc']) + 1 for c in cases], max_step=1,
                         prefixes=[(c['loc'], c['tim']) for c in cases], slots=n, lmax=512)
        assert (res['steps'].numpy() == 1).all()
        eng = list(s._engines.values())[0]
        ss = eng.slot_state.cpu().numpy()
        lp = eng.cand_lp.cpu().numpy()
        cid = eng.cand_id.cpu().numpy()
        seen = set()
        err = 0.0
        for b in range(eng.slots):
            if ss[b, 23] != 1:          # SS_GEN: this slot ran exactly one search
                continue
            i = int(ss[b, 22])          # SS_OD
            seen.add(i)
            c = cases[i]
            C = len(c['cand'])
            assert cid[b, :C].tolist() == c['cand']
            p = want[i][3].astype(np.float64)
            got_lp = lp[b, :C].astype(np.float64)      # log f + the parent's log_prob (0 for the start node)
            np.testing.assert_allclose(np.exp(got_lp), p, rtol=RTOL, err_msg='L=%d' % len(c['loc']))
            np.testing.assert_allclose(got_lp, np.log(p), rtol=RTOL, atol=1e-6, err_msg='L=%d' % len(c['loc']))
            err = max(err, float(np.abs(np.exp(got_lp) / p - 1).max()))
        assert seen == set(range(n))
        worst[fmt] = err
    print('[f-scores] out_fc x%g (p_max / p_min up to %.3g): worst relative error of p over prefixes L in %s: '
          'split-f16 %.3g, fp32 %.3g' % (sharp, spread, PREFIX_LENS, worst[1], worst[0]))


@pytest.mark.parametrize('sharp', [1.0, 100.0])
def test_f_scores_at_long_prefixes_predict_parts(world, sharp):
    """GeneratorV4.predict_parts (tsg_lstm_step_batched + tsg_score_candidates) at BJ dims against the oracle for
    prefixes of L = 1 .. 256: g, h, p and log p within 1e-4 relative."""
    w, model, cfg = world
    with _sharper(model, sharp):
        _predict_parts_f_scores(w, model, cfg, sharp)


def _predict_parts_f_scores(w, model, cfg, sharp):
    om = w.oracle_model(model, cfg)
    worst = 0.0
    for c in _prefix_cases(w, n_per_len=3, seed=12):
        dis, g, h, p = _oracle_parts(w, om, c)
        tl = torch.LongTensor(c['loc']).unsqueeze(0).to(DEV)
        tt = torch.LongTensor(c['tim']).unsqueeze(0).to(DEV)
        cs = torch.LongTensor(c['cand']).unsqueeze(0).to(DEV)
        cd = torch.from_numpy(dis).unsqueeze(0).to(DEV)
        dt = torch.LongTensor([c['des']]).to(DEV)
        gp, gg, gh = model.predict_parts(tl, tt, dt, cs, cd)
        msg = 'L=%d' % len(c['loc'])
        np.testing.assert_allclose(gg.cpu().numpy()[0], g, rtol=RTOL, err_msg=msg)
        np.testing.assert_allclose(gh.cpu().numpy()[0], h, rtol=RTOL, err_msg=msg)
        np.testing.assert_allclose(gp.cpu().numpy()[0], p, rtol=RTOL, err_msg=msg)
        np.testing.assert_allclose(torch.log(gp).cpu().numpy()[0], np.log(p), rtol=RTOL, atol=1e-6, err_msg=msg)
        worst = max(worst, float(np.abs(gp.cpu().numpy()[0] / p - 1).max()))
    print('[f-scores] predict_parts, out_fc x%g: worst relative error of p: %.3g' % (sharp, worst))
