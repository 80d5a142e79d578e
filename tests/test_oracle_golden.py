"""Pin the CPU oracle against the golden vectors captured from the reference."""
import numpy as np
import pytest
import torch

from oracle import model as om
from oracle import search as osr
from tests._fixtures import G11World, GoldenRegionWorld, GoldenRoadWorld, load_json, load_npz

RTOL = 1e-5   # oracle vs reference: same torch-CPU fp32 ops, only the op grouping differs


@pytest.fixture(scope='module')
def road():
    return GoldenRoadWorld()


@pytest.fixture(scope='module')
def region():
    return GoldenRegionWorld()


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_g1_gat_layer(case):
    z = load_npz('g1_gat.npz')
    t = lambda k: torch.as_tensor(z['%s_%s' % (case, k)])
    nh = z['%s_scoring_fn_source' % case].shape[1]
    fo = z['%s_scoring_fn_source' % case].shape[2]
    out = om.gat_forward(t('x'), torch.as_tensor(z['edge_index']), t('linear_proj.weight'), t('skip_proj.weight'),
                         t('scoring_fn_source'), t('scoring_fn_target'), t('bias'), nh, fo,
                         concat=bool(z['%s_concat' % case]))
    np.testing.assert_allclose(out.numpy(), z['%s_out' % case], rtol=RTOL, atol=1e-6)


PARAMS = ('linear_proj.weight', 'skip_proj.weight', 'scoring_fn_source', 'scoring_fn_target', 'bias')


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_g12_gat_layer_training(case):
    """Training mode: the reference layer under recorded dropout masks, its output and its autograd gradients."""
    z = load_npz('g12_gat_train.npz')
    t = lambda k: torch.as_tensor(z['%s_%s' % (case, k)])
    nh, fo = z['%s_scoring_fn_source' % case].shape[1:]
    x = t('x').clone().requires_grad_(True)
    par = {k: t(k).clone().requires_grad_(True) for k in PARAMS}
    out = om.gat_forward(x, torch.as_tensor(z['edge_index']), par['linear_proj.weight'], par['skip_proj.weight'],
                         par['scoring_fn_source'], par['scoring_fn_target'], par['bias'], nh, fo,
                         concat=bool(z['%s_concat' % case]), masks=(t('mask0'), t('mask1'), t('mask2')))
    np.testing.assert_allclose(out.detach().numpy(), z['%s_out' % case], rtol=RTOL, atol=1e-6)
    (out * t('up')).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), z['%s_grad_x' % case], rtol=1e-4, atol=1e-5)
    for k in PARAMS:
        g = par[k].grad if par[k].grad is not None else torch.zeros_like(par[k])
        np.testing.assert_allclose(g.numpy(), z['%s_grad_%s' % (case, k)], rtol=1e-4, atol=1e-5, err_msg=k)


def test_g13_function_h_training_loss(road):
    """DistanceGatFC.calculate_loss in training mode (distance_gat_fc.py:75-95) and every parameter's gradient."""
    z = load_npz('g13_h_train.npz')
    sd = {k: torch.as_tensor(v).clone().requires_grad_(True) for k, v in road.h_state.items()}
    fh = road.cfg['function_h']
    x = om.h_input_features(sd, torch.as_tensor(road.node_features), fh.get('no_gps_emb', False))
    emb = om.gat_forward(x, torch.as_tensor(road.edge_index), sd['gat_encoder.linear_proj.weight'],
                         sd['gat_encoder.skip_proj.weight'], sd['gat_encoder.scoring_fn_source'],
                         sd['gat_encoder.scoring_fn_target'], sd['gat_encoder.bias'], fh['num_of_heads'],
                         fh['embed_dim'], fh['concat'],
                         masks=tuple(torch.as_tensor(z['mask%d' % i]) for i in range(3)))
    cand, des, dis = torch.as_tensor(z['cand']), torch.as_tensor(z['des']), torch.as_tensor(z['dis'])
    score = torch.stack([om.h_scores(sd, emb, cand[b], int(des[b]), dis[b]) for b in range(cand.shape[0])])
    loss = torch.nn.functional.cross_entropy(score, torch.as_tensor(z['target']))
    assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-5 * max(1.0, abs(float(z["loss"])))
    loss.backward()
    for k, v in sd.items():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        scale = max(1e-6, float(np.abs(z['grad_' + k]).max()))
        np.testing.assert_allclose(g.numpy(), z['grad_' + k], rtol=1e-4, atol=1e-5 * scale + 1e-7, err_msg=k)


def test_g2_node_embedding(road, region):
    np.testing.assert_allclose(road.oracle_model().setup_node_emb().numpy(), road.npz['node_emb'],
                               rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(region.oracle_model().setup_node_emb().numpy(), region.npz['node_emb'],
                               rtol=RTOL, atol=1e-5)


def test_g234_predict(road):
    m = road.oracle_model()
    for c in load_json('g234_predict.json')['cases']:
        g = m.predict_g(c['loc'], c['tim'], c['cand']).numpy()
        h = m.predict_h(c['cand'], c['des'], c['dis']).numpy()
        p = m.predict(c['loc'], c['tim'], c['des'], c['cand'], np.asarray(c['dis'], dtype=np.float32)).numpy()
        np.testing.assert_allclose(g, c['g'], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(h, c['h'], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(p, c['p'], rtol=RTOL, atol=1e-7)


def test_g10_masked_batch_and_logits(road):
    """The reference's padded batch with trace_mask: the padded (zero) positions stay in the attention softmax
    (function_g_fc.py:33 assigns the masked_fill to `mask`); and FunctionGFC.forward's candidate logits."""
    m = road.oracle_model()
    g10 = load_json('g10_masked_batch.json')
    assert len(g10['batches']) >= 2
    shorter = 0
    for b in g10['batches']:
        T = len(b['loc'][0])
        for i in range(len(b['loc'])):
            n = int(sum(b['mask'][i]))
            shorter += int(n < T)
            loc, tim = b['loc'][i][:n], b['tim'][i][:n]
            p, g = m.predict_masked_row(loc, tim, b['des'][i], b['cand'][i], b['dis'][i], T)
            np.testing.assert_allclose(m.g_logits(loc, tim, b['cand'][i], T).numpy(), b['g_score'][i], rtol=RTOL, atol=1e-6)
            np.testing.assert_allclose(g.numpy(), b['g'][i], rtol=RTOL, atol=1e-7)
            np.testing.assert_allclose(p.numpy(), b['p'][i], rtol=RTOL, atol=1e-7)
    assert shorter > 0
    for c in g10['logits']:
        np.testing.assert_allclose(m.g_logits(c['loc'], c['tim'], c['cand']).numpy(), c['g_score'], rtol=RTOL, atol=1e-6)


def _check_search_case(got_loc, got_tim, stats, c):
    assert got_loc == c['loc']
    assert got_tim == c['tim']
    assert [p[0] for p in stats.pops] == c['pop_rid']
    np.testing.assert_allclose([p[1] for p in stats.pops], c['pop_prio'], rtol=1e-5, atol=1e-6)


def test_g5_astar(road):
    m = road.oracle_model()
    rtd = road.rtd.astype(int)
    cases = load_json('g5_astar.json')['cases']
    kinds = set()
    for c in cases:
        st = osr.SearchStats()
        loc, tim = osr.astar_search(m, road.adjacent_list, road.rid_gps, road.road_length, rtd,
                                    [c['start']], [c['start_time']], c['des'], c['default_len'], c['max_step'],
                                    stats=st)
        _check_search_case(loc, tim, st, c)
        kinds.add(st.status)
    assert kinds == {osr.REACHED, osr.FALLBACK, osr.UNTOUCHED}


def test_g11_bj_dims_config0():
    """The oracle at BJ model dims (hidden 256, 2-layer LSTM, GAT 5 x 256) against the REFERENCE's own run of
    BASELINE.json configs[0] (golden G11: search.py:240-424 driving generator_v4.py:56-72 under max_step 5000):
    trajectories, full pop order, priorities.  The reference took ~20 core-minutes for all 256 OD pairs; the CPU suite
    takes the 24 shortest searches, the median one (1 113 pops) and the longest one (1 982 pops); the GPU test
    (tests/test_gpu_g11.py) takes all 256."""
    w = G11World()
    model, cfg = w.model()
    m = w.oracle_model(model, cfg)
    rtd = w.net.road_time_distribution.astype(int)
    order = np.argsort(w.z['npop'], kind='stable')
    picked = [int(i) for i in order[:24]] + [int(order[128]), int(order[-1])]
    threads = torch.get_num_threads()
    torch.set_num_threads(1)          # batch-1 model calls: more threads only add synchronisation
    try:
        _g11_check(w, m, rtd, picked)
    finally:
        torch.set_num_threads(threads)


def _g11_check(w, m, rtd, picked):
    for i in picked:
        c = w.case(i)
        st = osr.SearchStats()
        loc, tim = osr.astar_search(m, w.net.adjacent_list, w.net.rid_gps, w.net.road_length, rtd,
                                    [int(w.od['start'][i])], [int(w.od['start_time'][i])], int(w.od['des'][i]),
                                    int(w.od['default_len'][i]), w.MAX_STEP, stats=st)
        assert loc == c['loc'] and tim == c['tim'], i
        assert [p[0] for p in st.pops] == c['pop_rid'], i
        np.testing.assert_allclose([p[1] for p in st.pops], c['pop_prio'], rtol=1e-5, atol=1e-6)


def test_g6_astar_in_region(road, region):
    m = road.oracle_model()
    rtd = road.rtd.astype(int)
    for c in load_json('g6_astar_in_region.json')['cases']:
        st = osr.SearchStats()
        rset = set(region.meta['region2rid'][str(c['region'])])
        loc, tim = osr.astar_search(m, road.adjacent_list, road.rid_gps, road.road_length, rtd,
                                    c['prefix_loc'], c['prefix_tim'], c['des'], c['default_len'], c['max_step'],
                                    region_rid_set=rset, stats=st)
        _check_search_case(loc, tim, st, c)


def test_g7_double_layer(road, region):
    road_m = road.oracle_model()
    reg_m = region.oracle_model()
    tables = region.tables(road)
    fail, rfail = set(), set()
    for c in region.meta['cases']:
        rtrace, flag = osr.region_astar_search(reg_m, tables['region_adjacent_list'], tables['region_dist'],
                                               tables['region_time_distribution'],
                                               tables['rid2region'][str(c['start'])], c['start_time'],
                                               tables['rid2region'][str(c['des'])], c['max_step'])
        assert rtrace == c['region_trace'] and flag == c['region_flag']
        rng = np.random.RandomState(c['seed'])
        loc, tim, is_astar = osr.double_layer_astar_search(reg_m, road_m, tables, c['start'], c['start_time'],
                                                           c['des'], c['default_len'], c['max_step'], rng=rng,
                                                           fail_log=fail, region_fail_log=rfail)
        assert loc == c['loc'] and tim == c['tim'] and is_astar == c['is_astar']
    assert sorted([list(map(int, x)) for x in fail]) == region.meta['fail_find_log']
    assert sorted([list(map(int, x)) for x in rfail]) == region.meta['region_fail_find_log']


def test_g8_random_sample(road, region):
    """road_random_sample / region_random_sample (search.py:614-716, :1163-1256): with torch's generator seeded
    like the reference run the oracle draws the same stream and must return the same trajectories; the
    probability vector of every draw is compared too."""
    g8 = load_json('g8_random_sample.json')
    m = road.oracle_model()
    rtd = road.rtd.astype(int)
    for c in g8['road']:
        log = []
        torch.manual_seed(c['seed'])
        loc, tim = osr.road_random_sample(m, road.adjacent_list, road.rid_gps, road.road_length, rtd,
                                          c['prefix_loc'], c['prefix_tim'], c['des'], c['default_len'], log=log)
        assert [int(x) for x in loc] == c['loc'] and [int(x) for x in tim] == c['tim']
        assert len(log) == len(c['draws'])
        for (p, i), (rp, ri) in zip(log, c['draws']):
            assert i == ri
            np.testing.assert_allclose(p, rp, rtol=1e-4, atol=1e-7)
    rm = region.oracle_model()
    t = region.tables(road)
    for c in g8['region']:
        log = []
        torch.manual_seed(c['seed'])
        loc, tim = osr.region_random_sample(rm, t['region_adjacent_list'], t['region_dist'],
                                            t['region_time_distribution'], c['prefix_loc'], c['prefix_tim'],
                                            c['des'], c['default_len'], log=log)
        assert [int(x) for x in loc] == c['loc'] and [int(x) for x in tim] == c['tim']
        assert [i for _, i in log] == [ri for _, ri in c['draws']]


def test_inverse_cdf_choice():
    p = np.array([0.25, 0.5, 0.25], dtype=np.float32)
    assert osr.inverse_cdf_choice(p, 0.0) == 0 and osr.inverse_cdf_choice(p, 0.2499) == 0
    assert osr.inverse_cdf_choice(p, 0.25) == 1 and osr.inverse_cdf_choice(p, 0.7499) == 1
    assert osr.inverse_cdf_choice(p, 0.75) == 2 and osr.inverse_cdf_choice(p, 0.99999) == 2
    assert osr.inverse_cdf_choice(np.array([0.3, 0.3], dtype=np.float32), 0.9) == 1   # sums short of u: last


def test_g9_beam_search(road):
    """Searcher.beam_search (search.py:137-238): trajectories, the size of the final draw set and the
    reference's IndexError when every branch dies."""
    om = road.oracle_model()
    cases = load_json('g9_beam_search.json')['cases']
    assert any('error' in c for c in cases) and any(c.get('index_bound', 1) > 1 for c in cases)
    for c in cases:
        rs = np.random.RandomState(c['seed'])
        info = {}
        args = (om, road.adjacent_list, road.rid_gps, c['prefix_loc'], c['prefix_tim'], c['des'])
        kw = dict(width=c['width'], max_step=c['max_step'], prob_threshold=c['prob_threshold'],
                  randint=rs.randint, info=info)
        if 'error' in c:
            with pytest.raises(IndexError):
                osr.beam_search(*args, **kw)
            continue
        loc, tim = osr.beam_search(*args, **kw)
        assert info['index_bound'] == c['index_bound']
        assert loc == c['loc'] and tim == c['tim']
