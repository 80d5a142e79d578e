"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors from the reference and
against the CPU oracle.  Tolerances: fp32 results within 1e-4 relative (north_star); road sequences
exact except at documented near-ties (first diverging pop has |delta priority| < 1e-5)."""
import numpy as np
import pytest
import torch

from tests._fixtures import (report_near_ties, GoldenRegionWorld, GoldenRoadWorld, build_generator, classify_traj, load_json,
                             load_npz)

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
RTOL = 1e-4


@pytest.fixture(scope='module')
def road():
    return GoldenRoadWorld()


@pytest.fixture(scope='module')
def road_model(road):
    return build_generator(road, DEV)


@pytest.fixture(scope='module')
def region():
    return GoldenRegionWorld()


def test_library_loads():
    from ts_trajgen_amd import _lib
    assert _lib.lib().tsg_version() == 2


@pytest.mark.parametrize('m,n,k', [(1, 1, 1), (130, 70, 33), (257, 1280, 30), (512, 256, 256), (1000, 40, 512)])
def test_gemm_f32(m, n, k):
    from ts_trajgen_amd.engine import gemm
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    b = torch.randn(k, n, generator=g)
    bias = torch.randn(n, generator=g)
    ref = (a.double() @ b.double() + bias.double()).float()
    out = gemm(a.to(DEV), b.to(DEV), bias.to(DEV)).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=RTOL, atol=1e-4)


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_g1_gat_layer(case):
    from ts_trajgen_amd.generator import GATLayerImp3
    z = load_npz('g1_gat.npz')
    nh, fo = z['%s_scoring_fn_source' % case].shape[1:]
    fin = z['%s_x' % case].shape[1]
    layer = GATLayerImp3(fin, fo, nh, DEV, concat=bool(z['%s_concat' % case]))
    layer.load_state_dict({k[2:]: torch.as_tensor(v) for k, v in z.items()
                           if k.startswith(case + '_') and k[2:] in layer.state_dict()})
    layer = layer.to(DEV).train(False)
    out, _ = layer((torch.as_tensor(z['%s_x' % case]).to(DEV), torch.as_tensor(z['edge_index']).to(DEV)))
    np.testing.assert_allclose(out.cpu().numpy(), z['%s_out' % case], rtol=RTOL, atol=1e-5)


@pytest.mark.parametrize('fin,nh,fo', [(30, 5, 256), (48, 3, 64), (64, 2, 128)])
def test_gat_layer_input_space_form_against_oracle(fin, nh, fo):
    """The road encoder's form of the layer (heads averaged, FIN <= 64, N >= (NH+1) FIN: attention-weighted sums of the
    input rows, then one product; tsg_gat.hip) against the oracle's restatement of gat.py:138-200 on a graph with
    what the goldens' graph has (duplicate edges, self loops, nodes without incoming edges) plus hubs whose
    in-degree exceeds one pass of the kernel's lane group (32 or 64 edges)."""
    from oracle.model import gat_forward
    from ts_trajgen_amd.generator import GATLayerImp3
    g = torch.Generator().manual_seed(fin * 100 + nh)
    n = 700
    src = torch.randint(0, n, (2400,), generator=g)
    trg = torch.randint(40, n, (2400,), generator=g)              # nodes 0..39: no incoming edge
    hub_src = torch.randint(0, n, (230,), generator=g)
    hub_trg = torch.cat([torch.full((150,), 50), torch.full((80,), 51)]).long()      # in-degree 150+ and 80+
    loops = torch.arange(60, 90)
    ei = torch.stack([torch.cat([src, hub_src, loops, src[:50]]), torch.cat([trg, hub_trg, loops, trg[:50]])])
    ei = ei[:, torch.randperm(ei.shape[1], generator=g)].contiguous()
    x = torch.randn((n, fin), generator=g)
    layer = GATLayerImp3(fin, fo, nh, DEV, concat=False)
    with torch.no_grad():
        layer.bias.copy_(torch.randn(layer.bias.shape, generator=g) * 0.1)
    want = gat_forward(x, ei, layer.linear_proj.weight.detach(), layer.skip_proj.weight.detach(),
                       layer.scoring_fn_source.detach(), layer.scoring_fn_target.detach(), layer.bias.detach(),
                       nh, fo, concat=False)
    layer = layer.to(DEV).train(False)
    out, _ = layer((x.to(DEV), ei.to(DEV)))
    want = want.detach().numpy()
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=RTOL, atol=1e-5 * max(1.0, float(np.abs(want).max())))


GAT_PARAMS = ('linear_proj.weight', 'skip_proj.weight', 'scoring_fn_source', 'scoring_fn_target', 'bias')


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_g12_gat_layer_training(case):
    """GATLayerImp3 in training mode on the HIP kernels (tsg_gat_train_forward / _backward through autograd) against
    the reference's output and autograd gradients under the same recorded dropout masks (golden G12)."""
    from ts_trajgen_amd.generator import GATLayerImp3
    z = load_npz('g12_gat_train.npz')
    nh, fo = z['%s_scoring_fn_source' % case].shape[1:]
    fin = z['%s_x' % case].shape[1]
    layer = GATLayerImp3(fin, fo, nh, DEV, concat=bool(z['%s_concat' % case]))
    layer.load_state_dict({k: torch.as_tensor(z['%s_%s' % (case, k)]) for k in GAT_PARAMS})
    layer = layer.to(DEV).train(True)
    layer.dropout_masks = tuple(torch.as_tensor(z['%s_mask%d' % (case, i)]) for i in range(3))
    x = torch.as_tensor(z['%s_x' % case]).to(DEV).requires_grad_(True)
    out, _ = layer((x, torch.as_tensor(z['edge_index']).to(DEV)))
    np.testing.assert_allclose(out.detach().cpu().numpy(), z['%s_out' % case], rtol=RTOL, atol=1e-5)
    (out * torch.as_tensor(z['%s_up' % case]).to(DEV)).sum().backward()

    def close(got, want, what):
        scale = max(1e-6, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=2e-5 * scale, err_msg=what)
    close(x.grad.cpu().numpy(), z['%s_grad_x' % case], 'x')
    for k in GAT_PARAMS:
        p = dict(layer.named_parameters())[k]
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        close(g, z['%s_grad_%s' % (case, k)], k)
    # inference mode still runs the inference kernels and ignores the masks
    layer.train(False)
    z1 = load_npz('g1_gat.npz')
    out1, _ = layer((torch.as_tensor(z1['%s_x' % case]).to(DEV), torch.as_tensor(z1['edge_index']).to(DEV)))
    np.testing.assert_allclose(out1.cpu().numpy(), z1['%s_out' % case], rtol=RTOL, atol=1e-5)


def test_g13_function_h_training_loss(road):
    """DistanceGatFC.calculate_loss in training mode and the gradient of every parameter of function H against the
    reference's (golden G13); then one optimiser step and update_node_emb (train_gan.py:284) run end to end."""
    z = load_npz('g13_h_train.npz')
    model = build_generator(road, DEV)
    h = model.function_h
    h.train(True)
    h.gat_encoder.dropout_masks = tuple(torch.as_tensor(z['mask%d' % i]) for i in range(3))
    cand, des, dis = (torch.as_tensor(z[k]).to(DEV) for k in ('cand', 'des', 'dis'))
    h.zero_grad()
    loss = h.calculate_loss(cand, des, dis, torch.as_tensor(z['target']).to(DEV))
    assert abs(float(loss.detach()) - float(z["loss"])) <= RTOL * max(1.0, abs(float(z["loss"])))
    loss.backward()
    for k, p in h.named_parameters():
        want = z['grad_' + k]
        g = p.grad.cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        scale = max(1e-6, float(np.abs(want).max()))
        np.testing.assert_allclose(g, want, rtol=RTOL, atol=2e-5 * scale + 1e-7, err_msg=k)
    before = h.gat_encoder.linear_proj.weight.detach().clone()
    torch.optim.SGD(h.parameters(), lr=1e-3).step()
    assert not torch.equal(before, h.gat_encoder.linear_proj.weight.detach())
    h.train(False)
    h.update_node_emb()
    assert bool(torch.isfinite(h.node_emb_feature).all())


def test_g2_node_embedding(road, road_model, region):
    road_model.function_h.node_emb_feature = None
    road_model.function_h._setup_node_emb()
    np.testing.assert_allclose(road_model.function_h.node_emb_feature.cpu().numpy(), road.npz['node_emb'],
                               rtol=RTOL, atol=1e-5)
    rm = build_generator(region, DEV)
    rm.function_h._setup_node_emb()
    np.testing.assert_allclose(rm.function_h.node_emb_feature.cpu().numpy(), region.npz['node_emb'],
                               rtol=RTOL, atol=1e-4)


def test_g234_predict(road_model):
    for c in load_json('g234_predict.json')['cases']:
        tl = torch.LongTensor(c['loc']).unsqueeze(0).to(DEV)
        tt = torch.LongTensor(c['tim']).unsqueeze(0).to(DEV)
        cs = torch.LongTensor(c['cand']).unsqueeze(0).to(DEV)
        cd = torch.FloatTensor(c['dis']).unsqueeze(0).to(DEV)
        dt = torch.LongTensor([c['des']]).to(DEV)
        p, g, h = road_model.predict_parts(tl, tt, dt, cs, cd)
        np.testing.assert_allclose(g.cpu().numpy()[0], c['g'], rtol=RTOL)
        np.testing.assert_allclose(h.cpu().numpy()[0], c['h'], rtol=RTOL)
        np.testing.assert_allclose(p.cpu().numpy()[0], c['p'], rtol=RTOL)
        np.testing.assert_allclose(torch.log(p).cpu().numpy()[0], c['logp'], rtol=RTOL)
        # the halves under their reference names (function_g_fc.py:206, distance_gat_fc.py:136)
        g2 = road_model.function_g.predict_g(tl, tt, dt, cs, cd)
        h2 = road_model.function_h.predict_next(cs, dt, cd)
        np.testing.assert_allclose(g2.cpu().numpy()[0], c['g'], rtol=RTOL)
        np.testing.assert_allclose(h2.cpu().numpy()[0], c['h'], rtol=RTOL)
        # scores before the softmax (predict_cache :118, inference-mode forward :44) and predict (:97)
        fh = road_model.function_h
        sc = fh.predict_cache(cs.flatten(), dt.repeat(cs.shape[1]), cd.flatten())
        np.testing.assert_allclose(torch.softmax(sc, 0).cpu().numpy(), c['h'], rtol=RTOL)
        assert torch.equal(sc, fh.forward(cs.flatten(), dt.repeat(cs.shape[1]), cd.flatten()))
        np.testing.assert_allclose(fh.predict(cs, dt, cd).cpu().numpy()[0], c['h'], rtol=RTOL)


def test_predict_batched_with_mask(road_model):
    """(B, T) batches with trace_mask against the REFERENCE's own masked-batch outputs (golden G10): packed LSTM,
    last real state as the query, and the padded zero rows left inside the attention softmax
    (function_g_fc.py:31-35,162-180: the masked_fill result is assigned to `mask`).  Also FunctionGFC.forward's
    candidate logits (function_g_fc.py:184-187), masked and unmasked, and GeneratorV4.forward's mix (:54)."""
    g10 = load_json('g10_masked_batch.json')
    for b in g10['batches']:
        loc, tim = torch.LongTensor(b['loc']).to(DEV), torch.LongTensor(b['tim']).to(DEV)
        mask = torch.LongTensor(b['mask']).to(DEV)
        cs, cd = torch.LongTensor(b['cand']).to(DEV), torch.FloatTensor(b['dis']).to(DEV)
        dt = torch.LongTensor(b['des']).to(DEV)
        p = road_model.predict(loc, tim, dt, cs, cd, trace_mask=mask)
        np.testing.assert_allclose(p.cpu().numpy(), np.array(b['p']), rtol=RTOL)
        g = road_model.function_g.predict_g(loc, tim, dt, cs, cd, trace_mask=mask)
        np.testing.assert_allclose(g.cpu().numpy(), np.array(b['g']), rtol=RTOL)
        sc = road_model.function_g.forward(loc, tim, dt, cs, cd, trace_mask=mask, train=False)
        np.testing.assert_allclose(sc.cpu().numpy(), np.array(b['g_score']), rtol=RTOL, atol=1e-5)
        mix = road_model.forward(loc, tim, dt, cs, cd, trace_mask=mask)
        np.testing.assert_allclose(mix.cpu().numpy(), np.array(b['mix']), rtol=RTOL)
        # a mask wider than the longest row: the reference fails on the shapes (function_g_fc.py:33); so do we
        wide = torch.cat([mask, torch.zeros_like(mask[:, :1])], dim=1)
        with pytest.raises(RuntimeError):
            road_model.predict(torch.cat([loc, loc[:, :1]], dim=1), torch.cat([tim, tim[:, :1]], dim=1), dt, cs, cd,
                               trace_mask=wide)
    for c in g10['logits']:
        sc = road_model.function_g.forward(torch.LongTensor([c['loc']]).to(DEV), torch.LongTensor([c['tim']]).to(DEV),
                                           torch.LongTensor([c['des']]).to(DEV), torch.LongTensor([c['cand']]).to(DEV),
                                           torch.FloatTensor([c['dis']]).to(DEV), None, False)
        np.testing.assert_allclose(sc.cpu().numpy()[0], c['g_score'], rtol=RTOL, atol=1e-5)


def test_engine_reports_a_search_beyond_its_expansion_records(road, road_model):
    """A direct user of the C ABI who asks for more expansions than exp_cap holds gets an error (the OD pair ends
    as OVERFLOW and the engine's error flag is raised), not a silently shorter search."""
    from ts_trajgen_amd.search import Searcher
    c = load_json('g5_astar.json')['cases'][0]
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    s.generate(road_model, [c['start']], [c['start_time']], [c['des']], [c['default_len']], max_step=50, slots=4)
    eng = list(s._engines.values())[0]
    assert eng.exp_cap == 51
    eng.max_step = 10 ** 6                    # get past the host-side check of AStarEngine.run
    with pytest.raises(RuntimeError, match='workspace cap'):
        eng.run([0, 1], [c['start']], [c['start_time']], [c['des']], [c['default_len']], [500])


def classify(res, i, c, near_tie=1e-5):
    """'exact', or 'near-tie' when the first diverging pop differs in priority by < near_tie (else raises)."""
    return classify_traj(res, i, c['loc'], c['tim'], c['pop_rid'], c['pop_prio'], near_tie)


def test_g5_astar(road, road_model):
    from ts_trajgen_amd.search import Searcher
    cases = load_json('g5_astar.json')['cases']
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    res = s.generate(road_model, [c['start'] for c in cases], [c['start_time'] for c in cases],
                     [c['des'] for c in cases], [c['default_len'] for c in cases],
                     max_step=[c['max_step'] for c in cases], slots=16, poplog_cap=1024)
    kinds = [classify(res, i, c) for i, c in enumerate(cases)]
    report_near_ties('G5 road-level A* (64 searches)', kinds, recorded=0)
    # pop traces of the exact ones agree in order and (1e-4) in priority
    for i, c in enumerate(cases):
        if kinds[i] == 'exact':
            k = int(res['poplog_n'][i])
            assert res['poplog_rid'][i, :k].tolist() == c['pop_rid'][:1024]
            np.testing.assert_allclose(res['poplog_prio'][i, :k].numpy(), c['pop_prio'][:1024], rtol=RTOL, atol=1e-5)
    st = res['status'].tolist()
    assert {0, 1, 2} <= set(st)


def test_reconstruct_paths(road, road_model):
    """tsg_reconstruct_paths against a host walk of the same expansion records: one search per slot (slots >= OD
    pairs, so no slot is refilled), the path of each search's LAST expanded node, with and without one more
    point appended, and a request out of range."""
    from ts_trajgen_amd.search import Searcher
    cases = load_json('g5_astar.json')['cases'][:24]
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    res = s.generate(road_model, [c['start'] for c in cases], [c['start_time'] for c in cases],
                     [c['des'] for c in cases], [c['default_len'] for c in cases],
                     max_step=[c['max_step'] for c in cases], slots=32)
    eng = list(s._engines.values())[0]
    ss = eng.slot_state.cpu().numpy()
    em = eng.exp_meta.cpu().numpy()                      # [slots][exp_cap][parent, rid, tcode, depth]
    used = [b for b in range(eng.slots) if ss[b, 23] == 1 and ss[b, 4] > 0]     # SS_GEN == 1, SS_EXP_N > 0
    assert len(used) >= len(cases) - 4
    node = [int(ss[b, 4]) - 1 for b in used]
    lmax = 64
    rid, tim, ln = eng.reconstruct_paths(used, node, lmax=lmax)
    rid2, tim2, ln2 = eng.reconstruct_paths(used + [10 ** 6], node + [0], last_rid=[7] * len(used) + [7],
                                            last_tcode=[11] * len(used) + [11], lmax=lmax)
    rid, tim, ln, rid2, tim2, ln2 = [t.cpu().numpy() for t in (rid, tim, ln, rid2, tim2, ln2)]
    assert ln2[-1] == 0
    for k, (b, nd) in enumerate(zip(used, node)):
        want_r, want_t, cur = [], [], nd
        while cur >= 0:
            want_r.append(int(em[b, cur, 1]))
            want_t.append(int(em[b, cur, 2]))
            cur = int(em[b, cur, 0])
        want_r, want_t = want_r[::-1], want_t[::-1]
        assert len(want_r) == int(em[b, nd, 3]) == ln[k]
        od = int(ss[b, 22])                                 # SS_OD: the OD pair this slot ran
        assert want_r[0] == cases[od]['start']
        n = min(len(want_r), lmax)
        assert rid[k, :n].tolist() == want_r[:n] and tim[k, :n].tolist() == want_t[:n]
        assert ln2[k] == ln[k] + 1
        n2 = min(len(want_r) + 1, lmax)
        assert rid2[k, :n2].tolist() == (want_r + [7])[:n2] and tim2[k, :n2].tolist() == (want_t + [11])[:n2]
        for a, c in zip(want_r, want_r[1:]):
            assert c in road.adjacent_list[str(a)]


def test_superstep_in_two_calls(road, road_model):
    """tsg_frontier_expand_select + tsg_score_frontier in a loop == tsg_engine_run (and its HIP-graph replay)."""
    from ts_trajgen_amd.search import Searcher
    cases = load_json('g5_astar.json')['cases']
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    args = (road_model, [c['start'] for c in cases], [c['start_time'] for c in cases], [c['des'] for c in cases],
            [c['default_len'] for c in cases])
    kw = dict(max_step=[c['max_step'] for c in cases], slots=16)
    ref = s.generate(*args, **kw)
    eng = list(s._engines.values())[0]
    eng.use_graph, eng.split_calls = False, True
    got = s.generate(*args, **kw)
    assert list(s._engines.values())[0] is eng
    for k in ('rid', 'time', 'len', 'status', 'steps'):
        assert torch.equal(ref[k], got[k]), k
    eng.split_calls = False
    plain = s.generate(*args, **kw)                     # tsg_engine_run without the graph
    for k in ('rid', 'time', 'len', 'status', 'steps'):
        assert torch.equal(ref[k], plain[k]), k


def test_g5_single_od_api(road, road_model):
    """Reference call signature, one OD per call (search.py:240)."""
    from ts_trajgen_amd.search import Searcher
    cases = load_json('g5_astar.json')['cases'][:6]
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    for c in cases:
        loc, tim = s.astar_search(road_model, [c['start']], [c['start_time']], c['des'], c['default_len'],
                                  max_step=c['max_step'])
        assert loc == c['loc'] and tim == c['tim']


def test_g6_astar_in_region(road, road_model, region):
    from ts_trajgen_amd.search import Searcher
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    for c in load_json('g6_astar_in_region.json')['cases']:
        rset = set(region.meta['region2rid'][str(c['region'])])
        loc, tim = s.astar_search_in_region(road_model, c['prefix_loc'], c['prefix_tim'], c['des'],
                                            c['default_len'], rset, max_step=c['max_step'])
        assert loc == c['loc'] and tim == c['tim']


def test_g7_double_layer(road, road_model, region):
    """Two-stage search (search.py:859-934): region A*, border-road legs, last leg; per-OD RandomState."""
    from ts_trajgen_amd.search import DoubleLayerSearcher
    region_model = build_generator(region, DEV)
    t = region.tables(road)
    s = DoubleLayerSearcher(DEV, t['adjacent_list'], t['road_center_gps'], t['road_length'],
                            t['region_adjacent_list'], t['region_dist'], t['region_transfer_freq'],
                            t['rid2region'], t['region2rid'], road.rtd, region.npz['region_time_distribution'])
    cases = region.meta['cases']
    # region stage alone (search.py:936-1117)
    for c in cases[:6]:
        tr, fl = s.astar_search_in_region(region_model, t['rid2region'][str(c['start'])], c['start_time'],
                                          t['rid2region'][str(c['des'])], c['max_step'])
        assert tr == c['region_trace'] and fl == c['region_flag']
    loc, tim, fl = s.generate(region_model, road_model, [c['start'] for c in cases],
                              [c['start_time'] for c in cases], [c['des'] for c in cases],
                              [c['default_len'] for c in cases], max_step=[c['max_step'] for c in cases],
                              seeds=[c['seed'] for c in cases])
    # every leg is a search of its own with its own near-ties, and a two-stage result cannot be classified through
    # one pop log: the bar here is identity with the reference, trajectory by trajectory
    for i, c in enumerate(cases):
        assert loc[i] == c['loc'] and tim[i] == c['tim'] and fl[i] == c['is_astar'], i
    assert sorted([list(map(int, x)) for x in s.fail_find_log]) == region.meta['fail_find_log']
    assert sorted([list(map(int, x)) for x in s.region_fail_find_log]) == region.meta['region_fail_find_log']
    # output rows far too short for the legs: the overflowing legs are run again with a wider row, nothing is lost
    s2 = DoubleLayerSearcher(DEV, t['adjacent_list'], t['road_center_gps'], t['road_length'],
                             t['region_adjacent_list'], t['region_dist'], t['region_transfer_freq'],
                             t['rid2region'], t['region2rid'], road.rtd, region.npz['region_time_distribution'])
    loc2, tim2, fl2 = s2.generate(region_model, road_model, [c['start'] for c in cases],
                                  [c['start_time'] for c in cases], [c['des'] for c in cases],
                                  [c['default_len'] for c in cases], max_step=[c['max_step'] for c in cases],
                                  seeds=[c['seed'] for c in cases], leg_lmax=4)
    assert max(len(x) for x in loc) > 4
    assert loc2 == loc and tim2 == tim and fl2 == fl
    assert s2.generate(region_model, road_model, [], [], [], []) == ([], [], [])       # an empty shard
    for n_eng in (1, 2):                                                               # the road-level entry, empty too
        e0 = s2.road_search.generate(road_model, [], [], [], [], max_step=10, slots=8, streams=n_eng)
        assert tuple(e0['rid'].shape) == (0, 256) and e0['len'].numel() == 0 and e0['status'].numel() == 0
    # reference one-OD signature with the global numpy RNG (search.py:859)
    c = cases[3]
    np.random.seed(c['seed'])
    l1, t1, f1 = s.astar_search(region_model, road_model, c['start'], c['start_time'], c['des'], c['default_len'],
                                max_step=c['max_step'])
    assert l1 == c['loc'] and t1 == c['tim'] and f1 == c['is_astar']


def test_device_geodesic_against_oracle():
    """Row a6 on its own: the device's fp64 Vincenty inverse (tsg_geodesic_pairs, the function the engine calls for
    every candidate, search.py:355-356) against oracle/geodesic.py on 6 000 pairs: a Beijing-sized cloud, pairs a
    metre apart, identical points, points on one meridian / one parallel, and a few long hauls.  fp64 results
    agree to 1e-9 relative + 0.2 um (short pairs are ill-conditioned in any fp64 implementation), and the f32
    model input `kilometres * 10` is the same float, or the neighbouring one, for every pair."""
    import ctypes as C
    from oracle import geodesic as og
    from ts_trajgen_amd import _lib
    rng = np.random.RandomState(5)
    n = 6000
    lon = 116.25 + rng.rand(n) * 0.6
    lat = 39.79 + rng.rand(n) * 0.5
    a = rng.randint(0, n, size=n)
    b = rng.randint(0, n, size=n)
    b[:200] = a[:200]                                            # identical points
    near = np.arange(200, 1200)
    lon[b[near]] = lon[a[near]] + rng.randn(len(near)) * 1e-5     # about a metre apart
    lat[b[near]] = lat[a[near]] + rng.randn(len(near)) * 1e-5
    mer = np.arange(1200, 1400)
    lon[b[mer]] = lon[a[mer]]                                    # same meridian
    par = np.arange(1400, 1600)
    lat[b[par]] = lat[a[par]]                                    # same parallel
    far = np.arange(1600, 1700)
    lon[b[far]] = lon[a[far]] + 30.0 * rng.rand(len(far))        # long hauls (not antipodal)
    lat[b[far]] = lat[a[far]] - 40.0 * rng.rand(len(far))
    f = 1.0 / 298.257223563
    u = np.arctan((1.0 - f) * np.tan(lat * (np.pi / 180.0)))
    geo = torch.from_numpy(np.ascontiguousarray(np.stack([lon, lat, np.sin(u), np.cos(u)], axis=1))).to(DEV)
    ta = torch.from_numpy(a.astype(np.int32)).to(DEV)
    tb = torch.from_numpy(b.astype(np.int32)).to(DEV)
    out = torch.full((n,), -7.0, dtype=torch.float64, device=DEV)
    _lib.check(_lib.lib().tsg_geodesic_pairs(_lib.ptr(geo), n, n, _lib.ptr(ta), _lib.ptr(tb), _lib.ptr(out),
                                             _lib.current_stream()), 'tsg_geodesic_pairs')
    got = out.cpu().numpy()
    want = np.array([og.geodesic_m(lat[i], lon[i], lat[j], lon[j]) for i, j in zip(a, b)])
    assert (got[a == b] == 0.0).all() and (want[a == b] == 0.0).all()
    assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max() and (np.abs(got - want) <= 1e-9 * want + 2e-7).all()
    g32, w32 = (got / 1000.0 * 10.0).astype(np.float32), (want / 1000.0 * 10.0).astype(np.float32)
    one_ulp = np.spacing(np.maximum(np.abs(w32), np.float32(1e-30)))
    assert (np.abs(g32 - w32) <= one_ulp).all()
    assert (g32 == w32).mean() > 0.99
    # an index outside the table is reported, not read
    bad = torch.tensor([n, -1], dtype=torch.int32, device=DEV)
    out2 = torch.zeros(2, dtype=torch.float64, device=DEV)
    _lib.check(_lib.lib().tsg_geodesic_pairs(_lib.ptr(geo), n, 2, _lib.ptr(bad), _lib.ptr(ta), _lib.ptr(out2),
                                             _lib.current_stream()), 'tsg_geodesic_pairs')
    assert out2.cpu().tolist() == [-1.0, -1.0]


def _midpoint_uniforms(cases, n_draws):
    """Uniforms that make the inverse-CDF rule reproduce the golden draws: the midpoint of the drawn
    candidate's CDF interval (golden probabilities, fp32 running sum)."""
    u = np.zeros((len(cases), n_draws), dtype=np.float32)
    for b, c in enumerate(cases):
        for k, (p, idx) in enumerate(c['draws']):
            cum = np.cumsum(np.asarray(p, dtype=np.float32), dtype=np.float32)
            lo = 0.0 if idx == 0 else float(cum[idx - 1])
            u[b, k] = 0.5 * (lo + float(cum[idx]))
    return u


def test_g8_random_sample(road, road_model, region):
    """Rollouts (search.py:614-716, :1163-1256) against the reference goldens: same draws -> same
    trajectories, and the probability vector of every draw within 1e-4 relative."""
    from ts_trajgen_amd.search import DoubleLayerSearcher
    g8 = load_json('g8_random_sample.json')
    t = region.tables(road)
    s = DoubleLayerSearcher(DEV, t['adjacent_list'], t['road_center_gps'], t['road_length'],
                            t['region_adjacent_list'], t['region_dist'], t['region_transfer_freq'],
                            t['rid2region'], t['region2rid'], road.rtd, region.npz['region_time_distribution'])
    cases = g8['road']
    nd = max(20, max(c['default_len'] for c in cases))
    res = s.road_search.random_sample_batch(road_model, [(c['prefix_loc'], c['prefix_tim']) for c in cases],
                                            [c['des'] for c in cases], [c['default_len'] for c in cases],
                                            uniforms=_midpoint_uniforms(cases, nd), return_probs=True)
    probs = res['probs'].cpu().numpy()
    for b, c in enumerate(cases):
        assert res['loc'][b] == c['loc'] and res['tim'][b] == c['tim'], b
        assert int(res['n_drawn'][b]) == len(c['draws'])
        for k, (p, _) in enumerate(c['draws']):
            np.testing.assert_allclose(probs[b, k, :len(p)], p, rtol=1e-4, atol=1e-7)
    region_model = build_generator(region, DEV)
    cases = g8['region']
    res = s.region_random_sample_batch(region_model, [(c['prefix_loc'], c['prefix_tim']) for c in cases],
                                       [c['des'] for c in cases], [c['default_len'] for c in cases],
                                       uniforms=_midpoint_uniforms(cases, 20))
    for b, c in enumerate(cases):
        assert res['loc'][b] == c['loc'] and res['tim'][b] == c['tim'], b
    # reference one-row signatures (torch.rand draws: only the shape of the result is checked)
    c = g8['road'][1]
    loc, tim = s.road_random_sample(road_model, c['prefix_loc'], c['prefix_tim'], c['des'], c['default_len'])
    assert loc[:len(c['prefix_loc'])] == c['prefix_loc'][:len(loc)] and len(loc) == len(tim)
    c = g8['region'][0]
    loc, tim = s.region_random_sample(region_model, c['prefix_loc'], c['prefix_tim'], c['des'], c['default_len'])
    assert len(loc) == len(tim) and len(loc) >= 1


def test_random_sample_vs_oracle(road, road_model):
    """512 rollouts with random uniforms against the oracle using the same inverse-CDF rule; a row may differ
    only where a uniform falls within 1e-5 of a CDF boundary."""
    from oracle import search as osr
    from ts_trajgen_amd.search import Searcher
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    rng = np.random.RandomState(3)
    n = 512
    keys = [int(k) for k in road.adjacent_list.keys()]
    start = [keys[i] for i in rng.randint(len(keys), size=n)]
    des = rng.randint(road.n, size=n)
    code = rng.randint(0, 2880, size=n)
    dl = rng.randint(1, 35, size=n)
    u = rng.rand(n, 40).astype(np.float32)
    res = s.random_sample_batch(road_model, [([start[i]], [int(code[i])]) for i in range(n)], des, dl, uniforms=u)
    m = road.oracle_model()
    rtd = road.rtd.astype(int)
    bad = 0
    for i in range(0, n, 4):          # every 4th row through the (slow) oracle
        log = []
        loc, tim = osr.road_random_sample(m, road.adjacent_list, road.rid_gps, road.road_length, rtd, [start[i]],
                                          [int(code[i])], int(des[i]), int(dl[i]),
                                          choose=lambda p, k, i=i: osr.inverse_cdf_choice(p.numpy(), u[i, k]),
                                          log=log)
        if [int(x) for x in loc] != res['loc'][i] or [int(x) for x in tim] != res['tim'][i]:
            near = any(np.min(np.abs(np.cumsum(np.float32(p), dtype=np.float32) - u[i, k])) < 1e-5
                       for k, (p, _) in enumerate(log))
            assert near, i
            bad += 1
    assert bad <= 2


def test_g9_beam_search(road, road_model):
    """Searcher.beam_search (search.py:137-238) against the reference goldens: batched, then one call with the
    reference signature and the global numpy RNG, then the reference's IndexError."""
    from ts_trajgen_amd.search import Searcher
    cases = load_json('g9_beam_search.json')['cases']
    s = Searcher(DEV, road.adjacent_list, road.rid_gps, road.road_length, road.rtd)
    same = 0
    groups = {}
    for i, c in enumerate(cases):
        groups.setdefault((c['width'], c['max_step'], c['prob_threshold']), []).append(i)
    for (w, ms, thr), idx in groups.items():
        locs, tims, err = s.beam_search_batch(road_model, [(cases[i]['prefix_loc'], cases[i]['prefix_tim']) for i in idx],
                                              [cases[i]['des'] for i in idx], width=w, max_step=ms, prob_threshold=thr,
                                              seeds=[cases[i]['seed'] for i in idx])
        for k, i in enumerate(idx):
            c = cases[i]
            if 'error' in c:
                assert err[k] == c['error']
                same += 1
            else:
                assert err[k] is None
                assert locs[k] == c['loc'] and tims[k] == c['tim'], i      # identical to the reference, every case
                same += 1
    assert same == len(cases)
    c = next(c for c in cases if 'loc' in c and c['index_bound'] > 1)
    np.random.seed(c['seed'])
    loc, tim = s.beam_search(road_model, c['prefix_loc'], c['prefix_tim'], c['des'], width=c['width'],
                             max_step=c['max_step'], prob_threshold=c['prob_threshold'])
    assert loc == c['loc'] and tim == c['tim']
    c = next(c for c in cases if 'error' in c)
    with pytest.raises(IndexError):
        s.beam_search(road_model, c['prefix_loc'], c['prefix_tim'], c['des'], width=c['width'], max_step=c['max_step'])
