"""BASELINE.json configs[3] and configs[4] on MI355X, through the C ABI.

* configs[3]: two-stage generation at Porto scale -- a 10 x 10 tiling of the map into ~100 regions, region model
  128/32/128 with `no_gps_emb` on features summed from the road model's GAT embeddings
  (prepare_region_feature.py:69-75), region-level A* then road-level A* leg by leg (search.py:859-934).
  A few OD pairs against the CPU oracle (bounded max_step), all of them against the properties a two-stage
  trajectory must have.
* configs[4]: the 500 000-segment stress city (GAT hidden 128, trajectories up to 256 roads): the encoder against
  properties the oracle-checked small cases share (finite, in-degree-0 rows = ELU(skip + bias)), searches against
  the per-trajectory properties, with a state-page pool well below the worst case.
"""
import copy

import numpy as np
import pytest
import torch

from tests.test_gpu_fullsize import BJ_DIMS, _check_properties, _model, _oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _region_model(net, road_model, reg):
    from ts_trajgen_amd import dataset as ds
    from ts_trajgen_amd.generator import GeneratorV4
    road_model.function_h._setup_node_emb()
    feats = ds.region_features_from_road_emb(road_model.function_h.node_emb_feature[:net.num_roads],
                                             reg['rid2region'], reg['num_regions'])
    feats = torch.cat([feats, torch.zeros((1, feats.shape[1]), device=feats.device)], dim=0)   # pad region
    r = reg['num_regions']
    cfg = ds._gen_config((128, 32, 128, 128, 10, 5), DEV, True)          # our_model_generate.py:62-83
    data_feature = {'road_num': r + 1, 'time_size': 2881, 'road_pad': r, 'time_pad': 2880,
                    'adj_mx': reg['region_adj_mx'], 'node_features': feats.cpu(), 'img_width': 52, 'img_height': 43}
    torch.manual_seed(1)
    model = GeneratorV4(config=cfg, data_feature=data_feature).to(DEV)
    model.function_h.node_features = model.function_h.node_features.to(DEV)
    model.train(False)
    return model, cfg, data_feature


def test_config3_two_stage_porto_scale():
    from oracle import search as osearch
    from oracle.model import OracleGenerator
    from ts_trajgen_amd import synth
    from ts_trajgen_amd.search import DoubleLayerSearcher
    net = synth.porto_scale_network()
    road_model, cfg = _model(net)
    reg = synth.region_tables(net, tiles=10)
    assert 80 <= reg['num_regions'] <= 100
    region_model, rcfg, rdf = _region_model(net, road_model, reg)
    s = DoubleLayerSearcher(DEV, net.adjacent_list, net.rid_gps, net.road_length, reg['region_adjacent_list'],
                            reg['region_dist'], reg['region_transfer_freq'], reg['rid2region'], reg['region2rid'],
                            net.road_time_distribution, reg['region_time_distribution'])
    od = synth.make_od_pairs(net, 160, seed=7, workers=1)
    n = len(od['des'])
    max_step = 2000
    locs, tims, flags = s.generate(region_model, road_model, od['start'], od['start_time'], od['des'],
                                   od['default_len'], max_step=max_step, seeds=list(range(n)))
    # properties: every hop is a successor, time codes follow search.py:376-384, no road twice within a leg is
    # not required across legs (a later leg may revisit), the trajectory starts at the origin
    from ts_trajgen_amd.synth import csr_from_adjacent_list
    nr = net.num_roads
    indptr, indices = csr_from_adjacent_list(net.adjacent_list, nr)
    rtd = net.road_time_distribution.astype(int)
    reached = 0
    for i in range(n):
        r, t = locs[i], tims[i]
        assert r[0] == od['start'][i] and t[0] == od['start_time'][i] and len(r) == len(t)
        woff = 1440 if t[0] >= 1440 else 0
        for k in range(1, len(r)):
            assert r[k] in indices[indptr[r[k - 1]]:indptr[r[k - 1] + 1]], (i, k)
            assert 0 <= t[k] - woff < 1440
        reached += int(r[-1] == od['des'][i])
        assert flags[i] in (0, 1)
    assert reached > n // 4, reached
    # a few OD pairs against the oracle's two-stage search (same per-OD RandomState)
    o_road = _oracle(net, road_model, cfg)
    adj = rdf['adj_mx']
    g = {k: v.detach().cpu().numpy() for k, v in region_model.function_g.state_dict().items()}
    h = {k: v.detach().cpu().numpy() for k, v in region_model.function_h.state_dict().items()}
    o_region = OracleGenerator(g, h, float(region_model.w1.item()), rdf['node_features'].numpy(),
                               np.stack([adj.row, adj.col]), 5, 128, 2, False, True)
    tables = {'adjacent_list': net.adjacent_list, 'road_center_gps': net.rid_gps, 'road_length': net.road_length,
              'region_adjacent_list': reg['region_adjacent_list'], 'region_dist': reg['region_dist'],
              'region_transfer_freq': reg['region_transfer_freq'], 'rid2region': reg['rid2region'],
              'region2rid': reg['region2rid'], 'road_time_distribution': rtd,
              'region_time_distribution': reg['region_time_distribution'].astype(int)}
    sub = list(range(6))
    small = 120
    l2, t2, f2 = s.generate(region_model, road_model, od['start'][sub], od['start_time'][sub], od['des'][sub],
                            od['default_len'][sub], max_step=small, seeds=sub)
    for i in sub:
        loc, tim, fl = osearch.double_layer_astar_search(o_region, o_road, tables, int(od['start'][i]),
                                                         int(od['start_time'][i]), int(od['des'][i]),
                                                         int(od['default_len'][i]), max_step=small,
                                                         rng=np.random.RandomState(i))
        assert [int(x) for x in loc] == l2[i] and [int(x) for x in tim] == t2[i] and fl == f2[i], i


def test_config4_stress_city():
    from ts_trajgen_amd import synth
    from ts_trajgen_amd.generator import GeneratorV4
    from ts_trajgen_amd.search import Searcher
    net = synth.stress_network()
    assert net.num_roads == 500000
    cfg = copy.deepcopy(BJ_DIMS)
    cfg['function_h']['embed_dim'] = 128
    torch.manual_seed(0)
    model = GeneratorV4(config=cfg, data_feature=net.data_feature(torch)).to(DEV)
    model.function_h.node_features = model.function_h.node_features.to(DEV)
    model.train(False)
    model.function_h._setup_node_emb()
    emb = model.function_h.node_emb_feature
    assert emb.shape[0] >= net.num_roads and emb.shape[1] == 128 and bool(torch.isfinite(emb).all())
    # a node without incoming edges aggregates nothing: its row is ELU(x . mean_h(Ws_h)^T + bias) (gat.py:93-114)
    indeg = np.bincount(net.adj_mx.col, minlength=emb.shape[0])[:net.num_roads]
    lone = np.flatnonzero(indeg == 0)[:64]
    assert len(lone) > 0
    x = model.function_h.input_features()[torch.as_tensor(lone, device=DEV)]
    enc = model.function_h.gat_encoder
    ws = enc.skip_proj.weight.detach().view(enc.num_of_heads, enc.num_out_features, -1).mean(dim=0)
    want = torch.nn.functional.elu(x @ ws.t() + enc.bias.detach())
    got = emb[torch.as_tensor(lone, device=DEV)]
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    od = synth.make_od_pairs(net, 768, seed=2, default_len='lattice')
    assert int(od['default_len'].max()) <= 256
    s = Searcher(DEV, net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution)
    max_step, lmax = 5000, 256
    pps = (1 + max_step + 63) // 64
    res = s.generate(model, od['start'], od['start_time'], od['des'], od['default_len'], max_step=max_step,
                     slots=768, lmax=lmax, pool_pages=768 * pps // 2, ramp_steps=500)
    _check_properties(net, od, res, max_step, lmax)
