"""Generation driver end to end (row f2): dataset files + checkpoints on disk -> output CSV, checked
against the oracle's two-stage search (search.py:859-934 restated in oracle/search.py) on the same files."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _oracle_generator(model, cfg, data_feature):
    from oracle.model import OracleGenerator
    adj = data_feature['adj_mx']
    fh = cfg['function_h']
    g = {k: v.detach().cpu().numpy() for k, v in model.function_g.state_dict().items()}
    h = {k: v.detach().cpu().numpy() for k, v in model.function_h.state_dict().items()}
    return OracleGenerator(g, h, float(model.w1.item()), data_feature['node_features'].cpu().numpy(),
                           np.stack([adj.row, adj.col]), fh['num_of_heads'], fh['embed_dim'],
                           cfg['function_g']['lstm_layer_num'], fh['concat'], fh.get('no_gps_emb', False))


def test_driver_matches_oracle(tmp_path, monkeypatch):
    from oracle import search as osearch
    from ts_trajgen_amd import dataset as ds
    from ts_trajgen_amd import generate as drv
    from ts_trajgen_amd import synth
    name, seed, max_step = 'Synth', 5, 150
    root, save = str(tmp_path / 'data'), str(tmp_path / 'save')
    net = synth.street_grid_network(grid=8, seed=1)
    reg = synth.region_tables(net, tiles=3)
    dims = {'road': [32, 8, 32, 32, 4, 5], 'region': [16, 8, 16, 16, 4, 5]}
    feats = np.random.RandomState(1).rand(reg['num_regions'], 20).astype(np.float32)
    pt = ds.save_dataset(root, name, net, reg, feats, dims=dims)
    od = synth.make_od_pairs(net, 24, seed=4)
    ds.write_test_csv(pt['test_csv'], ['t%d' % i for i in range(24)],
                      [[int(o)] * (int(n) - 1) + [int(d)] if n > 1 else [int(o)]
                       for o, d, n in zip(od['start'], od['des'], od['default_len'])],
                      [[int(t)] * max(int(n), 1) for t, n in zip(od['start_time'], od['default_len'])])
    # checkpoints in the reference layout (our_model_generate.py:35-39): one state_dict per sub-module
    data = ds.load_dataset(root, name, device='cpu')
    cfg, rcfg = ds.gen_configs(name, 'cpu', data['meta'])
    torch.manual_seed(seed)
    road = drv.build_generator(cfg, data['data_feature'], 'cpu')
    region = drv.build_generator(rcfg, data['region_data_feature'], 'cpu')
    os.makedirs(os.path.join(save, name))
    torch.save(road.function_g.state_dict(), os.path.join(save, name, 'function_g_fc.pt'))
    torch.save(road.function_h.state_dict(), os.path.join(save, name, 'gat_fc.pt'))
    torch.save(region.function_g.state_dict(), os.path.join(save, name, 'region_function_g_fc.pt'))
    torch.save(region.function_h.state_dict(), os.path.join(save, name, 'region_gat_fc.pt'))

    monkeypatch.chdir(tmp_path)                       # ./log/*.json land in the temp dir
    rc = drv.main(['--dataset_name', name, '--device', DEV, '--data_root', root, '--save_dir', save,
                   '--max_step', str(max_step), '--seed', str(seed)])
    assert rc == 0
    ids, locs, tims = ds.read_generate_csv(pt['generate_csv'])
    assert [str(i) for i in ids] == ['t%d' % i for i in range(24)]

    o_road = _oracle_generator(road, cfg, data['data_feature'])
    o_region = _oracle_generator(region, rcfg, data['region_data_feature'])
    tables = dict(data['searcher'])
    tables['road_time_distribution'] = tables['road_time_distribution'].astype(int)
    tables['region_time_distribution'] = tables['region_time_distribution'].astype(int)
    _, start, code, des, dl = ds.read_test_csv(pt['test_csv'])
    fail, rfail, flags = set(), set(), []
    for i in range(24):
        loc, tim, fl = osearch.double_layer_astar_search(o_region, o_road, tables, int(start[i]), int(code[i]),
                                                         int(des[i]), int(dl[i]), max_step=max_step,
                                                         rng=np.random.RandomState(seed + i), fail_log=fail,
                                                         region_fail_log=rfail)
        assert [int(x) for x in loc] == locs[i] and [int(x) for x in tim] == tims[i], i     # identical, every OD pair
        flags.append(fl)
    with open(tmp_path / 'log' / 'find_error_log.json') as f:
        assert sorted(map(tuple, json.load(f))) == sorted((int(a), int(b)) for a, b in fail)
    with open(tmp_path / 'log' / 'region_find_error_log.json') as f:
        assert sorted(map(tuple, json.load(f))) == sorted((int(a), int(b)) for a, b in rfail)
