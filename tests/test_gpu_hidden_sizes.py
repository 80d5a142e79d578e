"""Hidden sizes other than the ones the goldens and BASELINE.json use, through the C ABI on MI355X.

The engine's kernels are instantiated per padded hidden size hp (a multiple of 64, at most 512: DESIGN.md section 7):
the scoring kernel holds hp / 64 words of a state row per lane (1, 2, 4 or 8, with a 16-byte-per-lane form at
hp = 256 only), the recurrence walks hp / 32 column tiles and hp / 32 (layer 0) or hp / 16 (layer 1) k-tiles.  The
goldens run hp = 64 (hidden 32) and 128, the BJ-dims tests hp = 256.  Here: hidden 192 (the 4-word form without the
16-byte loads), 320 (5 of 8 words used), 500 (padded to 512, the maximum) and 512, each for both forms of the
recurrence, against the CPU oracle on a small street grid: trajectories, time codes and pop order identical or a
proven near-tie (first diverging pop within 1e-5 relative in priority).
"""
import copy

import numpy as np
import pytest
import torch

from tests._fixtures import classify_traj, report_near_ties

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
MAX_STEP = 150
N_OD = 8


def _cfg(hidden):
    return {
        'function_g': {'road_emb_size': 96, 'time_emb_size': 20, 'hidden_size': hidden, 'dropout_p': 0.6,
                       'lstm_layer_num': 2, 'pretrain_road_rep': None, 'dis_weight': 0.5, 'device': DEV},
        'function_h': {'embed_dim': 64, 'gps_emb_dim': 10, 'num_of_heads': 2, 'concat': False, 'device': DEV,
                       'distance_mode': 'l2'},
        'dis_weight': 0.45,
    }


@pytest.fixture(scope='module')
def grid():
    from ts_trajgen_amd import synth
    net = synth.street_grid_network(12)
    od = synth.make_od_pairs(net, N_OD, seed=5, workers=1)
    return net, od


def _oracle(net, model, cfg):
    from oracle.model import OracleGenerator
    g_state = {k: v.detach().cpu() for k, v in model.function_g.state_dict().items()}
    h_state = {k: v.detach().cpu() for k, v in model.function_h.state_dict().items()}
    ei = np.stack([net.adj_mx.row, net.adj_mx.col]).astype(np.int64)
    om = OracleGenerator(g_state, h_state, float(model.w1.detach().cpu()[0]), net.node_features, ei,
                         cfg['function_h']['num_of_heads'], cfg['function_h']['embed_dim'],
                         cfg['function_g']['lstm_layer_num'])
    om.setup_node_emb()
    return om


@pytest.mark.parametrize('hidden', [192, 320, 500, 512])
def test_other_hidden_sizes_match_oracle(grid, hidden):
    from oracle import search as osr
    from ts_trajgen_amd.generator import GeneratorV4
    from ts_trajgen_amd.search import Searcher
    net, od = grid
    cfg = copy.deepcopy(_cfg(hidden))
    torch.manual_seed(hidden)
    model = GeneratorV4(config=cfg, data_feature=net.data_feature(torch)).to(DEV)
    model.function_h.node_features = model.function_h.node_features.to(DEV)
    model.train(False)
    om = _oracle(net, model, cfg)
    rtd = net.road_time_distribution.astype(int)
    want = []
    for i in range(N_OD):
        st = osr.SearchStats()
        loc, tim = osr.astar_search(om, net.adjacent_list, net.rid_gps, net.road_length, rtd, [int(od['start'][i])],
                                    [int(od['start_time'][i])], int(od['des'][i]), int(od['default_len'][i]),
                                    MAX_STEP, stats=st)
        want.append((loc, tim, st))
    for state_format in (1, 0):
        s = Searcher(DEV, net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution)
        s.state_format = state_format
        res = s.generate(model, od['start'], od['start_time'], od['des'], od['default_len'], max_step=MAX_STEP,
                         slots=8, poplog_cap=MAX_STEP + 8)
        kinds = []
        for i, (loc, tim, st) in enumerate(want):
            kind = classify_traj(res, i, loc, tim, [p[0] for p in st.pops], [p[1] for p in st.pops])
            if kind == 'exact':
                assert int(res['steps'][i]) == st.steps
            kinds.append(kind)
        report_near_ties('hidden %d, state_format %d against the oracle' % (hidden, state_format), kinds, recorded=0)


def test_hidden_beyond_the_limit_is_refused(grid):
    """hidden > 512: the C ABI reports TSG_E_INVALID before anything is launched on the engine."""
    from ts_trajgen_amd.generator import GeneratorV4
    from ts_trajgen_amd.search import Searcher
    net, od = grid
    cfg = copy.deepcopy(_cfg(576))
    torch.manual_seed(1)
    model = GeneratorV4(config=cfg, data_feature=net.data_feature(torch)).to(DEV)
    model.function_h.node_features = model.function_h.node_features.to(DEV)
    model.train(False)
    s = Searcher(DEV, net.adjacent_list, net.rid_gps, net.road_length, net.road_time_distribution)
    with pytest.raises(RuntimeError, match='TSG_E_'):
        s.generate(model, od['start'], od['start_time'], od['des'], od['default_len'], max_step=10, slots=8)
    torch.cuda.synchronize()
