"""Loaders for tests/golden/* (vectors produced by tools/gen_golden.py from the reference)."""
import json
import os

import numpy as np
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def load_npz(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def split_state(npz):
    g = {k[2:]: v for k, v in npz.items() if k.startswith('g.')}
    h = {k[2:]: v for k, v in npz.items() if k.startswith('h.')}
    return g, h


class GoldenRoadWorld(object):
    """The small 'messy' 16x16 street grid + road model every search golden was made on."""

    def __init__(self):
        self.net = load_json('road_net.json')
        self.npz = load_npz('road_model.npz')
        self.g_state, self.h_state = split_state(self.npz)
        self.w1 = float(self.npz['w1'][0])
        self.n = self.net['num_roads']
        self.adjacent_list = self.net['adjacent_list']
        self.rid_gps = self.net['rid_gps']
        self.road_length = self.net['road_length']
        self.rtd = self.npz['rtd']
        self.node_features = self.npz['node_features']
        self.edge_index = self.npz['edge_index']
        self.cfg = load_json('g234_predict.json')['config']

    def adj_mx(self):
        ei = self.edge_index
        return sp.coo_matrix((np.ones(ei.shape[1], dtype=np.float32), (ei[0], ei[1])),
                             shape=(self.n + 1, self.n + 1))

    def data_feature(self, torch):
        return {'road_num': self.n + 1, 'time_size': 2881, 'road_pad': self.n, 'time_pad': 2880,
                'adj_mx': self.adj_mx(), 'node_features': torch.as_tensor(self.node_features),
                'img_width': self.net['img_width'], 'img_height': self.net['img_height']}

    def oracle_model(self):
        from oracle.model import OracleGenerator
        fh = self.cfg['function_h']
        return OracleGenerator(self.g_state, self.h_state, self.w1, self.node_features, self.edge_index,
                               fh['num_of_heads'], fh['embed_dim'], self.cfg['function_g']['lstm_layer_num'],
                               fh['concat'], fh.get('no_gps_emb', False))


class GoldenRegionWorld(object):
    def __init__(self):
        self.meta = load_json('g7_double_layer.json')
        self.npz = load_npz('region_model.npz')
        self.g_state, self.h_state = split_state(self.npz)
        self.w1 = float(self.npz['w1'][0])
        self.cfg = self.meta['config']
        self.num_regions = self.meta['num_regions']
        self.node_features = self.npz['node_features']
        self.edge_index = self.npz['edge_index']

    def adj_mx(self):
        ei = self.edge_index
        n = self.num_regions
        return sp.coo_matrix((np.ones(ei.shape[1], dtype=np.float32), (ei[0], ei[1])), shape=(n + 1, n + 1))

    def data_feature(self, torch):
        n = self.num_regions
        return {'road_num': n + 1, 'time_size': 2881, 'road_pad': n, 'time_pad': 2880,
                'adj_mx': self.adj_mx(), 'node_features': torch.as_tensor(self.node_features),
                'img_width': 52, 'img_height': 43}

    def oracle_model(self):
        from oracle.model import OracleGenerator
        fh = self.cfg['function_h']
        return OracleGenerator(self.g_state, self.h_state, self.w1, self.node_features, self.edge_index,
                               fh['num_of_heads'], fh['embed_dim'], self.cfg['function_g']['lstm_layer_num'],
                               fh['concat'], fh.get('no_gps_emb', False))

    def tables(self, road_world):
        return {'adjacent_list': road_world.adjacent_list, 'road_center_gps': road_world.rid_gps,
                'road_length': road_world.road_length,
                'road_time_distribution': road_world.rtd.astype(int),
                'region_adjacent_list': self.meta['region_adjacent_list'],
                'region_dist': self.npz['region_dist'],
                'region_transfer_freq': self.meta['region_transfer_freq'],
                'rid2region': self.meta['rid2region'], 'region2rid': self.meta['region2rid'],
                'region_time_distribution': self.npz['region_time_distribution'].astype(int)}


def build_generator(world, device):
    """Our GeneratorV4 (HIP path) loaded with a golden world's weights, eval mode, on `device`."""
    import torch
    from ts_trajgen_amd.generator import GeneratorV4
    cfg = json.loads(json.dumps(world.cfg))
    cfg['function_g']['device'] = device
    cfg['function_h']['device'] = device
    model = GeneratorV4(config=cfg, data_feature=world.data_feature(torch))
    model.function_g.load_state_dict({k: torch.as_tensor(v) for k, v in world.g_state.items()})
    model.function_h.load_state_dict({k: torch.as_tensor(v) for k, v in world.h_state.items()})
    with torch.no_grad():
        model.w1.fill_(world.w1)
    model = model.to(device)
    model.function_h.node_features = model.function_h.node_features.to(device)
    model.train(False)
    return model


def classify_traj(res, i, loc, tim, pop_rid, pop_prio, near_tie=1e-5):
    """How search i of a result dict (with its pop log) compares with the expected (loc, tim) and the expected
    pop trace: 'exact', or 'near-tie' when the FIRST diverging pop differs in priority by < near_tie (relative
    to max(1, |priority|)): the documented exception of the north star.  Anything else raises: a different
    trajectory is never accepted without that proof."""
    n = int(res['len'][i])
    if res['rid'][i, :n].tolist() == [int(x) for x in loc] and res['time'][i, :n].tolist() == [int(x) for x in tim]:
        return 'exact'
    k = int(res['poplog_n'][i])
    assert k <= res['poplog_rid'].shape[1], 'pop log of search %d truncated: raise poplog_cap' % i
    got_r = res['poplog_rid'][i, :k].tolist()
    got_p = res['poplog_prio'][i, :k].tolist()
    for j in range(min(len(got_r), len(pop_rid))):
        if got_r[j] != int(pop_rid[j]):
            want = float(pop_prio[j])
            assert abs(got_p[j] - want) < near_tie * max(1.0, abs(want)), \
                'search %d diverges at pop %d by %g (not a near-tie)' % (i, j, got_p[j] - want)
            return 'near-tie'
    raise AssertionError('search %d: same pop order but a different trajectory' % i)


class G11World(object):
    """BASELINE.json configs[0] as the REFERENCE ran it (tools/gen_golden.py::gen_bj_dims_config0): 32 x 32 street
    grid, BJ model dims, reference constructors under torch.manual_seed(0), 256 OD pairs (RandomState(2)),
    max_step 5000.  The fixture holds trajectories and full pop traces, no weights: the mirrors reproduce the init
    from the same seed and `weight_sums` proves that they did."""

    BJ_CFG = {   # our_model_generate.py:41-59
        'function_g': {'road_emb_size': 256, 'time_emb_size': 50, 'hidden_size': 256, 'dropout_p': 0.6,
                       'lstm_layer_num': 2, 'pretrain_road_rep': None, 'dis_weight': 0.5, 'device': 'cpu'},
        'function_h': {'embed_dim': 256, 'gps_emb_dim': 10, 'num_of_heads': 5, 'concat': False, 'device': 'cpu',
                       'distance_mode': 'l2'},
        'dis_weight': 0.45,
    }
    MAX_STEP = 5000

    def __init__(self):
        from ts_trajgen_amd import synth
        z = load_npz('g11_bj_dims_config0.npz')
        self.z = z
        self.net = synth.street_grid_network(32)
        self.od = synth.make_od_pairs(self.net, 256, seed=2, workers=1)
        self.len_ptr = np.concatenate([[0], np.cumsum(z['len'])]).astype(np.int64)
        self.pop_ptr = np.concatenate([[0], np.cumsum(z['npop'])]).astype(np.int64)

    def case(self, i):
        a, b = self.len_ptr[i], self.len_ptr[i + 1]
        p, q = self.pop_ptr[i], self.pop_ptr[i + 1]
        z = self.z
        return {'loc': z['loc'][a:b].tolist(), 'tim': z['tim'][a:b].tolist(),
                'pop_rid': z['pop_rid'][p:q].astype(np.int64).tolist(), 'pop_prio': z['pop_prio'][p:q]}

    def model(self, device='cpu'):
        """Our GeneratorV4 with the reference constructors' init under torch.manual_seed(0) (built on the CPU, like
        the reference run, then moved); checked against the fixture's weight sums."""
        import copy
        import torch
        from ts_trajgen_amd.generator import GeneratorV4
        cfg = copy.deepcopy(self.BJ_CFG)
        torch.manual_seed(0)
        model = GeneratorV4(config=cfg, data_feature=self.net.data_feature(torch))
        model.train(False)
        sd = model.state_dict()
        for name, want in zip(self.z['weight_sum_names'].tolist(), self.z['weight_sums'].tolist()):
            got = float(sd[name].detach().double().sum())
            assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), 'init differs from the reference run: %s' % name
        if device != 'cpu':
            cfg['function_g']['device'] = device
            cfg['function_h']['device'] = device
            model = model.to(device)
            model.function_h.device = device
            model.function_h.gat_encoder.device = device
            model.function_h.Apt = model.function_h.Apt.to(device)
            model.function_h.node_features = model.function_h.node_features.to(device)
            model.train(False)
        return model, cfg

    def oracle_model(self, model, cfg):
        from oracle.model import OracleGenerator
        g_state = {k: v.detach().cpu() for k, v in model.function_g.state_dict().items()}
        h_state = {k: v.detach().cpu() for k, v in model.function_h.state_dict().items()}
        ei = np.stack([self.net.adj_mx.row, self.net.adj_mx.col]).astype(np.int64)
        om = OracleGenerator(g_state, h_state, float(model.w1.detach().cpu()[0]), self.net.node_features, ei,
                             cfg['function_h']['num_of_heads'], cfg['function_h']['embed_dim'],
                             cfg['function_g']['lstm_layer_num'])
        om.setup_node_emb()
        return om


def report_near_ties(name, kinds, recorded):
    """Print the exact / near-tie counts of a classified comparison and hold the near-ties to the number recorded
    when the test was last looked at (a silent drift from 0 to a few proven near-ties would otherwise pass)."""
    exact, near = kinds.count('exact'), kinds.count('near-tie')
    print('[near-ties] %s: %d exact, %d near-tie of %d (recorded: %d)' % (name, exact, near, len(kinds), recorded))
    assert exact + near == len(kinds)
    assert near <= recorded, '%s: %d near-ties, recorded %d' % (name, near, recorded)
    return near
