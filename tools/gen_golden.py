"""Generate tests/golden/* by importing the REFERENCE (/root/reference) in the build container.

Run here only (the reference never travels): `python tools/gen_golden.py`.
The reference's `search.py` does `from geopy import distance` (search.py:7); geopy is absent, so a
stub module exposing `distance.distance(p, q).kilometers` is injected BEFORE the import and backed
by oracle/geodesic.py (parity at that boundary is unpinned, see oracle/__init__.py).
Only inputs and outputs are stored (weights, graphs, probabilities, trajectories, pop traces);
no reference source text is copied anywhere.
"""
import json
import os
import queue
import sys
import types

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import numpy as np
import scipy.sparse as sp
import torch

from oracle import geodesic
from ts_trajgen_amd import synth

_geopy = types.ModuleType('geopy')
_dist = types.ModuleType('geopy.distance')
_dist.distance = geodesic._Distance
_geopy.distance = _dist
sys.modules['geopy'] = _geopy
sys.modules['geopy.distance'] = _dist

import search as ref_search                                   # noqa: E402
from generator.gat import GATLayerImp3                        # noqa: E402
from generator.generator_v4 import GeneratorV4                # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def sd_np(module, prefix=''):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


# ----------------------------------------------------------------------------- G1: GAT layer
def gen_g1():
    rng = np.random.RandomState(11)
    n = 65
    e = 200
    src = rng.randint(0, n - 5, size=e)          # nodes n-5.. never a source
    trg = rng.randint(5, n, size=e)              # nodes 0..4 have in-degree 0
    src[:6] = src[6:12]                          # duplicate edges
    trg[:6] = trg[6:12]
    src[12] = trg[12] = 20                       # self loop
    edge_index = np.stack([src, trg]).astype(np.int64)
    out = {}
    for name, fin, fout, concat in (('a', 20, 32, False), ('b', 32, 32, False), ('c', 12, 16, True)):
        torch.manual_seed(100 + fin)
        layer = GATLayerImp3(num_in_features=fin, num_out_features=fout, num_of_heads=5, device='cpu',
                             concat=concat)
        with torch.no_grad():
            layer.bias.normal_(0, 0.1)
        layer.train(False)
        x = torch.randn(n, fin)
        with torch.no_grad():
            y, _ = layer((x, torch.from_numpy(edge_index)))
        out['%s_x' % name] = x.numpy()
        out['%s_out' % name] = y.numpy()
        out['%s_concat' % name] = np.array(int(concat))
        for k, v in sd_np(layer).items():
            out['%s_%s' % (name, k)] = v
    out['edge_index'] = edge_index
    np.savez_compressed(os.path.join(OUT, 'g1_gat.npz'), **out)
    print('G1 done')


# ----------------------------------------------------------------------------- G12: GAT layer, training mode
class _MaskSeq(object):
    """Stands in for the layer's nn.Dropout (gat.py:63): the k-th call multiplies by the k-th recorded mask."""

    def __init__(self, masks):
        self.masks, self.i = masks, 0

    def __call__(self, t):
        m = self.masks[self.i]
        self.i += 1
        return t * m.view(t.shape)


def _set_dropout(layer, masks):
    layer._modules.pop('dropout', None)
    object.__setattr__(layer, 'dropout', _MaskSeq(masks))


def gen_g12():
    """The reference layer in training mode with RECORDED dropout masks (p = 0.6, kept entries scaled by 2.5) in
    place of its nn.Dropout draws, and the gradients its autograd returns for loss = sum(out * G)."""
    z = np.load(os.path.join(OUT, 'g1_gat.npz'))
    edge_index = z['edge_index']
    e = edge_index.shape[1]
    out = {'edge_index': edge_index}
    for name, fin, fout, concat in (('a', 20, 32, False), ('b', 32, 32, False), ('c', 12, 16, True)):
        torch.manual_seed(100 + fin)
        layer = GATLayerImp3(num_in_features=fin, num_out_features=fout, num_of_heads=5, device='cpu', concat=concat)
        with torch.no_grad():
            layer.bias.normal_(0, 0.1)
        layer.train(True)
        n = z['%s_x' % name].shape[0]
        g = torch.Generator().manual_seed(500 + fin)
        keep = 0.4
        masks = [(torch.rand(s, generator=g) < keep).float() / keep for s in ((n, fin), (n, 5 * fout), (e, 5))]
        _set_dropout(layer, masks)
        x = torch.tensor(z['%s_x' % name], requires_grad=True)
        y, _ = layer((x, torch.from_numpy(edge_index)))
        up = torch.randn(y.shape, generator=g)
        (y * up).sum().backward()
        out['%s_x' % name] = x.detach().numpy()
        out['%s_out' % name] = y.detach().numpy()
        out['%s_up' % name] = up.numpy()
        out['%s_concat' % name] = np.array(int(concat))
        for i, m in enumerate(masks):
            out['%s_mask%d' % (name, i)] = m.numpy()
        out['%s_grad_x' % name] = x.grad.numpy()
        for k, v in layer.named_parameters():
            out['%s_%s' % (name, k)] = v.detach().numpy()
            out['%s_grad_%s' % (name, k)] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
    np.savez_compressed(os.path.join(OUT, 'g12_gat_train.npz'), **out)
    print('G12 done')


def gen_g13(net, model):
    """DistanceGatFC.calculate_loss (distance_gat_fc.py:75-95) in training mode, the GAT's dropouts replaced by
    recorded masks: the loss and the gradient of every parameter of function H."""
    h = model.function_h
    h.train(True)
    n = h.node_features.shape[0]
    e = h.Apt.shape[1]
    g = torch.Generator().manual_seed(77)
    keep = 0.4
    nh, fo = h.num_of_heads, h.embed_dim
    masks = [(torch.rand(s, generator=g) < keep).float() / keep for s in ((n, h.feature_dim), (n, nh * fo), (e, nh))]
    _set_dropout(h.gat_encoder, masks)
    rng = np.random.RandomState(5)
    b, c = 24, 4
    cand = torch.from_numpy(rng.randint(0, net.num_roads, size=(b, c)).astype(np.int64))
    des = torch.from_numpy(rng.randint(0, net.num_roads, size=b).astype(np.int64))
    dis = torch.from_numpy(rng.uniform(0.0, 60.0, size=(b, c)).astype(np.float32))
    target = torch.from_numpy(rng.randint(0, c, size=b).astype(np.int64))
    h.zero_grad()
    loss = h.calculate_loss(cand, des, dis, target)
    loss.backward()
    out = {'cand': cand.numpy(), 'des': des.numpy(), 'dis': dis.numpy(), 'target': target.numpy(),
           'loss': np.array(float(loss))}
    for i, m in enumerate(masks):
        out['mask%d' % i] = m.numpy()
    for k, v in h.named_parameters():
        out['grad_' + k] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy()
    np.savez_compressed(os.path.join(OUT, 'g13_h_train.npz'), **out)
    h.zero_grad()
    print('G13 done, loss %.6f' % float(loss))


# ----------------------------------------------------------------------------- shared small world
ROAD_CFG = {
    'function_g': {'road_emb_size': 32, 'time_emb_size': 8, 'hidden_size': 32, 'dropout_p': 0.6,
                   'lstm_layer_num': 2, 'pretrain_road_rep': None, 'dis_weight': 0.5, 'device': 'cpu'},
    'function_h': {'embed_dim': 32, 'gps_emb_dim': 4, 'num_of_heads': 5, 'concat': False, 'device': 'cpu',
                   'distance_mode': 'l2'},
    'dis_weight': 0.45,
}
REGION_CFG = {
    'function_g': {'road_emb_size': 16, 'time_emb_size': 8, 'hidden_size': 16, 'dropout_p': 0.6,
                   'lstm_layer_num': 2, 'pretrain_road_rep': None, 'dis_weight': 0.5, 'device': 'cpu'},
    'function_h': {'embed_dim': 16, 'gps_emb_dim': 4, 'num_of_heads': 5, 'concat': False, 'device': 'cpu',
                   'distance_mode': 'l2', 'no_gps_emb': True},
    'dis_weight': 0.45,
}


def messy_network(seed=5):
    """16x16 street grid with single-successor roads, dead ends and unreachable roads mixed in."""
    net = synth.street_grid_network(16, seed=1)
    rng = np.random.RandomState(seed)
    n = net.num_roads
    adj = {k: list(v) for k, v in net.adjacent_list.items()}
    unreachable = set(rng.choice(n, 4, replace=False).tolist())
    for k in list(adj.keys()):
        adj[k] = [c for c in adj[k] if c not in unreachable]
    for r in rng.choice(n, n // 8, replace=False):
        if len(adj.get(str(r), [])) > 1:
            adj[str(r)] = [adj[str(r)][int(rng.randint(len(adj[str(r)])))]]
    for r in rng.choice(n, n // 30, replace=False):
        adj.pop(str(r), None)
    adj = {k: v for k, v in adj.items() if len(v) > 0}
    rows = [int(k) for k, v in adj.items() for _ in v]
    cols = [c for k, v in adj.items() for c in v]
    net.adjacent_list = adj
    net.adj_mx = sp.coo_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(n + 1, n + 1))
    net.meta['unreachable'] = sorted(unreachable)
    return net


def build_road_model(net):
    torch.manual_seed(0)
    model = GeneratorV4(config=ROAD_CFG, data_feature=net.data_feature(torch))
    with torch.no_grad():
        model.function_h.gat_encoder.bias.normal_(0, 0.1)
        # spread the logits so that the search is not a pure breadth-first sweep
        model.function_g.out_fc.weight.mul_(8.0)
        model.function_h.out_fc.weight.mul_(4.0)
    model.train(False)
    return model


def net_json(net):
    return {'num_roads': net.num_roads, 'adjacent_list': net.adjacent_list, 'rid_gps': net.rid_gps,
            'road_length': net.road_length, 'img_width': net.img_width, 'img_height': net.img_height,
            'unreachable': net.meta.get('unreachable', [])}


class RecordingPQ(queue.PriorityQueue):
    """queue.PriorityQueue that logs every get() as (priority, rid)."""
    log = None

    def get(self, *a, **kw):
        item = super().get(*a, **kw)
        if RecordingPQ.log is not None:
            RecordingPQ.log.append((float(item[0]), int(item[1].rid)))
        return item


def nonstale(log):
    seen = set()
    out = []
    for p, r in log:
        if r not in seen:
            seen.add(r)
            out.append((r, p))
    return out


def gen_model_goldens():
    net = messy_network()
    model = build_road_model(net)
    n = net.num_roads
    rng = np.random.RandomState(7)
    out = {}
    for k, v in sd_np(model.function_g).items():
        out['g.' + k] = v
    for k, v in sd_np(model.function_h).items():
        out['h.' + k] = v
    out['w1'] = model.w1.detach().numpy()
    out['node_features'] = net.node_features
    out['edge_index'] = np.stack([net.adj_mx.row, net.adj_mx.col]).astype(np.int64)
    out['rtd'] = net.road_time_distribution
    with torch.no_grad():
        model.function_h._setup_node_emb()
        out['node_emb'] = model.function_h.node_emb_feature.numpy()                      # G2 (GAT through H)
        # G2/G3/G4 cases: random prefixes that follow the adjacency, candidates = successors
        cases = []
        keys = [int(k) for k in net.adjacent_list.keys() if len(net.adjacent_list[k]) > 1]
        for ci in range(32):
            L = 1 + ci % 12
            r = int(keys[rng.randint(len(keys))])
            loc = [r]
            while len(loc) < L:
                nb = net.adjacent_list.get(str(loc[-1]))
                if not nb:
                    break
                loc.append(int(nb[rng.randint(len(nb))]))
            if str(loc[-1]) not in net.adjacent_list or len(net.adjacent_list[str(loc[-1])]) < 2:
                loc = loc[:1]
            t0 = int(rng.randint(0, 2880))
            tim = [min(t0 + i, 2879) for i in range(len(loc))]
            cand = net.adjacent_list[str(loc[-1])]
            des = int(rng.randint(n))
            dis = [geodesic.candidate_dis(net.rid_gps[str(des)], net.rid_gps[str(c)]) for c in cand]
            tl = torch.LongTensor(loc).unsqueeze(0)
            tt = torch.LongTensor(tim).unsqueeze(0)
            cs = torch.LongTensor(cand).unsqueeze(0)
            cd = torch.FloatTensor(dis).unsqueeze(0)
            dt = torch.LongTensor([des])
            g = model.function_g.predict_g(tl, tt, dt, cs, cd)
            h = model.function_h.predict_next(cs, dt, cd)
            p = model.predict(trace_loc=tl, trace_time=tt, des=dt, candidate_set=cs, candidate_dis=cd)
            cases.append({'loc': loc, 'tim': tim, 'des': des, 'cand': [int(c) for c in cand],
                          'dis': [float(np.float32(d)) for d in dis],
                          'g': g.squeeze(0).tolist(), 'h': h.squeeze(0).tolist(), 'p': p.squeeze(0).tolist(),
                          'logp': torch.log(p).squeeze(0).tolist()})
    np.savez_compressed(os.path.join(OUT, 'road_model.npz'), **out)
    with open(os.path.join(OUT, 'road_net.json'), 'w') as f:
        json.dump(net_json(net), f)
    with open(os.path.join(OUT, 'g234_predict.json'), 'w') as f:
        json.dump({'config': ROAD_CFG, 'cases': cases}, f)
    print('G2-G4 done')
    return net, model


# ----------------------------------------------------------------------------- G5/G6: road A*
def gen_search_goldens(net, model):
    ref_search.PriorityQueue = RecordingPQ
    searcher = ref_search.Searcher('cpu', net.adjacent_list, net.rid_gps, net.road_length,
                                   net.road_time_distribution)
    n = net.num_roads
    rng = np.random.RandomState(9)
    od = synth.make_od_pairs(net, 64, seed=2)
    cases = []
    unreachable = net.meta['unreachable']
    for i in range(64):
        o, d, t, dl = int(od['start'][i]), int(od['des'][i]), int(od['start_time'][i]), int(od['default_len'][i])
        max_step = 5000
        if i % 8 == 3:
            max_step = int(rng.randint(3, 40))               # budget exhausted -> fallback / untouched
        if i % 16 == 5:
            d = int(unreachable[i % len(unreachable)])       # des with no incoming edge
            dl = int(rng.randint(3, 12))
        if i == 10:
            d = o                                            # start == des
        if i % 4 == 1:
            dl = int(rng.randint(2, 9))                      # short default_len: exercises the fallback snapshot
        RecordingPQ.log = []
        with torch.no_grad():
            loc, tim = searcher.astar_search(model, [o], [t], d, dl, max_step=max_step)
        pops = nonstale(RecordingPQ.log)
        cases.append({'start': o, 'start_time': t, 'des': d, 'default_len': dl, 'max_step': max_step,
                      'loc': [int(x) for x in loc], 'tim': [int(x) for x in tim],
                      'pop_rid': [p[0] for p in pops], 'pop_prio': [p[1] for p in pops]})
    with open(os.path.join(OUT, 'g5_astar.json'), 'w') as f:
        json.dump({'cases': cases}, f)
    print('G5 done', sum(c['loc'][-1] == c['des'] for c in cases), 'reached of', len(cases))

    # G6: in-region search continuing a multi-element prefix
    rt = synth.region_tables(net, tiles=2)
    cases6 = []
    for i in range(16):
        base = cases[(i * 3) % 64]
        if len(base['loc']) < 4:
            continue
        k = 2 + i % 3
        prefix_loc, prefix_tim = base['loc'][:k], base['tim'][:k]
        reg = rt['rid2region'][str(prefix_loc[-1])]
        region_set = set(rt['region2rid'][str(reg)])
        members = sorted(region_set)
        d = int(members[rng.randint(len(members))]) if i % 2 == 0 else int(rng.randint(n))
        dl = int(rng.randint(k + 1, k + 10))
        RecordingPQ.log = []
        with torch.no_grad():
            loc, tim = searcher.astar_search_in_region(model, prefix_loc, prefix_tim, d, dl, region_set,
                                                       max_step=300)
        pops = nonstale(RecordingPQ.log)
        cases6.append({'prefix_loc': prefix_loc, 'prefix_tim': prefix_tim, 'des': d, 'default_len': dl,
                       'region': int(reg), 'max_step': 300, 'loc': [int(x) for x in loc],
                       'tim': [int(x) for x in tim], 'pop_rid': [p[0] for p in pops],
                       'pop_prio': [p[1] for p in pops]})
    with open(os.path.join(OUT, 'g6_astar_in_region.json'), 'w') as f:
        json.dump({'tiles': 2, 'cases': cases6}, f)
    print('G6 done', len(cases6))
    return rt


# ----------------------------------------------------------------------------- G7: two-stage
def build_region_model(net, model, rt):
    nreg = rt['num_regions']
    n = net.num_roads
    with torch.no_grad():
        model.function_h._setup_node_emb()
        node_emb = model.function_h.node_emb_feature
        mat = np.zeros((nreg, n))
        for rid, reg in rt['rid2region'].items():
            mat[reg][int(rid)] = 1.0
        region_feature = torch.matmul(torch.FloatTensor(mat), node_emb)     # prepare_region_feature.py:69-75
    region_df = {'road_num': nreg + 1, 'time_size': 2881, 'road_pad': nreg, 'time_pad': 2880,
                 'adj_mx': rt['region_adj_mx'], 'node_features': region_feature, 'img_width': 52, 'img_height': 43}
    torch.manual_seed(1)
    region_model = GeneratorV4(config=REGION_CFG, data_feature=region_df)
    with torch.no_grad():
        region_model.function_h.gat_encoder.bias.normal_(0, 0.1)
        region_model.function_g.out_fc.weight.mul_(8.0)
    region_model.train(False)
    return region_model, region_feature


def gen_double_layer(net, model, rt):
    nreg = rt['num_regions']
    n = net.num_roads
    region_model, region_feature = build_region_model(net, model, rt)
    ref_search.PriorityQueue = queue.PriorityQueue
    searcher = ref_search.DoubleLayerSearcher('cpu', net.adjacent_list, net.rid_gps, net.road_length,
                                              rt['region_adjacent_list'], rt['region_dist'],
                                              rt['region_transfer_freq'], rt['rid2region'], rt['region2rid'],
                                              net.road_time_distribution, rt['region_time_distribution'])
    od = synth.make_od_pairs(net, 24, seed=4)
    cases = []
    for i in range(24):
        o, d, t, dl = int(od['start'][i]), int(od['des'][i]), int(od['start_time'][i]), int(od['default_len'][i])
        np.random.seed(i)
        with torch.no_grad():
            loc, tim, is_astar = searcher.astar_search(region_model, model, o, t, d, dl, max_step=400)
        # the region-level trace alone (same seed state is irrelevant: no RNG in the region stage)
        with torch.no_grad():
            rtrace, flag = searcher.astar_search_in_region(region_model, rt['rid2region'][str(o)], t,
                                                           rt['rid2region'][str(d)], 400)
        cases.append({'start': o, 'start_time': t, 'des': d, 'default_len': dl, 'max_step': 400, 'seed': i,
                      'loc': [int(x) for x in loc], 'tim': [int(x) for x in tim], 'is_astar': int(is_astar),
                      'region_trace': [int(x) for x in rtrace], 'region_flag': int(flag)})
    out = {}
    for k, v in sd_np(region_model.function_g).items():
        out['g.' + k] = v
    for k, v in sd_np(region_model.function_h).items():
        out['h.' + k] = v
    out['w1'] = region_model.w1.detach().numpy()
    out['node_features'] = region_feature.numpy()
    out['edge_index'] = np.stack([rt['region_adj_mx'].row, rt['region_adj_mx'].col]).astype(np.int64)
    with torch.no_grad():
        region_model.function_h._setup_node_emb()
        out['node_emb'] = region_model.function_h.node_emb_feature.numpy()
    out['region_dist'] = rt['region_dist']
    out['region_time_distribution'] = rt['region_time_distribution']
    np.savez_compressed(os.path.join(OUT, 'region_model.npz'), **out)
    with open(os.path.join(OUT, 'g7_double_layer.json'), 'w') as f:
        json.dump({'config': REGION_CFG, 'tiles': 2, 'num_regions': nreg,
                   'rid2region': rt['rid2region'], 'region2rid': rt['region2rid'],
                   'region_adjacent_list': rt['region_adjacent_list'],
                   'region_transfer_freq': rt['region_transfer_freq'],
                   'fail_find_log': sorted([[int(a), int(b)] for a, b in searcher.fail_find_log]),
                   'region_fail_find_log': sorted([[int(a), int(b)] for a, b in searcher.region_fail_find_log]),
                   'cases': cases}, f)
    print('G7 done', sum(c['loc'][-1] == c['des'] for c in cases), 'reached of', len(cases))


# ----------------------------------------------------------------------------- G8: stochastic rollouts
class _RecordingMultinomial(object):
    """Wraps torch.Tensor.multinomial while a rollout runs: records (probabilities, drawn index)."""
    log = None
    orig = torch.Tensor.multinomial

    @staticmethod
    def patched(self, *a, **kw):
        out = _RecordingMultinomial.orig(self, *a, **kw)
        if _RecordingMultinomial.log is not None:
            _RecordingMultinomial.log.append(([float(x) for x in self.detach().tolist()], int(out.item())))
        return out


def gen_random_sample(net, model, rt):
    """road_random_sample (search.py:614-716) and region_random_sample (:1163-1256) under
    torch.manual_seed(1000 + i); stored: inputs, outputs and every (probability vector, drawn index)."""
    region_model, _ = build_region_model(net, model, rt)
    ref = dict(np.load(os.path.join(OUT, 'region_model.npz')))
    for k, v in sd_np(region_model.function_g).items():
        assert np.array_equal(ref['g.' + k], v), k            # same region model as G7
    ref_search.PriorityQueue = queue.PriorityQueue
    searcher = ref_search.DoubleLayerSearcher('cpu', net.adjacent_list, net.rid_gps, net.road_length,
                                              rt['region_adjacent_list'], rt['region_dist'],
                                              rt['region_transfer_freq'], rt['rid2region'], rt['region2rid'],
                                              net.road_time_distribution, rt['region_time_distribution'])
    road_searcher = ref_search.Searcher('cpu', net.adjacent_list, net.rid_gps, net.road_length,
                                        net.road_time_distribution)
    with open(os.path.join(OUT, 'g5_astar.json')) as f:
        g5 = json.load(f)['cases']
    torch.Tensor.multinomial = _RecordingMultinomial.patched
    rng = np.random.RandomState(21)
    n = net.num_roads
    road_cases, region_cases = [], []
    try:
        for i in range(40):
            base = g5[(i * 5) % len(g5)]
            k = 1 if i % 3 == 0 else min(len(base['loc']), 2 + i % 4)         # prefix length
            loc, tim = base['loc'][:k], base['tim'][:k]
            des = base['des'] if i % 5 != 4 else int(rng.randint(n))
            dl = int(rng.randint(2, 30)) if i % 4 else int(rng.randint(21, 40))
            _RecordingMultinomial.log = []
            torch.manual_seed(1000 + i)
            with torch.no_grad():
                o_loc, o_tim = road_searcher.road_random_sample(model, loc, tim, des, dl)
            road_cases.append({'prefix_loc': loc, 'prefix_tim': tim, 'des': int(des), 'default_len': dl,
                               'seed': 1000 + i, 'loc': [int(x) for x in o_loc], 'tim': [int(x) for x in o_tim],
                               'draws': _RecordingMultinomial.log})
        nreg = rt['num_regions']
        for i in range(16):
            s0, d0 = int(rng.randint(nreg)), int(rng.randint(nreg))
            t0 = int(rng.randint(0, 2880))
            loc, tim = [s0], [t0]
            if i % 3 == 1 and str(s0) in rt['region_adjacent_list']:
                nxt = int(list(rt['region_adjacent_list'][str(s0)].keys())[0])
                loc, tim = [s0, nxt], [t0, min(t0 + 3, 2879) if t0 < 1440 else min(t0 + 3, 2879)]
            dl = int(rng.randint(2, 8))
            _RecordingMultinomial.log = []
            torch.manual_seed(2000 + i)
            with torch.no_grad():
                o_loc, o_tim = searcher.region_random_sample(region_model, loc, tim, d0, dl)
            region_cases.append({'prefix_loc': loc, 'prefix_tim': tim, 'des': d0, 'default_len': dl,
                                 'seed': 2000 + i, 'loc': [int(x) for x in o_loc], 'tim': [int(x) for x in o_tim],
                                 'draws': _RecordingMultinomial.log})
    finally:
        torch.Tensor.multinomial = _RecordingMultinomial.orig
        _RecordingMultinomial.log = None
    with open(os.path.join(OUT, 'g8_random_sample.json'), 'w') as f:
        json.dump({'road': road_cases, 'region': region_cases}, f)
    print('G8 done: road reached', sum(c['loc'][-1] == c['des'] for c in road_cases), 'of', len(road_cases),
          '; region reached', sum(c['loc'][-1] == c['des'] for c in region_cases), 'of', len(region_cases),
          '; draws', sum(len(c['draws']) for c in road_cases), sum(len(c['draws']) for c in region_cases))


# ----------------------------------------------------------------------------- G9: beam search
def gen_beam_search(net, model):
    """Searcher.beam_search (search.py:137-238) under np.random.seed(3000 + i); stored: inputs, the returned
    trajectory and the bound `index` of the final np.random.randint(index) (how many end nodes were within
    prob_threshold of the best)."""
    ref = dict(np.load(os.path.join(OUT, 'road_model.npz')))
    for k, v in sd_np(model.function_g).items():
        assert np.array_equal(ref['g.' + k], v), k            # same road model as G2-G8
    searcher = ref_search.Searcher('cpu', net.adjacent_list, net.rid_gps, net.road_length,
                                   net.road_time_distribution)
    with open(os.path.join(OUT, 'g5_astar.json')) as f:
        g5 = json.load(f)['cases']
    bounds = []
    orig_randint = np.random.randint

    def recording_randint(*a, **kw):
        bounds.append(int(a[0]))
        return orig_randint(*a, **kw)
    cases = []
    np.random.randint = recording_randint
    try:
        for i in range(32):
            base = g5[(i * 3) % len(g5)]
            k = 1 if i % 3 else min(len(base['loc']), 2 + i % 3)              # prefix length
            loc, tim = base['loc'][:k], base['tim'][:k]
            width = [15, 3, 1, 6][i % 4]
            max_step = [50, 12, 30, 5][(i // 4) % 4]
            thr = 1e-4 if i % 5 else 0.5                                      # 0.5: several end nodes qualify
            del bounds[:]
            np.random.seed(3000 + i)
            case = {'prefix_loc': loc, 'prefix_tim': tim, 'des': int(base['des']), 'width': width,
                    'max_step': max_step, 'prob_threshold': thr, 'seed': 3000 + i}
            try:
                with torch.no_grad():
                    o_loc, o_tim = searcher.beam_search(model, loc, tim, base['des'], width=width,
                                                        max_step=max_step, prob_threshold=thr)
                case.update({'loc': [int(x) for x in o_loc], 'tim': [int(x) for x in o_tim],
                             'index_bound': bounds[-1]})
            except IndexError:      # every branch died: the reference indexes an empty end_beam (search.py:230)
                case.update({'error': 'IndexError'})
            cases.append(case)
    finally:
        np.random.randint = orig_randint
    with open(os.path.join(OUT, 'g9_beam_search.json'), 'w') as f:
        json.dump({'cases': cases}, f)
    ok = [c for c in cases if 'loc' in c]
    print('G9 done: reached', sum(c['loc'][-1] == c['des'] for c in ok), 'of', len(cases), '; errors',
          len(cases) - len(ok), '; index bounds', sorted(set(c['index_bound'] for c in ok)))


# ----------------------------------------------------------------------------- G10: masked batches, logits before the softmax
def gen_masked_batch(net, model):
    """GeneratorV4.predict / FunctionGFC.predict_g / FunctionGFC.forward for a padded (B, T) batch WITH trace_mask
    (function_g_fc.py:162-180: packed LSTM, last valid hidden state as the query, attention over every position
    of the re-padded LSTM output -- the masked_fill of :33 is assigned to `mask`, so padded (zero) rows stay in
    the softmax), and FunctionGFC.forward's candidate logits for the unmasked one-row cases."""
    cases = json.load(open(os.path.join(OUT, 'g234_predict.json')))['cases']
    counts = [len(c['cand']) for c in cases]
    out = {'batches': [], 'logits': []}
    with torch.no_grad():
        for ncand in sorted(set(counts)):
            grp = [c for c in cases if len(c['cand']) == ncand]
            if len(grp) < 2:
                continue
            T = max(len(c['loc']) for c in grp)
            loc = torch.full((len(grp), T), net.num_roads, dtype=torch.long)     # road_pad / time_pad in the padding
            tim = torch.full((len(grp), T), 2880, dtype=torch.long)
            mask = torch.zeros((len(grp), T), dtype=torch.long)
            for i, c in enumerate(grp):
                loc[i, :len(c['loc'])] = torch.LongTensor(c['loc'])
                tim[i, :len(c['tim'])] = torch.LongTensor(c['tim'])
                mask[i, :len(c['loc'])] = 1
            cs = torch.LongTensor([c['cand'] for c in grp])
            cd = torch.FloatTensor([c['dis'] for c in grp])
            dt = torch.LongTensor([c['des'] for c in grp])
            p = model.predict(trace_loc=loc, trace_time=tim, des=dt, candidate_set=cs, candidate_dis=cd,
                              trace_mask=mask)
            g = model.function_g.predict_g(loc, tim, dt, cs, cd, trace_mask=mask)
            score = model.function_g.forward(loc, tim, dt, cs, cd, trace_mask=mask, train=False)
            mix = model.forward(trace_loc=loc, trace_time=tim, des=dt, candidate_set=cs, candidate_dis=cd,
                                trace_mask=mask)
            out['batches'].append({'loc': loc.tolist(), 'tim': tim.tolist(), 'mask': mask.tolist(),
                                   'cand': cs.tolist(), 'dis': cd.tolist(), 'des': dt.tolist(),
                                   'p': p.tolist(), 'g': g.tolist(), 'g_score': score.tolist(), 'mix': mix.tolist()})
        for c in cases[:12]:
            tl = torch.LongTensor(c['loc']).unsqueeze(0)
            tt = torch.LongTensor(c['tim']).unsqueeze(0)
            cs = torch.LongTensor(c['cand']).unsqueeze(0)
            cd = torch.FloatTensor(c['dis']).unsqueeze(0)
            dt = torch.LongTensor([c['des']])
            score = model.function_g.forward(tl, tt, dt, cs, cd, None, False)
            out['logits'].append({'loc': c['loc'], 'tim': c['tim'], 'cand': c['cand'], 'dis': c['dis'],
                                  'des': c['des'], 'g_score': score.squeeze(0).tolist()})
    with open(os.path.join(OUT, 'g10_masked_batch.json'), 'w') as f:
        json.dump(out, f)
    print('G10 done', len(out['batches']), 'batches', [len(b['loc']) for b in out['batches']])


# ----------------------------------------------------------------------------- G11: the reference at BJ dims on configs[0]
BJ_CFG = {   # our_model_generate.py:41-59
    'function_g': {'road_emb_size': 256, 'time_emb_size': 50, 'hidden_size': 256, 'dropout_p': 0.6,
                   'lstm_layer_num': 2, 'pretrain_road_rep': None, 'dis_weight': 0.5, 'device': 'cpu'},
    'function_h': {'embed_dim': 256, 'gps_emb_dim': 10, 'num_of_heads': 5, 'concat': False, 'device': 'cpu',
                   'distance_mode': 'l2'},
    'dis_weight': 0.45,
}


def _bj_worker(args):
    lo, hi = args
    torch.set_num_threads(1)
    net = synth.street_grid_network(32)
    torch.manual_seed(0)
    model = GeneratorV4(config=BJ_CFG, data_feature=net.data_feature(torch))
    model.train(False)
    od = synth.make_od_pairs(net, 256, seed=2)
    ref_search.PriorityQueue = RecordingPQ
    searcher = ref_search.Searcher('cpu', net.adjacent_list, net.rid_gps, net.road_length,
                                   net.road_time_distribution)
    res = []
    for i in range(lo, hi):
        RecordingPQ.log = []
        with torch.no_grad():
            loc, tim = searcher.astar_search(model, [int(od['start'][i])], [int(od['start_time'][i])],
                                             int(od['des'][i]), int(od['default_len'][i]), max_step=5000)
        pops = nonstale(RecordingPQ.log)
        res.append((i, [int(x) for x in loc], [int(x) for x in tim], [p[0] for p in pops], [p[1] for p in pops]))
    probe = {k: float(v.detach().double().sum()) for k, v in model.state_dict().items()
             if k in ('function_g.lstm.weight_ih_l0', 'function_g.out_fc.weight',
                      'function_h.gat_encoder.linear_proj.weight', 'function_h.emb_fc.weight')}
    return res, probe


def gen_bj_dims_config0(workers=6):
    """BASELINE.json configs[0] as written, run by the REFERENCE: 32 x 32 street grid (N = 1 984), BJ model dims,
    reference constructors under torch.manual_seed(0), 256 OD pairs (RandomState(2)), max_step 5000.  Stores the
    trajectories and the pop traces (rid int16 / priority f32); no weights (41 MB): the mirrors reproduce the
    init from the same seed, and `weight_sums` lets the test check that they did."""
    import multiprocessing as mp
    chunks = [(i, min(256, i + 4)) for i in range(0, 256, 4)]
    with mp.get_context('fork').Pool(workers) as pool:
        parts = pool.map(_bj_worker, chunks, chunksize=1)
    rows = sorted([r for part, _ in parts for r in part])
    probe = parts[0][1]
    lens = np.array([len(r[1]) for r in rows], dtype=np.int32)
    npop = np.array([len(r[3]) for r in rows], dtype=np.int32)
    out = {'len': lens, 'loc': np.concatenate([np.asarray(r[1], dtype=np.int32) for r in rows]),
           'tim': np.concatenate([np.asarray(r[2], dtype=np.int32) for r in rows]),
           'npop': npop, 'pop_rid': np.concatenate([np.asarray(r[3], dtype=np.int16) for r in rows]),
           'pop_prio': np.concatenate([np.asarray(r[4], dtype=np.float32) for r in rows]),
           'weight_sum_names': np.array(sorted(probe.keys())),
           'weight_sums': np.array([probe[k] for k in sorted(probe.keys())], dtype=np.float64)}
    np.savez_compressed(os.path.join(OUT, 'g11_bj_dims_config0.npz'), **out)
    print('G11 done: 256 OD pairs, mean length %.1f, mean pops %.0f' % (lens.mean(), npop.mean()))



if __name__ == '__main__':
    only = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if only == 'g12':                     # adds G12 / G13 next to the existing files (same deterministic world)
        gen_g12()
        net = messy_network()
        gen_g13(net, build_road_model(net))
        sys.exit(0)
    if only == 'g10':                     # adds G10 next to the existing files (same deterministic world)
        net = messy_network()
        gen_masked_batch(net, build_road_model(net))
        sys.exit(0)
    if only == 'g11':                     # ~20 minutes on 8 cores: the reference itself at BJ dims
        gen_bj_dims_config0()
        sys.exit(0)
    if only == 'g9':                      # adds G9 next to the existing files (same deterministic world)
        net = messy_network()
        gen_beam_search(net, build_road_model(net))
        sys.exit(0)
    if only == 'g8':                      # adds G8 next to the existing files (same deterministic world)
        net = messy_network()
        model = build_road_model(net)
        gen_random_sample(net, model, synth.region_tables(net, tiles=2))
    else:
        gen_g1()
        net, model = gen_model_goldens()
        rt = gen_search_goldens(net, model)
        gen_double_layer(net, model, rt)
        gen_random_sample(net, model, rt)
        gen_beam_search(net, model)
        gen_masked_batch(net, model)
        gen_bj_dims_config0()
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))
