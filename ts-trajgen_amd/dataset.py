"""On-disk formats of the generation path -> host tables.

The reference reads its inputs inline in our_model_generate.py:189-391; this module reads (and, for
synthetic data and tests, writes) the same files with the same names and layouts:

  adjacent_list.json          {"rid": [next rid, ...]}; roads without successors have no key
                              (script/preprocess_pretrain_input.py:66-76)
  rid_gps.json                {"rid": [lon, lat]}                      (:92-102)
  road_length.json            {"rid": metres}                          (script/construct_region_dist.py:16-25)
  road_time_distribution.npy  (24, N) seconds per hour-of-day, 1 = unknown
                              (script/generate_time_distribution.py:10,42-50)
  adjacent_mx.npz             scipy COO (N+1, N+1), row = from, col = to   (pretrain_gat_fc.py:127-141)
  node_feature.pt             float tensor (N, F), last two columns lon_grid, lat_grid  (:143-195)
  region family               region2rid / rid2region / region_adjacent_list {a: {b: [border rids]}}
                              (script/construct_traffic_zone_relation.py:27-75), region_transfer_prob
                              {a: {b: {transfer_rid, transfer_freq}}} (script/count_region_transfer.py:32-70),
                              region_dist.npy (R, R) metres, region_time_distribution.npy (24, R),
                              region_adj_mx.npz, region_feature.pt
  <test>.csv                  traj_id,rid_list,time_list (comma-joined strings, ISO-8601 times)
  <generate>.csv              traj_id,"r1,r2,...","t1,t2,..."          (our_model_generate.py:413-428)

`.pt` files are read with `torch.load(weights_only=True)` only.
"""
import json
import math
import os

import numpy as np
import scipy.sparse as sp
import torch

from .utils import encode_time

ARCHIVE = 'TS_TrajGen_data_archive'                 # our_model_generate.py:25

# file names per dataset (our_model_generate.py:32-39,191-391); '{d}' = dataset sub-folder of the archive
_NAMES = {
    'BJ_Taxi': dict(prefix='', region='kaffpa_tarjan_', sub='', region_dist='kaffpa_tarjan_region_dist.npy',
                    test='chaoyang_traj_mm_test.csv', out='TS_TrajGen_chaoyang_generate.csv'),
    'Porto_Taxi': dict(prefix='porto_', region='porto_', sub='', region_dist='porto_region_count_dist.npy',
                       test='porto_mm_test.csv', out='TS_TrajGen_generate.csv'),
    'Xian': dict(prefix='', region='', sub='Xian', region_dist='region_count_dist.npy',
                 test='xianshi_partA_mm_test.csv', out='TS_TrajGen_generate.csv'),
}

# model dimensions per dataset (our_model_generate.py:40-83, 92-135, 144-187): road (G r/t/h, H dim/gps/heads),
# region (same)
_DIMS = {
    'BJ_Taxi': ((256, 50, 256, 256, 10, 5), (128, 32, 128, 128, 5, 5)),
    'Porto_Taxi': ((256, 50, 256, 256, 10, 5), (64, 16, 64, 64, 5, 5)),
    'Xian': ((128, 32, 128, 128, 5, 4), (64, 16, 64, 68, 5, 4)),
}
# (road_num, img_width, img_height) (our_model_generate.py:203-213, 273-293, 343; utils/map_manager.py:45-54)
_GEOM = {
    'BJ_Taxi': (40306, math.ceil(0.2507 / 0.005) + 1, math.ceil(0.21 / 0.005) + 1),
    'Porto_Taxi': (11095, 134, 47),
    'Xian': (17378, math.ceil((109.0499449 - 108.8093988) / 0.005) + 1,
             math.ceil((34.29639324 - 34.17026046) / 0.005) + 1),
}


def names(dataset_name):
    """File-name table of a dataset; unknown names use the Xian layout (a sub-folder of the archive)."""
    if dataset_name in _NAMES:
        return dict(_NAMES[dataset_name])
    n = dict(_NAMES['Xian'])
    n['sub'] = dataset_name
    n['test'] = dataset_name + '_mm_test.csv'
    return n


def paths(data_root, dataset_name):
    nm = names(dataset_name)
    arch = os.path.join(data_root, ARCHIVE, nm['sub']) if nm['sub'] else os.path.join(data_root, ARCHIVE)
    p, r = nm['prefix'], nm['region']

    def reg(s):
        return os.path.join(arch, r + 'region_' + s)

    out = {
        'archive': arch,
        'node_feature': os.path.join(arch, p + 'node_feature.pt'),
        'adjacent_mx': os.path.join(arch, p + 'adjacent_mx.npz'),
        'adjacent_list': os.path.join(arch, p + 'adjacent_list.json'),
        'rid_gps': os.path.join(arch, p + 'rid_gps.json'),
        'road_length': os.path.join(arch, p + 'road_length.json'),
        'road_time_distribution': os.path.join(arch, p + 'road_time_distribution.npy'),
        'region_adj_mx': reg('adj_mx.npz'),
        'region_feature': reg('feature.pt'),
        'region_adjacent_list': reg('adjacent_list.json'),
        'region_transfer_prob': reg('transfer_prob.json'),
        'region_time_distribution': reg('time_distribution.npy'),
        'region_dist': os.path.join(arch, nm['region_dist']),
        'region2rid': os.path.join(arch, r + 'region2rid.json'),
        'rid2region': os.path.join(arch, r + 'rid2region.json'),
        'test_csv': os.path.join(data_root, dataset_name, nm['test']),
        'generate_csv': os.path.join(data_root, dataset_name, nm['out']),
        'meta': os.path.join(arch, p + 'dataset_meta.json'),
    }
    return out


def _gen_config(dims, device, no_gps_emb):
    er, et, hid, emb, gps, heads = dims
    fh = {'embed_dim': emb, 'gps_emb_dim': gps, 'num_of_heads': heads, 'concat': False, 'device': device,
          'distance_mode': 'l2'}
    if no_gps_emb:
        fh['no_gps_emb'] = True
    return {'function_g': {'road_emb_size': er, 'time_emb_size': et, 'hidden_size': hid, 'dropout_p': 0.6,
                           'lstm_layer_num': 2, 'pretrain_road_rep': None, 'dis_weight': 0.5, 'device': device},
            'function_h': fh, 'dis_weight': 0.45}


def gen_configs(dataset_name, device, meta=None):
    """(gen_config, region_gen_config) of our_model_generate.py:40-187; `meta['dims']` overrides them for
    datasets the reference does not name."""
    if meta and 'dims' in meta:
        road, region = tuple(meta['dims']['road']), tuple(meta['dims']['region'])
    else:
        road, region = _DIMS.get(dataset_name, _DIMS['Xian'])
    return _gen_config(road, device, False), _gen_config(region, device, True)


def _json(path):
    with open(path, 'r') as f:
        return json.load(f)


def load_dataset(data_root, dataset_name, device='cpu'):
    """Everything our_model_generate.py:189-391 loads, as one dict:
    data_feature / region_data_feature (GeneratorV4 inputs), the DoubleLayerSearcher tables, and the paths."""
    pt = paths(data_root, dataset_name)
    meta = _json(pt['meta']) if os.path.exists(pt['meta']) else {}
    node_features = torch.load(pt['node_feature'], map_location=device, weights_only=True)
    region_features = torch.load(pt['region_feature'], map_location=device, weights_only=True)
    adj_mx = sp.load_npz(pt['adjacent_mx'])
    region_adj_mx = sp.load_npz(pt['region_adj_mx'])
    region2rid = _json(pt['region2rid'])
    road_num, img_w, img_h = _GEOM.get(dataset_name, (adj_mx.shape[0] - 1, None, None))
    if 'road_num' in meta:
        road_num, img_w, img_h = meta['road_num'], meta['img_width'], meta['img_height']
    if road_num + 1 != adj_mx.shape[0]:
        raise ValueError('adjacent_mx is %s but the dataset has %d roads (+1 pad)' % (adj_mx.shape, road_num))
    time_size = 2880
    region_num = len(region2rid)
    r_w, r_h = (52, 43) if dataset_name == 'BJ_Taxi' else (img_w, img_h)       # our_model_generate.py:237-238
    return {
        'paths': pt, 'meta': meta,
        'data_feature': {'road_num': road_num + 1, 'time_size': time_size + 1, 'road_pad': road_num,
                         'time_pad': time_size, 'adj_mx': adj_mx, 'node_features': node_features,
                         'img_width': img_w, 'img_height': img_h},
        'region_data_feature': {'road_num': region_num + 1, 'time_size': time_size + 1, 'road_pad': region_num,
                                'time_pad': time_size, 'adj_mx': region_adj_mx, 'node_features': region_features,
                                'img_width': r_w, 'img_height': r_h},
        'searcher': {
            'adjacent_list': _json(pt['adjacent_list']), 'road_center_gps': _json(pt['rid_gps']),
            'road_length': _json(pt['road_length']),
            'region_adjacent_list': _json(pt['region_adjacent_list']),
            'region_dist': np.load(pt['region_dist']),
            'region_transfer_freq': _json(pt['region_transfer_prob']),
            'rid2region': _json(pt['rid2region']), 'region2rid': region2rid,
            'road_time_distribution': np.load(pt['road_time_distribution']),
            'region_time_distribution': np.load(pt['region_time_distribution']),
        },
    }


def region_features_from_road_emb(node_emb_feature, rid2region, num_regions):
    """Region input features = sum of the member roads' GAT embeddings (prepare_region_feature.py:62-77:
    `region2rid_mat @ node_emb_feature`, done here as an index_add)."""
    n = node_emb_feature.shape[0]
    reg = torch.zeros(n, dtype=torch.long, device=node_emb_feature.device)
    keep = torch.zeros(n, dtype=torch.bool, device=node_emb_feature.device)
    for rid, g in rid2region.items():
        if int(rid) < n:
            reg[int(rid)] = int(g)
            keep[int(rid)] = True
    out = torch.zeros((num_regions, node_emb_feature.shape[1]), dtype=node_emb_feature.dtype,
                      device=node_emb_feature.device)
    out.index_add_(0, reg[keep], node_emb_feature[keep])
    return out


def save_dataset(data_root, dataset_name, net, regions, region_features, dims=None):
    """Write a RoadNetwork (synth.py) + region tables in the reference's on-disk layout."""
    pt = paths(data_root, dataset_name)
    os.makedirs(pt['archive'], exist_ok=True)
    os.makedirs(os.path.dirname(pt['test_csv']), exist_ok=True)

    def dump(key, obj):
        with open(pt[key], 'w') as f:
            json.dump(obj, f)

    dump('adjacent_list', net.adjacent_list)
    dump('rid_gps', net.rid_gps)
    dump('road_length', net.road_length)
    np.save(pt['road_time_distribution'], net.road_time_distribution)
    sp.save_npz(pt['adjacent_mx'], sp.coo_matrix(net.adj_mx))
    torch.save(torch.as_tensor(net.node_features, dtype=torch.float32), pt['node_feature'])
    dump('region2rid', regions['region2rid'])
    dump('rid2region', regions['rid2region'])
    dump('region_adjacent_list', regions['region_adjacent_list'])
    dump('region_transfer_prob', regions['region_transfer_freq'])
    np.save(pt['region_dist'], regions['region_dist'])
    np.save(pt['region_time_distribution'], regions['region_time_distribution'])
    sp.save_npz(pt['region_adj_mx'], sp.coo_matrix(regions['region_adj_mx']))
    torch.save(torch.as_tensor(region_features, dtype=torch.float32).cpu(), pt['region_feature'])
    meta = {'road_num': int(net.num_roads), 'img_width': int(net.img_width), 'img_height': int(net.img_height)}
    if dims is not None:
        meta['dims'] = dims
    dump('meta', meta)
    return pt


# ---- trajectory CSVs ---------------------------------------------------------------------------
def decode_time(code, day='2018-10-01'):
    """An ISO-8601 stamp whose `encode_time` is `code` (weekday for codes < 1440, a Saturday above)."""
    base = '2018-10-06' if code >= 1440 else day
    m = int(code) % 1440
    return '%sT%02d:%02d:00Z' % (base, m // 60, m % 60)


def write_test_csv(path, traj_ids, rid_lists, time_code_lists):
    """The map-matched test file the driver reads: traj_id,rid_list,time_list with comma-joined strings
    (read back with row['rid_list'].split(','), our_model_generate.py:418-420)."""
    with open(path, 'w') as f:
        f.write('traj_id,rid_list,time_list\n')
        for tid, rids, codes in zip(traj_ids, rid_lists, time_code_lists):
            f.write('%s,"%s","%s"\n' % (tid, ','.join(str(int(r)) for r in rids),
                                        ','.join(decode_time(int(c)) for c in codes)))


def read_test_csv(path):
    """-> (traj_ids, start_rids, start_codes, des_rids, default_lens): what the OD loop takes from each row
    (our_model_generate.py:417-426: first rid, first encoded time, last rid, len(rid_list))."""
    import pandas as pd
    df = pd.read_csv(path)
    ids, start, code, des, dl = [], [], [], [], []
    for tid, rl, tl in zip(df['traj_id'], df['rid_list'], df['time_list']):
        rids = [int(i) for i in str(rl).split(',')]
        ids.append(tid)
        start.append(rids[0])
        des.append(rids[-1])
        dl.append(len(rids))
        code.append(encode_time(str(tl).split(',')[0]))
    return (ids, np.asarray(start, dtype=np.int32), np.asarray(code, dtype=np.int32),
            np.asarray(des, dtype=np.int32), np.asarray(dl, dtype=np.int32))


def write_generate_csv(path, traj_ids, locs, tims):
    """our_model_generate.py:413-428."""
    with open(path, 'w') as f:
        f.write('traj_id,rid_list,time_list\n')
        for tid, loc, tim in zip(traj_ids, locs, tims):
            f.write('{},\"{}\",\"{}\"\n'.format(str(tid), ','.join([str(int(r)) for r in loc]),
                                                ','.join([str(int(t)) for t in tim])))


def read_generate_csv(path):
    import pandas as pd
    df = pd.read_csv(path)
    return (list(df['traj_id']), [[int(i) for i in str(r).split(',')] for r in df['rid_list']],
            [[int(i) for i in str(t).split(',')] for t in df['time_list']])
