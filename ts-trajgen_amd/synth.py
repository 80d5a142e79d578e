"""Synthetic road networks, OD pairs and region partitions in the reference's input formats.

The reference ships no data (README.md:9 points at an external archive), so every config of
BASELINE.json is synthetic.  What this module emits has exactly the shapes the reference loads at
our_model_generate.py:191-259: `adjacent_list` (dict str -> list[int], roads without successors have
no key), `rid_gps` (dict str -> [lon, lat]), `road_length` (dict str -> metres),
`road_time_distribution` ((24, N), ones = unknown), `adj_mx` (scipy COO, row = from, col = to,
shape (N+1, N+1)) and `node_features` ((N, 12): 10 attributes + lon_grid + lat_grid).
Recipes follow SURVEY.md section 8(d).
"""
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp

from . import geo

LON0, LAT0, CELL = 116.25, 39.79, 0.005        # our_model_generate.py:207-211


@dataclass
class RoadNetwork:
    num_roads: int
    adjacent_list: dict
    rid_gps: dict
    road_length: dict
    road_time_distribution: np.ndarray
    adj_mx: sp.coo_matrix
    node_features: np.ndarray
    img_width: int
    img_height: int
    meta: dict = field(default_factory=dict)

    def data_feature(self, torch_module=None):
        """The `data_feature` dict of our_model_generate.py:214-223."""
        nf = self.node_features
        if torch_module is not None:
            nf = torch_module.as_tensor(nf, dtype=torch_module.float32)
        n = self.num_roads
        return {'road_num': n + 1, 'time_size': 2880 + 1, 'road_pad': n, 'time_pad': 2880,
                'adj_mx': self.adj_mx, 'node_features': nf,
                'img_width': self.img_width, 'img_height': self.img_height}


def _finish(n, succ, mid_lon, mid_lat, length, grid, seed, meta):
    rng = np.random.RandomState(seed)
    adjacent_list = {str(r): [int(c) for c in succ[r]] for r in range(n) if len(succ[r]) > 0}
    rid_gps = {str(r): [float(mid_lon[r]), float(mid_lat[r])] for r in range(n)}
    road_length = {str(r): float(length[r]) for r in range(n)}
    rtd = np.ones((24, n), dtype=np.float64)
    mask = rng.rand(24, n) < 0.5
    rtd[mask] = rng.randint(10, 121, size=int(mask.sum()))
    rows = np.concatenate([np.full(len(succ[r]), r, dtype=np.int64) for r in range(n)]) if n else np.zeros(0, np.int64)
    cols = np.concatenate([np.asarray(succ[r], dtype=np.int64) for r in range(n)]) if n else np.zeros(0, np.int64)
    adj_mx = sp.coo_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)), shape=(n + 1, n + 1))
    feats = np.empty((n, 12), dtype=np.float32)
    feats[:, :10] = rng.rand(n, 10).astype(np.float32)
    feats[:, 10] = np.clip(np.floor((mid_lon - LON0) / CELL + 1e-9), 0, grid)
    feats[:, 11] = np.clip(np.floor((mid_lat - LAT0) / CELL + 1e-9), 0, grid)
    meta = dict(meta)
    meta['num_edges'] = int(len(rows))
    return RoadNetwork(n, adjacent_list, rid_gps, road_length, rtd, adj_mx, feats, grid + 1, grid + 1, meta)


def street_grid_network(grid=32, seed=1):
    """Config 1: `grid` x `grid` intersections, ONE segment per street (N = 2*grid*(grid-1));
    a -> b iff the two segments share an intersection, successors sorted by id."""
    seg_ends = []
    for j in range(grid):
        for i in range(grid - 1):
            seg_ends.append(((i, j), (i + 1, j)))
    for j in range(grid - 1):
        for i in range(grid):
            seg_ends.append(((i, j), (i, j + 1)))
    n = len(seg_ends)
    at = {}
    for s, (u, v) in enumerate(seg_ends):
        at.setdefault(u, []).append(s)
        at.setdefault(v, []).append(s)
    succ = []
    for s, (u, v) in enumerate(seg_ends):
        nb = sorted(set(at[u] + at[v]) - {s})
        succ.append(nb)
    ends = np.asarray(seg_ends, dtype=np.float64)            # (n, 2, 2)
    lon = LON0 + ends[:, :, 0] * CELL
    lat = LAT0 + ends[:, :, 1] * CELL
    length = geo.geodesic_m(lat[:, 0], lon[:, 0], lat[:, 1], lon[:, 1])
    return _finish(n, succ, lon.mean(axis=1), lat.mean(axis=1), length, grid, seed,
                   {'kind': 'street_grid', 'grid': grid})


def directed_grid_network(grid, n_target=None, e_target=None, seed=3):
    """Configs 2/3/5: directed no-U-turn segment graph on a `grid` x `grid` lattice, thinned to
    `n_target` segments and `e_target` turn edges (uniformly, every segment that had a successor
    keeps at least one)."""
    rng = np.random.RandomState(seed)
    segs = []
    for j in range(grid):
        for i in range(grid):
            if i + 1 < grid:
                segs.append((i, j, i + 1, j))
                segs.append((i + 1, j, i, j))
            if j + 1 < grid:
                segs.append((i, j, i, j + 1))
                segs.append((i, j + 1, i, j))
    segs = np.asarray(segs, dtype=np.int64)
    if n_target is not None and n_target < len(segs):
        keep = np.sort(rng.choice(len(segs), n_target, replace=False))
        segs = segs[keep]
    n = len(segs)
    node_id = segs[:, 0] * grid + segs[:, 1]                 # tail intersection
    head_id = segs[:, 2] * grid + segs[:, 3]
    order = np.argsort(node_id, kind='stable')
    starts = np.searchsorted(node_id[order], np.arange(grid * grid + 1))
    src_list, dst_list = [], []
    for a in range(n):
        h = head_id[a]
        outs = order[starts[h]:starts[h + 1]]
        outs = outs[head_id[outs] != node_id[a]]             # no U-turn
        src_list.append(np.full(len(outs), a, dtype=np.int64))
        dst_list.append(np.sort(outs))
    src = np.concatenate(src_list)
    dst = np.concatenate(dst_list)
    if e_target is not None and e_target < len(src):
        first = np.ones(len(src), dtype=bool)
        first[1:] = src[1:] != src[:-1]
        # protect one uniformly chosen successor per segment
        prot = np.zeros(len(src), dtype=bool)
        grp_start = np.flatnonzero(first)
        grp_len = np.diff(np.append(grp_start, len(src)))
        prot[grp_start + (rng.rand(len(grp_start)) * grp_len).astype(np.int64)] = True
        free = np.flatnonzero(~prot)
        n_drop = len(src) - e_target
        n_drop = min(n_drop, len(free))
        drop = rng.choice(free, n_drop, replace=False)
        keep_e = np.ones(len(src), dtype=bool)
        keep_e[drop] = False
        src, dst = src[keep_e], dst[keep_e]
    succ = [[] for _ in range(n)]
    bounds = np.searchsorted(src, np.arange(n + 1))
    for a in range(n):
        succ[a] = dst[bounds[a]:bounds[a + 1]].tolist()
    lon_a = LON0 + segs[:, 0] * CELL
    lat_a = LAT0 + segs[:, 1] * CELL
    lon_b = LON0 + segs[:, 2] * CELL
    lat_b = LAT0 + segs[:, 3] * CELL
    length = geo.geodesic_m(lat_a, lon_a, lat_b, lon_b)
    # nudge the midpoint of the two directions apart so they do not share a coordinate
    mid_lon = 0.5 * (lon_a + lon_b) + 1e-5 * (segs[:, 3] - segs[:, 1])
    mid_lat = 0.5 * (lat_a + lat_b) - 1e-5 * (segs[:, 2] - segs[:, 0])
    net = _finish(n, succ, mid_lon, mid_lat, length, grid, seed, {'kind': 'directed_grid', 'grid': grid})
    net.meta['segs'] = segs
    return net


def porto_scale_network():
    """Config 2: N = 11 095, E ~ 25 k (54 x 54 lattice)."""
    return directed_grid_network(54, n_target=11095, e_target=25000, seed=3)


def bj_scale_network():
    """Config 3: N = 40 306, E ~ 100 k (102 x 102 lattice)."""
    return directed_grid_network(102, n_target=40306, e_target=100000, seed=3)


def stress_network():
    """Config 5: N = 500 000 segments, E = 1.25 M turn edges (356 x 356 lattice)."""
    return directed_grid_network(356, n_target=500000, e_target=1250000, seed=3)


def lattice_default_len(net, origins, dests, cap=256):
    """default_len for graphs too large for all-pairs BFS: lattice (Manhattan) distance between the segments'
    start intersections + 1, clipped to [2, cap] (config 5 bounds trajectories at 256 roads)."""
    segs = np.asarray(net.meta['segs'])
    o, d = segs[np.asarray(origins)], segs[np.asarray(dests)]
    hops = np.abs(o[:, 0] - d[:, 0]) + np.abs(o[:, 1] - d[:, 1]) + 1
    return np.clip(hops, 2, cap).astype(np.int32)


def csr_from_adjacent_list(adjacent_list, n):
    """Order-preserving CSR of the successor lists (roads absent from the dict get degree 0)."""
    indptr = np.zeros(n + 1, dtype=np.int32)
    for r in range(n):
        lst = adjacent_list.get(str(r))
        indptr[r + 1] = indptr[r] + (len(lst) if lst else 0)
    indices = np.empty(int(indptr[-1]), dtype=np.int32)
    for r in range(n):
        lst = adjacent_list.get(str(r))
        if lst:
            indices[indptr[r]:indptr[r + 1]] = lst
    return indptr, indices


def _bfs_chunk(args):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import shortest_path
    indptr, indices, n, src, rows, dcol = args
    graph = csr_matrix((np.ones(len(indices), dtype=np.float32), indices, indptr), shape=(n, n))
    dist = shortest_path(graph, method='D', directed=True, unweighted=True, indices=src)
    return dist[rows, dcol]


def hop_distance(net, origins, dests, unreachable=-1, chunk=256, workers=1, native=True):
    """BFS hop count from each origin to its destination (number of transitions).
    native: libtsg_host.so's tsg_host_hop_distance (one early-exit BFS per OD pair, `workers` pthreads joined
    before it returns; 131 072 OD pairs at BJ scale in seconds where the scipy form below takes minutes);
    otherwise scipy's all-destinations BFS per distinct origin, `workers` > 1 forking a process pool (call it
    before the process has touched the GPU).  Both give the same numbers (tests/test_host.py)."""
    n = net.num_roads
    indptr, indices = csr_from_adjacent_list(net.adjacent_list, n)
    if native:
        import ctypes as C
        from . import _lib
        o32 = np.ascontiguousarray(origins, dtype=np.int32)
        d32 = np.ascontiguousarray(dests, dtype=np.int32)
        ip = np.ascontiguousarray(indptr, dtype=np.int32)
        ix = np.ascontiguousarray(indices if len(indices) else np.zeros(1), dtype=np.int32)
        res = np.zeros(len(o32), dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = _lib.host_lib().tsg_host_hop_distance(n, p(ip), p(ix), len(o32), p(o32), p(d32), p(res), max(1, int(workers)))
        if rc != 0:
            raise ValueError('tsg_host_hop_distance failed (%d): a road id outside [0, %d)?' % (rc, n))
        return np.where(res >= 0, res, unreachable).astype(np.int64)
    origins = np.asarray(origins, dtype=np.int64)
    dests = np.asarray(dests, dtype=np.int64)
    out = np.full(len(origins), unreachable, dtype=np.int64)
    uniq, inv = np.unique(origins, return_inverse=True)
    order = np.argsort(inv, kind='stable')
    bounds = np.searchsorted(inv[order], np.arange(0, len(uniq) + chunk, chunk))
    jobs, sels = [], []
    for k, c0 in enumerate(range(0, len(uniq), chunk)):
        sel = order[bounds[k]:bounds[k + 1]]
        sels.append(sel)
        jobs.append((indptr, indices, n, uniq[c0:c0 + chunk], inv[sel] - c0, dests[sel]))
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context('fork').Pool(min(workers, len(jobs))) as pool:
            results = pool.map(_bfs_chunk, jobs)
    else:
        results = [_bfs_chunk(j) for j in jobs]
    for sel, d in zip(sels, results):
        out[sel] = np.where(np.isfinite(d), d, unreachable).astype(np.int64)
    return out


def make_od_pairs(net, n_od, seed=2, default_len='hops', unreachable_len=20, workers=1, shard=None):
    """OD pairs uniform over segments (o != d), start time codes U{0..2879}.

    default_len: 'hops' = BFS hop distance + 1 (exact, slow for many ODs), or an int constant.
    shard = (rank, world): return only OD pairs rank, rank + world, ... of the n_od (the same pairs as slicing the
    full result, without computing the other ranks' hop distances).
    Returns dict of int32 arrays: start, des, start_time, default_len.
    """
    rng = np.random.RandomState(seed)
    n = net.num_roads
    o = rng.randint(0, n, size=n_od)
    d = rng.randint(0, n - 1, size=n_od)
    d = np.where(d >= o, d + 1, d)
    t = rng.randint(0, 2880, size=n_od)
    if shard is not None:
        o, d, t = o[shard[0]::shard[1]], d[shard[0]::shard[1]], t[shard[0]::shard[1]]
        n_od = len(o)
    if default_len == 'hops':
        hops = hop_distance(net, o, d, workers=workers)
        dl = np.where(hops >= 0, hops + 1, unreachable_len)
    elif default_len == 'lattice':
        dl = lattice_default_len(net, o, d)
    else:
        dl = np.full(n_od, int(default_len))
    return {'start': o.astype(np.int32), 'des': d.astype(np.int32),
            'start_time': t.astype(np.int32), 'default_len': dl.astype(np.int32)}


def region_tables(net, tiles=10):
    """Config 4: a `tiles` x `tiles` partition of the map into contiguous blocks, plus the derived
    tables DoubleLayerSearcher takes (search.py:826-828).  Formats follow
    script/construct_traffic_zone_relation.py:27-75 (region_adjacent_list:
    {region: {next_region: [border rids]}}) and script/count_region_transfer.py:32-70
    (region_transfer_freq: {region: {next_region: {'transfer_rid': [...], 'transfer_freq': [...]}}}),
    here with uniform frequencies."""
    n = net.num_roads
    grid = net.meta['grid']
    lon = np.array([net.rid_gps[str(r)][0] for r in range(n)])
    lat = np.array([net.rid_gps[str(r)][1] for r in range(n)])
    span = (grid - 1) * CELL + 1e-9
    tx = np.minimum(((lon - LON0) / span * tiles).astype(np.int64), tiles - 1)
    ty = np.minimum(((lat - LAT0) / span * tiles).astype(np.int64), tiles - 1)
    raw = np.clip(tx, 0, tiles - 1) * tiles + np.clip(ty, 0, tiles - 1)
    uniq, region_of = np.unique(raw, return_inverse=True)
    num_regions = len(uniq)
    rid2region = {str(r): int(region_of[r]) for r in range(n)}
    region2rid = {str(g): [] for g in range(num_regions)}
    for r in range(n):
        region2rid[str(int(region_of[r]))].append(r)
    region_adjacent_list = {}
    for r in range(n):
        for c in net.adjacent_list.get(str(r), []):
            a, b = int(region_of[r]), int(region_of[c])
            if a != b:
                region_adjacent_list.setdefault(str(a), {}).setdefault(str(b), [])
                if c not in region_adjacent_list[str(a)][str(b)]:
                    region_adjacent_list[str(a)][str(b)].append(c)
    region_transfer_freq = {a: {b: {'transfer_rid': list(v), 'transfer_freq': [1] * len(v)}
                                for b, v in d.items()} for a, d in region_adjacent_list.items()}
    clon = np.array([lon[region_of == g].mean() for g in range(num_regions)])
    clat = np.array([lat[region_of == g].mean() for g in range(num_regions)])
    region_dist = geo.geodesic_m(clat[:, None], clon[:, None], clat[None, :], clon[None, :])
    region_time_distribution = np.ones((24, num_regions), dtype=np.float64)
    rows, cols = [], []
    for a, d in region_adjacent_list.items():
        for b in d.keys():
            rows.append(int(a))
            cols.append(int(b))
    region_adj_mx = sp.coo_matrix((np.ones(len(rows), dtype=np.float32), (rows, cols)),
                                  shape=(num_regions + 1, num_regions + 1))
    return {'num_regions': num_regions, 'rid2region': rid2region, 'region2rid': region2rid,
            'region_adjacent_list': region_adjacent_list, 'region_transfer_freq': region_transfer_freq,
            'region_dist': region_dist, 'region_time_distribution': region_time_distribution,
            'region_adj_mx': region_adj_mx, 'region_gps': np.stack([clon, clat], axis=1)}
