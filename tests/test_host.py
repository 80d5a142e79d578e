"""CPU-only checks: substrate construction, the C-ABI library's exports, host-side layout helpers."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests._fixtures import GoldenRoadWorld

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_csr_bit_exact():
    """G8: indices[indptr[r]:indptr[r+1]] == adjacent_list[str(r)] for every r."""
    from ts_trajgen_amd.synth import csr_from_adjacent_list, street_grid_network, porto_scale_network
    w = GoldenRoadWorld()
    for adj, n in ((w.adjacent_list, w.n), (street_grid_network(8).adjacent_list, 112),
                   (porto_scale_network().adjacent_list, 11095)):
        indptr, indices = csr_from_adjacent_list(adj, n)
        assert indptr.dtype == np.int32 and indices.dtype == np.int32
        for r in range(n):
            assert indices[indptr[r]:indptr[r + 1]].tolist() == list(adj.get(str(r), []))
        assert indptr[-1] == sum(len(v) for v in adj.values())


def test_library_exports_every_declared_symbol():
    """libtsg.so loads and exports each function include/tsg.h declares (no compute call)."""
    lib_path = os.path.join(ROOT, 'ts-trajgen_amd', 'csrc', 'libtsg.so')
    if not os.path.exists(lib_path):
        pytest.fail('libtsg.so is not built: run __graft_entry__.build()')
    header = open(os.path.join(ROOT, 'include', 'tsg.h')).read()
    declared = set(re.findall(r'^int (tsg_\w+)\(', header, flags=re.M))
    assert len(declared) >= 10
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), name
    from ts_trajgen_amd import _lib
    assert set(_lib.EXPORTS) == declared
    assert lib.tsg_version() == _lib.ABI_VERSION == 2
    header_version = int(re.search(r'#define TSG_ABI_VERSION (\d+)', header).group(1))
    assert header_version == _lib.ABI_VERSION


def test_build_entry_point_runs():
    """__graft_entry__.build() is what the driver calls on the CPU every round: make is a no-op on a built tree, the
    package imports and both libraries load."""
    import __graft_entry__ as g
    g.build()


def test_host_library_exports_every_declared_symbol():
    """libtsg_host.so (plain C, no GPU) loads and exports what include/tsg_host.h declares."""
    lib_path = os.path.join(ROOT, 'ts-trajgen_amd', 'csrc', 'libtsg_host.so')
    if not os.path.exists(lib_path):
        pytest.fail('libtsg_host.so is not built: run __graft_entry__.build()')
    header = open(os.path.join(ROOT, 'include', 'tsg_host.h')).read()
    declared = set(re.findall(r'^int (tsg_host_\w+)\(', header, flags=re.M))
    assert declared == {'tsg_host_border_orders', 'tsg_host_hop_distance'}
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_header(tmp_path):
    """The ctypes mirrors are as large as the C structs of include/tsg.h (compiled here with gcc)."""
    import subprocess
    from ts_trajgen_amd import _lib
    names = [('tsg_gat_desc', _lib.GatDesc), ('tsg_gat_train', _lib.GatTrain), ('tsg_model', _lib.Model), ('tsg_lstm_step', _lib.LstmStep),
             ('tsg_score_batch', _lib.ScoreBatch), ('tsg_graph', _lib.Graph), ('tsg_od_batch', _lib.OdBatch),
             ('tsg_engine', _lib.Engine), ('tsg_rollout', _lib.Rollout)]
    src = tmp_path / 'sizes.c'
    src.write_text('#include <stdio.h>\n#include "tsg.h"\nint main(void) {\n' +
                   ''.join('  printf("%%zu\\n", sizeof(%s));\n' % n for n, _ in names) + '  return 0;\n}\n')
    exe = tmp_path / 'sizes'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    for (n, cls), sz in zip(names, sizes):
        assert ctypes.sizeof(cls) == sz, n


def test_gate_permutation_roundtrip():
    from ts_trajgen_amd.engine import gate_permutation
    for hidden in (16, 32, 128, 256):
        hp = (hidden + 63) // 64 * 64
        src = gate_permutation(hidden, hp)
        valid = src[src >= 0]
        assert sorted(valid.tolist()) == list(range(4 * hidden))
        # the four gates of unit u sit 32 columns apart inside one 128-column tile
        for u in (0, hidden - 1):
            cols = [int(np.flatnonzero(src == g * hidden + u)[0]) for g in range(4)]
            assert [c - cols[0] for c in cols] == [0, 32, 64, 96] and cols[0] // 128 == u // 32


def test_cost_table_matches_reference_expression():
    """search.py:376-381 incl. Python's banker's rounding."""
    from oracle.search import travel_cost
    w = GoldenRoadWorld()
    rtd = w.rtd.astype(int)
    n = w.n
    default = np.array([int(round(w.road_length[str(r)] / 8.334)) for r in range(n)])
    cost = np.where(rtd != 1, rtd, default[None, :])
    for hour in (0, 7, 23):
        for rid in range(0, n, 37):
            assert cost[hour, rid] == travel_cost(rtd, w.road_length, hour, rid)


def test_host_geodesic_matches_oracle():
    from oracle import geodesic
    from ts_trajgen_amd import geo
    rng = np.random.RandomState(0)
    for _ in range(200):
        a = (116.25 + rng.rand() * 0.3, 39.79 + rng.rand() * 0.3)
        b = (116.25 + rng.rand() * 0.3, 39.79 + rng.rand() * 0.3)
        assert abs(geodesic.candidate_dis(a, b) - geo.candidate_dis(a, [b])[0]) < 1e-9
    assert geo.candidate_dis((116.3, 39.9), [(116.3, 39.9)])[0] == 0.0
    # known value: one degree of latitude at the equator on WGS-84 is 110574.389 m
    assert abs(geodesic.geodesic_m(0.0, 0.0, 1.0, 0.0) - 110574.389) < 1e-2


def test_ordered_draw_equals_randomstate_choice():
    """The two-stage search's border-road order (search.py:900-902) is drawn without RandomState.choice's per-call
    overhead: same uniforms from the same generator, same result, same generator state afterwards."""
    from ts_trajgen_amd.search import ordered_draw
    rs = np.random.RandomState(0)
    for t in range(400):
        n = int(rs.randint(1, 40))
        a = rs.randint(0, 10000, size=n)
        f = np.ones(n) if t % 3 == 0 else rs.randint(1, 50, size=n).astype(float)
        p = f / np.sum(f)
        r1, r2 = np.random.RandomState(t), np.random.RandomState(t)
        for _ in range(2):
            assert np.array_equal(r1.choice(a, n, p=p, replace=False), ordered_draw(r2, a, p))
        assert r1.randint(1 << 30) == r2.randint(1 << 30)


def test_border_orders_equal_randomstate_choice():
    """The C host helper (libtsg_host.so) draws the border-road orders of many OD pairs at once: for every OD pair the
    k-th order is what the k-th RandomState(seed).choice(..., replace=False, p=...) call of the reference returns
    (search.py:897-902), element for element, for seeds up to 2^32 - 1 and lists of 1..300 roads."""
    from ts_trajgen_amd.search import border_orders
    rs = np.random.RandomState(1)
    pairs_a, pairs_p = [], []
    for t in range(60):
        n = int(rs.choice([1, 2, 3, 7, 40, 300]))
        pairs_a.append(rs.randint(0, 100000, size=n))
        f = np.ones(n) if t % 3 == 0 else rs.randint(1, 50, size=n).astype(float)
        pairs_p.append(f / np.sum(f))
    pair_ptr = np.zeros(len(pairs_a) + 1, dtype=np.int64)
    np.cumsum([len(a) for a in pairs_a], out=pair_ptr[1:])
    seeds = np.concatenate([np.arange(200), [2 ** 32 - 1, 123456789, 2 ** 31]]).astype(np.int64)
    legs = [rs.randint(0, len(pairs_a), size=int(rs.randint(0, 6))) for _ in seeds]
    leg_ptr = np.zeros(len(seeds) + 1, dtype=np.int64)
    np.cumsum([len(l) for l in legs], out=leg_ptr[1:])
    out_ptr, out_idx = border_orders(seeds, leg_ptr, np.concatenate(legs).astype(np.int32), pair_ptr,
                                     np.concatenate(pairs_p))
    for i, sd in enumerate(seeds):
        r = np.random.RandomState(int(sd))
        for k, q in enumerate(legs[i]):
            l = int(leg_ptr[i]) + k
            want = r.choice(pairs_a[q], len(pairs_a[q]), p=pairs_p[q], replace=False)
            got = pairs_a[q][out_idx[out_ptr[l]:out_ptr[l + 1]]]
            assert np.array_equal(want, got), (i, k)


def test_native_hop_distance_equals_scipy():
    """tsg_host_hop_distance (early-exit BFS per OD pair, pthreads) against scipy's BFS on the 16 x 16 golden street
    grid (dead ends, unreachable destinations) and on a thinned synthetic graph; origin == destination gives 0."""
    from ts_trajgen_amd import synth
    for net in (synth.street_grid_network(12), synth.porto_scale_network()):
        rs = np.random.RandomState(4)
        n = net.num_roads
        o = rs.randint(0, n, size=600)
        d = rs.randint(0, n, size=600)
        d[:5] = o[:5]
        want = synth.hop_distance(net, o, d, native=False)
        for workers in (1, 3):
            got = synth.hop_distance(net, o, d, workers=workers)
            assert np.array_equal(want, got)
        assert (want[:5] == 0).all()
    with pytest.raises(ValueError):
        synth.hop_distance(net, [0], [n])


def test_border_orders_rejects_what_numpy_rejects():
    """A probability vector with a zero, a negative or a NaN entry, and a seed outside [0, 2^32): numpy's
    RandomState.choice(replace=False, p) / RandomState(seed) raise ValueError; so does the C helper (its draw loop
    would otherwise never end on such a vector)."""
    from ts_trajgen_amd.search import border_orders
    leg_ptr, leg_pair, pair_ptr = np.array([0, 1]), np.array([0], dtype=np.int32), np.array([0, 3])
    for p in ([0.5, 0.0, 0.5], [0.7, -0.2, 0.5], [0.5, float('nan'), 0.5]):
        with pytest.raises(ValueError):
            np.random.RandomState(0).choice(3, 3, p=np.array(p), replace=False)
        with pytest.raises(ValueError):
            border_orders([0], leg_ptr, leg_pair, pair_ptr, np.array(p))
    for seed in (-1, 2 ** 32):
        with pytest.raises(ValueError):
            np.random.RandomState(seed)
        with pytest.raises(ValueError):
            border_orders([seed], leg_ptr, leg_pair, pair_ptr, np.array([0.2, 0.3, 0.5]))
    out_ptr, out_idx = border_orders([5], leg_ptr, leg_pair, pair_ptr, np.array([0.2, 0.3, 0.5]))
    assert sorted(out_idx[:3].tolist()) == [0, 1, 2]


class _QNode(object):
    """SearchNode's ordering (search.py:48-52): by the node's CURRENT log_prob."""
    def __init__(self, i, lp):
        self.i, self.lp = i, lp

    def __lt__(self, other):
        return self.lp[self.i] < other.lp[other.i]


def _queue_script(seed, n_ops, n_nodes, grow):
    """A random script of pushes / pops / log_prob updates with many equal priorities, and what heapq pops."""
    import heapq
    rng = np.random.RandomState(seed)
    lp = np.zeros(n_nodes, dtype=np.float32)
    objs = [_QNode(i, lp) for i in range(n_nodes)]
    heap, lines, want = [], [], []
    longest = 0
    for i in range(n_nodes):
        lp[i] = np.float32(rng.randint(0, 50)) / np.float32(8.0)
        lines.append('U %d %r' % (i, float(lp[i])))
    for _ in range(n_ops):
        r = rng.rand()
        if r < grow or not heap:
            node = int(rng.randint(0, n_nodes))
            prio = float(np.float32(rng.randint(0, 40)) / np.float32(4.0))       # few distinct values: ties
            heapq.heappush(heap, (prio, objs[node]))
            lines.append('P %r %d' % (prio, node))
            longest = max(longest, len(heap))
        elif r < grow + 0.05:
            node = int(rng.randint(0, n_nodes))
            lp[node] = np.float32(rng.randint(0, 50)) / np.float32(8.0)          # a better path found (search.py:405)
            lines.append('U %d %r' % (node, float(lp[node])))
        else:
            want.append(heapq.heappop(heap)[1].i)
            lines.append('O')
    while heap:
        want.append(heapq.heappop(heap)[1].i)
        lines.append('O')
    return '\n'.join(lines) + '\n', want, longest


@pytest.mark.parametrize('blocked,lin_base,n_ops,grow', [
    (4095, 8 * (585 + 2) + 8, 30000, 0.62),       # the shipped layout: the queue passes 4095 entries into the plain part
    (63, 8 * (9 + 2) + 8, 6000, 0.55),            # two block levels only: the boundary and the plain part in every sift
    (7, 8 * (1 + 2) + 8, 3000, 0.55),             # one block
])
def test_blocked_heap_equals_heapq(tmp_path, blocked, lin_base, n_ops, grow):
    """The queue of the frontier kernel (csrc/tsg_heap.h: blocked layout, several levels per round trip) pops what
    CPython's heapq pops, under the tuple order (priority, SearchNode) with ties broken by the CURRENT log_prob."""
    import subprocess
    exe = tmp_path / 'heap_harness'
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-DTSG_HEAP_BLOCKED=%d' % blocked,
                           '-DTSG_HEAP_LIN_BASE=%d' % lin_base, '-I', os.path.join(ROOT, 'ts-trajgen_amd', 'csrc'),
                           os.path.join(ROOT, 'tests', 'heap_harness.cpp'), '-o', str(exe)])
    for seed in (1, 2):
        script, want, longest = _queue_script(seed, n_ops, 512, grow)
        assert longest > 2 * blocked                  # the queue did grow past the blocked levels
        out = subprocess.run([str(exe), str(1 << 16), '512'], input=script, capture_output=True, text=True, check=True)
        got = out.stdout.split('\n')
        assert got[-2] == 'n 0'
        assert [int(x) for x in got[:-2]] == want


def test_plan_engine_sizes():
    """Sizes Searcher.generate picks for a large batch (search.plan_engine): the bench's arithmetic."""
    from ts_trajgen_amd.search import plan_engine
    page, pps, fixed = 279040, 80, 1003140
    slots, pool, ramp = plan_engine(306e9, fixed, page, pps, 131072, 2080)
    assert slots == 24576 and ramp == 1500                      # the cap binds, the pool is below the worst case
    assert slots * fixed + pool * page + 131072 * 2080 <= 0.93 * 306e9 < slots * fixed + (pool + 1) * page + 131072 * 2080
    s2, p2, r2 = plan_engine(306e9, fixed, page, pps, 3000, 2080)
    assert s2 == 2048 and p2 == s2 * pps and r2 == 0            # fewer OD pairs than places: worst case reserved
    s3, p3, r3 = plan_engine(40e9, fixed, page, pps, 10 ** 6, 2080)
    assert 1024 <= s3 < 24576 and s3 % 1024 == 0 and p3 < s3 * pps and r3 == 1500
    assert s3 * fixed + p3 * page <= 0.93 * 40e9 - 10 ** 6 * 2080
    s4, p4, r4 = plan_engine(1e9, fixed, page, pps, 10 ** 6, 2080)   # next to nothing free: one whole search at least
    assert s4 >= 1 and p4 >= pps
