"""ctypes binding of csrc/libtsg.so (the C ABI declared in include/tsg.h).

There is deliberately no fallback: if the library is missing or a symbol is absent, importing a
compute entry point raises.  Structures mirror include/tsg.h field for field.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TSG_LIB_OVERRIDE') or os.path.join(_HERE, 'csrc', 'libtsg.so')  # override: ablation builds only

TSG_MAX_LAYERS = 4
TSG_MAX_DEGREE = 64
TSG_H_HIDDEN = 10
STATUS_UNTOUCHED, STATUS_REACHED, STATUS_FALLBACK, STATUS_OVERFLOW = 0, 1, 2, 0x100
GAT_STATIC_SCRATCH_BYTES = 8 << 20     # TSG_GAT_STATIC_SCRATCH_BYTES

ABI_VERSION = 2         # TSG_ABI_VERSION of include/tsg.h
EXPORTS = ('tsg_version', 'tsg_gemm_f32', 'tsg_gat_forward', 'tsg_gat_forward_static', 'tsg_gat_train_forward', 'tsg_gat_train_backward', 'tsg_gat_build_input', 'tsg_lstm_step_batched',
           'tsg_score_candidates', 'tsg_engine_reset', 'tsg_engine_run', 'tsg_engine_run_profiled', 'tsg_engine_graph_create', 'tsg_engine_graph_launch',
           'tsg_engine_graph_destroy', 'tsg_frontier_expand_select',
           'tsg_score_frontier', 'tsg_rollout_run', 'tsg_reconstruct_paths', 'tsg_h_pair_scores', 'tsg_geodesic_pairs')

_p = C.c_void_p
_i32 = C.c_int32
_sz = C.c_size_t


class GatDesc(C.Structure):
    _fields_ = [('n_nodes', _i32), ('n_edges', _i32), ('fin', _i32), ('heads', _i32), ('fout', _i32),
                ('concat', _i32), ('skip_identity', _i32), ('reserved', _i32),
                ('x', _p), ('w_proj_t', _p), ('w_skip_t', _p), ('a_src', _p), ('a_trg', _p), ('bias', _p),
                ('in_ptr', _p), ('in_src', _p), ('proj', _p), ('s_src', _p), ('s_trg', _p), ('gmax', _p),
                ('out', _p), ('proj_bytes', _sz), ('s_bytes', _sz), ('out_bytes', _sz),
                ('skip', _p), ('w_skip_mean', _p), ('skip_bytes', _sz), ('w_skip_mean_bytes', _sz),
                ('w_att', _p), ('w_att_bytes', _sz)]


class GatTrain(C.Structure):
    _fields_ = [('n_nodes', _i32), ('n_edges', _i32), ('fin', _i32), ('heads', _i32), ('fout', _i32),
                ('concat', _i32), ('skip_identity', _i32), ('reserved', _i32),
                ('in_ptr', _p), ('in_src', _p), ('out_ptr', _p), ('out_slot', _p), ('slot_trg', _p),
                ('x', _p), ('m_in', _p), ('m_proj', _p), ('m_att', _p), ('w_proj_t', _p), ('w_skip_t', _p),
                ('a_src', _p), ('a_trg', _p), ('bias', _p),
                ('x_d', _p), ('p_d', _p), ('skip', _p), ('s_src', _p), ('s_trg', _p), ('att', _p), ('att_d', _p),
                ('gmax', _p), ('out', _p), ('d_out', _p), ('d_o', _p), ('d_p', _p), ('d_pre', _p), ('ds_src', _p),
                ('ds_trg', _p), ('d_a_src', _p), ('d_a_trg', _p), ('d_bias', _p),
                ('in_bytes', _sz), ('node_feat_bytes', _sz), ('score_bytes', _sz), ('edge_bytes', _sz),
                ('out_bytes', _sz)]


class Model(C.Structure):
    _fields_ = [('n_roads', _i32), ('n_time', _i32), ('hp', _i32), ('n_layers', _i32),
                ('xproj', _p), ('tproj', _p), ('whh0_t', _p),
                ('wl_t', _p * TSG_MAX_LAYERS), ('bl', _p * TSG_MAX_LAYERS),
                ('out_w', _p), ('out_b', _p), ('pfrom', _p), ('pto', _p), ('hparams', _p),
                ('whh0_s', _p), ('wl_s', _p * TSG_MAX_LAYERS)]


class LstmStep(C.Structure):
    _fields_ = [('layer', _i32), ('n_rows_host', _i32), ('n_rows_dev', _p), ('prev_idx', _p), ('out_idx', _p),
                ('rid', _p), ('tcode', _p), ('h_below', _p), ('h', _p), ('c', _p), ('state_rows', _sz)]


class ScoreBatch(C.Structure):
    _fields_ = [('n_rows', _i32), ('chain_ptr', _p), ('chain', _p), ('h_top', _p), ('cand_ptr', _p),
                ('cand', _p), ('dis', _p), ('des', _p), ('prob', _p), ('prob_g', _p), ('prob_h', _p),
                ('logit_g', _p), ('logit_h', _p)]


class Graph(C.Structure):
    _fields_ = [('n_roads', _i32), ('max_degree', _i32), ('indptr', _p), ('indices', _p), ('geo', _p),
                ('cost', _p), ('label', _p), ('dist_table', _p), ('pair_cost', _p), ('empty_key', _p)]


class OdBatch(C.Structure):
    _fields_ = [('n_od', _i32), ('lmax', _i32), ('prefix_ptr', _p), ('prefix_rid', _p), ('prefix_time', _p),
                ('des', _p), ('default_len', _p), ('max_step', _p), ('allow_label', _p),
                ('out_rid', _p), ('out_time', _p), ('out_len', _p), ('out_status', _p), ('out_steps', _p),
                ('poplog_cap', _i32), ('poplog_rid', _p), ('poplog_prio', _p), ('poplog_n', _p)]


class Engine(C.Structure):
    _fields_ = [('slots', _i32), ('exp_cap', _i32), ('node_cap', _i32), ('heap_cap', _i32),
                ('slot_state', _p), ('slot_state_bytes', _sz), ('heap', _p), ('heap_bytes', _sz),
                ('table', _p), ('table_bytes', _sz), ('nodes', _p), ('nodes_bytes', _sz),
                ('exp_meta', _p), ('exp_meta_bytes', _sz), ('exp_up', _p), ('exp_up_bytes', _sz),
                ('up_levels', _i32), ('prune_spread', C.c_float), ('state_format', _i32), ('max_pops', _i32),
                ('h', _p * TSG_MAX_LAYERS), ('c', _p * TSG_MAX_LAYERS), ('state_bytes', _sz),
                ('n_pages', _i32), ('pages_per_slot', _i32), ('admit_reserve', _i32), ('ramp_steps', _i32),
                ('slot_pages', _p), ('slot_pages_bytes', _sz), ('page_ring', _p), ('page_ring_bytes', _sz),
                ('retry', _p), ('retry_bytes', _sz), ('page_flags', _p), ('page_flags_bytes', _sz),
                ('items', _p), ('items_bytes', _sz), ('item_counts', _p), ('item_counts_bytes', _sz), ('cand_lp', _p), ('cand_lp_bytes', _sz),
                ('cand_id', _p), ('cand_id_bytes', _sz), ('row_meta', _p), ('row_meta_bytes', _sz),
                ('shared_counters', _p), ('counters', _p), ('counters_bytes', _sz)]


class Rollout(C.Structure):
    _fields_ = [('n_traj', _i32), ('lmax', _i32), ('n_draws', _i32), ('reserved', _i32),
                ('out_rid', _p), ('out_time', _p), ('len', _p), ('n_state', _p), ('n_drawn', _p), ('status', _p),
                ('des', _p), ('default_len', _p), ('start_code', _p), ('uniform', _p), ('prob_log', _p),
                ('h', _p * TSG_MAX_LAYERS), ('c', _p * TSG_MAX_LAYERS), ('state_bytes', _sz),
                ('prev_idx', _p), ('out_idx', _p), ('rid', _p), ('tcode', _p)]


_lib = None


def lib():
    """Load libtsg.so once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libtsg.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` '
                           'or `make -C ts-trajgen_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    # torch ships its own libamdhip64 (soname libamdhip64.so.7).  Loaded first, it satisfies libtsg's NEEDED entry
    # and both share one HIP runtime; loaded second, the process ends up with two runtimes and every launch
    # from libtsg fails with hipErrorNoDevice.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise RuntimeError('libtsg.so lacks symbol %s (stale build?)' % name)
    L.tsg_version.restype = C.c_int
    if L.tsg_version() != ABI_VERSION:      # struct layouts of this file are those of include/tsg.h at that version
        raise RuntimeError('libtsg.so has ABI version %d, this binding expects %d (stale build?)'
                           % (L.tsg_version(), ABI_VERSION))
    L.tsg_gemm_f32.argtypes = [_p, C.c_int, _p, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]
    L.tsg_gat_forward.argtypes = [C.POINTER(GatDesc), _p]
    L.tsg_gat_forward_static.argtypes = [C.POINTER(GatDesc), _p]
    L.tsg_gat_train_forward.argtypes = [C.POINTER(GatTrain), _p]
    L.tsg_gat_train_backward.argtypes = [C.POINTER(GatTrain), _p]
    L.tsg_gat_build_input.argtypes = [_p, C.c_int, C.c_int, _p, C.c_int, _p, C.c_int, C.c_int, _p, _p]
    L.tsg_lstm_step_batched.argtypes = [C.POINTER(Model), C.POINTER(LstmStep), _p]
    L.tsg_score_candidates.argtypes = [C.POINTER(Model), C.POINTER(ScoreBatch), _p]
    L.tsg_engine_reset.argtypes = [C.POINTER(Engine), C.POINTER(Graph), _p]
    L.tsg_engine_run.argtypes = [C.POINTER(Model), C.POINTER(Graph), C.POINTER(OdBatch), C.POINTER(Engine),
                                 _i32, _i32, _p]
    L.tsg_engine_run_profiled.argtypes = [C.POINTER(Model), C.POINTER(Graph), C.POINTER(OdBatch), C.POINTER(Engine),
                                          _i32, _i32, _p, C.POINTER(C.c_float), C.POINTER(_i32)]
    L.tsg_engine_graph_create.argtypes = [C.POINTER(Model), C.POINTER(Graph), C.POINTER(OdBatch), C.POINTER(Engine),
                                          _i32, _p, C.POINTER(_p)]
    L.tsg_engine_graph_launch.argtypes = [_p, _p]
    L.tsg_engine_graph_destroy.argtypes = [_p]
    L.tsg_frontier_expand_select.argtypes = [C.POINTER(Graph), C.POINTER(OdBatch), C.POINTER(Engine), _i32, _p]
    L.tsg_score_frontier.argtypes = [C.POINTER(Model), C.POINTER(Graph), C.POINTER(Engine), _i32, _p]
    L.tsg_reconstruct_paths.argtypes = [C.POINTER(Engine), _i32, _p, _p, _p, _p, _i32, _p, _p, _p, _p]
    L.tsg_h_pair_scores.argtypes = [C.POINTER(Model), _i32, _p, _p, _p, _p, _p]
    L.tsg_geodesic_pairs.argtypes = [_p, _i32, _i32, _p, _p, _p, _p]
    L.tsg_rollout_run.argtypes = [C.POINTER(Model), C.POINTER(Graph), C.POINTER(Rollout), _i32, _p]
    for name in EXPORTS:
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


_host = None


def host_lib():
    """libtsg_host.so: the plain-C host helper (exact numpy RandomState draws for the two-stage search)."""
    global _host
    if _host is None:
        path = os.path.join(_HERE, 'csrc', 'libtsg_host.so')
        if not os.path.exists(path):
            raise RuntimeError('libtsg_host.so is not built (%s): run `make -C ts-trajgen_amd/csrc`' % path)
        H = C.CDLL(path)
        H.tsg_host_border_orders.argtypes = [_i32, _p, _p, _p, _p, _p, _p, _p, _i32]
        H.tsg_host_border_orders.restype = C.c_int
        H.tsg_host_hop_distance.argtypes = [_i32, _p, _p, C.c_int64, _p, _p, _p, _i32]
        H.tsg_host_hop_distance.restype = C.c_int
        _host = H
    return _host


def check(rc, what):
    if rc != 0:
        names = {-1: 'TSG_E_INVALID', -2: 'TSG_E_TOO_SMALL', -3: 'TSG_E_LIMIT'}
        msg = names.get(rc, 'hipError %d' % (-(rc + 1000)) if rc <= -1000 else str(rc))
        raise RuntimeError('%s failed: %s' % (what, msg))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
