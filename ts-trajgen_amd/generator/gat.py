"""GATLayerImp3 drop-in (reference: generator/gat.py:13-68 constructor / parameters, :138-200 forward).

Parameters keep the reference's names and shapes (`linear_proj.weight`, `skip_proj.weight`,
`scoring_fn_source`, `scoring_fn_target`, `bias`) and are created in the reference's order so that the
same `torch.manual_seed` reproduces the same initial weights.  After `.train(False)` (the generation path: the
reference's dropouts are the identity) `forward` runs `tsg_gat_forward`; in training mode it runs
`tsg_gat_train_forward` under autograd (`tsg_gat_train_backward` + `tsg_gemm_f32` for the dense gradients), the three
dropouts of gat.py:149,155,180 drawn on the device from torch's generator (or taken from `dropout_masks`).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from .. import _lib


def csr_by_target(edge_index, n_nodes):
    """(2, E) COO [source; target] -> (in_ptr[N+1], in_src[E]) int32, incoming edges of each target in
    their original COO order (stable), which keeps the reference's scatter_add_ summation order."""
    ei = edge_index.detach().cpu().numpy() if torch.is_tensor(edge_index) else np.asarray(edge_index)
    src, trg = ei[0].astype(np.int64), ei[1].astype(np.int64)
    if len(src) and (src.min() < 0 or trg.min() < 0 or src.max() >= n_nodes or trg.max() >= n_nodes):
        raise ValueError('edge index out of range for %d nodes' % n_nodes)
    order = np.argsort(trg, kind='stable')
    in_ptr = np.zeros(n_nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(trg, minlength=n_nodes), out=in_ptr[1:])
    return in_ptr, src[order].astype(np.int32)


def train_graph(edge_index, n_nodes):
    """The graph as the training kernels want it (include/tsg.h, tsg_gat_train): edge slots = edges sorted stably by
    target; returns numpy (in_ptr, in_src, out_ptr, out_slot, slot_trg, order) with order[j] = COO index of slot j."""
    ei = edge_index.detach().cpu().numpy() if torch.is_tensor(edge_index) else np.asarray(edge_index)
    src, trg = ei[0].astype(np.int64), ei[1].astype(np.int64)
    in_ptr, in_src = csr_by_target(edge_index, n_nodes)
    order = np.argsort(trg, kind='stable')
    slot_trg = trg[order].astype(np.int32)
    out_slot = np.argsort(in_src, kind='stable').astype(np.int32)
    out_ptr = np.zeros(n_nodes + 1, dtype=np.int32)
    np.cumsum(np.bincount(in_src, minlength=n_nodes), out=out_ptr[1:])
    return in_ptr, in_src, out_ptr, out_slot, slot_trg, order


class _GATTrainFn(torch.autograd.Function):
    """GATLayerImp3.forward in training mode (gat.py:138-200) as one autograd node on the HIP kernels."""

    @staticmethod
    def forward(ctx, x, w_proj, w_skip, a_src, a_trg, bias, cfg):
        from ..engine import gemm  # noqa: F401  (the dense gradients in backward)
        L = _lib.lib()
        dev = x.device
        n, fin = x.shape
        nh, fo, concat = cfg['heads'], cfg['fout'], cfg['concat']
        e = int(cfg['in_src'].numel())
        f32 = torch.float32
        x = x.detach().contiguous().float()
        bufs = {
            'x_d': torch.empty((n, fin), dtype=f32, device=dev), 'p_d': torch.empty((n, nh * fo), dtype=f32, device=dev),
            's_src': torch.empty((n, nh), dtype=f32, device=dev), 's_trg': torch.empty((n, nh), dtype=f32, device=dev),
            'att': torch.empty((max(e, 1), nh), dtype=f32, device=dev),
            'att_d': torch.empty((max(e, 1), nh), dtype=f32, device=dev),
            'gmax': torch.empty(1, dtype=f32, device=dev),
            'out': torch.empty((n, nh * fo if concat else fo), dtype=f32, device=dev),
            'w_proj_t': w_proj.detach().t().contiguous().float(),
            'a_src': a_src.detach().reshape(nh, fo).contiguous().float(),
            'a_trg': a_trg.detach().reshape(nh, fo).contiguous().float(),
            'bias': bias.detach().contiguous().float() if bias is not None else None,
        }
        identity = fin == fo
        if not identity:
            bufs['w_skip_t'] = w_skip.detach().t().contiguous().float()
            bufs['skip'] = torch.empty((n, nh * fo), dtype=f32, device=dev)
        d = _lib.GatTrain()
        d.n_nodes, d.n_edges, d.fin, d.heads, d.fout = n, e, fin, nh, fo
        d.concat, d.skip_identity = int(bool(concat)), int(identity)
        for k in ('in_ptr', 'in_src', 'out_ptr', 'out_slot', 'slot_trg', 'm_in', 'm_proj', 'm_att'):
            setattr(d, k, _lib.ptr(cfg[k]))
        d.x = _lib.ptr(x)
        for k, v in bufs.items():
            setattr(d, k, _lib.ptr(v))
        d.in_bytes = n * fin * 4
        d.node_feat_bytes = n * nh * fo * 4
        d.score_bytes = n * nh * 4
        d.edge_bytes = max(e, 1) * nh * 4
        d.out_bytes = bufs['out'].numel() * 4
        with torch.cuda.device(dev):
            _lib.check(L.tsg_gat_train_forward(C.byref(d), _lib.current_stream()), 'tsg_gat_train_forward')
        ctx.cfg, ctx.bufs, ctx.x, ctx.w = cfg, bufs, x, (w_proj.detach().contiguous().float(),
                                                      None if identity else w_skip.detach().contiguous().float())
        ctx.desc = d
        ctx.has_bias = bias is not None
        return bufs['out']

    @staticmethod
    def backward(ctx, d_out):
        from ..engine import gemm
        L = _lib.lib()
        cfg, b, d = ctx.cfg, ctx.bufs, ctx.desc
        dev = d_out.device
        n, fin, nh, fo = d.n_nodes, d.fin, d.heads, d.fout
        e = max(d.n_edges, 1)
        f32 = torch.float32
        d_out = d_out.contiguous().float()
        g = {'d_o': torch.empty((n, nh * fo), dtype=f32, device=dev), 'd_p': torch.empty((n, nh * fo), dtype=f32, device=dev),
             'd_pre': torch.zeros((e, nh), dtype=f32, device=dev), 'ds_src': torch.empty((n, nh), dtype=f32, device=dev),
             'ds_trg': torch.empty((n, nh), dtype=f32, device=dev), 'd_a_src': torch.empty((nh, fo), dtype=f32, device=dev),
             'd_a_trg': torch.empty((nh, fo), dtype=f32, device=dev)}
        if ctx.has_bias:
            g['d_bias'] = torch.empty_like(b['bias'])
        d.d_out = _lib.ptr(d_out)
        for k, v in g.items():
            setattr(d, k, _lib.ptr(v))
        with torch.cuda.device(dev):
            _lib.check(L.tsg_gat_train_backward(C.byref(d), _lib.current_stream()), 'tsg_gat_train_backward')
        w_proj, w_skip = ctx.w
        x_d = b['x_d']
        d_w_proj = gemm(g['d_p'].t().contiguous(), x_d)             # [NH*F, N] . [N, FIN]
        d_x_d = gemm(g['d_p'], w_proj)                              # [N, NH*F] . [NH*F, FIN]
        d_w_skip = None
        if w_skip is None:                                          # FIN == F: the input itself was added to every head
            d_x_d = d_x_d + g['d_o'].view(n, nh, fo).sum(dim=1)
        else:
            d_w_skip = gemm(g['d_o'].t().contiguous(), x_d)
            d_x_d = d_x_d + gemm(g['d_o'], w_skip)
        d_x = d_x_d * cfg['m_in'] if cfg['m_in'] is not None else d_x_d
        return (d_x, d_w_proj, d_w_skip, g['d_a_src'].view(1, nh, fo), g['d_a_trg'].view(1, nh, fo),
                g.get('d_bias'), None)


class GATLayerImp3(nn.Module):
    def __init__(self, num_in_features, num_out_features, num_of_heads, device, concat=True, activation=nn.ELU(),
                 dropout_prob=0.6, add_skip_connection=True, bias=True, log_attention_weights=False):
        super().__init__()
        if not isinstance(activation, nn.ELU) or not add_skip_connection or log_attention_weights:
            raise NotImplementedError('the HIP encoder implements the configuration the path uses: '
                                      'ELU activation, skip connection, no attention logging')
        self.num_of_heads = num_of_heads
        self.num_out_features = num_out_features
        self.num_in_features = num_in_features
        self.concat = concat
        self.device = device
        self.linear_proj = nn.Linear(num_in_features, num_of_heads * num_out_features, bias=False)
        self.scoring_fn_target = nn.Parameter(torch.Tensor(1, num_of_heads, num_out_features))
        self.scoring_fn_source = nn.Parameter(torch.Tensor(1, num_of_heads, num_out_features))
        if bias:
            self.bias = nn.Parameter(torch.Tensor(num_of_heads * num_out_features if concat else num_out_features))
        else:
            self.register_parameter('bias', None)
        self.skip_proj = nn.Linear(num_in_features, num_of_heads * num_out_features, bias=False)
        self.dropout_prob = dropout_prob
        self._csr_key = None
        self._csr = None
        self._tg_key = None
        self._tg = None
        self.dropout_masks = None   # training mode: (m_in (N, FIN), m_proj (N, NH*F), m_att (E, NH) in COO edge order),
        #                             each scaled by 1 / (1 - p); None = drawn from torch's generator on the device
        nn.init.xavier_uniform_(self.linear_proj.weight)
        nn.init.xavier_uniform_(self.scoring_fn_target)
        nn.init.xavier_uniform_(self.scoring_fn_source)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def _graph(self, edge_index, n_nodes, dev):
        key = (edge_index.data_ptr(), tuple(edge_index.shape), n_nodes, str(dev))
        if self._csr_key != key:
            in_ptr, in_src = csr_by_target(edge_index, n_nodes)
            self._csr = (torch.from_numpy(in_ptr).to(dev), torch.from_numpy(in_src).to(dev))
            self._csr_key = key
        return self._csr

    def _transposed_weights(self):
        """[FIN][NH*F] copies of the two projection weights, rebuilt only when a weight has changed (in-place
        updates bump `_version`; a re-assigned parameter has another data_ptr)."""
        ws = (self.linear_proj.weight, self.skip_proj.weight)
        key = tuple((w.data_ptr(), w._version, str(w.device)) for w in ws)
        if getattr(self, '_wt_key', None) != key:
            self._wt = tuple(w.detach().t().contiguous().float() for w in ws)
            self._wt_key = key
        return self._wt

    def _train_graph(self, edge_index, n_nodes, dev):
        key = (edge_index.data_ptr(), tuple(edge_index.shape), n_nodes, str(dev))
        if self._tg_key != key:
            parts = train_graph(edge_index, n_nodes)
            self._tg = [torch.from_numpy(np.ascontiguousarray(p)).to(dev) for p in parts[:5]] + [
                torch.from_numpy(parts[5]).to(dev)]
            self._tg_key = key
        return self._tg

    def _forward_train(self, data):
        x, edge_index = data
        if edge_index.shape[0] != 2:
            raise AssertionError('Expected edge index with shape=(2,E) got %s' % (tuple(edge_index.shape),))
        if not x.is_cuda:
            raise RuntimeError('GATLayerImp3.forward needs CUDA/HIP tensors (no CPU fallback)')
        dev = x.device
        n, fin = x.shape
        nh, fo = self.num_of_heads, self.num_out_features
        in_ptr, in_src, out_ptr, out_slot, slot_trg, order = self._train_graph(edge_index, n, dev)
        e = int(in_src.numel())
        if self.dropout_masks is not None:
            m_in, m_proj, m_att = [m.to(dev).float().contiguous() for m in self.dropout_masks]
        else:
            keep = 1.0 - self.dropout_prob
            m_in = (torch.rand((n, fin), device=dev) < keep).float() / keep
            m_proj = (torch.rand((n, nh * fo), device=dev) < keep).float() / keep
            m_att = (torch.rand((e, nh), device=dev) < keep).float() / keep
        cfg = {'heads': nh, 'fout': fo, 'concat': bool(self.concat), 'in_ptr': in_ptr, 'in_src': in_src,
               'out_ptr': out_ptr, 'out_slot': out_slot, 'slot_trg': slot_trg,
               'm_in': m_in.reshape(n, fin), 'm_proj': m_proj.reshape(n, nh * fo),
               'm_att': m_att.reshape(e, nh)[order].contiguous()}
        out = _GATTrainFn.apply(x, self.linear_proj.weight, self.skip_proj.weight, self.scoring_fn_source,
                                self.scoring_fn_target, self.bias, cfg)
        return out, edge_index

    def forward(self, data):
        if self.training:
            return self._forward_train(data)
        with torch.no_grad():
            return self._forward_eval(data)

    def _forward_eval(self, data):
        x, edge_index = data
        if edge_index.shape[0] != 2:
            raise AssertionError('Expected edge index with shape=(2,E) got %s' % (tuple(edge_index.shape),))
        if not x.is_cuda:
            raise RuntimeError('GATLayerImp3.forward needs CUDA/HIP tensors (no CPU fallback)')
        L = _lib.lib()
        dev = x.device
        x = x.contiguous().float()
        n, fin = x.shape
        nh, fo = self.num_of_heads, self.num_out_features
        in_ptr, in_src = self._graph(edge_index, n, dev)
        w_proj_t, w_skip_t = self._transposed_weights()
        a_src = self.scoring_fn_source.detach().reshape(nh, fo).contiguous()
        a_trg = self.scoring_fn_target.detach().reshape(nh, fo).contiguous()
        proj = torch.empty((n, nh * fo), dtype=torch.float32, device=dev)
        s_src = torch.empty((n, nh), dtype=torch.float32, device=dev)
        s_trg = torch.empty((n, nh), dtype=torch.float32, device=dev)
        gmax = torch.empty(1, dtype=torch.float32, device=dev)
        out = torch.empty((n, nh * fo if self.concat else fo), dtype=torch.float32, device=dev)
        d = _lib.GatDesc()
        d.n_nodes, d.n_edges, d.fin, d.heads, d.fout = n, int(in_src.numel()), fin, nh, fo
        d.concat = int(bool(self.concat))
        d.skip_identity = int(fin == fo)
        d.x, d.w_proj_t, d.w_skip_t = _lib.ptr(x), _lib.ptr(w_proj_t), _lib.ptr(w_skip_t)
        d.a_src, d.a_trg = _lib.ptr(a_src), _lib.ptr(a_trg)
        d.bias = _lib.ptr(self.bias.detach().contiguous()) if self.bias is not None else None
        d.in_ptr, d.in_src = _lib.ptr(in_ptr), _lib.ptr(in_src)
        d.proj, d.s_src, d.s_trg, d.gmax, d.out = (_lib.ptr(proj), _lib.ptr(s_src), _lib.ptr(s_trg),
                                                    _lib.ptr(gmax), _lib.ptr(out))
        if fin != fo:       # skip projection workspaces (x . Ws^T + bias on the matrix cores)
            skip = torch.empty_like(out)
            d.skip, d.skip_bytes = _lib.ptr(skip), skip.numel() * 4
            if not self.concat:
                w_mean = torch.empty((fin, fo), dtype=torch.float32, device=dev)
                d.w_skip_mean, d.w_skip_mean_bytes = _lib.ptr(w_mean), w_mean.numel() * 4
        w_att = torch.empty((fin, 2 * nh), dtype=torch.float32, device=dev)
        d.w_att, d.w_att_bytes = _lib.ptr(w_att), w_att.numel() * 4
        d.proj_bytes = proj.numel() * 4
        d.s_bytes = s_src.numel() * 4
        d.out_bytes = out.numel() * 4
        with torch.cuda.device(dev):
            _lib.check(L.tsg_gat_forward(C.byref(d), _lib.current_stream()), 'tsg_gat_forward')
        return out, edge_index
