"""GATLayerImp3 drop-in (reference: generator/gat.py:13-68 constructor / parameters, :138-200 forward).

Parameters keep the reference's names and shapes (`linear_proj.weight`, `skip_proj.weight`,
`scoring_fn_source`, `scoring_fn_target`, `bias`) and are created in the reference's order so that the
same `torch.manual_seed` reproduces the same initial weights.  `forward` is inference-only (the
reference's dropouts are the identity after `.train(False)`) and runs `tsg_gat_forward`.
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

    @torch.no_grad()
    def forward(self, data):
        x, edge_index = data
        if self.training:
            raise RuntimeError('the HIP GAT encoder is inference-only: call .train(False) first '
                               '(our_model_generate.py:399)')
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
        w_proj_t = self.linear_proj.weight.detach().t().contiguous()
        w_skip_t = self.skip_proj.weight.detach().t().contiguous()
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
