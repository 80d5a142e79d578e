"""GeneratorV4 drop-in (reference: generator/generator_v4.py:16-29 constructor, :31-72 forward/predict).

`predict` keeps the reference signature and returns softmax((1-w1)*softmax(G) + w1*softmax(H)) over the
candidates, computed by libtsg.so (incremental LSTM cell on MFMA, attention, gathered out_fc rows,
factored H).  Device tables are (re)built lazily and invalidated by `load_state_dict` / `.to()`.
"""
import weakref

import torch
import torch.nn as nn

from .distance_gat_fc import DistanceGatFC
from .function_g_fc import FunctionGFC


OWNERS = weakref.WeakValueDictionary()


def owner_of(half, attr):
    g = OWNERS.get(id(half))
    return g if g is not None and getattr(g, attr, None) is half else None


class GeneratorV4(nn.Module):
    def __init__(self, config, data_feature):
        super().__init__()
        self.dis_weight = config['dis_weight']
        self.w1 = nn.Parameter(torch.tensor([self.dis_weight]))
        self.function_g = FunctionGFC(config['function_g'], data_feature)
        self.function_h = DistanceGatFC(config['function_h'], data_feature)
        self._tables = None
        self._tables_key = None
        self._adopt()

    def _adopt(self):
        """predict_g / predict_next of the two halves run through the device tables of the whole model: the halves
        find it here (weak values, keyed by the half's id: no module cycle, nothing extra to pickle)."""
        OWNERS[id(self.function_g)] = self
        OWNERS[id(self.function_h)] = self

    # -- device tables ---------------------------------------------------------------------------
    def _key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def tables(self):
        """ModelTables for the current parameters (rebuilt when any parameter changed or moved)."""
        from ..engine import ModelTables
        if self.training:
            raise RuntimeError('generation is inference-only: call .train(False) first '
                               '(our_model_generate.py:399)')
        self._adopt()
        key = self._key()
        if self._tables is None or self._tables_key != key:
            self.function_h.node_emb_feature = None
            self._tables = ModelTables(self)
            self._tables_key = key
        return self._tables

    # -- reference API ---------------------------------------------------------------------------
    def predict_parts(self, trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask=None, logits=False):
        """(p, g, h): the mixed probability and the two inner softmaxes (predict_g / predict_next); with `logits`
        also the candidate scores of G and H before their softmax (FunctionGFC.forward / DistanceGatFC.forward)."""
        return self.tables().predict(trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask, logits)

    def predict(self, trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask=None, cache=True):
        if not cache:
            raise NotImplementedError()          # as the reference (generator_v4.py:49-50)
        return self.predict_parts(trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask)[0]

    def forward(self, trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask=None, cache=True):
        """(1-w1)*g + w1*h before the outer softmax (generator_v4.py:54)."""
        if not cache:
            raise NotImplementedError()
        _, g, h = self.predict_parts(trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask)[:3]
        w1 = self.w1.detach()
        return torch.add((1 - w1) * g, w1 * h)
