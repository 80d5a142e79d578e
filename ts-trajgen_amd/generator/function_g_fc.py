"""FunctionGFC drop-in (function G; reference: generator/function_g_fc.py:105-136 constructor).

Holds the reference's parameters under the reference's names (`road_emb`, `time_emb`, `lstm`,
`intra_attn.w1/w2/w3` -- unused by the reference too, function_g_fc.py:17-19 -- and `out_fc`).  The
arithmetic (incremental LSTM cell on MFMA, attention over the prefix, gathered out_fc rows) lives in
libtsg.so and is driven by GeneratorV4 / the engine, which need G and H together; `forward` / `predict` /
`predict_g` (function_g_fc.py:138-221) return G's share of that evaluation."""
import torch.nn as nn


class IntraAttention(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.w1 = nn.Linear(in_features=hidden_size, out_features=1, bias=False)
        self.w2 = nn.Linear(in_features=hidden_size, out_features=hidden_size, bias=False)
        self.w3 = nn.Linear(in_features=hidden_size, out_features=hidden_size, bias=False)


class FunctionGFC(nn.Module):
    def __init__(self, config, data_feature):
        super().__init__()
        self.road_emb_size = config['road_emb_size']
        self.time_emb_size = config['time_emb_size']
        self.hidden_size = config['hidden_size']
        self.lstm_layer_num = config['lstm_layer_num']
        self.dropout_p = config['dropout_p']
        self.dis_weight = config['dis_weight']
        self.input_size = self.road_emb_size + self.time_emb_size
        self.road_emb = nn.Embedding(num_embeddings=data_feature['road_num'], embedding_dim=self.road_emb_size,
                                     padding_idx=data_feature['road_pad'])
        self.time_emb = nn.Embedding(num_embeddings=data_feature['time_size'], embedding_dim=self.time_emb_size,
                                     padding_idx=data_feature['time_pad'])
        self.lstm = nn.LSTM(input_size=self.input_size, hidden_size=self.hidden_size,
                            num_layers=self.lstm_layer_num, batch_first=True)
        self.intra_attn = IntraAttention(hidden_size=self.hidden_size)
        self.dropout = nn.Dropout(p=self.dropout_p)
        self.out_fc = nn.Linear(in_features=self.hidden_size, out_features=data_feature['road_num'])

    def predict_g(self, trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask=None):
        """softmax over the candidates of function G alone (function_g_fc.py:206-221); same arguments."""
        from .generator_v4 import owner_of
        owner = owner_of(self, 'function_g')
        if owner is None:
            raise RuntimeError('FunctionGFC.predict_g runs through the device tables of its GeneratorV4: '
                               'construct the module as GeneratorV4(config, data_feature).function_g')
        return owner.predict_parts(trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask)[1]

    def forward(self, trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask=None, train=True):
        """The candidates' scores before the softmax: out_fc(attention-pooled LSTM state)[candidate_set]
        (function_g_fc.py:138-187), for a module in inference mode (the dropouts of :160,182 are then the
        identity; `train` is unused by the reference too).  Computed by the scoring kernel, which gathers only
        the candidates' rows of out_fc."""
        if self.training:
            raise NotImplementedError('FunctionGFC.forward with live dropout (training) is outside the generation '
                                      'path: call .train(False) first (our_model_generate.py:399)')
        from .generator_v4 import owner_of
        owner = owner_of(self, 'function_g')
        if owner is None:
            raise RuntimeError('FunctionGFC.forward runs through the device tables of its GeneratorV4: '
                               'construct the module as GeneratorV4(config, data_feature).function_g')
        return owner.predict_parts(trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask,
                                   logits=True)[3]

    def predict(self, trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask=None):
        """softmax of forward() over the candidates (function_g_fc.py:189-204)."""
        return self.predict_g(trace_loc, trace_time, des, candidate_set, candidate_dis, trace_mask)
