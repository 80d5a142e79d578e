"""CPU restatement of GeneratorV4.predict (function G + function H + GAT).  TEST INFRASTRUCTURE.

See oracle/__init__.py.  Arithmetic is fp32 on torch-CPU tensors, eval mode (all
dropouts are the identity, as after ``.train(False)`` in our_model_generate.py:399).
Weights come in as a flat ``state_dict``-style mapping with the reference's key
names so the same fixture drives the reference, this oracle
and the HIP path.
"""
import torch
import torch.nn.functional as F


def gat_forward(x, edge_index, linear_proj_w, skip_proj_w, scoring_src, scoring_trg, bias,
                num_heads, out_features, concat=False, masks=None):
    """generator/gat.py:138-200 (GATLayerImp3.forward) + :84-114 (skip_concat_bias).

    x (N, FIN) f32, edge_index (2, E) int64 with row 0 = source, row 1 = target.  Eval mode by default; with
    `masks` = (m_in (N, FIN), m_proj (N, NH, F), m_att (E, NH)), each already scaled by 1 / (1 - p), the three
    dropouts of training mode (gat.py:149, :155, :180) multiply by them: every op is a differentiable torch op, so
    autograd through this function is the training-mode backward.
    """
    n = x.shape[0]
    nh, fo = num_heads, out_features
    if masks is not None:
        x = x * masks[0]                                                     # gat.py:149
    proj = F.linear(x, linear_proj_w).view(-1, nh, fo)                       # gat.py:153
    if masks is not None:
        proj = proj * masks[1].view(-1, nh, fo)                              # gat.py:155
    s_src = (proj * scoring_src).sum(dim=-1)                                 # gat.py:164
    s_trg = (proj * scoring_trg).sum(dim=-1)                                 # gat.py:165
    src, trg = edge_index[0], edge_index[1]
    e_src = s_src.index_select(0, src)                                       # gat.py:276
    e_trg = s_trg.index_select(0, trg)                                       # gat.py:277
    proj_lift = proj.index_select(0, src)                                    # gat.py:278
    e = F.leaky_relu(e_src + e_trg, 0.2)                                     # gat.py:174
    e = e - e.max()                                                          # gat.py:221 (GLOBAL max)
    ex = e.exp()                                                             # gat.py:222
    denom = torch.zeros((n, nh), dtype=ex.dtype)
    denom.scatter_add_(0, trg.unsqueeze(-1).expand_as(ex), ex)               # gat.py:246
    att = ex / (denom.index_select(0, trg) + 1e-16)                          # gat.py:251,230
    if masks is not None:
        att = att * masks[2]                                                 # gat.py:180
    weighted = proj_lift * att.unsqueeze(-1)                                 # gat.py:188
    out = torch.zeros((n, nh, fo), dtype=x.dtype)
    out.scatter_add_(0, trg.view(-1, 1, 1).expand_as(weighted), weighted)    # gat.py:263
    if fo == x.shape[-1]:                                                    # gat.py:94-97
        out = out + x.unsqueeze(1)
    else:                                                                    # gat.py:101
        out = out + F.linear(x, skip_proj_w).view(-1, nh, fo)
    if concat:
        out = out.view(-1, nh * fo)                                          # gat.py:106
    else:
        out = out.mean(dim=1)                                                # gat.py:109
    if bias is not None:
        out = out + bias                                                     # gat.py:112
    return F.elu(out)                                                        # gat.py:114


def h_input_features(sd, node_features, no_gps_emb):
    """distance_gat_fc.py:191-199: [lon_embed(x[:,-2]), lat_embed(x[:,-1]), x[:, :-2]]."""
    if no_gps_emb:
        return node_features
    lon = node_features[:, -2].long()
    lat = node_features[:, -1].long()
    return torch.cat([sd['lon_embed.weight'][lon], sd['lat_embed.weight'][lat], node_features[:, :-2]], dim=1)


def h_node_embedding(sd, node_features, edge_index, num_heads, embed_dim, concat=False, no_gps_emb=False):
    """distance_gat_fc.py:189-203 (_setup_node_emb): the cached GAT node embedding (N, F)."""
    x = h_input_features(sd, node_features, no_gps_emb)
    return gat_forward(x, edge_index,
                       sd['gat_encoder.linear_proj.weight'], sd['gat_encoder.skip_proj.weight'],
                       sd['gat_encoder.scoring_fn_source'], sd['gat_encoder.scoring_fn_target'],
                       sd['gat_encoder.bias'], num_heads, embed_dim, concat)


def h_scores(sd, node_emb, cand, des, dis):
    """distance_gat_fc.py:118-134 (predict_cache): unnormalised H score per candidate.

    cand (C,) int64, des int, dis (C,) f32.
    """
    from_emb = node_emb[cand]
    to_emb = node_emb[torch.full_like(cand, des)]
    emb_out = torch.relu(F.linear(torch.cat([from_emb, to_emb], dim=1), sd['emb_fc.weight'], sd['emb_fc.bias']))
    dis_out = torch.relu(F.linear(dis.unsqueeze(1), sd['dis_fc.weight'], sd['dis_fc.bias']))
    return F.linear(torch.cat([emb_out, dis_out], dim=1), sd['out_fc.weight'], sd['out_fc.bias']).squeeze(1)


def lstm_forward(sd, x, num_layers):
    """nn.LSTM (batch_first, zero initial state) as called at function_g_fc.py:169.

    x (L, I) f32 -> top-layer hidden states (L, H).  PyTorch gate order i, f, g, o;
    gates = W_ih x + b_ih + W_hh h + b_hh.
    """
    inp = x
    for layer in range(num_layers):
        w_ih = sd['lstm.weight_ih_l%d' % layer]
        w_hh = sd['lstm.weight_hh_l%d' % layer]
        b_ih = sd['lstm.bias_ih_l%d' % layer]
        b_hh = sd['lstm.bias_hh_l%d' % layer]
        hid = w_hh.shape[1]
        h = torch.zeros(hid)
        c = torch.zeros(hid)
        igates = F.linear(inp, w_ih, b_ih)          # (L, 4H), batched over the sequence like ATen does
        outs = []
        for t in range(inp.shape[0]):
            gates = igates[t] + F.linear(h, w_hh, b_hh)
            i, f, g, o = gates.chunk(4)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, dim=0)
    return inp


def g_attn_hidden(sd, trace_loc, trace_tim, num_layers):
    """function_g_fc.py:155-180 + IntraAttention :31-35 for one un-padded trajectory prefix."""
    loc = torch.as_tensor(trace_loc, dtype=torch.long)
    tim = torch.as_tensor(trace_tim, dtype=torch.long)
    x = torch.cat([sd['road_emb.weight'][loc], sd['time_emb.weight'][tim]], dim=1)   # :155-158
    hs = lstm_forward(sd, x, num_layers)                                             # :169
    q = hs[-1]                                                                       # :179
    w = torch.softmax(hs @ q, dim=0)                                                 # :31,34
    return (w.unsqueeze(1) * hs).sum(dim=0)                                          # :35


def g_attn_hidden_masked(sd, trace_loc, trace_tim, num_layers, padded_len):
    """The same for one row of a padded batch WITH trace_mask (function_g_fc.py:162-180): the packed LSTM runs
    over the row's real positions, pad_packed_sequence re-pads its output with ZERO rows up to the longest row
    of the batch (`padded_len`), the query is the last real state, and -- the masked_fill of :33 being assigned
    to `mask`, not to the weights -- the softmax of :34 runs over all `padded_len` positions, the zero rows
    included (logit 0)."""
    loc = torch.as_tensor(trace_loc, dtype=torch.long)
    tim = torch.as_tensor(trace_tim, dtype=torch.long)
    x = torch.cat([sd['road_emb.weight'][loc], sd['time_emb.weight'][tim]], dim=1)
    hs = lstm_forward(sd, x, num_layers)
    q = hs[-1]                                                                       # :173-177
    key = torch.cat([hs, torch.zeros((padded_len - hs.shape[0], hs.shape[1]))], dim=0)   # :167
    w = torch.softmax(key @ q, dim=0)                                                # :31,34 (mask has no effect)
    return (w.unsqueeze(1) * key).sum(dim=0)                                         # :35


def g_scores(sd, trace_loc, trace_tim, cand, num_layers):
    """function_g_fc.py:184-186: full-vocabulary FC then gather the candidates' logits."""
    a = g_attn_hidden(sd, trace_loc, trace_tim, num_layers)
    score = F.linear(a, sd['out_fc.weight'], sd['out_fc.bias'])
    return score[cand]


class OracleGenerator(object):
    """generator/generator_v4.py:16-72 restated on plain weight dicts (eval mode, cache=True)."""

    def __init__(self, g_state, h_state, w1, node_features, edge_index, num_heads, embed_dim,
                 lstm_layers=2, concat=False, no_gps_emb=False):
        self.g = {k: torch.as_tensor(v).float() for k, v in g_state.items()}
        self.h = {k: torch.as_tensor(v).float() for k, v in h_state.items()}
        self.w1 = torch.tensor([float(w1)], dtype=torch.float32)
        self.node_features = torch.as_tensor(node_features).float()
        self.edge_index = torch.as_tensor(edge_index).long()
        self.num_heads = num_heads
        self.embed_dim = embed_dim
        self.lstm_layers = lstm_layers
        self.concat = concat
        self.no_gps_emb = no_gps_emb
        self.node_emb = None

    def setup_node_emb(self):
        if self.node_emb is None:
            self.node_emb = h_node_embedding(self.h, self.node_features, self.edge_index, self.num_heads,
                                             self.embed_dim, self.concat, self.no_gps_emb)
        return self.node_emb

    def predict_g(self, trace_loc, trace_tim, cand):
        c = torch.as_tensor(cand, dtype=torch.long)
        return torch.softmax(g_scores(self.g, trace_loc, trace_tim, c, self.lstm_layers), dim=0)  # :221

    def g_logits(self, trace_loc, trace_tim, cand, padded_len=None):
        """FunctionGFC.forward (function_g_fc.py:138-187): candidate scores before the softmax; with `padded_len`
        the row sits in a masked batch padded to that length."""
        c = torch.as_tensor(cand, dtype=torch.long)
        if padded_len is None:
            return g_scores(self.g, trace_loc, trace_tim, c, self.lstm_layers)
        a = g_attn_hidden_masked(self.g, trace_loc, trace_tim, self.lstm_layers, padded_len)
        return F.linear(a, self.g['out_fc.weight'], self.g['out_fc.bias'])[c]

    def predict_masked_row(self, trace_loc, trace_tim, des, cand, dis, padded_len):
        """GeneratorV4.predict for one row of a masked batch padded to `padded_len` (see g_attn_hidden_masked)."""
        g = torch.softmax(self.g_logits(trace_loc, trace_tim, cand, padded_len), dim=0)
        h = self.predict_h(cand, des, dis)
        return torch.softmax(torch.add((1 - self.w1) * g, self.w1 * h), dim=0), g

    def predict_h(self, cand, des, dis):
        c = torch.as_tensor(cand, dtype=torch.long)
        d = torch.as_tensor(dis, dtype=torch.float32)
        return torch.softmax(h_scores(self.h, self.setup_node_emb(), c, int(des), d), dim=0)     # :157

    def predict(self, trace_loc, trace_tim, des, cand, dis):
        """generator_v4.py:46-54,71-72.  Returns the (C,) f32 probability vector."""
        g = self.predict_g(trace_loc, trace_tim, cand)
        h = self.predict_h(cand, des, dis)
        return torch.softmax(torch.add((1 - self.w1) * g, self.w1 * h), dim=0)
