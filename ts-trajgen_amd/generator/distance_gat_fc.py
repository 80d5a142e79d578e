"""DistanceGatFC drop-in (function H; reference: generator/distance_gat_fc.py:9-42 constructor,
:118-157 predict_cache/predict_next, :189-223 node-embedding cache, :44-95 forward / calculate_loss in training
mode)."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from .gat import GATLayerImp3


class DistanceGatFC(nn.Module):
    def __init__(self, config, data_feature):
        super().__init__()
        self.embed_dim = config['embed_dim']
        self.num_of_heads = config['num_of_heads']
        self.concat = config['concat']
        self.device = config['device']
        self.gps_emb_dim = config['gps_emb_dim']
        self.distance_mode = config['distance_mode']
        self.no_gps_emb = config.get('no_gps_emb', False)
        self.adj_mx = data_feature.get('adj_mx')
        self.Apt = torch.LongTensor([self.adj_mx.row.tolist(), self.adj_mx.col.tolist()]).to(self.device)
        self.node_features = data_feature['node_features']
        if not self.no_gps_emb:
            self.feature_dim = self.node_features.shape[1] - 2 + self.gps_emb_dim * 2
        else:
            self.feature_dim = self.node_features.shape[1]
        self.gat_encoder = GATLayerImp3(num_in_features=self.feature_dim, num_out_features=self.embed_dim,
                                        num_of_heads=self.num_of_heads, concat=self.concat, device=self.device)
        if not self.no_gps_emb:
            self.lat_embed = nn.Embedding(num_embeddings=data_feature['img_height'], embedding_dim=self.gps_emb_dim)
            self.lon_embed = nn.Embedding(num_embeddings=data_feature['img_width'], embedding_dim=self.gps_emb_dim)
        self.emb_fc = nn.Linear(in_features=self.embed_dim * 2, out_features=10)
        self.dis_fc = nn.Linear(in_features=1, out_features=10)
        self.out_fc = nn.Linear(in_features=20, out_features=1)
        self.loss_func = torch.nn.CrossEntropyLoss()
        self.node_emb_feature = None

    def _dev(self):
        return self.emb_fc.weight.device

    @torch.no_grad()
    def input_features(self):
        """[lon_embed | lat_embed | raw[:, :-2]] per node (distance_gat_fc.py:191-199), built on the device."""
        dev = self._dev()
        nf = self.node_features.to(dev).contiguous().float()
        if self.no_gps_emb:
            return nf
        L = _lib.lib()
        n, fraw = nf.shape
        x = torch.empty((n, self.feature_dim), dtype=torch.float32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.tsg_gat_build_input(_lib.ptr(nf), n, fraw, _lib.ptr(self.lon_embed.weight.detach()),
                                             self.lon_embed.num_embeddings,
                                             _lib.ptr(self.lat_embed.weight.detach()),
                                             self.lat_embed.num_embeddings, self.gps_emb_dim, _lib.ptr(x),
                                             _lib.ptr(flag), _lib.current_stream()), 'tsg_gat_build_input')
        if int(flag.item()) != 0:
            raise IndexError('lon/lat grid cell outside the lon_embed / lat_embed tables')
        return x

    def update_node_emb(self):
        dev = self._dev()
        if dev.type != 'cuda':
            raise RuntimeError('DistanceGatFC needs its parameters on a HIP device (no CPU fallback)')
        x = self.input_features()
        self.node_emb_feature = self.gat_encoder((x, self.Apt.to(dev)))[0]

    def _setup_node_emb(self):
        if self.node_emb_feature is None:
            self.update_node_emb()

    def predict_next(self, candidate_set, des, candidate_distance):
        """softmax over the candidates of function H alone (distance_gat_fc.py:136-157); same arguments."""
        from .generator_v4 import owner_of
        owner = owner_of(self, 'function_h')
        if owner is None:
            raise RuntimeError('DistanceGatFC.predict_next runs through the device tables of its GeneratorV4: '
                               'construct the module as GeneratorV4(config, data_feature).function_h')
        # the scoring kernel evaluates G and H together; G gets a one-point trace (the first candidate, time 0)
        loc = candidate_set[:, :1]
        return owner.predict_parts(loc, torch.zeros_like(loc), des, candidate_set, candidate_distance)[2]

    def _owner_tables(self, what):
        from .generator_v4 import owner_of
        owner = owner_of(self, 'function_h')
        if owner is None:
            raise RuntimeError('DistanceGatFC.%s runs through the device tables of its GeneratorV4: construct the '
                               'module as GeneratorV4(config, data_feature).function_h' % what)
        return owner.tables()

    def predict_cache(self, from_node, to_node, distance):
        """Score of (from, to, distance) triples before the softmax, on the cached node embeddings
        (distance_gat_fc.py:118-134)."""
        return self._owner_tables('predict_cache').h_pair_scores(from_node, to_node, distance)

    def forward(self, from_node, to_node, distance):
        """distance_gat_fc.py:44-74 for a module in inference mode: the reference re-encodes the graph on every call,
        which gives the cached embeddings again as long as the parameters have not changed (the device tables are
        rebuilt when they have).  In training mode: _forward_train (live dropout inside the GAT, differentiable)."""
        if self.training:
            return self._forward_train(from_node, to_node, distance)
        return self._owner_tables('forward').h_pair_scores(from_node, to_node, distance)

    def _forward_train(self, from_node, to_node, distance):
        """distance_gat_fc.py:44-74 in training mode, differentiable: the graph is re-encoded by the GAT layer's
        training kernels (live dropout, autograd node), the embedding look-ups and the 20-unit head around it are
        torch ops on the device (a few thousand multiply-adds per batch)."""
        dev = self._dev()
        if dev.type != 'cuda':
            raise RuntimeError('DistanceGatFC needs its parameters on a HIP device (no CPU fallback)')
        nf = self.node_features.to(dev).float()
        if not self.no_gps_emb:
            in_feature = torch.cat([self.lon_embed(nf[:, -2].long()), self.lat_embed(nf[:, -1].long()), nf[:, :-2]],
                                   dim=1)
        else:
            in_feature = nf
        encode_feature = self.gat_encoder((in_feature, self.Apt.to(dev)))[0]
        from_node_emb = encode_feature[from_node.to(dev)]
        to_node_emb = encode_feature[to_node.to(dev)]
        emb_fc_out = torch.relu(self.emb_fc(torch.cat([from_node_emb, to_node_emb], dim=1)))
        dis_fc_out = torch.relu(self.dis_fc(distance.to(dev).float().unsqueeze(1)))
        return self.out_fc(torch.cat([emb_fc_out, dis_fc_out], dim=1)).squeeze(1)

    def calculate_loss(self, candidate_set, des, candidate_distance, target):
        """Cross entropy of the candidates' scores (distance_gat_fc.py:75-95); same arguments."""
        c = candidate_set.shape[1]
        score = self.forward(candidate_set.flatten(), des.unsqueeze(1).repeat(1, c).flatten(),
                             candidate_distance.flatten())
        return self.loss_func(score.reshape(candidate_set.shape[0], c), target.to(score.device))

    def predict(self, candidate_set, des, candidate_distance):
        """softmax over the candidates of forward() (distance_gat_fc.py:97-116)."""
        c = candidate_set.shape[1]
        score = self.forward(candidate_set.flatten(), des.unsqueeze(1).repeat(1, c).flatten(),
                             candidate_distance.flatten())
        return torch.softmax(score.reshape(candidate_set.shape[0], c), dim=1)
