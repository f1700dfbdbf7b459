"""Hybrid (graph + BERT) recommender heads — mirrors `/root/reference/src/models/hybrid.py:13-181`.

``HybridCBRS`` takes four inputs (user-graph, item-graph, user-BERT, item-BERT):

    ug -> dense1a, ig -> dense1b, ub -> dense2a, ib -> dense2b
    feature_based:  x1 = dense3a([ug || ig]),  x2 = dense3b([ub || ib])
    entity based:   x1 = dense3a([ug || ub]),  x2 = dense3b([ig || ib])
    out = clf([x1 || x2])

with FusionLayer('concatenate') everywhere (the only setting of the BASELINE configs);
'attention' fusion and residual heads are out of scope.  ``HybridBertGNN`` = propagation ->
lookup -> HybridCBRS; the factory generates ``HybridBert{GCN,GAT,GraphSage,LightGCN}``.

The BERT blocks may be given per batch ([B, 768] host arrays, as the reference's Sequence
does, `datasets.py:65-66`) or as a resident table + ids (``bert_table``), which removes the
12.6 MB host->device copy per 2 048-pair batch.
"""
import abc

import torch

from deep_cbrs_amar_renaissance_amd.engine import Model, ids_to_device, to_device_tensor
from deep_cbrs_amar_renaissance_amd.layers.fusion import FusionLayer
from deep_cbrs_amar_renaissance_amd.models.dense import build_dense_network, build_dense_classifier
from deep_cbrs_amar_renaissance_amd.models.gnn import GCN, GAT, GraphSage, LightGCN, DGCF
from deep_cbrs_amar_renaissance_amd.models.basic import _out_of_scope, _TWO_STEP, _TWO_WAY


class HybridCBRS(Model):
    """Hybrid recommender system that receives inputs from two sources."""

    def __init__(
            self,
            feature_based=True,
            dense_units=((512, 256, 128), (512, 256, 128), (64, 64)),
            clf_units=(64, 64),
            activation='relu',
            fusion_method='concatenate',
            residual=False,
            **kwargs
    ):
        super().__init__()
        self.feature_based = feature_based
        if feature_based:
            self.fuse1a, self.fuse1b, self.fuse2 = FusionLayer('concatenate'), FusionLayer('concatenate'), FusionLayer(fusion_method)
        else:
            self.fuse1a, self.fuse1b, self.fuse2 = FusionLayer(fusion_method), FusionLayer(fusion_method), FusionLayer('concatenate')
        if residual:
            if dense_units[2][-1] != clf_units[-1]:
                raise ValueError("The last dense units before the last fusion layer "
                                 "must be equal to the last classifier units for residual connections")
            raise NotImplementedError("residual heads are out of scope for the HIP path (SURVEY.md §8f N4)")
        self.dense_units = [list(d) for d in dense_units]
        if any(len(d) == 0 for d in self.dense_units):
            raise NotImplementedError("every hybrid branch needs at least one layer")
        self.dense1a = build_dense_network(dense_units[0], activation=activation)
        self.dense1b = build_dense_network(dense_units[0], activation=activation)
        self.dense2a = build_dense_network(dense_units[1], activation=activation)
        self.dense2b = build_dense_network(dense_units[1], activation=activation)
        self.dense3a = build_dense_network(dense_units[2], activation=activation)
        self.dense3b = build_dense_network(dense_units[2], activation=activation)
        self.residual = self.activation = None
        self.clf = build_dense_classifier(clf_units, n_classes=1, activation=activation)

    def build(self, input_shape):
        self.build_head(input_shape[0][-1], input_shape[2][-1])

    def build_head(self, g_dim, b_dim):
        """Create all weights for known graph / BERT input widths."""
        d1 = self.dense1a.build_chain(g_dim)
        self.dense1b.build_chain(g_dim)
        d2 = self.dense2a.build_chain(b_dim)
        self.dense2b.build_chain(b_dim)
        d3 = self.dense3a.build_chain(2 * d1 if self.feature_based else d1 + d2)
        self.dense3b.build_chain(2 * d2 if self.feature_based else d1 + d2)
        self.clf.build_chain(2 * d3)
        self.built = True

    def call(self, inputs, g_ids=None, b_ids=None, **kwargs):
        """inputs = (ug, ig, ub, ib) blocks; ``g_ids`` / ``b_ids`` = (user ids, item ids) turn the graph /
        BERT inputs into tables to gather from."""
        ug, ig, ub, ib = [to_device_tensor(t) for t in inputs]
        gu, gi = g_ids if g_ids is not None else (None, None)
        bu, bi = b_ids if b_ids is not None else (None, None)
        m = gu.numel() if gu is not None else ug.shape[0]
        d1, d2, d3 = self.dense_units[0][-1], self.dense_units[1][-1], self.dense_units[2][-1]
        dev = ug.device
        if self.feature_based:
            f1 = torch.empty((m, 2 * d1), dtype=torch.float32, device=dev)     # [ug || ig]
            f2 = torch.empty((m, 2 * d2), dtype=torch.float32, device=dev)     # [ub || ib]
            self.dense1a(ug, out=f1[:, :d1], ids=gu)
            self.dense1b(ig, out=f1[:, d1:], ids=gi)
            self.dense2a(ub, out=f2[:, :d2], ids=bu)
            self.dense2b(ib, out=f2[:, d2:], ids=bi)
        else:
            f1 = torch.empty((m, d1 + d2), dtype=torch.float32, device=dev)    # [ug || ub]
            f2 = torch.empty((m, d1 + d2), dtype=torch.float32, device=dev)    # [ig || ib]
            self.dense1a(ug, out=f1[:, :d1], ids=gu)
            self.dense2a(ub, out=f1[:, d1:], ids=bu)
            self.dense1b(ig, out=f2[:, :d1], ids=gi)
            self.dense2b(ib, out=f2[:, d1:], ids=bi)
        x = torch.empty((m, 2 * d3), dtype=torch.float32, device=dev)
        self.dense3a(f1, out=x[:, :d3])
        self.dense3b(f2, out=x[:, d3:])
        return self.clf(x)


class HybridBertGNN(Model, abc.ABC):
    def __init__(
            self,
            dense_units=(32, 16),
            clf_units=(16, 16),
            feature_based=False,
            activation='relu',
            fusion_method='concatenate',
            residual=False,
            **kwargs
    ):
        super().__init__()
        self.rs = HybridCBRS(
            feature_based=feature_based,
            dense_units=dense_units,
            clf_units=clf_units,
            activation=activation,
            fusion_method=fusion_method,
            residual=residual
        )
        self.bert_table = None
        self.built = True

    def set_bert_table(self, table):
        """Keep the [|U|+|I|, 768] BERT rows resident in HBM; batches may then pass ids only."""
        self.bert_table = to_device_tensor(table)

    def call(self, inputs, **kwargs):
        updated_embeddings = self.gnn(None)
        return self.embed_recommend(updated_embeddings, inputs)

    def embed_recommend(self, embeddings, inputs):
        """inputs = (user ids, item ids, user BERT block, item BERT block); the BERT blocks may be None
        when a resident table was registered with :meth:`set_bert_table`."""
        ug, ig, ub, ib = inputs
        g_ids = (ids_to_device(ug), ids_to_device(ig))
        if ub is None and ib is None:
            if self.bert_table is None:
                raise ValueError("no BERT blocks in the batch and no resident table registered")
            return self.rs([embeddings, embeddings, self.bert_table, self.bert_table], g_ids=g_ids, b_ids=g_ids)
        return self.rs([embeddings, embeddings, ub, ib], g_ids=g_ids)

    def _hoist_begin(self, hoist):
        self.gnn.hoist = bool(hoist)

    def _hoist_end(self):
        self.gnn.hoist = False
        self.gnn._hoisted = None


def BasicGNNFactory(name, Parent, GNN):
    def __init__(self, *args, **kwargs):
        Parent.__init__(self, **kwargs)
        self.gnn = self.gnn_class(*args, **kwargs)
        self.gnn.gnn_layers._build_layers(self.gnn.gnn_layers.layer_widths())

    return type(name, (Parent,), {"gnn_class": GNN, "__init__": __init__})


class HybridBertTSGNN(HybridBertGNN):
    pass


class HybridBertTWGNN(HybridBertGNN):
    pass


HYBRID_GNNS = [
    (HybridBertGNN, [GCN, GAT, GraphSage, LightGCN, DGCF], None),
    (HybridBertTSGNN, [_out_of_scope(n) for n in _TWO_STEP], lambda name: 'HybridBertTS' + name[7:]),
    (HybridBertTWGNN, [_out_of_scope(n) for n in _TWO_WAY], lambda name: 'HybridBertTW' + name[6:]),
]


def generate_hybrids():
    for parent, gnns, name_getter in HYBRID_GNNS:
        for gnn in gnns:
            name = name_getter(gnn.__name__) if name_getter is not None else 'HybridBert' + gnn.__name__
            globals()[name] = BasicGNNFactory(name, parent, gnn)


generate_hybrids()
