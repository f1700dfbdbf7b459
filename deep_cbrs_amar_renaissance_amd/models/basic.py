"""Basic recommender heads — mirrors `/root/reference/src/models/basic.py:11-120`.

``BasicRS``: two dense towers (user, item), Concatenate, dense classifier ending in
Dense(1, sigmoid).  ``BasicGNN``: full-graph propagation -> embedding lookup of the batch's
(user, item) ids -> BasicRS.  The class factory at the bottom generates ``BasicGCN``,
``BasicGAT``, ``BasicGraphSage``, ``BasicLightGCN`` (and the out-of-scope names, which resolve
but raise on construction) exactly like `basic.py:90-120`.

Device mapping: the lookup is fused into the first tower layer's load (`amar_dense_f32` with
``ids``), each tower's last layer writes its half of the concatenation buffer directly.
"""
import abc

import numpy as np
import torch

from deep_cbrs_amar_renaissance_amd.engine import Model, ids_to_device, to_device_tensor
from deep_cbrs_amar_renaissance_amd.models.dense import build_dense_network, build_dense_classifier
from deep_cbrs_amar_renaissance_amd.models.gnn import GCN, GAT, GraphSage, LightGCN, DGCF


class BasicRS(Model):
    def __init__(
            self,
            dense_units=(512, 256, 128),
            clf_units=(64, 64),
            activation='relu',
            **kwargs
    ):
        """
        :param dense_units: units of the user / item towers.
        :param clf_units: units of the classifier (a final Dense(1, sigmoid) is appended).
        :param activation: hidden activation.
        :param kwargs: unused.
        """
        super().__init__()
        self.dense_units = list(dense_units)
        self.unet = build_dense_network(dense_units, activation=activation)
        self.inet = build_dense_network(dense_units, activation=activation)
        self.clf = build_dense_classifier(clf_units, n_classes=1, activation=activation)

    def build(self, input_shape):
        self.build_head(input_shape[0][-1], input_shape[1][-1])

    def build_head(self, u_dim, i_dim):
        """Create all weights for known input widths (Keras would do it at the first call)."""
        d = self.unet.build_chain(u_dim)
        self.inet.build_chain(i_dim)
        self.clf.build_chain(2 * d)
        self.built = True

    def call(self, inputs, u_ids=None, i_ids=None, **kwargs):
        """inputs = (u, i): [B, F] feature blocks — or, with ``u_ids``/``i_ids``, two tables to gather from."""
        u, i = inputs
        u, i = to_device_tensor(u), to_device_tensor(i)
        if len(self.dense_units) == 0:
            raise NotImplementedError("BasicRS needs at least one tower layer")
        if not self.built:
            self.build_head(u.shape[1], i.shape[1])
        tu = self.unet.apply2(u, ids_a=u_ids)
        ti = self.inet.apply2(i, ids_a=i_ids)
        return self.clf.apply2(tu, ti)                      # Concatenate([u, i]) happens in the kernel's gather

    def towers(self, u_table, i_table):
        """Per-ENTITY tower outputs: Dense stacks act row by row, so unet/inet can run once per user/item
        instead of once per pair; `score_towers` then only gathers and classifies."""
        return self.unet.apply2(u_table), self.inet.apply2(i_table)

    def score_towers(self, tu, ti, u_ids, i_ids, u_base=0, i_base=0):
        return self.clf.apply2(tu, ti, ids_a=u_ids, base_a=u_base, ids_b=i_ids, base_b=i_base)


class BasicGNN(Model, abc.ABC):
    def __init__(
            self,
            dense_units=(32, 16),
            clf_units=(16, 16),
            activation='relu',
            **kwargs
    ):
        super().__init__()
        self.rs = BasicRS(dense_units, clf_units, activation=activation)
        self.n_users = self.n_items = None      # optional: lets the hoisted mode run each tower on its own rows only
        self._towers = None
        self.built = True

    def call(self, inputs, **kwargs):
        updated_embeddings = self.gnn(None)
        return self.embed_recommend(updated_embeddings, inputs)

    def embed_recommend(self, embeddings, inputs):
        """Look up the user / item rows of `embeddings` and score them: inputs = (user ids, item ids).

        Faithful mode gathers per pair.  Hoisted mode (predict) computes the two towers once per entity —
        same kernel, same per-row arithmetic, so both modes return identical bits."""
        u, i = ids_to_device(inputs[0]), ids_to_device(inputs[1])
        if not self.gnn.hoist:
            return self.rs([embeddings, embeddings], u_ids=u, i_ids=i)
        key = (self.weights_version, embeddings.data_ptr())
        if self._towers is None or self._towers[0] != key:
            n = embeddings.shape[0]
            nu = self.n_users if self.n_users is not None else n
            lo = nu if self.n_users is not None else 0
            hi = nu + self.n_items if (self.n_users is not None and self.n_items is not None) else n
            tu, ti = self.rs.towers(embeddings[:nu], embeddings[lo:hi])
            self._towers = (key, tu, ti, lo)
        _, tu, ti, lo = self._towers
        return self.rs.score_towers(tu, ti, u, i, 0, lo)

    def _hoist_begin(self, hoist):
        self.gnn.hoist = bool(hoist)

    def _hoist_end(self):
        self.gnn.hoist = False
        self.gnn._hoisted = None
        self._towers = None


class BasicTSGNN(BasicGNN):
    pass


class BasicTWGNN(BasicGNN):
    pass


class BasicKnowledgeGCN(BasicGNN):
    pass


def _out_of_scope(name):
    class _OutOfScope:
        def __init__(self, *args, **kwargs):
            raise NotImplementedError("{} is out of scope for the HIP path (SURVEY.md §2 row 6)".format(name))
    _OutOfScope.__name__ = name
    return _OutOfScope


def BasicGNNFactory(name, Parent, GNN):
    def __init__(self, *args, **kwargs):
        Parent.__init__(self, **kwargs)
        self.gnn = self.gnn_class(*args, **kwargs)
        self.gnn.gnn_layers._build_layers(self.gnn.gnn_layers.layer_widths())
        self.rs.build_head(self.gnn.output_dim(), self.gnn.output_dim())

    return type(name, (Parent,), {"gnn_class": GNN, "__init__": __init__})


_TWO_STEP = ['TwoStepGCN', 'TwoStepGraphSage', 'TwoStepGAT', 'TwoStepLightGCN', 'TwoStepDGCF']
_TWO_WAY = ['TwoWayGCN', 'TwoWayGraphSage', 'TwoWayGAT', 'TwoWayLightGCN', 'TwoWayDGCF']

BASIC_GNNS = [
    (BasicGNN, [GCN, GAT, GraphSage, LightGCN, DGCF], None),
    (BasicTSGNN, [_out_of_scope(n) for n in _TWO_STEP], lambda name: 'BasicTS' + name[7:]),
    (BasicTWGNN, [_out_of_scope(n) for n in _TWO_WAY], lambda name: 'BasicTW' + name[6:]),
]


def generate_basics():
    for parent, gnns, name_getter in BASIC_GNNS:
        for gnn in gnns:
            name = name_getter(gnn.__name__) if name_getter is not None else 'Basic' + gnn.__name__
            globals()[name] = BasicGNNFactory(name, parent, gnn)


generate_basics()
