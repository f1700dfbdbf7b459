"""Two-Step GNN stacks — mirrors `/root/reference/src/models/tsgnn.py:11-265`.

Step one propagates a trainable [|I|+|P|, d] table over the item-property graph (items first: `loaders.py:62-68` keeps
item indices un-offset for 'unary-kg'); its first |I| rows, reduced by `item_node`, become the item half of the
user-item node table whose user half [|U|, d2] is trained by step two (`HalfInputSequentialGNN`, gnn.py:87-150).
Every convolution of both steps runs on the same HIP kernels as the single-graph models (models/gnn.py).
"""
import abc

from deep_cbrs_amar_renaissance_amd.engine import Model, L2
from deep_cbrs_amar_renaissance_amd.layers.dgcf_conv import DGCFConv
from deep_cbrs_amar_renaissance_amd.layers.gat_conv import GATConv
from deep_cbrs_amar_renaissance_amd.layers.gcn_conv import GCNConv
from deep_cbrs_amar_renaissance_amd.layers.graphsage_conv import GraphSageConv
from deep_cbrs_amar_renaissance_amd.layers.lightgcn_conv import LightGCNConv
from deep_cbrs_amar_renaissance_amd.models.gnn import SequentialGNN, HalfInputSequentialGNN, _Hoisted


def _regularizer(l2_regularizer):
    if isinstance(l2_regularizer, str):                       # PyYAML reads '1e-4' (no dot) as a string
        l2_regularizer = float(l2_regularizer)
    return L2(l2_regularizer) if l2_regularizer is not None else None


class TwoStepGNN(Model, _Hoisted, abc.ABC):
    def __init__(
            self,
            n_users,
            n_items,
            adj_matrices,
            n_hops,
            embedding_dim=8,
            item_node="mean",
            final_node="concatenation",
            dropout=None,
            l2_regularizer=None,
            cache_neighbours=False,
            **kwargs
    ):
        """
        :param n_users: number of users (rows trained by step two).
        :param n_items: number of items (leading rows of the item-property graph).
        :param adj_matrices: (user-item adjacency, item-property adjacency), already pre-processed for the layer type.
        :param n_hops: number of convolution layers of EACH step.
        :param embedding_dim: width of the item-property node table.
        :param item_node: reduction of step one ('mean' by default, tsgnn.py:19).
        :param final_node: reduction of step two.
        :param dropout, l2_regularizer, cache_neighbours: see models.gnn.GNN.
        :param kwargs: unused.
        """
        super().__init__()
        regularizer = _regularizer(l2_regularizer)
        if len(adj_matrices) != 2:
            raise ValueError('Exactly two adjacency matrix are needed!')
        adj_ui_matrix, adj_kg_matrix = adj_matrices

        step_one = [self.build_gnn_layer(i, regularizer=regularizer) for i in range(n_hops)]
        self.step_one_gnn_layers = SequentialGNN(
            adj_kg_matrix, step_one,
            embedding_dim=embedding_dim, final_node=item_node,
            dropout=dropout, regularizer=regularizer, cache_neighbours=cache_neighbours
        )
        self.n_embeddings = n_items

        # widths of the second stack (tsgnn.py:65-75): it continues the n_hiddens list with the width step one hands over
        second_embedding_dim = embedding_dim
        if hasattr(self, 'n_hiddens'):
            if n_hops == len(self.n_hiddens):
                if item_node == 'concatenation':
                    second_embedding_dim = embedding_dim * (n_hops + 1)
                    self.n_hiddens.extend([second_embedding_dim for _ in range(n_hops)])
                else:
                    self.n_hiddens.extend([embedding_dim for _ in range(n_hops)])
        step_two = [self.build_gnn_layer(i + n_hops, regularizer=regularizer) for i in range(n_hops)]
        # the user table carries no regulariser in the reference (tsgnn.py:77-81 does not pass one)
        self.step_two_gnn_layers = HalfInputSequentialGNN(
            adj_ui_matrix, step_two, n_users,
            embedding_dim=second_embedding_dim, final_node=final_node,
            dropout=dropout, cache_neighbours=cache_neighbours
        )
        if self.step_one_gnn_layers.output_dim() != second_embedding_dim:
            raise ValueError("step one hands over {}-wide item rows, the user table is {} wide".format(
                self.step_one_gnn_layers.output_dim(), second_embedding_dim))
        self.built = True
        self._init_hoist()

    @abc.abstractmethod
    def build_gnn_layer(self, i, **kwargs):
        pass

    def output_dim(self):
        return self.step_two_gnn_layers.output_dim()

    def build_layers(self):
        for seq in (self.step_one_gnn_layers, self.step_two_gnn_layers):
            seq._build_layers(seq.layer_widths())

    def _run(self):
        x = self.step_one_gnn_layers(None)
        return self.step_two_gnn_layers(x[:self.n_embeddings])

    def call(self, inputs=None, **kwargs):
        """[|U|+|I|, F_out] node representations; `inputs` is ignored (tsgnn.py:99-101)."""
        return self._maybe_hoisted(self._run)


class TwoStepGCN(TwoStepGNN):
    def __init__(self, n_users, n_items, adj_matrices, n_hiddens=(8, 8, 8), **kwargs):
        self.n_hiddens = list(n_hiddens)
        adj_matrices = [GCNConv.preprocess(matrix) for matrix in adj_matrices]       # tsgnn.py:122
        super().__init__(n_users, n_items, adj_matrices, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GCNConv(self.n_hiddens[i], activation='relu', kernel_regularizer=regularizer, bias_regularizer=regularizer)


class TwoStepGraphSage(TwoStepGNN):
    def __init__(self, n_users, n_items, adj_matrices, n_hiddens=(8, 8, 8), aggregate='mean', **kwargs):
        self.n_hiddens = list(n_hiddens)
        self.aggregate = aggregate
        super().__init__(n_users, n_items, adj_matrices, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GraphSageConv(self.n_hiddens[i], activation='relu', aggregate=self.aggregate,
                             kernel_regularizer=regularizer, bias_regularizer=regularizer)


class TwoStepGAT(TwoStepGNN):
    def __init__(self, n_users, n_items, adj_matrix, n_hiddens=(8, 8, 8), dropout_rate=0.0, **kwargs):
        self.n_hiddens = list(n_hiddens)
        self.dropout_rate = dropout_rate
        super().__init__(n_users, n_items, adj_matrix, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GATConv(self.n_hiddens[i], dropout_rate=self.dropout_rate, activation='relu',
                       kernel_regularizer=regularizer, bias_regularizer=regularizer)


class TwoStepLightGCN(TwoStepGNN):
    def __init__(self, n_users, n_items, adj_matrix, n_layers=3, **kwargs):
        kwargs['final_node'] = 'mean'                                                 # tsgnn.py:222
        adj_matrix = [LightGCNConv.preprocess(matrix) for matrix in adj_matrix]       # tsgnn.py:225
        super().__init__(n_users, n_items, adj_matrix, n_layers, **kwargs)

    def build_gnn_layer(self, i, **kwargs):
        return LightGCNConv()


class TwoStepDGCF(TwoStepGNN):
    def __init__(self, n_users, n_items, adj_matrix, n_layers=3, **kwargs):
        kwargs['final_node'] = 'mean'                                                 # tsgnn.py:252
        crosshop_matrix = [DGCFConv.preprocess(matrix) for matrix in adj_matrix]      # tsgnn.py:255
        super().__init__(n_users, n_items, crosshop_matrix, n_layers, **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return DGCFConv(regularizer)
