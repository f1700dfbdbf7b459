"""Two-Way GNN stacks — mirrors `/root/reference/src/models/twgnn.py:12-274`.

Way one propagates a trainable table over the user-property graph (users first, `preprocess.py:9-41`), way two another
one over the item-property graph (items first); the leading |U| and |I| rows of their `user_item_node` reductions are
stacked into the user-item node table that `FullInputSequentialGNN` (gnn.py:153-207) propagates over the user-item
graph.  Every convolution runs on the same HIP kernels as the single-graph models (models/gnn.py).
"""
import abc

import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Model
from deep_cbrs_amar_renaissance_amd.layers.dgcf_conv import DGCFConv
from deep_cbrs_amar_renaissance_amd.layers.gat_conv import GATConv
from deep_cbrs_amar_renaissance_amd.layers.gcn_conv import GCNConv
from deep_cbrs_amar_renaissance_amd.layers.graphsage_conv import GraphSageConv
from deep_cbrs_amar_renaissance_amd.layers.lightgcn_conv import LightGCNConv
from deep_cbrs_amar_renaissance_amd.models.gnn import SequentialGNN, FullInputSequentialGNN, _Hoisted
from deep_cbrs_amar_renaissance_amd.models.tsgnn import _regularizer


class TwoWayGNN(Model, _Hoisted, abc.ABC):
    def __init__(
            self,
            n_users,
            n_items,
            adj_matrices,
            n_hops,
            embedding_dim=8,
            user_item_node="mean",
            final_node="concatenation",
            dropout=None,
            l2_regularizer=None,
            cache_neighbours=False,
            **kwargs
    ):
        """
        :param n_users: number of users (leading rows of the user-property graph).
        :param n_items: number of items (leading rows of the item-property graph).
        :param adj_matrices: (user-item, item-property, user-property) adjacencies, pre-processed for the layer type.
        :param n_hops: number of convolution layers of each of the three stacks.
        :param embedding_dim: width of both property-graph node tables.
        :param user_item_node: reduction of the two ways ('mean' by default, twgnn.py:20).
        :param final_node: reduction of the user-item stack.
        :param dropout, l2_regularizer, cache_neighbours: see models.gnn.GNN.
        :param kwargs: unused.
        """
        super().__init__()
        regularizer = _regularizer(l2_regularizer)
        if len(adj_matrices) != 3:
            raise ValueError('Exactly three adjacency matrix are needed!')
        adj_ui_matrix, adj_ip_matrix, adj_up_matrix = adj_matrices

        way_one = [self.build_gnn_layer(i, regularizer=regularizer) for i in range(n_hops)]
        self.way_one_gnn_layers = SequentialGNN(
            adj_up_matrix, way_one,
            embedding_dim=embedding_dim, final_node=user_item_node,
            dropout=dropout, regularizer=regularizer, cache_neighbours=cache_neighbours
        )
        way_two = [self.build_gnn_layer(i, regularizer=regularizer) for i in range(n_hops)]
        self.way_two_gnn_layers = SequentialGNN(
            adj_ip_matrix, way_two,
            embedding_dim=embedding_dim, final_node=user_item_node,
            dropout=dropout, regularizer=regularizer, cache_neighbours=cache_neighbours
        )
        self.n_items = n_items
        self.n_users = n_users

        # widths of the user-item stack (twgnn.py:74-80) continue the n_hiddens list
        if hasattr(self, 'n_hiddens'):
            if n_hops == len(self.n_hiddens):
                if user_item_node == 'concatenation':
                    second_embedding_dim = embedding_dim * (n_hops + 1)
                    self.n_hiddens.extend([second_embedding_dim for _ in range(n_hops)])
                else:
                    self.n_hiddens.extend([embedding_dim for _ in range(n_hops)])
        step_two = [self.build_gnn_layer(i + n_hops, regularizer=regularizer) for i in range(n_hops)]
        self.step_two_gnn_layers = FullInputSequentialGNN(
            adj_ui_matrix, step_two,
            final_node=final_node, dropout=dropout, cache_neighbours=cache_neighbours,
            input_dim=self.way_one_gnn_layers.output_dim()
        )
        self.built = True
        self._init_hoist()

    @abc.abstractmethod
    def build_gnn_layer(self, i, **kwargs):
        pass

    def output_dim(self):
        return self.step_two_gnn_layers.output_dim()

    def build_layers(self):
        for seq in (self.way_one_gnn_layers, self.way_two_gnn_layers, self.step_two_gnn_layers):
            seq._build_layers(seq.layer_widths())

    def _run(self):
        users = self.way_one_gnn_layers(None)
        items = self.way_two_gnn_layers(None)
        if users.shape[1] != items.shape[1]:
            raise ValueError("the two ways hand over {} and {} wide rows".format(users.shape[1], items.shape[1]))
        x = torch.empty((self.n_users + self.n_items, users.shape[1]), dtype=torch.float32, device=users.device)
        capi.copy_columns(users[:self.n_users], x[:self.n_users])
        capi.copy_columns(items[:self.n_items], x[self.n_users:])
        return self.step_two_gnn_layers(x)

    def call(self, inputs=None, **kwargs):
        """[|U|+|I|, F_out] node representations; `inputs` is ignored (twgnn.py:98-105)."""
        return self._maybe_hoisted(self._run)


class TwoWayGCN(TwoWayGNN):
    def __init__(self, n_users, n_items, adj_matrices, n_hiddens=(8, 8, 8), **kwargs):
        self.n_hiddens = list(n_hiddens)
        adj_matrices = [GCNConv.preprocess(matrix) for matrix in adj_matrices]       # twgnn.py:127
        super().__init__(n_users, n_items, adj_matrices, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GCNConv(self.n_hiddens[i], activation='relu', kernel_regularizer=regularizer, bias_regularizer=regularizer)


class TwoWayGraphSage(TwoWayGNN):
    def __init__(self, n_users, n_items, adj_matrices, n_hiddens=(8, 8, 8), aggregate='mean', **kwargs):
        self.n_hiddens = list(n_hiddens)
        self.aggregate = aggregate
        super().__init__(n_users, n_items, adj_matrices, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GraphSageConv(self.n_hiddens[i], activation='relu', aggregate=self.aggregate,
                             kernel_regularizer=regularizer, bias_regularizer=regularizer)


class TwoWayGAT(TwoWayGNN):
    def __init__(self, n_users, n_items, adj_matrix, n_hiddens=(8, 8, 8), dropout_rate=0.0, **kwargs):
        self.n_hiddens = list(n_hiddens)
        self.dropout_rate = dropout_rate
        super().__init__(n_users, n_items, adj_matrix, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GATConv(self.n_hiddens[i], dropout_rate=self.dropout_rate, activation='relu',
                       kernel_regularizer=regularizer, bias_regularizer=regularizer)


class TwoWayLightGCN(TwoWayGNN):
    def __init__(self, n_users, n_items, adj_matrix, n_layers=3, **kwargs):
        kwargs['final_node'] = 'mean'                                                 # twgnn.py:227
        adj_matrix = [LightGCNConv.preprocess(matrix) for matrix in adj_matrix]       # twgnn.py:230
        super().__init__(n_users, n_items, adj_matrix, n_layers, **kwargs)

    def build_gnn_layer(self, i, **kwargs):
        return LightGCNConv()


class TwoWayDGCF(TwoWayGNN):
    def __init__(self, n_users, n_items, adj_matrix, n_layers=3, **kwargs):
        kwargs['final_node'] = 'mean'                                                 # twgnn.py:257
        crosshop_matrix = [DGCFConv.preprocess(matrix) for matrix in adj_matrix]      # twgnn.py:260
        super().__init__(n_users, n_items, crosshop_matrix, n_layers, **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return DGCFConv(regularizer)
