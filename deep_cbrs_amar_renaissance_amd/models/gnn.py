"""Graph propagation models — mirrors `/root/reference/src/models/gnn.py:13-84, 210-388`.

``SequentialGNN`` owns the trainable [N, d] node table and the constant adjacency (in HBM as
CSR) and runs the convolution stack over the FULL graph; its output does not depend on the
batch (`gnn.py:263-264` passes ``None``).  Every layer writes its result straight into its
column slice of one [N, d(L+1)] buffer, which makes the 'concatenation' reduction free.

All-GCN stacks take a fused route: one `amar_rowwise_xw_f32` prologue (X_0.W_1, plus the X_0
slice copy) and then ONE kernel per layer, whose epilogue applies bias + ReLU, stores the
slice and multiplies by the next layer's kernel.  LightGCN stacks keep the running layer sum in
the SpMM epilogue ('mean' reduction).  GraphSAGE / GAT consume the raw edge list (duplicates
kept, `utilities/math.py`), exactly as the reference hands Spektral an un-normalised matrix
(`gnn.py:316-319, 349-352`).

``HalfInputSequentialGNN`` / ``FullInputSequentialGNN`` (`gnn.py:87-207`, TwoStep / TwoWay stacks) run the same
propagation on a node table that is (partly) handed in by the caller.
"""
import abc

import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Model, L2
from deep_cbrs_amar_renaissance_amd.layers.gat_conv import GATConv
from deep_cbrs_amar_renaissance_amd.layers.gcn_conv import GCNConv
from deep_cbrs_amar_renaissance_amd.layers.dgcf_conv import DGCFConv
from deep_cbrs_amar_renaissance_amd.layers.graphsage_conv import GraphSageConv
from deep_cbrs_amar_renaissance_amd.layers.lightgcn_conv import LightGCNConv
from deep_cbrs_amar_renaissance_amd.layers.reduction import ReductionLayer
from deep_cbrs_amar_renaissance_amd.utilities.math import convert_to_tensor, spmm_kind


class SequentialGNN(Model):
    def __init__(
            self,
            adj_matrix,
            seq_layers,
            embedding_dim=8,
            final_node='concatenation',
            dropout=None,
            regularizer=None,
            cache_neighbours=False
    ):
        """
        :param adj_matrix: the (sparse) graph adjacency matrix, already pre-processed for the layer type.
        :param seq_layers: list of GNN layers.
        :param embedding_dim: width of the trainable node table.
        :param final_node: 'concatenation', 'sum', 'mean', 'w-sum' or 'last'.
        :param dropout: must be None (no `dropout` key exists in config.yaml; training-only anyway).
        :param regularizer: regulariser object carried by the node table.
        :param cache_neighbours: must be False (the reference raises NotImplementedError too, gnn.py:52-53).
        """
        super().__init__()
        self.embeddings = self.add_weight('embeddings', (adj_matrix.shape[0], embedding_dim),
                                          'glorot_uniform', regularizer)
        self._init_stack(adj_matrix, seq_layers, final_node, dropout, cache_neighbours)

    def _init_stack(self, adj_matrix, seq_layers, final_node, dropout, cache_neighbours):
        if cache_neighbours:
            raise NotImplementedError("Multi-hops neighbours caching is not yet completely supported!")
        if dropout:
            raise NotImplementedError("dropout is a training-time feature; inference treats it as identity")
        self.cache_neighbours = cache_neighbours
        # GraphSAGE / GAT ignore edge values and add their own self loop
        edge_list = any(isinstance(l, (GraphSageConv, GATConv)) for l in seq_layers)
        self.adj_matrix = convert_to_tensor(adj_matrix, with_values=not edge_list, drop_diagonal=edge_list)
        self.dropout = None
        self.final_node = final_node
        self.reduce = ReductionLayer(final_node)
        self.seq_layers = torch.nn.ModuleList(seq_layers)
        self.reduce.build_weights(len(seq_layers) + 1)          # 'w-sum': one weight per term X_0 .. X_L (reduction.py:45-52)
        self.built = True

    @property
    def n_hops(self):
        return len(self.seq_layers)

    def __len__(self):
        return self.n_hops

    def input_width(self):
        return int(self.embeddings.shape[1])

    def layer_widths(self):
        widths = [self.input_width()]
        for layer in self.seq_layers:
            widths.append(int(layer.channels) if layer.channels is not None else widths[-1])
        return widths

    def output_dim(self):
        widths = self.layer_widths()
        return sum(widths) if self.final_node == 'concatenation' else widths[-1]

    def _build_layers(self, widths):
        for layer, f_in in zip(self.seq_layers, widths[:-1]):
            if not layer.built:
                layer.build([(self.adj_matrix.shape[0], f_in), self.adj_matrix.shape])
                layer.built = True

    def call(self, inputs=None, **kwargs):
        return self._propagate(self.embeddings)

    def _propagate(self, x, with_layers=False):
        """The convolution stack + reduction over the node table `x` [N, widths[0]] (gnn.py:74-84).
        with_layers: also return the [N, sum(widths)] buffer of every layer's output (None on LightGCN's running-sum
        route, which never materialises it) — what the reverse pass of training.py reads."""
        a = self.adj_matrix
        n = x.shape[0]
        if n != a.shape[0]:
            raise ValueError("node table has {} rows, the adjacency matrix {}".format(n, a.shape[0]))
        widths = self.layer_widths()
        if int(x.shape[1]) != widths[0]:
            raise ValueError("node table is {} wide, the stack was built for {}".format(int(x.shape[1]), widths[0]))
        self._build_layers(widths)
        dev = x.device
        layers = list(self.seq_layers)

        if self.final_node == 'mean' and layers and all(isinstance(l, LightGCNConv) for l in layers):
            # running sum S_l = S_{l-1} + X_l in the SpMM epilogue; the last layer divides by L+1
            acc = x
            for k, layer in enumerate(layers):
                last = k == len(layers) - 1
                acc_out = torch.empty((n, widths[0]), dtype=torch.float32, device=dev)
                nxt = None if last else torch.empty((n, widths[0]), dtype=torch.float32, device=dev)
                kind = spmm_kind(a, widths[0])
                if kind == 'xs':
                    capi.spmm_xs(a.tiled_image(widths[0]), x, nxt, acc_in=acc, acc_out=acc_out,
                                 acc_div=len(layers) + 1 if last else None)
                elif kind == 'sj':
                    capi.spmm_sj(a.sliced(widths[0]), x, nxt, acc_in=acc, acc_out=acc_out,
                                 acc_div=len(layers) + 1 if last else None)
                else:
                    capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, nxt, acc_in=acc, acc_out=acc_out,
                                  acc_div=len(layers) + 1 if last else None)
                x, acc = nxt, acc_out
            return (acc, None) if with_layers else acc

        cat = torch.empty((n, sum(widths)), dtype=torch.float32, device=dev)
        offs = [sum(widths[:k]) for k in range(len(widths) + 1)]
        slices = [cat[:, offs[k]:offs[k + 1]] for k in range(len(widths))]

        # (widths the kernels are not instantiated for run layer by layer as column chunks, without the fused next X.W)
        if layers and all(isinstance(l, GCNConv) for l in layers) and all(w in (4, 8, 16, 32, 64) for w in widths[1:]):
            h = torch.empty((n, widths[1]), dtype=torch.float32, device=dev)
            # value-free XS image: the chain of gathered tables stays pre-scaled by d^-1/2 (first one in this X.W launch, the
            # following ones in the combine kernel's epilogue)
            pre = all(spmm_kind(a, widths[k + 1]) == 'xs' for k in range(len(layers))) and \
                all(a.tiled_image(widths[k + 1]).row_scale is not None for k in range(len(layers)))
            capi.rowwise_xw(x, layers[0].kernel, h, copy_to=slices[0], row_scale=a.tiled_image(widths[1]).row_scale if pre else None)
            for k, layer in enumerate(layers):
                nxt = layers[k + 1] if k + 1 < len(layers) else None
                h_next = torch.empty((n, widths[k + 2]), dtype=torch.float32, device=dev) if nxt is not None else None
                kind = spmm_kind(a, widths[k + 1])
                if kind == 'xs':
                    # rows_needed (set by a scoring runner that knows which rows its towers read): the LAST layer's rows past it —
                    # the property rows of a user-item-property graph, a quarter of its tiles — are not computed
                    capi.spmm_xs(a.tiled_image(widths[k + 1]), h, slices[k + 1], bias=layer.bias, relu=True,
                                 Wnext=nxt.kernel if nxt is not None else None, Hnext=h_next,
                                 prescaled=pre, scale_next=pre and nxt is not None,
                                 rows_needed=getattr(self, 'rows_needed', None) if nxt is None else None)
                elif kind == 'sj':
                    capi.spmm_sj(a.sliced(widths[k + 1]), h, slices[k + 1], bias=layer.bias, relu=True,
                                 Wnext=nxt.kernel if nxt is not None else None, Hnext=h_next)
                else:
                    capi.gcn_layer(a.rowptr, a.colidx, a.vals, h, layer.bias, slices[k + 1],
                                   Wnext=nxt.kernel if nxt is not None else None, Hnext=h_next)
                h = h_next
        else:
            capi.copy_columns(x, slices[0])
            dense_in = x if x.stride(0) == widths[0] else None      # a dense copy of the layer's input, where one exists
            for k, layer in enumerate(layers):
                if dense_in is not None and hasattr(layer, 'wants_dense_input') and layer.wants_dense_input(a, widths[k]):
                    # GraphSAGE on the LDS-tiled walk: gathers from a dense table, the launch also leaves the next layer's
                    dense_next = torch.empty((n, widths[k + 1]), dtype=torch.float32, device=dev) if k + 1 < len(layers) else None
                    layer([dense_in, a], out=slices[k + 1], dense_out=dense_next)
                    dense_in = dense_next
                else:
                    layer([slices[k], a], out=slices[k + 1])
                    dense_in = None

        out = self._reduce(cat, slices, widths)
        return (out, cat) if with_layers else out

    def _reduce(self, cat, slices, widths):
        """ReductionLayer (reduction.py:9-33) over the column slices of `cat`."""
        if self.final_node == 'concatenation':
            return cat
        if self.final_node == 'last':
            return slices[-1]
        return self.reduce.reduce_slices(cat, widths)        # 'sum' / 'mean' / 'w-sum'


class HalfInputSequentialGNN(SequentialGNN):
    """`gnn.py:87-150`: the first `n_random_embeddings` rows of the node table are trainable, the remaining rows are
    handed in by the caller (TwoStep: users are trained here, items come out of the item-property stack)."""

    def __init__(self, adj_matrix, seq_layers, n_random_embeddings, final_node='concatenation', dropout=None,
                 embedding_dim=8, regularizer=None, cache_neighbours=False):
        Model.__init__(self)
        self.embeddings = self.add_weight('embeddings', (n_random_embeddings, embedding_dim), 'glorot_uniform', regularizer)
        self._init_stack(adj_matrix, seq_layers, final_node, dropout, cache_neighbours)

    def call(self, inputs=None, **kwargs):
        e = self.embeddings
        if inputs is None or inputs.shape[1] != e.shape[1] or e.shape[0] + inputs.shape[0] != self.adj_matrix.shape[0]:
            raise ValueError("HalfInputSequentialGNN: inputs must be [{}, {}]".format(
                self.adj_matrix.shape[0] - e.shape[0], e.shape[1]))
        x = torch.empty((self.adj_matrix.shape[0], e.shape[1]), dtype=torch.float32, device=e.device)
        capi.copy_columns(e.detach(), x[:e.shape[0]])
        capi.copy_columns(inputs, x[e.shape[0]:])
        return self._propagate(x)


class FullInputSequentialGNN(SequentialGNN):
    """`gnn.py:153-207`: no table of its own, the whole node table is handed in (TwoWay's user-item stack).  The input
    width is fixed by the first call, or by `input_dim` when the owner knows it up front."""

    def __init__(self, adj_matrix, seq_layers, final_node='concatenation', dropout=None, cache_neighbours=False,
                 input_dim=None):
        Model.__init__(self)
        self.input_dim = input_dim
        self._init_stack(adj_matrix, seq_layers, final_node, dropout, cache_neighbours)

    def input_width(self):
        if self.input_dim is None:
            raise ValueError("FullInputSequentialGNN: the input width is unknown before the first call")
        return int(self.input_dim)

    def call(self, x=None, **kwargs):
        if x is None:
            raise ValueError("FullInputSequentialGNN needs the node table as its input")
        if self.input_dim is None:
            self.input_dim = int(x.shape[1])
        return self._propagate(x)


class _Hoisted:
    """One propagation per weight state: legal because the node representations do not depend on the batch
    (gnn.py:263-264 passes None), see BasicGNN.predict."""

    def _init_hoist(self):
        self._hoisted = None
        self.hoist = False

    def _maybe_hoisted(self, run):
        if not self.hoist:
            return run()
        version = self.weights_version
        if self._hoisted is None or self._hoisted[0] != version:
            self._hoisted = (version, run())
        return self._hoisted[1]


class GNN(Model, _Hoisted, abc.ABC):
    def __init__(
            self,
            adj_matrix,
            n_hops,
            embedding_dim=8,
            final_node="concatenation",
            dropout=None,
            l2_regularizer=None,
            cache_neighbours=False,
            **kwargs
    ):
        """
        :param adj_matrix: the (sparse) graph adjacency matrix.
        :param n_hops: number of convolution layers.
        :param embedding_dim: width of the trainable node table.
        :param final_node: 'concatenation', 'sum', 'mean', 'w-sum' or 'last'.
        :param dropout: see SequentialGNN.
        :param l2_regularizer: L2 factor carried by the node table and the layers' weights (may be None).
        :param cache_neighbours: see SequentialGNN.
        :param kwargs: unused (the driver passes the whole `model:` config section, experiment.py:146-153).
        """
        super().__init__()
        if isinstance(l2_regularizer, str):                   # PyYAML reads '1e-4' (no dot) as a string
            l2_regularizer = float(l2_regularizer)
        regularizer = L2(l2_regularizer) if l2_regularizer is not None else None
        gnn_layers = [self.build_gnn_layer(i, regularizer=regularizer) for i in range(n_hops)]
        self.gnn_layers = SequentialGNN(
            adj_matrix, gnn_layers,
            embedding_dim=embedding_dim, final_node=final_node,
            dropout=dropout, regularizer=regularizer, cache_neighbours=cache_neighbours
        )
        self.built = True
        self._init_hoist()

    @abc.abstractmethod
    def build_gnn_layer(self, i, **kwargs):
        pass

    def output_dim(self):
        return self.gnn_layers.output_dim()

    def build_layers(self):
        self.gnn_layers._build_layers(self.gnn_layers.layer_widths())

    def call(self, inputs=None, **kwargs):
        """Node representations [N, F_out]; `inputs` is ignored like in the reference (gnn.py:263-264)."""
        return self._maybe_hoisted(lambda: self.gnn_layers(None))


class GCN(GNN):
    def __init__(self, adj_matrix, n_hiddens=(8, 8, 8), **kwargs):
        self.n_hiddens = list(n_hiddens)
        adj_matrix = GCNConv.preprocess(adj_matrix)           # gnn.py:283
        super().__init__(adj_matrix, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GCNConv(self.n_hiddens[i], activation='relu',
                       kernel_regularizer=regularizer, bias_regularizer=regularizer)


class GAT(GNN):
    def __init__(self, adj_matrix, n_hiddens=(8, 8, 8), dropout_rate=0.0, **kwargs):
        self.n_hiddens = list(n_hiddens)
        self.dropout_rate = dropout_rate
        super().__init__(adj_matrix, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GATConv(self.n_hiddens[i], dropout_rate=self.dropout_rate, activation='relu',
                       kernel_regularizer=regularizer, bias_regularizer=regularizer)


class GraphSage(GNN):
    def __init__(self, adj_matrix, n_hiddens=(8, 8, 8), aggregate='mean', **kwargs):
        self.n_hiddens = list(n_hiddens)
        self.aggregate = aggregate
        super().__init__(adj_matrix, len(self.n_hiddens), **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return GraphSageConv(self.n_hiddens[i], activation='relu', aggregate=self.aggregate,
                             kernel_regularizer=regularizer, bias_regularizer=regularizer)


class LightGCN(GNN):
    def __init__(self, adj_matrix, n_layers=3, **kwargs):
        kwargs['final_node'] = 'mean'                          # gnn.py:378
        adj_matrix = LightGCNConv.preprocess(adj_matrix)       # gnn.py:381
        super().__init__(adj_matrix, n_layers, **kwargs)

    def build_gnn_layer(self, i, **kwargs):
        return LightGCNConv()


class DGCF(GNN):
    def __init__(self, adj_matrix, n_layers=3, **kwargs):
        kwargs['final_node'] = 'mean'                          # gnn.py:405
        crosshop_matrix = DGCFConv.preprocess(adj_matrix)      # gnn.py:408
        super().__init__(crosshop_matrix, n_layers, **kwargs)

    def build_gnn_layer(self, i, regularizer=None, **kwargs):
        return DGCFConv(regularizer)
