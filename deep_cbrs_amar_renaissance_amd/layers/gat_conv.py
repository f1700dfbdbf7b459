"""GATConv: the Spektral layer the reference instantiates at `src/models/gnn.py:321-328`.

Spektral 1.x single-mode sparse path (``_call_single``), attn_heads=1, concat_heads=True,
dropout_rate=0.0, add_self_loops=True:

    H = X . W                                   W  [F, 1, C]
    e_ij  = LeakyReLU_0.2( H_i . a_self + H_j . a_neigh )        over A's edges (duplicates kept) + (i, i)
    alpha = exp(e - max_i) / ( sum_i exp(e - max_i) + 1e-9 )      unsorted_segment_softmax over targets
    X'_i  = act( sum_j alpha_ij H_j + b )

On the device: `amar_rowwise_xw_f32` (H and the two attention scalars per node) followed by
`amar_gat_layer_f32` (two passes over the row: max of the neighbour scalars, then the weighted sum); on graphs whose
node table exceeds the per-XCD L2s, `amar_gat_xs_f32` (XCD-sliced image, exact online softmax over (max, sum, weighted
sum) triples).
"""
import os

import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer
from deep_cbrs_amar_renaissance_amd.utilities.math import spmm_kind


class GATConv(Layer):
    def __init__(self, channels, attn_heads=1, concat_heads=True, dropout_rate=0.5, return_attn_coef=False,
                 add_self_loops=True, activation=None, use_bias=True, kernel_regularizer=None,
                 bias_regularizer=None, attn_kernel_regularizer=None, **kwargs):
        super().__init__()
        if attn_heads != 1 or return_attn_coef:
            raise NotImplementedError("the HIP GAT layer implements attn_heads=1 without returned coefficients")
        if dropout_rate:
            raise NotImplementedError("attention dropout is a training-time feature; the reference uses 0.0 (config.yaml:20)")
        if activation != 'relu' or not use_bias:
            raise NotImplementedError("the HIP GAT layer is built for activation='relu', use_bias=True")
        self.channels, self.add_self_loops = channels, add_self_loops
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        self.attn_kernel_regularizer = attn_kernel_regularizer
        self.kernel = self.attn_kernel_self = self.attn_kernel_neighs = self.bias = None

    def build(self, input_shape):
        f_in = input_shape[0][-1]
        c = self.channels
        self.kernel = self.add_weight('kernel', (f_in, 1, c), 'glorot_uniform', self.kernel_regularizer)
        self.attn_kernel_self = self.add_weight('attn_kernel_self', (c, 1, 1), 'glorot_uniform', self.attn_kernel_regularizer)
        self.attn_kernel_neighs = self.add_weight('attn_kernel_neighs', (c, 1, 1), 'glorot_uniform', self.attn_kernel_regularizer)
        self.bias = self.add_weight('bias', (c,), 'zeros', self.bias_regularizer)

    def call(self, inputs, out=None, **kwargs):
        x, a = inputs
        n, c = a.shape[0], self.channels
        h = torch.empty((n, c), dtype=torch.float32, device=x.device)
        s_self = torch.empty(n, dtype=torch.float32, device=x.device)
        s_neigh = torch.empty(n, dtype=torch.float32, device=x.device)
        capi.rowwise_xw(x, self.kernel.view(-1, c), h, a_self=self.attn_kernel_self.view(c),
                        a_neigh=self.attn_kernel_neighs.view(c), s_self=s_self, s_neigh=s_neigh)
        if out is None:
            out = torch.empty((n, c), dtype=torch.float32, device=x.device)
        # large graphs: XCD-sliced form, exact online softmax.  ml1m(s=64): C = 8 0.54 ms against 0.94 (row kernel), C = 16 0.94 / 1.08;
        # at C = 32 the XS form (4 entries per step) loses, 2.20 / 1.37, and is only used for the row blocks of a partition
        kind = spmm_kind(a, c)
        large = kind == 'xs' or (c == 32 and kind == 'csr' and not os.environ.get('AMAR_SPMM_KIND') and
                                 a.shape[0] == a.shape[1] and a.shape[1] * c * 4 >= (16 << 20))
        lt = a.tiled_gat_image(c) if c in (8, 16, 32) and large else None
        if lt is not None:
            # the LDS-tiled walk with additive softmax weights against a per-row bound (amar_gat_lt_f32)
            capi.gat_lt(lt, a, h, s_self, s_neigh, self.bias, out, self_loop=self.add_self_loops)
        elif c in (8, 16) and kind == 'xs':
            capi.gat_xs(a.xcd_sliced(), h, s_self, s_neigh, self.bias, out, self_loop=self.add_self_loops)
        else:
            capi.gat_layer(a.rowptr, a.colidx, h, s_self, s_neigh, self.bias, out, self_loop=self.add_self_loops)
        return out
