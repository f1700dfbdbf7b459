"""GraphSageConv: the Spektral layer the reference instantiates at `src/models/gnn.py:354-361`.

Spektral 1.x semantics, aggregate='mean' (`config.yaml:18`):

    a    <- add_self_loops(a)
    agg  =  unsorted_segment_mean( X[source], target )       edge values ignored, duplicates counted
    X'   =  act( l2_normalize( [X || agg] . W + b ) )         W [2 F, C]; normalise BEFORE the activation

On the device: one fused row kernel (`amar_sage_layer_f32`) while the node table fits the per-XCD L2s; beyond that
(utilities.math.spmm_kind) the mean aggregate runs on the XCD-sliced value-free SpMM (`amar_spmm_xs_f32` with
diag = 1, row scale = 1 / count) followed by `amar_dense_f32` and `amar_l2norm_fwd_f32`.

Spektral is not installed here, so three points are explicit switches: the added
self loop (``self_loops``), the concat order ``[x, agg]`` and normalise-before-activation (fixed).
"""
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer
from deep_cbrs_amar_renaissance_amd.utilities.math import spmm_kind


class GraphSageConv(Layer):
    def __init__(self, channels, aggregate='mean', activation=None, use_bias=True, kernel_regularizer=None,
                 bias_regularizer=None, self_loops=True, **kwargs):
        super().__init__()
        if aggregate != 'mean':
            raise NotImplementedError("only aggregate='mean' (config.yaml:18) has a HIP kernel")
        if activation != 'relu' or not use_bias:
            raise NotImplementedError("the HIP GraphSAGE layer is built for activation='relu', use_bias=True")
        self.channels, self.aggregate, self.self_loops = channels, aggregate, self_loops
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        self.kernel = self.bias = None

    def build(self, input_shape):
        f_in = input_shape[0][-1]
        self.kernel = self.add_weight('kernel', (2 * f_in, self.channels), 'glorot_uniform', self.kernel_regularizer)
        self.bias = self.add_weight('bias', (self.channels,), 'zeros', self.bias_regularizer)

    def call(self, inputs, out=None, **kwargs):
        x, a = inputs
        if a.vals is not None:
            raise ValueError("GraphSageConv expects the raw edge list (DeviceCSR without values)")
        n, f = a.shape[0], x.shape[1]
        if out is None:
            out = torch.empty((n, self.channels), dtype=torch.float32, device=x.device)
        if spmm_kind(a, f) != 'xs':
            capi.sage_layer(a.rowptr, a.colidx, x, self.kernel, self.bias, out, self_loop=self.self_loops)
            return out
        xa = torch.empty((n, 2 * f), dtype=torch.float32, device=x.device)
        capi.copy_columns(x, xa[:, :f])
        capi.spmm_xs(a.xcd_sliced_mean(self.self_loops), x, xa[:, f:], prescaled=True)
        z = torch.empty((n, self.channels), dtype=torch.float32, device=x.device)
        capi.dense(xa, self.kernel, self.bias, z, act=None)
        nrm = torch.empty_like(z)
        inv = torch.empty(n, dtype=torch.float32, device=x.device)
        capi.l2norm_fwd(z, nrm, inv, out, act='relu')
        return out
