"""GCNConv: the Spektral layer the reference instantiates at `src/models/gnn.py:289-295`.

    X' = act( A_hat . (X . W) + b ),   A_hat = gcn_filter(A)

Spektral order (dense product first, then the sparse product, bias, activation) is kept.  On
the device this is `amar_rowwise_xw_f32` + `amar_gcn_layer_f32`; inside a SequentialGNN the
dense product of layer l+1 is folded into the epilogue of layer l (see models/gnn.py).
"""
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer
from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter, spmm_kind


class GCNConv(Layer):
    def __init__(self, channels, activation=None, use_bias=True, kernel_regularizer=None, bias_regularizer=None,
                 **kwargs):
        super().__init__()
        if activation != 'relu' or not use_bias:
            raise NotImplementedError("the HIP GCN layer is built for activation='relu', use_bias=True (gnn.py:289-295)")
        self.channels = channels
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        self.kernel = self.bias = None

    def build(self, input_shape):
        f_in = input_shape[0][-1]
        self.kernel = self.add_weight('kernel', (f_in, self.channels), 'glorot_uniform', self.kernel_regularizer)
        self.bias = self.add_weight('bias', (self.channels,), 'zeros', self.bias_regularizer)

    def call(self, inputs, out=None, **kwargs):
        x, a = inputs
        n = a.shape[0]
        h = torch.empty((n, self.channels), dtype=torch.float32, device=x.device)
        capi.rowwise_xw(x, self.kernel, h)
        if out is None:
            out = torch.empty((n, self.channels), dtype=torch.float32, device=x.device)
        kind = spmm_kind(a, self.channels)
        if kind == 'xs':
            capi.spmm_xs(a.tiled_image(self.channels), h, out, bias=self.bias, relu=True)
        elif kind == 'sj':
            capi.spmm_sj(a.sliced(self.channels), h, out, bias=self.bias, relu=True)
        else:
            capi.gcn_layer(a.rowptr, a.colidx, a.vals, h, self.bias, out)
        return out

    @staticmethod
    def preprocess(a):
        return gcn_filter(a)
