"""LightGCNConv — mirrors `/root/reference/src/layers/lightgcn_conv.py:6-58`.

    X' = D^-1/2 (A + I) D^-1/2 X          (no weights, no bias, no activation)

The reference calls Spektral's ``ops.modal_dot(a, x)`` (`lightgcn_conv.py:53`), i.e.
``tf.sparse.sparse_dense_matmul``; here it is one `amar_spmm_csr_f32` launch.
"""
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer
from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter, spmm_kind


class LightGCNConv(Layer):
    def __init__(self, activity_regularizer=None, **kwargs):
        super().__init__()
        self.activity_regularizer = activity_regularizer
        self.channels = None                      # output width == input width

    def build(self, input_shape):
        assert len(input_shape) >= 2

    def call(self, inputs, out=None, acc_in=None, acc_out=None, acc_div=None, **kwargs):
        x, a = inputs
        if out is None and acc_out is None:
            out = torch.empty((a.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        kind = spmm_kind(a, x.shape[1])
        if kind == 'xs':
            capi.spmm_xs(a.tiled_image(x.shape[1]), x, out, acc_in=acc_in, acc_out=acc_out, acc_div=acc_div)
        elif kind == 'sj':
            capi.spmm_sj(a.sliced(x.shape[1]), x, out, acc_in=acc_in, acc_out=acc_out, acc_div=acc_div)
        else:
            capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, out, acc_in=acc_in, acc_out=acc_out, acc_div=acc_div)
        return out

    @staticmethod
    def preprocess(a):
        return gcn_filter(a)
