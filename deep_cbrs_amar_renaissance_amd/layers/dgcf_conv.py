"""DGCFConv — mirrors `/root/reference/src/layers/dgcf_conv.py:11-102` (Deoscillated adaptive Graph CF).

    X' = A_dgcf . ( X * sigmoid(w) )          w [N, 1] trainable, initialised to ones ("LocalityAdaptive")

``preprocess`` (host, scipy, once per model — as in the reference) builds
A_dgcf = gcn_filter(A) + highpass(gcn_filter(A . A)) + I, where the high-pass filter keeps the cross-hop
entries above the threshold, out of (1e-1, 1e-2, 1e-3, 5e-4), whose kept-entry count is closest in ratio to
nnz(gcn_filter(A)).  On the device the layer is `amar_locality_scale_f32` + the SpMM of the other layers.
"""
import numpy as np
import torch
from scipy import sparse

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer
from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter, spmm_kind

EPSILONS = (1e-1, 1e-2, 1e-3, 5e-4)


class DGCFConv(Layer):
    def __init__(self, regularizer=None, **kwargs):
        super().__init__()
        self.regularizer = regularizer
        self.channels = None                      # output width == input width
        self.w = None

    def build(self, input_shape):
        n_nodes = input_shape[0][0]
        self.w = self.add_weight('w', (n_nodes, 1), 'ones', self.regularizer)

    def call(self, inputs, out=None, **kwargs):
        x, a = inputs
        gated = torch.empty((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        capi.locality_scale(x, self.w.view(-1), gated)
        if out is None:
            out = torch.empty((a.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
        if spmm_kind(a, x.shape[1]) == 'xs':
            capi.spmm_xs(a.tiled_image(x.shape[1]), gated, out)
        else:
            capi.spmm_csr(a.rowptr, a.colidx, a.vals, gated, out)
        return out

    @staticmethod
    def high_pass_filter(adjacency, crosshop):
        """dgcf_conv.py:50-80.  A threshold that keeps nothing gets an infinite ratio (the reference would divide by zero)."""
        edges = len(adjacency.data)
        filtered = [crosshop.multiply(crosshop > eps).tocsr() for eps in EPSILONS]
        counts = [len(m.data) for m in filtered]
        ratios = [np.inf if c == 0 else (edges / c if edges > c else c / edges) for c in counts]
        print('Edges: {}'.format(edges))
        print('Cross edges: {}'.format(counts))
        print('Found ratios: {}'.format(ratios))
        return filtered[int(np.argmin(ratios))]

    @staticmethod
    def preprocess(a):
        a = sparse.csr_matrix(a)
        crosshop = a.dot(a)
        a, crosshop = gcn_filter(a), gcn_filter(crosshop)
        crosshop = DGCFConv.high_pass_filter(a, crosshop)
        out = (a + crosshop + sparse.eye(a.shape[0], dtype=np.float32)).tocsr().astype(np.float32)
        out.sum_duplicates()
        return out
