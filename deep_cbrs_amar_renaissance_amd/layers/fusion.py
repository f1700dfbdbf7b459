"""FusionLayer — mirrors `/root/reference/src/layers/fusion.py:5-68`.

'concatenate' joins two [B, *] feature blocks along the feature axis (`fusion.py:51-53`); in the
scoring head this is again a layout decision (producers write adjacent column slices).

'attention' (`fusion.py:54-68`, used by econfigs/hybrid-gnn-tweaks*.yaml): the narrower block is first
projected to the wider one's width (``proj_weight``, no bias), then with x = stack([a, b], axis=1)

    att = softmax(tanh(x . att_weight), axis=1)          a two-way softmax PER FEATURE
    out = sum(att * x, axis=1)                           = wa * a + (1 - wa) * b,  wa = sigmoid(tanh(a.W) - tanh(b.W))

Two `amar_dense_f32` products and `amar_attention_mix_f32`.
"""
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer


class FusionLayer(Layer):
    def __init__(self, method='concatenate'):
        super().__init__()
        if method not in ['concatenate', 'attention']:
            raise ValueError("Unknown concatenation method called {}".format(method))
        self.method = method
        self.proj_first = None
        self.proj_weight = self.att_weight = None

    def build(self, input_shape):
        a_shape, b_shape = input_shape
        if self.method != 'attention':
            return
        da, db = int(a_shape[-1]), int(b_shape[-1])
        if da != db:
            self.proj_first = da < db                       # the narrower block is projected
            self.proj_weight = self.add_weight('proj_weight', (min(da, db), max(da, db)), 'glorot_uniform')
        self.att_weight = self.add_weight('att_weight', (max(da, db), max(da, db)), 'glorot_uniform')

    def output_dim(self, da, db):
        return da + db if self.method == 'concatenate' else max(da, db)

    def project(self, a, b):
        """The two blocks at equal width (fusion.py:57-61)."""
        if self.proj_first is None:
            return a, b
        src = a if self.proj_first else b
        out = torch.empty((src.shape[0], self.proj_weight.shape[1]), dtype=torch.float32, device=src.device)
        capi.dense(src, self.proj_weight, None, out, act=None)
        return (out, b) if self.proj_first else (a, out)

    def call(self, inputs, **kwargs):
        a, b = inputs
        if self.method == 'concatenate':
            out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
            capi.copy_columns(a, out[:, :a.shape[1]])
            capi.copy_columns(b, out[:, a.shape[1]:])
            return out
        if not self.built:
            self.build([a.shape, b.shape])
            self.built = True
        a, b = self.project(a, b)
        ta, tb = torch.empty_like(a), torch.empty_like(b)
        capi.dense(a, self.att_weight, None, ta, act=None)
        capi.dense(b, self.att_weight, None, tb, act=None)
        out = torch.empty((a.shape[0], a.shape[1]), dtype=torch.float32, device=a.device)
        capi.attention_mix(a, b, ta, tb, out)
        return out
