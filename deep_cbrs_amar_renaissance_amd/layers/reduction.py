"""ReductionLayer — mirrors `/root/reference/src/layers/reduction.py:5-33`.

Reduces the per-layer node representations [X_0 .. X_L]:
'concatenation' (default, `config.yaml:12`), 'sum', 'mean', 'last'.  Inside SequentialGNN the
concatenation is a layout decision (every layer writes its own column slice of one buffer), so
this layer only does work when it is called stand-alone on separate tensors.  'w-sum'
(WeightedSum) is out of scope.
"""
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer


class ReductionLayer(Layer):
    def __init__(self, method='concatenate', regularizer=None):
        super().__init__()
        if method == 'w-sum':
            raise NotImplementedError("'w-sum' (WeightedSum) is out of scope for the HIP path")
        if method not in ('concatenation', 'sum', 'mean', 'last'):
            raise ValueError('Reduction method not supported: ' + method)
        self.method = method

    def call(self, inputs, **kwargs):
        if self.method == 'last':
            return inputs[-1]
        n = inputs[0].shape[0]
        widths = [int(t.shape[1]) for t in inputs]
        cat = torch.empty((n, sum(widths)), dtype=torch.float32, device=inputs[0].device)
        off = 0
        for t, w in zip(inputs, widths):
            capi.copy_columns(t, cat[:, off:off + w])
            off += w
        if self.method == 'concatenation':
            return cat
        if len(set(widths)) != 1:
            raise ValueError("'{}' needs layers of equal width".format(self.method))
        out = torch.empty((n, widths[0]), dtype=torch.float32, device=cat.device)
        capi.reduce_layers(cat, len(inputs), widths[0], out, mean=self.method == 'mean')
        return out
