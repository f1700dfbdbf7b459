"""ReductionLayer — mirrors `/root/reference/src/layers/reduction.py:5-55`.

Reduces the per-layer node representations [X_0 .. X_L]: 'concatenation' (default, `config.yaml:12`), 'sum', 'mean', 'last' and
'w-sum' (`WeightedSum`, reduction.py:36-55: sum_l w_l^2 X_l with learnable weights initialised to ones).  Inside SequentialGNN the
concatenation is a layout decision (every layer writes its own column slice of one buffer), so this layer only does work when it is
called stand-alone on separate tensors; the stack asks it for the 'w-sum' weights (`build_weights`) and reduces the slices itself.
"""
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer


class ReductionLayer(Layer):
    def __init__(self, method='concatenate', regularizer=None):
        super().__init__()
        if method not in ('concatenation', 'sum', 'mean', 'last', 'w-sum'):
            raise ValueError('Reduction method not supported: ' + method)
        self.method = method
        self.regularizer = regularizer
        self.w = None

    def build_weights(self, n_layers):
        """WeightedSum.build (reduction.py:45-52): 'reduction-weights' [n_layers, 1, 1], ones."""
        if self.method == 'w-sum' and self.w is None:
            self.add_weight('w', (n_layers, 1, 1), 'ones', self.regularizer)
        return self.w

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
        return self.reduce_slices(cat, widths)

    def reduce_slices(self, cat, widths):
        """'sum' / 'mean' / 'w-sum' over the equal-width column blocks of `cat`."""
        if len(set(widths)) != 1:
            raise ValueError("'{}' needs layers of equal width".format(self.method))
        out = torch.empty((cat.shape[0], widths[0]), dtype=torch.float32, device=cat.device)
        if self.method == 'w-sum':
            capi.reduce_layers_wsum(cat, len(widths), widths[0], self.build_weights(len(widths)).view(-1), out)
        else:
            capi.reduce_layers(cat, len(widths), widths[0], out, mean=self.method == 'mean')
        return out
