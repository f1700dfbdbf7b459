"""FusionLayer — mirrors `/root/reference/src/layers/fusion.py:5-68`.

'concatenate' joins two [B, *] feature blocks along the feature axis (`fusion.py:51-53`); in the
scoring head this is again a layout decision (producers write adjacent column slices).
'attention' (`fusion.py:54-68`) is only used by the hybrid-gnn-tweaks configs and is out of scope.
"""
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer


class FusionLayer(Layer):
    def __init__(self, method='concatenate'):
        super().__init__()
        if method not in ['concatenate', 'attention']:
            raise ValueError("Unknown concatenation method called {}".format(method))
        if method == 'attention':
            raise NotImplementedError("FusionLayer('attention') is out of scope for the HIP path (SURVEY.md §8f N4)")
        self.method = method

    def call(self, inputs, **kwargs):
        a, b = inputs
        out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
        capi.copy_columns(a, out[:, :a.shape[1]])
        capi.copy_columns(b, out[:, a.shape[1]:])
        return out
