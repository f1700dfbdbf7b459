"""Dense stacks of the scoring heads — mirrors `/root/reference/src/models/dense.py:4-17`.

``Dense`` is Keras' layer (act(x . W + b), W [in, out] glorot_uniform, b zeros) running on
`amar_dense_f32` (fp32 MFMA).  ``Sequential`` threads two layout hooks through a stack:
``ids`` lets the FIRST layer gather its input rows from a table (the embedding lookup of
`basic.py:73-74` fused into the GEMM's load), and ``out`` lets the LAST layer write into a
column slice of a wider buffer (the Concatenate of `basic.py:35`).
"""
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Layer, to_device_tensor


class Dense(Layer):
    def __init__(self, units, activation=None, **kwargs):
        super().__init__()
        if activation not in capi.ACT_CODES:
            raise NotImplementedError("activation '{}' has no HIP epilogue (relu, sigmoid, linear do)".format(activation))
        self.units, self.activation = int(units), activation
        self.kernel = self.bias = None

    def build(self, input_shape):
        self.kernel = self.add_weight('kernel', (input_shape[-1], self.units), 'glorot_uniform')
        self.bias = self.add_weight('bias', (self.units,), 'zeros')

    def call(self, x, out=None, ids=None, **kwargs):
        m = ids.numel() if ids is not None else x.shape[0]
        if out is None:
            out = torch.empty((m, self.units), dtype=torch.float32, device=x.device)
        k, n = self.kernel.shape
        if capi.dense_split_supported(int(k), int(n)) and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and x.is_cuda:
            # wide layers (the 768 -> 256 of the content towers): products on the bf16 matrix instruction with both operands split
            # three ways — f32-accurate (include/amar_hip.h); the kernel's split image is rebuilt when the weights change
            version = (id(self.kernel), self.kernel._version)
            cache = self.__dict__.get('_split_image')
            if cache is None or cache[0] != version:
                image = torch.from_numpy(capi.dense_split_pack(self.kernel.detach().cpu().numpy())).to(self.kernel.device)
                cache = self.__dict__['_split_image'] = (version, image)
            capi.dense_split(x, cache[1], int(k), int(n), self.bias, out, act=self.activation, ids=ids)
            return out
        capi.dense(x, self.kernel, self.bias, out, act=self.activation, ids=ids)
        return out


class Sequential(Layer):
    def __init__(self, layers):
        super().__init__()
        self.layers = torch.nn.ModuleList(layers)

    def call(self, x, out=None, ids=None, **kwargs):
        x = to_device_tensor(x)
        n = len(self.layers)
        if n == 0:
            if ids is not None or out is not None:
                raise ValueError("an empty dense stack cannot gather or redirect its output")
            return x
        for k, layer in enumerate(self.layers):
            x = layer(x, out=out if k == n - 1 else None, ids=ids if k == 0 else None)
        return x

    # -- fused execution ------------------------------------------------------------------------
    def _packed(self):
        """(device blob, dims, activations) of the stack in MFMA fragment order, re-packed when weights change."""
        version = self.weights_version
        if getattr(self, '_pack_cache', None) is None or self._pack_cache[0] != version:
            blob, dims = capi.chain_pack([l.kernel.detach().cpu().numpy() for l in self.layers],
                                         [l.bias.detach().cpu().numpy() for l in self.layers])
            self._pack_cache = (version, torch.from_numpy(blob).to(self.layers[0].kernel.device), dims,
                                [l.activation for l in self.layers])
        return self._pack_cache[1:]

    def dims(self, in_dim):
        return [in_dim] + [l.units for l in self.layers]

    def apply2(self, a, b=None, ids_a=None, base_a=0, ids_b=None, base_b=0, out=None):
        """Run the stack on x = [a[ids_a - base_a] || b[ids_b - base_b]] (ids optional, b optional).

        One fused launch (`amar_chain_f32`) when every width fits the register-resident chain;
        otherwise gather/concatenate once and run `amar_dense_f32` layer by layer.
        """
        if len(self.layers) == 0:
            raise ValueError("an empty dense stack has nothing to apply")
        m = ids_a.numel() if ids_a is not None else a.shape[0]
        in_a, in_b = a.shape[1], (b.shape[1] if b is not None else 0)
        if not self.built:
            self.build_chain(in_a + in_b)
        dims = self.dims(in_a + in_b)
        if out is None:
            out = torch.empty((m, dims[-1]), dtype=torch.float32, device=a.device)
        out_ok = out.stride(0) % 4 == 0 or dims[-1] == 1
        if capi.chain_supported(dims, in_a, in_b) and out_ok:
            blob, _, acts = self._packed()
            capi.chain(a, blob, dims, acts, out, ids_a=ids_a, base_a=base_a, B=b, ids_b=ids_b, base_b=base_b)
            return out
        if isinstance(a, capi.ConcatTable):
            a = a.materialize()                                 # (only the fused chain reads per-layer tables in place)
        if b is None and base_a == 0:
            x, ids = a, ids_a                                   # the first dense layer gathers by itself
        else:
            x, ids = torch.empty((m, in_a + in_b), dtype=torch.float32, device=a.device), None
            capi.copy_columns(a, x[:, :in_a], ids=ids_a, base=base_a)
            if b is not None:
                capi.copy_columns(b, x[:, in_a:], ids=ids_b, base=base_b)
        n = len(self.layers)
        for k, layer in enumerate(self.layers):
            x = layer(x, out=out if k == n - 1 else None, ids=ids if k == 0 else None)
        return x

    def build_chain(self, in_dim):
        """Create every layer's weights for a known input width (no device work); returns the output width."""
        for layer in self.layers:
            if not layer.built:
                layer.build((None, in_dim))
                layer.built = True
            in_dim = layer.units
        self.built = True
        return in_dim

    @property
    def output_units(self):
        return self.layers[-1].units if len(self.layers) else None


def build_dense_network(units, **kwargs):
    return Sequential([Dense(u, **kwargs) for u in units])


def build_dense_classifier(units, n_classes, **kwargs):
    if n_classes != 1:
        raise NotImplementedError("softmax classifiers are not on the hot path (n_classes=1 everywhere in the reference)")
    clf_kwargs = dict(kwargs)
    clf_kwargs['activation'] = 'sigmoid'
    return Sequential([Dense(u, **kwargs) for u in units] + [Dense(n_classes, **clf_kwargs)])


def build_residual_dense_network(units, **kwargs):
    """dense.py:20-27: like build_dense_network, but the last layer has no activation (it is added to the skip paths)."""
    last_kwargs = dict(kwargs)
    last_kwargs['activation'] = None
    return Sequential([Dense(u, **kwargs) for u in units[:-1]] + [Dense(units[-1], **last_kwargs)])
