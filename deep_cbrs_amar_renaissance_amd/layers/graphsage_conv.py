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

    def _inv_count(self, a):
        """1 / (edges into the node (+1 with the self loop)); 0 for an empty segment, as unsorted_segment_mean yields."""
        cache = a.__dict__.setdefault('_sage_inv_count', {})
        if self.self_loops not in cache:
            deg = (a.rowptr[1:] - a.rowptr[:-1]).to(torch.float32)
            inv = 1.0 / (deg + 1.0) if self.self_loops else torch.where(deg > 0, 1.0 / deg.clamp(min=1.0), torch.zeros_like(deg))
            cache[self.self_loops] = inv.contiguous()
        return cache[self.self_loops]

    def wants_dense_input(self, a, f):
        """Whether the layer gathers on the LDS-tiled image with the fused tail: its input then should be a dense [n, f] table
        (a column slice of the concatenation buffer spreads four 32-byte rows over three 128-byte lines instead of one:
        ml1m(s=64) 0.31 against 0.25 ms per layer) and `dense_out` is filled by the same launch."""
        if spmm_kind(a, f) != 'xs' or f != self.channels or f not in (8, 16, 32):
            return False
        from deep_cbrs_amar_renaissance_amd.utilities.lds_tiled import LdsTiled
        return isinstance(a.tiled_mean_image(f, self.self_loops), LdsTiled)

    def call(self, inputs, out=None, dense_out=None, **kwargs):
        """dense_out: an optional dense [n, channels] buffer that receives a second copy of the result."""
        self._dense_filled = False
        y = self._call(inputs, out, dense_out)
        if dense_out is not None and not self._dense_filled:
            capi.copy_columns(y, dense_out)
        return y

    def _call(self, inputs, out, dense_out):
        x, a = inputs
        if a.vals is not None:
            raise ValueError("GraphSageConv expects the raw edge list (DeviceCSR without values)")
        n, f = a.shape[0], x.shape[1]
        if out is None:
            out = torch.empty((n, self.channels), dtype=torch.float32, device=x.device)
        kind = spmm_kind(a, f)
        if kind != 'xs' and f in (4, 8, 16, 32) and self.channels <= 64:
            capi.sage_layer(a.rowptr, a.colidx, x, self.kernel, self.bias, out, self_loop=self.self_loops)
            return out
        if kind == 'xs' and f == self.channels and f in (8, 16, 32):
            from deep_cbrs_amar_renaissance_amd.utilities.lds_tiled import LdsTiled
            img = a.tiled_mean_image(f, self.self_loops)
            if isinstance(img, LdsTiled) and (x.stride(0) == f or x.shape[0] * x.stride(0) * 4 < (1 << 32)):
                # mean aggregate and the layer's tail in ONE launch: the tile's sums never leave the workgroup
                capi.spmm_lt(img, x, out, prescaled=True, sage_tail=(self.kernel, self.bias), Hnext=dense_out)
                self._dense_filled = dense_out is not None
                return out
        fused_tail = capi.sage_tail_supported(f, self.channels)
        xa = torch.empty((n, f if fused_tail else 2 * f), dtype=torch.float32, device=x.device)
        agg = xa if fused_tail else xa[:, f:]
        if kind == 'xs':
            capi.spmm_xs(a.tiled_mean_image(f, self.self_loops), x, agg, prescaled=True)
        else:
            # widths the fused row kernel is not instantiated for (TwoStep / TwoWay 'concatenation' hand-over, 24 / 48):
            # neighbour sum as column chunks of the value-free SpMM, then (sum + own row) / count
            capi.spmm_csr(a.rowptr, a.colidx, None, x, agg)
            capi.row_affine(agg, self._inv_count(a), agg, b=x if self.self_loops else None)
        if fused_tail:
            capi.sage_tail(x, agg, self.kernel, self.bias, out)    # [x || agg] . W + b, l2-normalise, ReLU in one pass
            return out
        capi.copy_columns(x, xa[:, :f])
        z = torch.empty((n, self.channels), dtype=torch.float32, device=x.device)
        capi.dense(xa, self.kernel, self.bias, z, act=None)
        nrm = torch.empty_like(z)
        inv = torch.empty(n, dtype=torch.float32, device=x.device)
        capi.l2norm_fwd(z, nrm, inv, out, act='relu')
        return out
