"""Training step for the Basic GNN recommenders — what ``model.fit`` runs per batch.

Mirrors what Keras does under `Experimenter.train` (`/root/reference/src/experiment.py:155-188`):
binary cross-entropy (`config.yaml:58`) on the sigmoid scores of `BasicGNN.call` (full-graph
propagation re-run for EVERY batch, `basic.py:61-63`), plus the L2 regularisers carried by the
node table and the GCN kernels/biases (`gnn.py:45, 293-294`), differentiated and applied with
Adam (`config.yaml:52-56`; Keras defaults beta_2 = 0.999, epsilon = 1e-7).

Forward activations come from the same HIP kernels as inference; the reverse pass uses the
training kernels of `csrc/amar_train.hip` (activation backward, two-stage deterministic weight
gradients, row scatter-add for the embedding lookup, Adam) and reuses the forward SpMM for
A_hat^T . dZ (A_hat is symmetric) and the forward GEMM for dX = dZ . W^T.

Implemented for GCN, GraphSAGE, GAT, LightGCN and DGCF stacks under every reduction ('concatenation', 'mean', 'sum',
'w-sum', 'last'), alone (models/gnn.py) or chained as TwoStep / TwoWay models (models/tsgnn.py, twgnn.py): `_StackTape` is the
forward-with-kept-activations + reverse pass of ONE stack, and the Trainer chains the tapes the way the model chains the
stacks (the gradient of a stack's leading rows is lifted back to its full node table).  GraphSAGE trains on an unfused
forward that keeps what the reverse pass needs ([x || mean] and the normalised pre-activation); its aggregate
(A + I) / count is symmetric up to the row scale, so the reverse aggregate is the same value-free SpMM.  The hybrid head
(HybridCBRS: 'concatenate' / 'attention' fusion, residual classifier, both feature_based settings) trains on the same
Dense tapes; its BERT inputs are constants.  GAT (1 head) trains on the inference kernels plus `amar_gat_bwd_f32`, which
forms the softmax / attention-scalar gradients row-wise for both edge directions (symmetric edge multiset, no float
atomics).
"""
import os
import types

import numpy as np
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import ids_to_device, stage_ids, to_device_tensor
from deep_cbrs_amar_renaissance_amd.layers.dgcf_conv import DGCFConv
from deep_cbrs_amar_renaissance_amd.layers.gat_conv import GATConv
from deep_cbrs_amar_renaissance_amd.layers.gcn_conv import GCNConv
from deep_cbrs_amar_renaissance_amd.layers.graphsage_conv import GraphSageConv
from deep_cbrs_amar_renaissance_amd.layers.lightgcn_conv import LightGCNConv
from deep_cbrs_amar_renaissance_amd.utilities.math import spmm_kind


def _spmm(a, x, out):
    """out = A_hat . x with whichever image of A_hat the forward pass uses for this width."""
    kind = spmm_kind(a, x.shape[1])
    if kind == 'xs':
        capi.spmm_xs(a.tiled_image(x.shape[1]), x, out)
    elif kind == 'sj':
        capi.spmm_sj(a.sliced(x.shape[1]), x, out)
    else:
        capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, out)
    return out


class _DenseTape:
    """Forward of a Dense stack that keeps every layer's input and output, and its reverse pass."""

    def __init__(self, stack):
        self.layers = list(stack.layers)
        self.inputs, self.outputs = [], []
        self._workspaces = {}
        self.defer_reduce = False                                    # set by Trainer._graph_body (see capi.DeferredGradient)

    def _workspace(self, k, m, kk, n, device):
        """The workspace of layer k's fused reverse pass for batches of m rows (allocated at the first batch of that shape — an eager
        one — and reused by every later batch, captured ones included)."""
        key = (k, int(m), int(kk), int(n))
        if key not in self._workspaces:
            self._workspaces[key] = capi.dense_bwd_workspace(m, kk, n, device)
        return self._workspaces[key]

    def _stack_spec(self, x, ids, out_last):
        """The arguments of the one-launch forward (capi.dense_stack) for this call, or None where the stack runs layer by layer."""
        m = int(ids.numel()) if ids is not None else int(x.shape[0])
        dims = [int(self.layers[0].kernel.shape[0])] + [int(l.units) for l in self.layers]
        if not (capi.dense_stack_enabled() and capi.dense_stack_supported(dims) and m > 0):
            return None
        dev = x.device
        outs = [torch.empty((m, l.units), dtype=torch.float32, device=dev) for l in self.layers]
        if out_last is not None:
            outs[-1] = out_last
        xin = torch.empty((m, dims[0]), dtype=torch.float32, device=dev) if ids is not None else None
        return dict(X=x, weights=[l.kernel.detach() for l in self.layers], biases=[l.bias.detach() for l in self.layers],
                    acts=[l.activation for l in self.layers], outs=outs, ids=ids, xcopy=xin)

    def _stack_done(self, spec):
        outs = spec['outs']
        self.inputs, self.outputs = [spec['xcopy'] if spec['ids'] is not None else spec['X']] + outs[:-1], outs
        return outs[-1]

    @staticmethod
    def forward_pair(first, first_args, second, second_args):
        """Two independent stacks (the user and the item tower) in ONE launch where both take the one-launch forward
        (capi.dense_stack_pair; AMAR_DENSE_PAIR=0: one launch each).  *_args = (x, ids, out_last)."""
        s0, s1 = first._stack_spec(*first_args), second._stack_spec(*second_args)
        if s0 is not None and s1 is not None and os.environ.get('AMAR_DENSE_PAIR', '1') != '0':
            capi.dense_stack_pair(s0, s1)
            return first._stack_done(s0), second._stack_done(s1)
        return first.forward(*first_args), second.forward(*second_args)

    def forward(self, x, ids=None, out_last=None):
        """y = stack(x[ids]) keeping every layer's input and output.  ids: gather the input rows first; out_last: where the last
        layer's output goes (a column slice of a concatenation buffer).  Stacks of at most four layers no wider than 128 run as ONE
        launch (amar_dense_stack_f32, round 4: gather, layers and the concat store together); others layer by layer."""
        spec = self._stack_spec(x, ids, out_last)
        if spec is not None:
            capi.dense_stack(**spec)
            return self._stack_done(spec)
        m = int(ids.numel()) if ids is not None else int(x.shape[0])
        dev = x.device
        dims = [int(self.layers[0].kernel.shape[0])] + [int(l.units) for l in self.layers]
        outs = [torch.empty((m, l.units), dtype=torch.float32, device=dev) for l in self.layers]
        if out_last is not None:
            outs[-1] = out_last
        if ids is not None:
            gathered = torch.empty((m, dims[0]), dtype=torch.float32, device=dev)
            capi.copy_columns(x, gathered, ids=ids)
            x = gathered
        self.inputs, self.outputs = [], []
        for layer, y in zip(self.layers, outs):
            capi.dense(x, layer.kernel, layer.bias, y, act=layer.activation)
            self.inputs.append(x)
            self.outputs.append(y)
            x = y
        return x

    def _stack_bwd_spec(self, dy, last_is_dz, need_input_grad, dx_out):
        """The arguments of the one-launch reverse pass (capi.dense_stack_bwd) for this call, or None where it runs layer by layer."""
        m = int(dy.shape[0])
        dims = [int(self.layers[0].kernel.shape[0])] + [int(l.units) for l in self.layers]
        if not (capi.dense_bwd_enabled() and capi.dense_stack_bwd_supported(dims, m)):
            return None
        dev = dy.device
        key = ('stack', m)
        if key not in self._workspaces:
            self._workspaces[key] = capi.dense_stack_bwd_workspace(m, dims, dev)
        dws = [torch.empty_like(l.kernel) for l in self.layers]
        dbs = [torch.empty_like(l.bias) for l in self.layers]
        dx0 = (dx_out if dx_out is not None else torch.empty((m, dims[0]), dtype=torch.float32, device=dev)) if need_input_grad else None
        return dict(dYtop=dy, Ytop=None if last_is_dz else self.outputs[-1], inputs=self.inputs, weights=[l.kernel.detach() for l in self.layers],
                    acts=[l.activation for l in self.layers], workspace=self._workspaces[key], dWs=dws, dbs=dbs, dX0=dx0, defer=self.defer_reduce)

    def _stack_bwd_done(self, spec, lazy, grads):
        for k, layer in enumerate(self.layers):
            grads[layer.kernel], grads[layer.bias] = lazy[k] if lazy is not None else (spec['dWs'][k], spec['dbs'][k])
        return spec['dX0']

    @staticmethod
    def backward_pair(first, dy_first, second, dy_second, grads, need_input_grad=True, dx_out=(None, None)):
        """The reverse passes of two independent stacks in ONE launch where both take the one-launch form (capi.dense_stack_bwd_pair)."""
        s0 = first._stack_bwd_spec(dy_first, False, need_input_grad, dx_out[0])
        s1 = second._stack_bwd_spec(dy_second, False, need_input_grad, dx_out[1])
        if s0 is not None and s1 is not None and os.environ.get('AMAR_DENSE_PAIR', '1') != '0':
            lazy0, lazy1 = capi.dense_stack_bwd_pair(s0, s1)
            return first._stack_bwd_done(s0, lazy0, grads), second._stack_bwd_done(s1, lazy1, grads)
        return (first.backward(dy_first, grads, need_input_grad=need_input_grad, dx_out=dx_out[0]),
                second.backward(dy_second, grads, need_input_grad=need_input_grad, dx_out=dx_out[1]))

    def backward(self, dy, grads, last_is_dz=False, need_input_grad=True, dx_out=None):
        """dy: gradient w.r.t. the stack's output (or, with last_is_dz, already w.r.t. the last pre-activation).
        Fills grads[param] for every kernel/bias; returns the gradient w.r.t. the stack's input (None when
        need_input_grad is False: constant inputs such as the BERT rows)."""
        m = int(dy.shape[0])
        spec = self._stack_bwd_spec(dy, last_is_dz, need_input_grad, dx_out)
        if spec is not None:
            # the whole stack's reverse pass in ONE launch (amar_dense_stack_bwd_f32): dZ walks the layers in LDS
            return self._stack_bwd_done(spec, capi.dense_stack_bwd(**spec), grads)
        for k in range(len(self.layers) - 1, -1, -1):
            layer, x, y = self.layers[k], self.inputs[k], self.outputs[k]
            kk, n = layer.kernel.shape
            if capi.dense_bwd_enabled() and capi.dense_bwd_supported(kk, n) and x.shape[0] > 0:
                # act', dX, dW, db in two launches (amar_dense_bwd_f32; round 4 — four launches of 5-10 us each before)
                need_dx = not (k == 0 and not need_input_grad)
                act = None if (last_is_dz and k == len(self.layers) - 1) else layer.activation
                dw, db = torch.empty_like(layer.kernel), torch.empty_like(layer.bias)
                dx = (dx_out if (k == 0 and dx_out is not None) else torch.empty((x.shape[0], kk), dtype=torch.float32, device=x.device)) if need_dx else None
                lazy = capi.dense_bwd(x, y if act is not None else None, dy, layer.kernel.detach() if need_dx else None, act,
                                      self._workspace(k, x.shape[0], kk, n, x.device), dX=dx, dW=dw, db=db, defer=self.defer_reduce)
                # (defer_reduce: the partial sums stay in the workspace and the batch's ONE Adam launch adds them — no reduction launch)
                grads[layer.kernel], grads[layer.bias] = lazy if lazy is not None else (dw, db)
                if not need_dx:
                    return None
                dy = dx
                continue
            if last_is_dz and k == len(self.layers) - 1:
                dz = dy
            else:
                dz = torch.empty_like(y)
                capi.act_bwd(dy, y, dz, layer.activation)
            dw, db = torch.empty_like(layer.kernel), torch.empty_like(layer.bias)
            capi.wgrad(x, dz, dw, db)
            grads[layer.kernel], grads[layer.bias] = dw, db
            if k == 0 and not need_input_grad:
                return None
            dx = dx_out if (k == 0 and dx_out is not None) else torch.empty((x.shape[0], layer.kernel.shape[0]), dtype=torch.float32, device=x.device)
            capi.dense(dz, layer.kernel.detach(), None, dx, act=None, w_transposed=True)
            dy = dx
        return dy


def _concat(a, b):
    out = torch.empty((a.shape[0], a.shape[1] + b.shape[1]), dtype=torch.float32, device=a.device)
    capi.copy_columns(a, out[:, :a.shape[1]])
    capi.copy_columns(b, out[:, a.shape[1]:])
    return out


class _BasicHead:
    """BasicRS (basic.py:11-37) with saved activations: towers on E[u], E[i]; classifier on their concatenation."""

    def __init__(self, rs):
        self.unet, self.inet, self.clf = _DenseTape(rs.unet), _DenseTape(rs.inet), _DenseTape(rs.clf)

    def forward(self, gu, gi, bert, ids=None):
        """ids = (u, i): gu and gi are node tables and the towers gather their rows themselves; the towers' last layers store straight
        into the two halves of the classifier's input (no concat copies)."""
        d = int(self.unet.layers[-1].units)
        b = int(ids[0].numel()) if ids is not None else int(gu.shape[0])
        cat = torch.empty((b, 2 * d), dtype=torch.float32, device=gu.device)
        _DenseTape.forward_pair(self.unet, (gu, ids[0] if ids is not None else None, cat[:, :d]),
                                self.inet, (gi, ids[1] if ids is not None else None, cat[:, d:]))      # (both towers: one launch)
        self.d = d
        return self.clf.forward(cat)

    def backward(self, dz, grads, need_input_grad=True, dx_out=(None, None)):
        """dz = dL/d(last pre-activation). Returns (dL/dE[u], dL/dE[i]) (None, None when the inputs are constants); dx_out: where to."""
        dcat = self.clf.backward(dz, grads, last_is_dz=True)
        return _DenseTape.backward_pair(self.unet, dcat[:, :self.d], self.inet, dcat[:, self.d:], grads, need_input_grad=need_input_grad, dx_out=dx_out)


class _FusionTape:
    """FusionLayer (fusion.py:5-68) with saved operands: concatenation, or the attention mix with its two weights."""

    def __init__(self, fuse):
        self.fuse = fuse

    def plan_joined(self, rows, wa, wb, device):
        """For a concatenating fusion: (buffer, left half, right half) for the producers of its operands to store into, else
        (None, None, None) — round 4: the two column copies of every `Concatenate` of a hybrid head were 6 of a batch's 54 launches
        (AMAR_FUSION_INPLACE=0: copies)."""
        if self.fuse.method != 'concatenate' or os.environ.get('AMAR_FUSION_INPLACE', '1') == '0':
            return None, None, None
        buf = torch.empty((rows, wa + wb), dtype=torch.float32, device=device)
        return buf, buf[:, :wa], buf[:, wa:]

    def forward(self, a, b, joined=None):
        """joined: the [rows, da + db] buffer whose two column halves a and b ALREADY are (their producers stored straight into
        it: `plan_joined`) — a concatenation then has nothing to copy."""
        f = self.fuse
        self.da, self.db = a.shape[1], b.shape[1]
        if f.method == 'concatenate':
            return joined if joined is not None else _concat(a, b)
        pa, pb = f.project(a, b)
        ta, tb = torch.empty_like(pa), torch.empty_like(pb)
        capi.dense(pa, f.att_weight, None, ta, act=None)
        capi.dense(pb, f.att_weight, None, tb, act=None)
        out = torch.empty_like(pa)
        capi.attention_mix(pa, pb, ta, tb, out)
        self.saved = (a, b, pa, pb, ta, tb)
        return out

    def backward(self, dout, grads):
        """Returns (dL/da, dL/db); fills grads for att_weight / proj_weight."""
        f = self.fuse
        if f.method == 'concatenate':
            return dout[:, :self.da], dout[:, self.da:]
        a, b, pa, pb, ta, tb = self.saved
        d_a, d_b, d_ta, d_tb = capi.attention_mix_bwd(dout, pa, pb, ta, tb)
        w = f.att_weight.detach()
        dw, dw2 = torch.empty_like(w), torch.empty_like(w)
        capi.wgrad(pa, d_ta, dw, None)
        capi.wgrad(pb, d_tb, dw2, None)
        capi.add_inplace(dw, dw2)
        grads[f.att_weight] = dw
        back = torch.empty_like(d_a)
        capi.dense(d_ta, w, None, back, act=None, w_transposed=True)
        capi.add_inplace(d_a, back)
        capi.dense(d_tb, w, None, back, act=None, w_transposed=True)
        capi.add_inplace(d_b, back)
        if f.proj_first is not None:                                 # the narrower block went through proj_weight first
            src, d_proj = (a, d_a) if f.proj_first else (b, d_b)
            dp = torch.empty_like(f.proj_weight)
            capi.wgrad(src, d_proj, dp, None)
            grads[f.proj_weight] = dp
            d_src = torch.empty((src.shape[0], src.shape[1]), dtype=torch.float32, device=src.device)
            capi.dense(d_proj, f.proj_weight.detach(), None, d_src, act=None, w_transposed=True)
            if f.proj_first:
                d_a = d_src
            else:
                d_b = d_src
        return d_a, d_b


class _HybridHead:
    """HybridCBRS (hybrid.py:13-89) with saved activations: both feature_based settings, 'concatenate' / 'attention'
    fusion, optional residual classifier.  The BERT rows are constants."""

    def __init__(self, rs):
        self.rs = rs
        self.fb = bool(rs.feature_based)
        names = ['dense1a', 'dense1b', 'dense2a', 'dense2b', 'dense3a', 'dense3b', 'clf'] + (['residual'] if rs.residual is not None else [])
        self.t = {name: _DenseTape(getattr(rs, name)) for name in names}
        self.f1a, self.f1b, self.f2 = _FusionTape(rs.fuse1a), _FusionTape(rs.fuse1b), _FusionTape(rs.fuse2)

    def forward(self, gu, gi, bert):
        ub, ib = bert
        t = self.t
        rows, dev = int(gu.shape[0]), gu.device
        w = {name: int(t[name].layers[-1].units) for name in ('dense1a', 'dense1b', 'dense2a', 'dense2b', 'dense3a', 'dense3b')}
        # feature based: (graph user, graph item) | (bert user, bert item); else per entity (hybrid.py:72-84).  Concatenating fusions
        # get their operands stored straight into the two halves of their output by the stacks that make them.
        feeds = (('dense1a', 'dense1b'), ('dense2a', 'dense2b')) if self.fb else (('dense1a', 'dense2a'), ('dense1b', 'dense2b'))
        ja, la, ra = self.f1a.plan_joined(rows, w[feeds[0][0]], w[feeds[0][1]], dev)
        jb, lb, rb = self.f1b.plan_joined(rows, w[feeds[1][0]], w[feeds[1][1]], dev)
        dest = {feeds[0][0]: la, feeds[0][1]: ra, feeds[1][0]: lb, feeds[1][1]: rb}
        g1, g2 = _DenseTape.forward_pair(t['dense1a'], (gu, None, dest['dense1a']), t['dense1b'], (gi, None, dest['dense1b']))      # (independent stacks: one launch)
        b1, b2 = _DenseTape.forward_pair(t['dense2a'], (ub, None, dest['dense2a']), t['dense2b'], (ib, None, dest['dense2b']))
        ins = ((g1, g2), (b1, b2)) if self.fb else ((g1, b1), (g2, b2))
        fa, fb = self.f1a.forward(*ins[0], joined=ja), self.f1b.forward(*ins[1], joined=jb)
        jc, lc, rc = self.f2.plan_joined(rows, w['dense3a'], w['dense3b'], dev) if 'residual' not in t else (None, None, None)
        x1, x2 = _DenseTape.forward_pair(t['dense3a'], (fa, None, lc), t['dense3b'], (fb, None, rc))
        x = self.f2.forward(x1, x2, joined=jc)
        if 'residual' in t:                                          # hybrid.py:86-89
            r = t['residual'].forward(x)
            self.s = torch.empty_like(r)
            capi.add3_act(r, x1, x2, self.s, act=self.rs.activation)
            x = self.s
        return t['clf'].forward(x)

    def backward(self, dz, grads, need_input_grad=True):
        t = self.t
        dx = t['clf'].backward(dz, grads, last_is_dz=True)
        skip = None
        if 'residual' in t:
            skip = torch.empty_like(dx)
            capi.act_bwd(dx, self.s, skip, self.rs.activation)       # d(residual(x) + x1 + x2): the same for all three terms
            dx = t['residual'].backward(skip, grads, last_is_dz=True)
        dx1, dx2 = self.f2.backward(dx, grads)
        if skip is not None:
            dx1, dx2 = dx1.contiguous().clone(), dx2.contiguous().clone()
            capi.add_inplace(dx1, skip)
            capi.add_inplace(dx2, skip)
        d3a, d3b = _DenseTape.backward_pair(t['dense3a'], dx1, t['dense3b'], dx2, grads)
        da = self.f1a.backward(d3a, grads)
        db = self.f1b.backward(d3b, grads)
        if self.fb:
            (dg1, dg2), (db1, db2) = da, db
        else:
            (dg1, db1), (dg2, db2) = da, db
        t['dense2a'].backward(db1, grads, need_input_grad=False)
        t['dense2b'].backward(db2, grads, need_input_grad=False)
        return _DenseTape.backward_pair(t['dense1a'], dg1, t['dense1b'], dg2, grads, need_input_grad=need_input_grad)


class _StackTape:
    """Forward of ONE convolution stack (SequentialGNN / HalfInput / FullInputSequentialGNN) that keeps what its reverse
    pass needs, and that reverse pass: d(loss)/d(reduced output) -> weight gradients + d(loss)/d(node table).

    Every layer's output lives in a column slice of one [N, sum(widths)] buffer `cat`; the reduction (reduction.py:9-33)
    is undone first (d_cat = d_out for 'concatenation', d_out / (L+1) per slice for 'mean', ...), then the layers run
    in reverse on the slices of (cat, d_cat).  All adjacency images are symmetric (config.yaml:36), so A^T . g reuses
    the forward SpMM."""

    KINDS = ((GCNConv, 'gcn'), (LightGCNConv, 'lightgcn'), (GraphSageConv, 'sage'), (GATConv, 'gat'), (DGCFConv, 'dgcf'))

    def __init__(self, seq):
        self.seq = seq
        layers = list(seq.seq_layers)
        self.kind = next((name for cls, name in self.KINDS if layers and all(isinstance(l, cls) for l in layers)), None)
        if self.kind is None:
            raise NotImplementedError("training needs a stack of one layer type (GCN, GraphSAGE, GAT, LightGCN or DGCF)")
        if seq.final_node not in ('concatenation', 'mean', 'sum', 'last', 'w-sum'):
            raise NotImplementedError("no reverse pass for the '{}' reduction".format(seq.final_node))
        if self.kind == 'sage':
            a = seq.adj_matrix
            deg = (a.rowptr[1:] - a.rowptr[:-1]).to(torch.float32)
            if len({bool(l.self_loops) for l in layers}) != 1:
                raise NotImplementedError("GraphSAGE layers with mixed self_loops settings")
            self.self_loops = bool(layers[0].self_loops)
            # unsorted_segment_mean: sum / count, 0 for an empty segment
            self.inv_cnt = (1.0 / (deg + 1.0)) if self.self_loops else torch.where(deg > 0, 1.0 / deg.clamp(min=1.0), torch.zeros_like(deg))
            self.inv_cnt = self.inv_cnt.contiguous()
        self.cat = self.tape = None
        self._workspaces = {}
        self.defer_reduce = False

    def _workspace(self, k, m, kk, n, device):
        key = (k, int(m), int(kk), int(n))
        if key not in self._workspaces:
            self._workspaces[key] = capi.dense_bwd_workspace(m, kk, n, device)
        return self._workspaces[key]

    def _slices(self, t):
        offs = self.offs
        return lambda k: t[:, offs[k]:offs[k + 1]]

    # -- forward ----------------------------------------------------------------------------------------------------
    def forward(self, x0=None):
        """Reduced node representations of the stack over the node table x0 (None: the stack's own table)."""
        seq = self.seq
        x0 = seq.embeddings if x0 is None else x0
        widths = self.widths = seq.layer_widths()
        seq._build_layers(widths)
        self.offs = [int(v) for v in np.cumsum([0] + widths)]
        if self.kind in ('gcn', 'lightgcn'):
            # the inference kernels: their outputs are all the reverse pass needs
            out, self.cat = seq._propagate(x0, with_layers=True)
            return out
        a = seq.adj_matrix
        n, dev = a.shape[0], x0.device
        cat = self.cat = torch.empty((n, self.offs[-1]), dtype=torch.float32, device=dev)
        sl = self._slices(cat)
        capi.copy_columns(x0, sl(0))
        self.tape = []
        for k, layer in enumerate(seq.seq_layers):
            f, c = widths[k], widths[k + 1]
            if self.kind == 'sage':
                # layer by layer, keeping [x || mean(x)] and the l2-normalised pre-activation
                xa = torch.empty((n, 2 * f), dtype=torch.float32, device=dev)
                capi.copy_columns(sl(k), xa[:, :f])
                ssum = torch.empty((n, f), dtype=torch.float32, device=dev)
                capi.spmm_csr(a.rowptr, a.colidx, None, sl(k), ssum)
                capi.row_affine(ssum, self.inv_cnt, xa[:, f:], b=sl(k) if self.self_loops else None)
                z = torch.empty((n, c), dtype=torch.float32, device=dev)
                capi.dense(xa, layer.kernel, layer.bias, z, act=None)
                nrm = torch.empty((n, c), dtype=torch.float32, device=dev)
                inv = torch.empty(n, dtype=torch.float32, device=dev)
                capi.l2norm_fwd(z, nrm, inv, sl(k + 1), act='relu')
                self.tape.append((xa, nrm, inv))
            elif self.kind == 'gat':
                # same kernels as inference, keeping H and the two attention scalars
                h = torch.empty((n, c), dtype=torch.float32, device=dev)
                s_self = torch.empty(n, dtype=torch.float32, device=dev)
                s_neigh = torch.empty(n, dtype=torch.float32, device=dev)
                capi.rowwise_xw(sl(k), layer.kernel.view(-1, c), h, a_self=layer.attn_kernel_self.view(c),
                                a_neigh=layer.attn_kernel_neighs.view(c), s_self=s_self, s_neigh=s_neigh)
                capi.gat_layer(a.rowptr, a.colidx, h, s_self, s_neigh, layer.bias, sl(k + 1), self_loop=layer.add_self_loops)
                self.tape.append((h, s_self, s_neigh))
            else:                                                    # dgcf: every layer's input stays in `cat` (the gate's gradient needs it)
                layer([sl(k), a], out=sl(k + 1))
        return seq._reduce(cat, [sl(k) for k in range(len(widths))], widths)

    # -- reverse ----------------------------------------------------------------------------------------------------
    def _expand(self, d_out, grads):
        """d(loss)/d(cat) from d(loss)/d(reduced output) ('w-sum': also the gradient of the reduction weights, into `grads`)."""
        final, widths = self.seq.final_node, self.widths
        n_terms = len(widths)
        if final == 'concatenation':
            return d_out
        if final == 'w-sum':
            w = self.seq.reduce.w
            d_cat = torch.empty((d_out.shape[0], self.offs[-1]), dtype=torch.float32, device=d_out.device)
            dw = torch.empty(n_terms, dtype=torch.float32, device=d_out.device)
            capi.reduce_layers_wsum_bwd(self.cat, n_terms, widths[0], w.view(-1), d_out, d_cat, dw)
            grads[w] = dw.view_as(w)
            return d_cat
        d_cat = torch.zeros((d_out.shape[0], self.offs[-1]), dtype=torch.float32, device=d_out.device)
        sl = self._slices(d_cat)
        if final == 'last':
            capi.add_inplace(sl(n_terms - 1), d_out)
        else:
            for k in range(n_terms):
                capi.add_inplace(sl(k), d_out, 1.0 / n_terms if final == 'mean' else 1.0)
        return d_cat

    def backward(self, d_out, grads):
        """Fills `grads` for the layers' weights; returns d(loss)/d(node table) [N, widths[0]] (a fresh buffer).
        `d_out` is consumed (it may be modified in place)."""
        seq, a = self.seq, self.seq.adj_matrix
        layers, widths = list(seq.seq_layers), self.widths
        n, dev = d_out.shape[0], d_out.device
        if self.kind == 'lightgcn' and self.cat is None:
            # running-sum route ('mean'): g0 = (I + A + A^2 + ...) d_out / (L + 1)
            n_terms = len(layers) + 1
            g0 = torch.zeros((n, widths[0]), dtype=torch.float32, device=dev)
            capi.add_inplace(g0, d_out, 1.0 / n_terms)
            acc = g0.clone()
            for _ in layers:
                nxt = torch.empty_like(acc)
                _spmm(a, acc, nxt)
                capi.add_inplace(g0, nxt)
                acc = nxt
            return g0
        e, de = self.cat, self._expand(d_out, grads)
        sl, dsl = self._slices(e), self._slices(de)
        for k in range(len(layers) - 1, -1, -1):
            layer = layers[k]
            f, c = widths[k], widths[k + 1]
            if self.kind == 'gcn':
                dzk = torch.empty((n, c), dtype=torch.float32, device=dev)
                dw, db = torch.empty_like(layer.kernel), torch.empty_like(layer.bias)
                fused = capi.dense_bwd_enabled() and capi.dense_bwd_supported(f, c) and n > 0
                if fused:                                             # act', its bias gradient and dZ in one launch
                    lazy = capi.dense_bwd(None, sl(k + 1), dsl(k + 1), None, 'relu', self._workspace(('b', k), n, 1, c, dev), db=db, dZ=dzk,
                                          defer=self.defer_reduce, K=1)
                    if lazy is not None:
                        db = lazy[1]
                else:
                    capi.act_bwd(dsl(k + 1), sl(k + 1), dzk, 'relu')
                dh = torch.empty((n, c), dtype=torch.float32, device=dev)
                _spmm(a, dzk, dh)                                     # A_hat^T = A_hat
                if fused:                                             # dW = X_k^T . dH, and dH . W^T added straight into the slice's gradient
                    lazy = capi.dense_bwd(sl(k), None, dh, layer.kernel.detach(), None, self._workspace(k, n, f, c, dev), dX=dsl(k), dW=dw,
                                          defer=self.defer_reduce, accumulate_dx=True)
                    if lazy is not None:
                        dw = lazy[0]
                else:
                    back = torch.empty((n, f), dtype=torch.float32, device=dev)
                    capi.wgrad(sl(k), dh, dw, None)
                    capi.dense(dh, layer.kernel.detach(), None, back, act=None, w_transposed=True)
                    capi.wgrad(None, dzk, None, db)
                    capi.add_inplace(dsl(k), back)
                grads[layer.kernel], grads[layer.bias] = dw, db
            elif self.kind == 'lightgcn':
                back = torch.empty((n, f), dtype=torch.float32, device=dev)
                _spmm(a, dsl(k + 1), back)
                capi.add_inplace(dsl(k), back)
            elif self.kind == 'sage':
                xa, nrm, inv = self.tape[k]
                dz = torch.empty((n, c), dtype=torch.float32, device=dev)
                capi.l2norm_bwd(dsl(k + 1), nrm, inv, dz, act='relu')
                dw, db = torch.empty_like(layer.kernel), torch.empty_like(layer.bias)
                dxa = torch.empty((n, 2 * f), dtype=torch.float32, device=dev)
                if capi.dense_bwd_enabled() and capi.dense_bwd_supported(2 * f, c) and n > 0:     # dW, db and dZ . W^T in one launch (round 4)
                    lazy = capi.dense_bwd(xa, None, dz, layer.kernel.detach(), None, self._workspace(k, n, 2 * f, c, dev), dX=dxa, dW=dw, db=db,
                                          defer=self.defer_reduce)
                    if lazy is not None:
                        dw, db = lazy
                else:
                    capi.wgrad(xa, dz, dw, db)
                    capi.dense(dz, layer.kernel.detach(), None, dxa, act=None, w_transposed=True)
                grads[layer.kernel], grads[layer.bias] = dw, db
                capi.add_inplace(dsl(k), dxa[:, :f])
                g = torch.empty((n, f), dtype=torch.float32, device=dev)
                capi.row_affine(dxa[:, f:], self.inv_cnt, g)               # d(mean)/d(sum)
                back = torch.empty((n, f), dtype=torch.float32, device=dev)
                capi.spmm_csr(a.rowptr, a.colidx, None, g, back)           # the edge multiset is symmetric
                capi.add_inplace(dsl(k), back)
                if self.self_loops:
                    capi.add_inplace(dsl(k), g)
            elif self.kind == 'gat':
                h, s_self, s_neigh = self.tape[k]
                w2d = layer.kernel.detach().view(f, c)
                dout, ds, dt, dh = capi.gat_bwd(a.rowptr, a.colidx, h, s_self, s_neigh, sl(k + 1), dsl(k + 1), layer.bias,
                                                layer.attn_kernel_self.detach().view(c), layer.attn_kernel_neighs.detach().view(c),
                                                self_loop=layer.add_self_loops)
                db = torch.empty_like(layer.bias)
                das, dan = torch.empty((c, 1), dtype=torch.float32, device=dev), torch.empty((c, 1), dtype=torch.float32, device=dev)
                fused = capi.dense_bwd_enabled() and capi.dense_bwd_supported(f, c) and n > 0
                if fused:                                             # H^T . ds as (ds^T . H)^T: a [1, c] weight gradient with X = ds, dZ = H — one launch each
                    lazy_s = capi.dense_bwd(ds.view(n, 1), None, h, None, None, self._workspace(('s', k), n, 1, c, dev), dW=das.view(1, c), defer=self.defer_reduce)
                    lazy_t = capi.dense_bwd(dt.view(n, 1), None, h, None, None, self._workspace(('t', k), n, 1, c, dev), dW=dan.view(1, c), defer=self.defer_reduce)
                    if lazy_s is not None:
                        das, dan = lazy_s[0], lazy_t[0]
                else:
                    capi.wgrad(h, ds.view(n, 1), das, None)
                    capi.wgrad(h, dt.view(n, 1), dan, None)
                dw = torch.empty((f, c), dtype=torch.float32, device=dev)
                if fused:
                    # round 4: the bias gradient in one launch, and dW = X_k^T . dH with dH . W^T added straight into the slice's gradient in
                    # one more (ten launches of a layer's reverse pass were weight-gradient partials and their reductions)
                    lazy_b = capi.dense_bwd(None, None, dout, None, None, self._workspace(('b', k), n, 1, c, dev), db=db, defer=self.defer_reduce, K=1)
                    lazy_w = capi.dense_bwd(sl(k), None, dh, w2d.contiguous(), None, self._workspace(k, n, f, c, dev), dX=dsl(k), dW=dw,
                                            defer=self.defer_reduce, accumulate_dx=True)
                    grads[layer.kernel] = lazy_w[0] if lazy_w is not None else dw.view_as(layer.kernel)
                    grads[layer.bias] = lazy_b[1] if lazy_b is not None else db
                else:
                    capi.wgrad(None, dout, None, db)
                    capi.wgrad(sl(k), dh, dw, None)
                    grads[layer.kernel], grads[layer.bias] = dw.view_as(layer.kernel), db
                    back = torch.empty((n, f), dtype=torch.float32, device=dev)
                    capi.dense(dh, w2d.contiguous(), None, back, act=None, w_transposed=True)
                    capi.add_inplace(dsl(k), back)
                grads[layer.attn_kernel_self] = das if isinstance(das, capi.DeferredGradient) else das.view_as(layer.attn_kernel_self)
                grads[layer.attn_kernel_neighs] = dan if isinstance(dan, capi.DeferredGradient) else dan.view_as(layer.attn_kernel_neighs)
            else:                                                    # dgcf
                back = torch.empty((n, f), dtype=torch.float32, device=dev)
                _spmm(a, dsl(k + 1), back)                # A_dgcf is symmetric
                dw = torch.empty(n, dtype=torch.float32, device=dev)
                capi.locality_scale_bwd(back, sl(k), layer.w.detach().view(-1), dsl(k), dw, accumulate=True)
                grads[layer.w] = dw.view_as(layer.w)
        g0 = torch.empty((n, widths[0]), dtype=torch.float32, device=dev)
        capi.copy_columns(dsl(0), g0)
        self.tape = self.cat = None
        return g0


def _require_symmetric(a):
    """The reverse pass of every stack reuses the forward product (or edge list) as its transpose: A^T = A.  That holds for
    `dataset.symmetric_adjacency: True` (config.yaml:36, every econfig); with False (preprocess.py:91 returns the
    un-symmetrised matrix) the gradients would be silently wrong, so such a graph is refused.  One host check per graph."""
    if getattr(a, '_symmetric_checked', None) is None:
        m = a.to_scipy()
        a._symmetric_checked = m.shape[0] == m.shape[1] and abs(m - m.T).max() <= 1e-6 * max(abs(m).max(), 1e-30)
    if not a._symmetric_checked:
        raise NotImplementedError("training needs a symmetric adjacency matrix (dataset.symmetric_adjacency: True): the reverse "
                                  "pass multiplies by A where A^T is due")


class Trainer:
    """Holds the Adam state of a Basic* / HybridBert* model (single-graph, TwoStep or TwoWay stacks) and performs training batches."""

    def __init__(self, model, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, bert_dim=None):
        gnn = model.gnn
        if hasattr(gnn, 'gnn_layers'):                               # one graph (gnn.py:210-264)
            self.layout, stacks = 'single', [gnn.gnn_layers]
        elif hasattr(gnn, 'step_one_gnn_layers'):                    # TwoStep (tsgnn.py:99-101)
            self.layout, stacks = 'two_step', [gnn.step_one_gnn_layers, gnn.step_two_gnn_layers]
        elif hasattr(gnn, 'way_one_gnn_layers'):                     # TwoWay (twgnn.py:98-105)
            self.layout, stacks = 'two_way', [gnn.way_one_gnn_layers, gnn.way_two_gnn_layers, gnn.step_two_gnn_layers]
        else:
            raise NotImplementedError("no training recipe for {}".format(type(gnn).__name__))
        for seq_ in stacks:
            _require_symmetric(seq_.adj_matrix)
        self.tapes = [_StackTape(seq) for seq in stacks]
        self.kind = self.tapes[-1].kind
        seq = stacks[-1]
        self.hybrid = hasattr(model.rs, 'dense1a')
        if not model.rs.built:
            if self.hybrid:
                if getattr(model, 'bert_table', None) is None and bert_dim is None:
                    raise ValueError("the hybrid head is not built yet: register the BERT table or pass bert_dim")
                model.rs.build_head(model.gnn.output_dim(), bert_dim if bert_dim is not None else model.bert_table.shape[1])
            else:
                model.rs.build_head(model.gnn.output_dim(), model.gnn.output_dim())
        self.model, self.seq = model, seq
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        self.t = 0
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.device = self.params[0].device
        self.m = {p: torch.zeros_like(p) for p in self.params}
        self.v = {p: torch.zeros_like(p) for p in self.params}
        self.head = _HybridHead(model.rs) if self.hybrid else _BasicHead(model.rs)

    @staticmethod
    def _l2(param):
        reg = getattr(param, 'regularizer', None)
        return float(reg.l2) if reg is not None else 0.0

    # -- one batch ------------------------------------------------------------------------------------------------
    def _bert_rows(self, ids, block):
        if block is not None:
            return block if isinstance(block, torch.Tensor) and block.is_cuda else to_device_tensor(block)
        table = getattr(self.model, 'bert_table', None)
        if table is None:
            raise ValueError("no BERT block in the batch and no resident table registered")
        rows = torch.empty((ids.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
        capi.copy_columns(table, rows, ids=ids)
        return rows

    def _forward_backward(self, u, i, yv, rows, ui=None):
        """Device-only body of a batch (no host synchronisation, fixed shapes -> capturable as a hipGraph):
        returns (per-pair loss terms [B], {param: gradient})."""
        b = u.numel()
        dev = self.device
        e = self._propagation_forward()                              # full-graph propagation, every batch (basic.py:61-63)
        f = e.shape[1]
        if isinstance(self.head, _BasicHead) and not self.hybrid:    # the towers gather E[u], E[i] themselves (one launch per stack)
            p = self.head.forward(e, e, rows, ids=(u, i))
        else:
            gu = torch.empty((b, f), dtype=torch.float32, device=dev)
            gi = torch.empty((b, f), dtype=torch.float32, device=dev)
            capi.copy_columns(e, gu, ids=u)
            capi.copy_columns(e, gi, ids=i)
            if self.hybrid:
                rows = (self._bert_rows(u, rows[0] if rows else None), self._bert_rows(i, rows[1] if rows else None))
            p = self.head.forward(gu, gi, rows)
        # ---- loss and its gradient through the final sigmoid
        dz = torch.empty((b, 1), dtype=torch.float32, device=dev)
        terms = torch.empty(b, dtype=torch.float32, device=dev)
        capi.bce_grad(p, yv, dz, terms)
        grads = {}
        de = torch.zeros((e.shape[0], f), dtype=torch.float32, device=dev)
        both = None
        if ui is not None and isinstance(self.head, _BasicHead) and not self.hybrid and 2 * b <= 8192:
            # ui = [u ; i] (a replayed batch's id buffer): the towers leave their input gradients in the two halves of one
            # buffer and ONE launch adds them to the node table's gradient (user and item ids never meet: same sums, same order)
            both = torch.empty((2 * b, f), dtype=torch.float32, device=dev)
            self.head.backward(dz, grads, dx_out=(both[:b], both[b:]))
            capi.scatter_add_rows(both, ui, de)
        else:
            dgu, dgi = self.head.backward(dz, grads)
            capi.scatter_add_rows(dgu, u, de)
            capi.scatter_add_rows(dgi, i, de)
        self._propagation_backward(e, de, grads)
        return terms, grads

    def loss_and_grads(self, u_ids, i_ids, y, bert=None):
        """Forward + reverse pass of one batch. Returns (data loss + regularisation loss, {param: gradient}).
        `bert` = (user block, item block) for the hybrid head (None: rows of the resident table)."""
        n_nodes = self.seq.adj_matrix.shape[0] if getattr(self, 'seq', None) is not None else None
        u, i = ids_to_device(u_ids, n_nodes), ids_to_device(i_ids, n_nodes)
        yv = to_device_tensor(np.asarray(y, dtype=np.float32) if not isinstance(y, torch.Tensor) else y)
        with torch.no_grad():
            terms, grads = self._forward_backward(u, i, yv, bert)
            loss = float(terms.sum().item()) / u.numel()
            for prm in self.params:
                l2 = self._l2(prm)
                if l2:
                    loss += l2 * float((prm.detach().double() ** 2).sum().item())
        return loss, grads

    # -- one batch as a hipGraph ------------------------------------------------------------------------------------
    def _graph_body(self):
        g = self._g
        tapes = self._all_tapes()
        for t in tapes:                                              # weight-gradient partials stay partial: the Adam launch below adds them
            t.defer_reduce = True
        try:
            terms, grads = self._forward_backward(g['u'], g['i'], g['y'], (g['ub'], g['ib']) if g['ub'] is not None else None, ui=g['ui'])
        finally:
            for t in tapes:
                t.defer_reduce = False
        capi.adam_advance(self._adam_state, self.lr, self.b1, self.b2)
        # one launch updates every parameter (a table of slots, uploaded by a captured copy from pinned memory: the
        # gradient buffers of this graph have fixed addresses) and adds the regularisation loss; one more adds the data loss
        entries = [(prm.data.view(-1), grads[prm] if isinstance(grads[prm], capi.DeferredGradient) else grads[prm].contiguous().view(-1),
                    self.m[prm].view(-1), self.v[prm].view(-1), self._l2(prm)) for prm in self.params]
        host, blocks = capi.adam_slot_table(entries)
        g['slot_host'][:host.numel()].copy_(host)                    # pinned buffer allocated before the capture began
        g['slot_bytes'] = int(host.numel())                          # uploaded ONCE, right after the capture (train_batch_graphed), into a
        g['keep'] = entries                                          # buffer allocated BEFORE it (memory of the capture's own pool is reused
        #                                                              by the graph's temporaries): the table never changes — the slots point
        #                                                              into tensors of the graph
        batch = float(g['u'].numel())
        capi.sum_into(terms, self._loss_sum)                         # sum of the per-pair terms = data loss x batch size
        capi.adam_multi(g['slot_dev'], len(entries), blocks, self._adam_state, self.b1, self.b2, self.eps,
                        reg_scale=batch, loss_acc=self._loss_sum)

    def train_batch_graphed(self, u_ids, i_ids, y, bert=None):
        """One training batch replayed from a hipGraph: the forward, the reverse pass and the Adam update are ~100
        small launches that are otherwise bound by host launch time.  The graph is captured at the second batch of a
        given size (the first one runs eagerly and warms every lazily built buffer); the running loss stays on the
        device (`pop_loss_sum`).  Batches of another size run eagerly."""
        b = len(y)
        dev = self.device
        with_blocks = bert is not None and bert[0] is not None
        key = (b, with_blocks)
        if not hasattr(self, '_graphs'):
            self._graphs, self._seen, self._eager_loss, self._dev_t = {}, set(), 0.0, None
            self._adam_state = torch.zeros(2, dtype=torch.float32, device=dev)
            self._loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
        g = self._graphs.get(key)
        if g is None:
            if key not in self._seen:                               # first batch of this shape: eager (real) step
                self._seen.add(key)
                self._eager_loss += self.train_batch(u_ids, i_ids, y, bert=bert) * b
                return
            d = int(np.asarray(bert[0]).shape[1]) if with_blocks else 0
            uiy = torch.zeros(3 * b, dtype=torch.int32, device=dev)  # u, i and the labels in ONE buffer: one upload per batch instead of three
            ui = uiy[:2 * b]                                          # (u and i side by side: one scatter of both towers' input gradients)
            g = self._g = {'uiy': uiy, 'ui': ui, 'u': ui[:b], 'i': ui[b:],
                           'y': uiy[2 * b:].view(torch.float32),
                           'ub': torch.zeros((b, d), dtype=torch.float32, device=dev) if with_blocks else None,
                           'ib': torch.zeros((b, d), dtype=torch.float32, device=dev) if with_blocks else None}
            g['slot_host'] = torch.empty(64 * len(self.params) + 64, dtype=torch.uint8).pin_memory()   # >= sizeof(amar_adam_slot) per parameter
            g['slot_dev'] = torch.empty(64 * len(self.params) + 64, dtype=torch.uint8, device=dev)
            # pinned staging for the batch's ids and labels, four sets in turn: the uploads are asynchronous, so the host prepares
            # batch k + 1 while the device still runs batch k (a pageable copy_ made the host wait for the stream every batch:
            # 0.13 ms of a 0.52 ms batch at ml1m(s=1))
            g['stage'] = []
            for _ in range(4):
                host = torch.empty(3 * b, dtype=torch.int32).pin_memory()
                g['stage'].append({'uiy': host, 'u': host[:b], 'i': host[b:2 * b], 'y': host[2 * b:].view(torch.float32), 'done': torch.cuda.Event()})
            g['turn'] = 0
            from deep_cbrs_amar_renaissance_amd.engine import capture_graph

            def body():
                with torch.no_grad():
                    self._graph_body()
            g['graph'], _ = capture_graph(body)
            g['slot_dev'][:g['slot_bytes']].copy_(g['slot_host'][:g['slot_bytes']])   # the Adam slot table of this graph (fixed addresses): once, not per replay
            self._graphs[key] = g
        self._g = g
        n_nodes = self.seq.adj_matrix.shape[0] if getattr(self, 'seq', None) is not None else None
        st = g['stage'][g['turn'] % len(g['stage'])]
        g['turn'] += 1
        st['done'].synchronize()                                     # (the uploads that last used this staging set have landed)
        on_device = [isinstance(src, torch.Tensor) and src.is_cuda for src in (u_ids, i_ids, y)]
        for name, src, dev_side in (('u', u_ids, on_device[0]), ('i', i_ids, on_device[1])):
            if dev_side:
                g[name].copy_(src)
            else:
                stage_ids(st[name], src, n_nodes)                    # range check + int32 on the host, into pinned memory
                if any(on_device):
                    g[name].copy_(st[name], non_blocking=True)
        if on_device[2]:
            g['y'].copy_(y)
        else:
            st['y'].numpy()[...] = y.numpy() if isinstance(y, torch.Tensor) else np.asarray(y, dtype=np.float32)
            if any(on_device):
                g['y'].copy_(st['y'], non_blocking=True)
        if not any(on_device):
            g['uiy'].copy_(st['uiy'], non_blocking=True)             # ids and labels of the batch: one asynchronous upload
        st['done'].record()
        if with_blocks:
            g['ub'].copy_(to_device_tensor(bert[0]))
            g['ib'].copy_(to_device_tensor(bert[1]))
        if self._dev_t != self.t:                                    # eager steps happened in between: resynchronise the counter
            self._adam_state[0] = float(self.t)
        g['graph'].replay()
        self.t += 1
        self._dev_t = self.t

    def _all_tapes(self):
        """Every tape of the trainer that owns a fused reverse pass (Dense stacks of the head, convolution stacks)."""
        out = list(getattr(self, 'tapes', []) or [])
        head = getattr(self, 'head', None)
        for name in ('unet', 'inet', 'clf'):
            if hasattr(head, name):
                out.append(getattr(head, name))
        out.extend(getattr(head, 't', {}).values() if isinstance(getattr(head, 't', None), dict) else [])
        return [t for t in out if hasattr(t, 'defer_reduce')]

    def pop_loss_sum(self):
        """Sum over the batches since the last call of (batch loss x batch size); one host synchronisation."""
        if not hasattr(self, '_graphs'):
            return 0.0
        total = self._eager_loss + float(self._loss_sum.item())
        self._eager_loss = 0.0
        self._loss_sum.zero_()
        return total

    def touch_parameters(self):
        """Bump the autograd version counters after graph replays (hoisting caches key on them)."""
        with torch.no_grad():
            for prm in self.params:
                prm.add_(0)

    def _propagation_forward(self):
        """E = gnn(None) with every stack's activations kept on its tape."""
        gnn, tapes = self.model.gnn, self.tapes
        if self.layout == 'single':
            return tapes[0].forward()
        if self.layout == 'two_step':
            items = tapes[0].forward()[:gnn.n_embeddings]
            users = gnn.step_two_gnn_layers.embeddings
            x0 = torch.empty((users.shape[0] + items.shape[0], users.shape[1]), dtype=torch.float32, device=self.device)
            capi.copy_columns(users.detach(), x0[:users.shape[0]])
            capi.copy_columns(items, x0[users.shape[0]:])
            return tapes[1].forward(x0)
        users, items = tapes[0].forward()[:gnn.n_users], tapes[1].forward()[:gnn.n_items]
        x0 = torch.empty((gnn.n_users + gnn.n_items, users.shape[1]), dtype=torch.float32, device=self.device)
        capi.copy_columns(users, x0[:gnn.n_users])
        capi.copy_columns(items, x0[gnn.n_users:])
        return tapes[2].forward(x0)

    def _lift(self, rows, n_nodes):
        """Gradient of a leading-rows slice: the rows, zero below (a stack hands over only its first |U| or |I| nodes)."""
        full = torch.zeros((n_nodes, rows.shape[1]), dtype=torch.float32, device=self.device)
        capi.copy_columns(rows, full[:rows.shape[0]])
        return full

    def _propagation_backward(self, e, de, grads):
        gnn, tapes = self.model.gnn, self.tapes
        if self.layout == 'single':
            grads[gnn.gnn_layers.embeddings] = tapes[0].backward(de, grads)
            return
        if self.layout == 'two_step':
            one, two = gnn.step_one_gnn_layers, gnn.step_two_gnn_layers
            dx0 = tapes[1].backward(de, grads)
            n_users = two.embeddings.shape[0]
            g_users = torch.empty_like(two.embeddings)
            capi.copy_columns(dx0[:n_users], g_users)
            grads[two.embeddings] = g_users
            grads[one.embeddings] = tapes[0].backward(self._lift(dx0[n_users:], one.adj_matrix.shape[0]), grads)
            return
        one, two = gnn.way_one_gnn_layers, gnn.way_two_gnn_layers
        dx0 = tapes[2].backward(de, grads)
        grads[one.embeddings] = tapes[0].backward(self._lift(dx0[:gnn.n_users], one.adj_matrix.shape[0]), grads)
        grads[two.embeddings] = tapes[1].backward(self._lift(dx0[gnn.n_users:], two.adj_matrix.shape[0]), grads)

    def apply_gradients(self, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        with torch.no_grad():
            for prm in self.params:
                g = grads[prm]
                capi.adam(prm.data.view(-1), g.contiguous().view(-1), self.m[prm].view(-1), self.v[prm].view(-1),
                          lr_t, self.b1, self.b2, self.eps, l2=self._l2(prm))
                prm._version  # noqa: B018  (data-level update; bump below keeps hoisting caches honest)
                prm.add_(0)                                            # bumps the autograd version counter

    def train_batch(self, u_ids, i_ids, y, bert=None):
        loss, grads = self.loss_and_grads(u_ids, i_ids, y, bert=bert)
        self.apply_gradients(grads)
        return loss


class HeadTrainer(Trainer):
    """BasicRS / HybridCBRS on pre-computed embedding rows (econfigs/basic-kge.yaml, hybrid-kge.yaml): the batch Sequence
    delivers the rows themselves (datasets.py:43-77), so only the Dense stacks (and fusion weights) train."""

    def __init__(self, model, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **unused):
        if not model.built:
            raise ValueError("build the head first (one forward call, as Experimenter.build_model does)")
        self.model = model
        self.hybrid = hasattr(model, 'dense1a')
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        self.t = 0
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.m = {p: torch.zeros_like(p) for p in self.params}
        self.v = {p: torch.zeros_like(p) for p in self.params}
        self.head = _HybridHead(model) if self.hybrid else _BasicHead(model)
        self.device = self.params[0].device
        self.tables = None

    # -- batches as ids against tables kept on the device (round 4) ---------------------------------------------------------------
    def set_tables(self, tables):
        """The embedding table(s) the batch Sequence gathers its rows from ([n, D] each: one for BasicRS, graph + BERT for HybridCBRS),
        uploaded once.  Batches are then (user ids, item ids, labels): the rows are gathered on the device inside a replayed hipGraph
        (Trainer.train_batch_graphed) instead of on the host and uploaded every batch — 25 epochs of HybridCBRS at ML-1M size took
        35 s that way, more than any graph model."""
        self.tables = [to_device_tensor(np.ascontiguousarray(t, dtype=np.float32)) for t in tables]
        self.seq = types.SimpleNamespace(adj_matrix=types.SimpleNamespace(shape=(min(int(t.shape[0]) for t in self.tables),)))   # (range check of the ids)

    def _rows_of(self, u, i):
        out = []
        for t in self.tables:
            for ids in (u, i):
                rows = torch.empty((ids.numel(), t.shape[1]), dtype=torch.float32, device=t.device)
                capi.copy_columns(t, rows, ids=ids)
                out.append(rows)
        return out                                                   # (user, item) per table: [gu, gi] or [gu, gi, bu, bi]

    def _forward_backward(self, u, i, yv, rows=None, ui=None):
        """Trainer's per-batch core for ids: gather the rows, head forward, BCE, head reverse pass (the inputs are constants)."""
        r = self._rows_of(u, i)
        b = u.numel()
        p = self.head.forward(r[0], r[1], (r[2], r[3]) if self.hybrid else None)
        dz = torch.empty((b, 1), dtype=torch.float32, device=p.device)
        terms = torch.empty(b, dtype=torch.float32, device=p.device)
        capi.bce_grad(p, yv, dz, terms)
        grads = {}
        self.head.backward(dz, grads, need_input_grad=False)
        return terms, grads

    def loss_and_grads(self, blocks, y):
        """blocks = (user rows, item rows) or (user graph, item graph, user BERT, item BERT), each [B, D]."""
        rows = [to_device_tensor(b) for b in blocks]
        yv = to_device_tensor(np.asarray(y, dtype=np.float32) if not isinstance(y, torch.Tensor) else y)
        b = rows[0].shape[0]
        with torch.no_grad():
            p = self.head.forward(rows[0], rows[1], (rows[2], rows[3]) if self.hybrid else None)
            dz = torch.empty((b, 1), dtype=torch.float32, device=p.device)
            terms = torch.empty(b, dtype=torch.float32, device=p.device)
            capi.bce_grad(p, yv, dz, terms)
            grads = {}
            self.head.backward(dz, grads, need_input_grad=False)
            loss = float(terms.sum().item()) / b
        return loss, grads

    def train_batch(self, *args, bert=None):
        """train_batch(blocks, y): the rows themselves; train_batch(u_ids, i_ids, y): ids against `set_tables` (eager: the first
        batch of a shape before Trainer.train_batch_graphed captures)."""
        if len(args) == 3:
            u_ids, i_ids, y = args
            n = self.seq.adj_matrix.shape[0]
            u, i = ids_to_device(u_ids, n), ids_to_device(i_ids, n)
            yv = to_device_tensor(np.asarray(y, dtype=np.float32) if not isinstance(y, torch.Tensor) else y)
            with torch.no_grad():
                terms, grads = self._forward_backward(u, i, yv)
                loss = float(terms.sum().item()) / u.numel()
                for prm in self.params:
                    l2 = self._l2(prm)
                    if l2:
                        loss += l2 * float((prm.detach().double() ** 2).sum().item())
        else:
            loss, grads = self.loss_and_grads(*args)
        self.apply_gradients(grads)
        return loss


def fit(model, sequence, epochs=1, callbacks=None, verbose=True, **kwargs):
    """Keras-style ``fit`` over a batch Sequence: ``epochs`` passes, ``on_epoch_end`` reshuffles (datasets.py:205-213)."""
    opt = getattr(model, 'optimizer', None)
    hp = {k: getattr(opt, k) for k in ('learning_rate', 'beta_1', 'beta_2', 'epsilon') if hasattr(opt, k)}
    if not hasattr(model, 'gnn'):                              # BasicRS / HybridCBRS on pre-computed rows: head-only training
        trainer = getattr(model, '_trainer', None)
        if trainer is None:
            if not model.built and len(sequence):
                model(sequence[0][0])                          # one forward call builds every weight (as Keras does)
            trainer = model._trainer = HeadTrainer(model, **hp)
        # the reference's Sequences of pre-computed rows (datasets.py:55-80) gather them on the host from one (BasicRS) or two
        # (HybridCBRS) tables indexed by node id: the tables go to the device once and the batches are read as ids
        # (AMAR_RESIDENT_ROWS=0: the batches as they come, eager)
        from deep_cbrs_amar_renaissance_amd.data.datasets import HybridUserItemEmbeddings, UserItemEmbeddings
        tables = None
        if os.environ.get('AMAR_RESIDENT_ROWS', '1') != '0':
            if type(sequence) is UserItemEmbeddings and not trainer.hybrid:
                tables = [sequence.embeddings]
            elif type(sequence) is HybridUserItemEmbeddings and trainer.hybrid:
                tables = [sequence.graph_embeddings, sequence.bert_embeddings]
            if tables is not None and not all(isinstance(t, np.ndarray) and t.ndim == 2 for t in tables):
                tables = None
        if tables is not None:
            src = getattr(trainer, '_table_sources', None)
            if src is None or len(src) != len(tables) or any(a is not b for a, b in zip(src, tables)):
                trainer.set_tables(tables)
                trainer._table_sources = tables
        use_graph = tables is not None and os.environ.get('AMAR_TRAIN_GRAPH', '1') != '0'
        history = []
        for epoch in range(int(epochs)):
            total, count = 0.0, 0
            for b in range(len(sequence)):
                if tables is not None:
                    r = sequence._batch_ratings(b)
                    u, i, y = r[:, 0], r[:, 1], r[:, 2]
                    if use_graph:
                        trainer.train_batch_graphed(u, i, y)
                    else:
                        total += trainer.train_batch(u, i, y) * len(y)
                else:
                    blocks, y = sequence[b]
                    total += trainer.train_batch(blocks, y) * len(y)
                count += len(y)
            if use_graph:
                total = trainer.pop_loss_sum()
                trainer.touch_parameters()
            history.append(total / max(count, 1))
            if verbose:
                print("Epoch {}/{} - loss: {:.4f}".format(epoch + 1, epochs, history[-1]))
            if hasattr(sequence, 'on_epoch_end'):
                sequence.on_epoch_end()
        return {'loss': history}
    # Hybrid batches of the reference's own Sequence (datasets.py:95-123 here, 112-115 there) carry, besides the ids, the BERT rows of the
    # batch's users and items — gathered on the host from ONE table indexed by node id and uploaded every batch (6 MB at batch 1 024).
    # That table is registered once on the device instead and the batches are read as ids only: the same rows, gathered there
    # (AMAR_RESIDENT_BERT=0: the batches as they come).  A replayed hybrid batch at ml1m(s=1): 0.41 against 0.71 ms.
    ids_only = model.resident_ids(sequence) if hasattr(model, 'resident_ids') else None
    trainer = getattr(model, '_trainer', None)
    if trainer is None:
        if hasattr(model.rs, 'dense1a') and not model.rs.built and len(sequence):
            first = sequence[0][0]
            if len(first) >= 4 and first[2] is not None:
                hp['bert_dim'] = int(np.asarray(first[2]).shape[1])
        trainer = model._trainer = Trainer(model, **hp)
    use_graph = os.environ.get('AMAR_TRAIN_GRAPH', '1') != '0'
    history = []
    for epoch in range(int(epochs)):
        total, count = 0.0, 0
        for b in range(len(sequence)):
            inputs, y = (ids_only if ids_only is not None else sequence)[b]
            u, i = inputs[0], inputs[1]
            bert = (inputs[2], inputs[3]) if len(inputs) >= 4 else None     # hybrid batches carry the BERT blocks (datasets.py:112-115)
            if use_graph:
                trainer.train_batch_graphed(u, i, y, bert=bert)
            else:
                total += trainer.train_batch(u, i, y, bert=bert) * len(y)
            count += len(y)
        if use_graph:
            total = trainer.pop_loss_sum()
            trainer.touch_parameters()
        history.append(total / max(count, 1))
        if verbose:
            print("Epoch {}/{} - loss: {:.4f}".format(epoch + 1, epochs, history[-1]))
        if hasattr(sequence, 'on_epoch_end'):
            sequence.on_epoch_end()
    return {'loss': history}
