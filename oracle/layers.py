"""Oracle rows A5 (GCN / LightGCN / GraphSAGE / GAT), A6 (reduction), Keras Dense.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function computes in the dtype of
its inputs: float32 is the reference-faithful run, float64 bounds the fp32 rounding error.
"""
import numpy as np
from scipy import sparse


def _act(x, activation):
    if activation in (None, 'linear'):
        return x
    if activation == 'relu':
        return np.maximum(x, 0)
    if activation == 'sigmoid':
        return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)
    raise ValueError("Unknown activation {}".format(activation))


def dense(x, kernel, bias, activation='relu'):
    """Keras Dense: act(x . W + b), W [in, out].   dense.py:4-17"""
    return _act(x @ kernel + bias, activation)


def dense_network(x, layers, activation='relu'):
    """build_dense_network (dense.py:4-7): every layer has `activation`."""
    for kernel, bias in layers:
        x = dense(x, kernel, bias, activation)
    return x


def dense_classifier(x, layers, activation='relu'):
    """build_dense_classifier (dense.py:10-17): hidden layers + Dense(1, sigmoid)."""
    for kernel, bias in layers[:-1]:
        x = dense(x, kernel, bias, activation)
    kernel, bias = layers[-1]
    return dense(x, kernel, bias, 'sigmoid')


def gcn_conv(x, a_hat, kernel, bias, activation='relu'):
    """Spektral GCNConv.call: act(A_hat . (x . W) + b).   gnn.py:289-295 (build), gnn.py:78 (call)"""
    a_hat = a_hat.astype(x.dtype)
    return _act(a_hat @ (x @ kernel) + bias, activation)


def lightgcn_conv(x, a_hat):
    """lightgcn_conv.py:51-54 — modal_dot(a, x), nothing else."""
    return a_hat.astype(x.dtype) @ x


def dgcf_conv(x, a_dgcf, w):
    """DGCFConv.call (dgcf_conv.py:32-36): LocalityAdaptive x * sigmoid(w) (w [N, 1], dgcf_conv.py:83-102), then A . x."""
    gate = 1.0 / (1.0 + np.exp(-w.astype(x.dtype)))
    return a_dgcf.astype(x.dtype) @ (x * gate)


def _segment_sum(values, targets, n):
    m = sparse.csr_matrix((np.ones(len(targets), dtype=values.dtype), (targets, np.arange(len(targets)))),
                          shape=(n, len(targets)))
    return m @ values


def sage_conv(x, row, col, kernel, bias, activation='relu', self_loops=True, aggregate='mean'):
    """Spektral 1.x GraphSageConv, aggregate='mean' (config.yaml:18; gnn.py:354-361).

    a <- add_self_loops(a); messages x[j] over the edge list (edge VALUES ignored, duplicate
    edges counted), unsorted_segment_mean over targets; out = [x || agg] . W + b;
    l2_normalize(axis=-1) = out * rsqrt(max(sum(out^2), 1e-12)); THEN the activation.
    Spektral's MessagePassing takes targets = indices[:,1], sources = indices[:,0].
    """
    if aggregate != 'mean':
        raise ValueError("oracle restates aggregate='mean' only")
    n = x.shape[0]
    if self_loops:
        from oracle.graph import add_self_loops_edges
        row, col = add_self_loops_edges(row, col, n)
    sources, targets = row, col
    summed = _segment_sum(x[sources], targets, n)
    count = np.bincount(targets, minlength=n).astype(x.dtype)
    agg = summed / np.maximum(count, 1)[:, None]          # segment_mean of an empty segment is 0
    out = np.concatenate([x, agg], axis=1) @ kernel + bias
    sq = np.sum(out * out, axis=1, keepdims=True)
    out = out * (1.0 / np.sqrt(np.maximum(sq, np.asarray(1e-12, dtype=x.dtype))))
    return _act(out.astype(x.dtype), activation)


def gat_conv(x, row, col, kernel, attn_self, attn_neigh, bias, activation='relu', self_loops=True):
    """Spektral 1.x GATConv._call_single, attn_heads=1, concat_heads, dropout 0 (gnn.py:321-328).

    h = x.W; e = LeakyReLU_0.2(h[target].a_self + h[source].a_neigh) over A's edges (duplicates
    kept) plus one self loop per node; alpha = exp(e - max_t) / (sum_t exp(e - max_t) + 1e-9);
    out[target] = sum alpha * h[source]; + b; activation.  targets = indices[:,1].
    """
    n = x.shape[0]
    if self_loops:
        from oracle.graph import add_self_loops_edges
        row, col = add_self_loops_edges(row, col, n)
    sources, targets = row, col
    h = x @ kernel
    s_self = h @ attn_self
    s_neigh = h @ attn_neigh
    e = s_self[targets] + s_neigh[sources]
    e = np.where(e > 0, e, np.asarray(0.2, dtype=x.dtype) * e)
    seg_max = np.full(n, -np.inf, dtype=x.dtype)
    np.maximum.at(seg_max, targets, e)
    ex = np.exp(e - seg_max[targets])
    denom = _segment_sum(ex[:, None], targets, n)[:, 0] + np.asarray(1e-9, dtype=x.dtype)
    alpha = ex / denom[targets]
    out = _segment_sum(alpha[:, None] * h[sources], targets, n) + bias
    return _act(out.astype(x.dtype), activation), alpha


def reduce_layers(hs, method='concatenation', w=None):
    """reduction.py:15-33; 'w-sum' = WeightedSum.call (reduction.py:54-55): reduce_sum(multiply(w * w, inputs), axis=0) with the
    learnable `w` [n_layers] (ones when None: its initial value, reduction.py:50)."""
    if method == 'w-sum':
        w = np.ones(len(hs), dtype=hs[0].dtype) if w is None else np.asarray(w, dtype=hs[0].dtype).reshape(-1)
        out = (w[0] * w[0]) * hs[0]
        for k, h in enumerate(hs[1:], start=1):
            out = out + (w[k] * w[k]) * h
        return out
    if method == 'concatenation':
        return np.concatenate(hs, axis=1)
    if method == 'sum':
        out = hs[0].copy()
        for h in hs[1:]:
            out = out + h
        return out
    if method == 'mean':
        out = hs[0].copy()
        for h in hs[1:]:
            out = out + h
        return out / np.asarray(len(hs), dtype=out.dtype)
    if method == 'last':
        return hs[-1]
    raise ValueError('Reduction method not supported: ' + method)
