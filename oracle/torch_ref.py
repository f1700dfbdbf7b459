"""Second, independent CPU implementation (torch-CPU index_add_/scatter) of the propagation.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  It shares no code with oracle/layers.py:
edges are an explicit (target, source, weight) list built here from the raw rating triples,
and every aggregation is an index_add_ / scatter_reduce.  tests/ require both implementations
to agree to 1e-6 relative on seeded ML-1M-shape inputs.
"""
import numpy as np
import torch


def edges_from_ratings(train, n_nodes, triples=None, n_users=0):
    """Symmetric positive-rating edge list (+ item-property links), duplicates kept."""
    pos = train[train[:, 2] == 1]
    r, c = [pos[:, 0]], [pos[:, 1]]
    if triples is not None:
        r.append(triples[:, 0] + n_users)
        c.append(triples[:, 1] + n_users)
    r, c = np.concatenate(r), np.concatenate(c)
    return torch.from_numpy(np.concatenate([r, c])).long(), torch.from_numpy(np.concatenate([c, r])).long()


def _gcn_weights(tgt, src, n, dtype):
    # A + I with duplicate edges summed, then D^-1/2 (A+I) D^-1/2 evaluated per edge in fp32 order
    loops = torch.arange(n)
    t = torch.cat([tgt, loops])
    s = torch.cat([src, loops])
    key = t * n + s
    uniq, inv = torch.unique(key, return_inverse=True)
    val = torch.zeros(len(uniq), dtype=dtype).index_add_(0, inv, torch.ones(len(key), dtype=dtype))
    t, s = uniq // n, uniq % n
    deg = torch.zeros(n, dtype=dtype).index_add_(0, t, val)
    dinv = deg.pow(-0.5)
    dinv[torch.isinf(dinv)] = 0
    return t, s, (dinv[t] * val) * dinv[s]


def _spmm(t, s, w, x, n):
    return torch.zeros(n, x.shape[1], dtype=x.dtype).index_add_(0, t, w[:, None] * x[s])


def _with_loops(tgt, src, n):
    keep = tgt != src
    loops = torch.arange(n)
    return torch.cat([tgt[keep], loops]), torch.cat([src[keep], loops])


def propagate(tgt, src, gnn, dtype=torch.float32, self_loops=True):
    n = gnn['embeddings'].shape[0]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    x = T(gnn['embeddings'])
    hs = [x]
    kind = gnn['kind']
    if kind in ('gcn', 'lightgcn'):
        t, s, w = _gcn_weights(tgt, src, n, dtype)
        for lw in gnn['layers']:
            if kind == 'gcn':
                x = torch.relu(_spmm(t, s, w, x @ T(lw['kernel']), n) + T(lw['bias']))
            else:
                x = _spmm(t, s, w, x, n)
            hs.append(x)
    else:
        t, s = _with_loops(tgt, src, n) if self_loops else (tgt, src)
        for lw in gnn['layers']:
            if kind == 'sage':
                cnt = torch.zeros(n, dtype=dtype).index_add_(0, t, torch.ones(len(t), dtype=dtype))
                agg = torch.zeros(n, x.shape[1], dtype=dtype).index_add_(0, t, x[s]) / cnt.clamp(min=1)[:, None]
                out = torch.cat([x, agg], 1) @ T(lw['kernel']) + T(lw['bias'])
                out = out * torch.rsqrt(out.pow(2).sum(1, keepdim=True).clamp(min=1e-12))
                x = torch.relu(out)
            elif kind == 'gat':
                h = x @ T(lw['kernel'])
                e = torch.nn.functional.leaky_relu((h @ T(lw['attn_self']))[t] + (h @ T(lw['attn_neigh']))[s], 0.2)
                m = torch.full((n,), -float('inf'), dtype=dtype).scatter_reduce(0, t, e, 'amax')
                ex = torch.exp(e - m[t])
                den = torch.zeros(n, dtype=dtype).index_add_(0, t, ex) + 1e-9
                x = torch.relu(torch.zeros(n, h.shape[1], dtype=dtype).index_add_(0, t, (ex / den[t])[:, None] * h[s])
                               + T(lw['bias']))
            else:
                raise ValueError(kind)
            hs.append(x)
    final = 'mean' if kind == 'lightgcn' else gnn.get('final_node', 'concatenation')
    if final == 'concatenation':
        return torch.cat(hs, 1).numpy()
    if final == 'last':
        return hs[-1].numpy()
    out = sum(hs)
    return (out / len(hs) if final == 'mean' else out).numpy()
