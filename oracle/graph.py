"""Oracle rows A1, A1', A2, A3: ratings -> adjacency -> normalised filter -> sorted COO.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  scipy is used the same way the reference
uses it, so that duplicate handling and float32 rounding order are the reference's.
"""
import numpy as np
from scipy import sparse


def remap_ratings(train_raw, test_raw):
    """Raw ids -> contiguous ids; items offset by |U|.   loaders.py:43-56

    users/items = ascending unique raw ids of the TRAIN file; test rows are looked up in them.
    """
    users, u_idx = np.unique(train_raw[:, 0], return_inverse=True)
    items, i_idx = np.unique(train_raw[:, 1], return_inverse=True)
    train = np.stack([u_idx, i_idx + len(users), train_raw[:, 2]], axis=1)
    tu = np.searchsorted(users, test_raw[:, 0])
    ti = np.searchsorted(items, test_raw[:, 1])
    if not (np.array_equal(users[tu], test_raw[:, 0]) and np.array_equal(items[ti], test_raw[:, 1])):
        raise ValueError("test ratings mention users/items that are absent from train")
    test = np.stack([tu, ti + len(users), test_raw[:, 2]], axis=1)
    return (train, test), (users, items)


def remap_props(props_raw, items):
    """Props triples -> (item index, prop index + |I|, 1).   loaders.py:60-68

    The relation column is dropped; duplicate (item, prop) pairs are kept.
    """
    it = np.searchsorted(items, props_raw[:, 0])
    if not np.array_equal(items[it], props_raw[:, 0]):
        raise ValueError("props file must be pre-filtered to train items (preprocess.py:173-198)")
    props, p_idx = np.unique(props_raw[:, 1], return_inverse=True)
    triples = np.stack([it, p_idx + len(items), np.ones(len(it), dtype=props_raw.dtype)], axis=1)
    return triples, props


def symmetrize(coo):
    """math.py:13-20 — concatenate (r,c) and (c,r); no dedupe."""
    return sparse.coo_matrix(
        (np.concatenate([coo.data, coo.data]),
         (np.concatenate([coo.row, coo.col]), np.concatenate([coo.col, coo.row]))),
        shape=coo.shape, dtype=coo.dtype)


def adjacency_unary(train, n_users, n_items, symmetric=True):
    """preprocess.py:68-86 — positive ratings only, value 1, float32 COO [N,N]."""
    pos = train[:, 2] == 1
    n = n_users + n_items
    a = sparse.coo_matrix((train[pos, 2], (train[pos, 0], train[pos, 1])), shape=[n, n], dtype=np.float32)
    return symmetrize(a) if symmetric else a


def adjacency_unary_uip(train, triples, n_users, n_items, n_props, symmetric=True):
    """preprocess.py:120-168 ('unary-uip') — UI positives + item-property links, one [N,N] COO."""
    pos = train[:, 2] == 1
    rows = np.concatenate([train[pos, 0], triples[:, 0] + n_users])
    cols = np.concatenate([train[pos, 1], triples[:, 1] + n_users])
    data = np.concatenate([train[pos, 2], triples[:, 2]])
    n = n_users + n_items + n_props
    a = sparse.coo_matrix((data, (rows, cols)), shape=[n, n], dtype=np.float32)
    return symmetrize(a) if symmetric else a


def adjacency_unary_kg(train, triples, n_users, n_items, n_props, symmetric=True):
    """preprocess.py:120-152 ('unary-kg') — the user-item graph and, separately, the item-property graph (items first)."""
    pos = train[:, 2] == 1
    n_ui, n_kg = n_users + n_items, n_items + n_props
    bi = sparse.coo_matrix((train[pos, 2], (train[pos, 0], train[pos, 1])), shape=[n_ui, n_ui], dtype=np.float32)
    kg = sparse.coo_matrix((triples[:, 2], (triples[:, 0], triples[:, 1])), shape=[n_kg, n_kg], dtype=np.float32)
    return (symmetrize(bi), symmetrize(kg)) if symmetric else (bi, kg)


def user_properties(ui_adj, ip_adj, n_users, n_items):
    """get_user_properties (preprocess.py:9-41), followed literally: stack the two graphs into one user-item-property
    matrix, square it, binarise, and copy the user x property blocks of the DENSE square (small graphs only)."""
    n_props = ip_adj.shape[0] - n_items
    n = n_users + n_items + n_props
    uip = sparse.coo_matrix((np.concatenate([ui_adj.data, ip_adj.data]),
                             (np.concatenate([ui_adj.row, ip_adj.row + n_users]),
                              np.concatenate([ui_adj.col, ip_adj.col + n_users]))), shape=(n, n))
    sq = uip.dot(uip)
    sq.data = np.ones(len(sq.data))
    sq = np.asarray(sq.todense())
    up = np.zeros((n_users + n_props, n_users + n_props))
    up[n_users:, :n_users] = sq[n_users + n_items:, :n_users]
    up[:n_users, n_users:] = sq[:n_users, n_users + n_items:]
    return sparse.coo_matrix(up)


def gcn_filter(a):
    """Spektral 1.x utils.convolution.gcn_filter (call sites gnn.py:283,381; lightgcn_conv.py:56-58).

    tocsr() (sums duplicates) -> diagonal += 1 -> D^-1/2 with inf -> 0 -> D.A.D -> sort_indices,
    all in the matrix dtype (float32 here).
    """
    out = a.tocsr().copy()
    out = (out + sparse.identity(out.shape[0], dtype=out.dtype, format='csr')).tocsr()
    with np.errstate(divide='ignore'):
        deg = np.power(np.array(out.sum(1)), -0.5).ravel().astype(out.dtype)
    deg[np.isinf(deg)] = 0.0
    d = sparse.diags(deg).astype(out.dtype)
    out = d.dot(out).dot(d).tocsr()
    out.sort_indices()
    return out


def reordered_coo(a):
    """math.py:37-56 — COO triplets in row-major order, duplicates KEPT (tf.sparse.reorder).

    Returns (row, col, val); this is the edge list GraphSAGE/GAT see and the nnz order of
    `tf.sparse.sparse_dense_matmul`.
    """
    c = a.tocoo()
    order = np.lexsort((c.col, c.row))
    return c.row[order].astype(np.int64), c.col[order].astype(np.int64), c.data[order].astype(np.float32)


def add_self_loops_edges(row, col, n):
    """Spektral ops.add_self_loops_indices: drop existing (i,i), append one (i,i) per node, reorder."""
    keep = row != col
    r = np.concatenate([row[keep], np.arange(n)])
    c = np.concatenate([col[keep], np.arange(n)])
    order = np.lexsort((c, r))
    return r[order], c[order]


def dgcf_adjacency(a):
    """DGCFConv.preprocess (dgcf_conv.py:38-48) with its high-pass filter (dgcf_conv.py:50-80).

    crosshop = A . A (duplicates of A summed by the product); both A and the crosshop matrix go through gcn_filter;
    the crosshop filter keeps the entries > eps for the eps in (1e-1, 1e-2, 1e-3, 5e-4) whose kept-entry count is
    closest in ratio to nnz(gcn_filter(A)) (first minimum); result = A_hat + filtered + I, float32 CSR.
    An eps that keeps nothing has an infinite ratio here (the reference would divide by zero).
    """
    a = sparse.csr_matrix(a)
    crosshop = a.dot(a)
    a_hat, cross_hat = gcn_filter(a), gcn_filter(crosshop)
    edges = len(a_hat.data)
    filtered = [cross_hat.multiply(cross_hat > eps).tocsr() for eps in (1e-1, 1e-2, 1e-3, 5e-4)]
    counts = [len(m.data) for m in filtered]
    ratios = [np.inf if c == 0 else (edges / c if edges > c else c / edges) for c in counts]
    best = int(np.argmin(ratios))
    out = (a_hat + filtered[best] + sparse.eye(a.shape[0], dtype=np.float32)).tocsr().astype(np.float32)
    out.sum_duplicates()
    return out
