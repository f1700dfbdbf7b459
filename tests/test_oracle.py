"""CPU: pins the oracle (hand-computed graphs, doc.pdf KATs, golden vectors, second implementation)."""
import glob
import os

import numpy as np
import pytest
import torch
from scipy import sparse

from oracle import graph as og, layers as ol, models as om, weights as ow, torch_ref
from tests import helpers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _sym(n, edges):
    r, c = np.array([e[0] for e in edges]), np.array([e[1] for e in edges])
    return og.symmetrize(sparse.coo_matrix((np.ones(len(r), dtype=np.float32), (r, c)), shape=(n, n)))


def test_gcn_filter_path_p3():
    """P3 path 0-1-2: degrees of A+I are (2,3,2) -> A_hat by hand."""
    a_hat = og.gcn_filter(_sym(3, [(0, 1), (1, 2)])).toarray()
    s6 = 1 / np.sqrt(6)
    want = np.array([[0.5, s6, 0], [s6, 1 / 3, s6], [0, s6, 0.5]])
    assert a_hat.dtype == np.float32 and np.allclose(a_hat, want, atol=1e-7)


def test_gcn_filter_star_and_lightgcn_mean():
    """Star K1,3 (centre 0): degrees (4,2,2,2); LightGCN 1 layer, mean of [X0, A_hat X0]."""
    adj = _sym(4, [(0, 1), (0, 2), (0, 3)])
    a_hat = og.gcn_filter(adj).toarray()
    s8 = 1 / np.sqrt(8)
    want = np.array([[.25, s8, s8, s8], [s8, .5, 0, 0], [s8, 0, .5, 0], [s8, 0, 0, .5]])
    assert np.allclose(a_hat, want, atol=1e-7)
    x0 = np.eye(4, dtype=np.float32)
    e = om.propagate(adj, {'kind': 'lightgcn', 'embeddings': x0, 'layers': [{}]})
    assert np.allclose(e, (np.eye(4) + want) / 2, atol=1e-7)


def test_duplicate_edges_summed_for_gcn_counted_for_sage():
    """Item 2 is linked to property 4 under two relations: GCN sums the duplicate (a=2), SAGE counts it twice."""
    adj = _sym(5, [(0, 2), (0, 3), (1, 2), (2, 4), (2, 4)])
    assert adj.nnz == 10                                     # symmetrised, duplicates kept (math.py:13-20)
    a_hat = og.gcn_filter(adj)
    assert a_hat.nnz == 8 + 5                                # 4 distinct undirected edges + 5 self loops
    deg = np.array([3, 2, 5, 2, 3], dtype=np.float64)        # row sums of A+I with A[2,4] = 2
    assert np.isclose(a_hat[2, 4], 2 / np.sqrt(deg[2] * deg[4]), atol=1e-7)
    row, col, _ = og.reordered_coo(adj)
    x = np.arange(10, dtype=np.float64).reshape(5, 2)
    w = np.concatenate([np.zeros((2, 2)), np.eye(2)])         # output = aggregate only
    out = ol.sage_conv(x, row, col, w, np.zeros(2), activation=None, self_loops=True)
    agg2 = (x[0] + x[1] + 2 * x[4] + x[2]) / 5                # neighbours 0,1,4,4 + self loop
    assert np.allclose(out[2], agg2 / np.linalg.norm(agg2))
    out_nl = ol.sage_conv(x, row, col, w, np.zeros(2), activation=None, self_loops=False)
    agg2 = (x[0] + x[1] + 2 * x[4]) / 4
    assert np.allclose(out_nl[2], agg2 / np.linalg.norm(agg2))


def test_gat_softmax_rows_sum_to_one_and_uniform_case():
    adj = _sym(4, [(0, 1), (0, 2), (0, 3)])
    row, col, _ = og.reordered_coo(adj)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((4, 3))
    out, alpha = ol.gat_conv(x, row, col, rng.standard_normal((3, 2)), rng.standard_normal(2), rng.standard_normal(2),
                             np.zeros(2), activation=None)
    r, c = og.add_self_loops_edges(row, col, 4)
    sums = np.bincount(c, weights=alpha, minlength=4)
    assert np.allclose(sums, 1.0, atol=1e-8)
    # zero attention vectors -> uniform attention -> plain mean of h over the neighbourhood incl. self
    w = rng.standard_normal((3, 2))
    out, _ = ol.gat_conv(x, row, col, w, np.zeros(2), np.zeros(2), np.zeros(2), activation=None)
    h = x @ w
    assert np.allclose(out[0], h.mean(axis=0), atol=1e-8) and np.allclose(out[1], (h[0] + h[1]) / 2, atol=1e-8)


def test_gcn_layer_by_hand():
    adj = _sym(3, [(0, 1), (1, 2)])
    x0 = np.array([[1, 0], [0, 1], [1, 1]], dtype=np.float32)
    w = np.array([[1, -1], [2, 0.5]], dtype=np.float32)
    b = np.array([0.1, -0.2], dtype=np.float32)
    gnn = {'kind': 'gcn', 'embeddings': x0, 'layers': [{'kernel': w, 'bias': b}], 'final_node': 'concatenation'}
    e = om.propagate(adj, gnn)
    s6 = 1 / np.sqrt(6)
    a_hat = np.array([[0.5, s6, 0], [s6, 1 / 3, s6], [0, s6, 0.5]])
    want = np.concatenate([x0, np.maximum(a_hat @ (x0 @ w) + b, 0)], axis=1)
    assert e.shape == (3, 4) and np.allclose(e, want, atol=1e-6)


# doc.pdf parameter counts: (kind, N, d, hiddens/layers, dense, clf, expected)
PARAM_KATS = [
    ('gcn', 9228, 8, (8, 8), [24, 24], [48, 48], 81121),
    ('sage', 9228, 8, (8, 8), [24, 24], [48, 48], 81249),
    ('gat', 9228, 8, (8, 8), [24, 24], [48, 48], 81153),
    ('lightgcn', 9228, 8, 2, [24, 24], [48, 48], 80209),
    ('gcn', 9228, 16, (16, 16), [48, 48], [64, 64], 168033),
    ('gcn', 26782, 16, (16, 16), [48, 48], [64, 64], 448897),
    ('gcn', 23843, 16, (16, 16), [48, 48], [64, 64], 401873),
]


@pytest.mark.parametrize('kind,n,d,hid,dense,clf,expected', PARAM_KATS)
def test_param_count_kats_oracle(kind, n, d, hid, dense, clf, expected):
    rng = np.random.default_rng(0)
    gnn = ow.gnn(rng, kind, n, d, hid if kind != 'lightgcn' else (), hid if kind == 'lightgcn' else 0)
    head = ow.basic_head(rng, ow.gnn_out_dim(gnn), dense, clf)
    assert om.count_params(gnn, head) == expected


def test_param_count_kats_hybrid_and_kge():
    rng = np.random.default_rng(0)
    gnn = ow.gnn(rng, 'gcn', 9228, 16, (16, 16))
    head = ow.hybrid_head(rng, 48, 768, [[48, 48], [256, 64], [64, 64]], [64, 64])
    assert om.count_params(gnn, head) == 619489                         # doc.pdf p.24 Table 7
    assert om.count_params(ow.basic_head(rng, 768, [512, 256, 128], [64, 64])) == 1136577      # p.22 Table 5


def test_dataset_shape_kats():
    """doc.pdf p.19 Table 3 statistics the generator's constants are taken from."""
    from deep_cbrs_amar_renaissance_amd.data import synthetic as sy
    assert abs(1 - sy.ML1M_RATINGS / (sy.ML1M_USERS * sy.ML1M_ITEMS) - 0.9509) < 1e-4
    assert abs(sy.ML1M_POSITIVE / sy.ML1M_USERS - 89.8) < 0.05
    assert abs(sy.ML1M_PROP_LINKS_RS2 / sy.ML1M_PROPS_RS2 - 4.0) < 0.01
    assert sy.ML1M_USERS + sy.ML1M_ITEMS == 9228 and 9228 + sy.ML1M_PROPS_RS2 == 26782


def _unflatten_gnn(z, kind):
    layers = []
    k = 0
    while any(key.startswith('gnn.layers.{}.'.format(k)) for key in z.files) or (kind == 'lightgcn' and k < 2):
        layers.append({key.split('.')[-1]: z[key] for key in z.files if key.startswith('gnn.layers.{}.'.format(k))})
        k += 1
    return {'kind': kind, 'embeddings': z['gnn.embeddings'], 'layers': layers,
            'final_node': 'mean' if kind == 'lightgcn' else 'concatenation'}


def _unflatten_net(z, prefix):
    out, k = [], 0
    while prefix + '.{}.0'.format(k) in z.files:
        out.append((z[prefix + '.{}.0'.format(k)], z[prefix + '.{}.1'.format(k)]))
        k += 1
    return out


def load_extra_golden(path):
    """Fixtures of the families added later (DGCF, hybrid-gnn-tweaks heads): (z, adj, gnn weights, head weights)."""
    z = np.load(path)
    n = int(z['n'])
    adj = sparse.coo_matrix((z['adj_data'], (z['adj_row'], z['adj_col'])), shape=(n, n))
    kind = 'dgcf' if 'dgcf' in os.path.basename(path) else 'gcn'
    gnn = _unflatten_gnn(z, kind)
    if kind == 'dgcf':
        gnn['final_node'] = 'mean'
    head = {}
    for name in sorted({key.split('.')[1] for key in z.files if key.startswith('head.')}):
        if name.startswith('fuse'):
            head[name] = {key.split('.')[-1]: z[key] for key in z.files if key.startswith('head.{}.'.format(name))}
        else:
            head[name] = _unflatten_net(z, 'head.' + name)
    return z, adj, gnn, head


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'extra_*.npz'))))
def test_oracle_reproduces_extra_golden(path):
    z, adj, gnn, head = load_extra_golden(path)
    if gnn['kind'] == 'dgcf':
        assert np.allclose(om.propagate(adj, gnn, np.float64), z['emb_f64'], rtol=0, atol=1e-15)
        got = om.basic_gnn_scores(adj, gnn, head, z['u_ids'], z['i_ids'], np.float64)
    else:
        got = om.hybrid_gnn_scores(adj, gnn, head, z['u_ids'], z['i_ids'], z['bert'], np.float64, feature_based=bool(z['feature_based']))
    assert np.allclose(got, z['scores_f64'], rtol=0, atol=1e-15)


def load_golden(path):
    z = np.load(path)
    kind = os.path.basename(path).split('_')[1]
    n = int(z['n'])
    adj = sparse.coo_matrix((z['adj_data'], (z['adj_row'], z['adj_col'])), shape=(n, n))
    head = {k: _unflatten_net(z, 'head.' + k) for k in ('unet', 'inet', 'clf')}
    return z, kind, adj, _unflatten_gnn(z, kind), head


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'basic_*.npz'))))
def test_oracle_reproduces_golden(path):
    z, kind, adj, gnn, head = load_golden(path)
    assert np.array_equal(om.propagate(adj, gnn, np.float64), z['emb_f64'])
    s64 = om.basic_gnn_scores(adj, gnn, head, z['u_ids'], z['i_ids'], np.float64)
    assert np.allclose(s64, z['scores_f64'], rtol=0, atol=1e-15)
    assert np.abs(om.basic_gnn_scores(adj, gnn, head, z['u_ids'], z['i_ids'], np.float32) - z['scores_f64']).max() < 1e-5
    faithful = om.basic_gnn_scores(adj, gnn, head, z['u_ids'], z['i_ids'], np.float64, batch=64)
    assert np.array_equal(faithful, s64)
    for k in (5, 10):
        tu, ti, _ = om.top_k(z['u_ids'], z['i_ids'], s64, z['users'], z['items'], k)
        assert np.array_equal(tu, z['top{}_users'.format(k)]) and np.array_equal(ti, z['top{}_items'.format(k)])


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(GOLDEN, 'basic_*.npz'))))
def test_second_implementation_matches_golden(path):
    """torch-CPU index_add_ implementation (oracle/torch_ref.py) vs the numpy/scipy oracle's golden output."""
    z, kind, adj, gnn, head = load_golden(path)
    half = adj.nnz // 2                                        # symmetrize() appends the transposed half
    tgt = torch.from_numpy(np.concatenate([adj.row[:half], adj.col[:half]]).astype(np.int64))
    src = torch.from_numpy(np.concatenate([adj.col[:half], adj.row[:half]]).astype(np.int64))
    got = torch_ref.propagate(tgt, src, gnn, torch.float64)
    assert helpers.rel_err(got, z['emb_f64']) < 1e-6


@pytest.mark.parametrize('kind', ['gcn', 'lightgcn', 'sage', 'gat'])
def test_two_implementations_agree_on_ml1m_shape(ml1m_s1, kind):
    rng = np.random.default_rng(42)
    adj = ml1m_s1['adj_uip']
    gnn = ow.gnn(rng, kind, adj.shape[0], 8, (8, 8), 2, bias_range=0.05)
    tgt, src = torch_ref.edges_from_ratings(ml1m_s1['train'], adj.shape[0], ml1m_s1['triples'], len(ml1m_s1['users']))
    e64 = om.propagate(adj, gnn, np.float64)
    assert helpers.rel_err(torch_ref.propagate(tgt, src, gnn, torch.float64), e64) < 1e-6
    assert helpers.rel_err(om.propagate(adj, gnn, np.float32), e64) < 5e-6          # fp32 error bound
    assert helpers.rel_err(torch_ref.propagate(tgt, src, gnn, torch.float32), e64) < 5e-6


def test_top_k_tie_rule():
    users, items = np.array([10, 20]), np.array([100, 200, 300])
    u = np.array([0, 0, 0, 1, 1])
    i = np.array([4, 3, 2, 2, 3])
    s = np.array([0.5, 0.5, 0.9, 0.1, 0.1], dtype=np.float32)
    tu, ti, ts = om.top_k(u, i, s, users, items, 2)
    assert tu.tolist() == [10, 10, 20, 20] and ti.tolist() == [100, 200, 100, 200]   # ties: item id ascending


def _by_user(users_out, items_out, scores_out):
    res = {}
    for u, i, s in zip(users_out.tolist(), items_out.tolist(), scores_out.tolist()):
        res.setdefault(u, []).append((i, s))
    return res


@pytest.mark.parametrize('k', [5, 10])
def test_top_k_reproduces_the_reference_functions_own_output(k):
    """tests/golden/topk_reference.npz holds inputs and outputs of the REFERENCE's `top_k_predictions`
    (`/root/reference/src/utilities/metrics.py:11-34`), run in the build container by tests/golden/make_topk_reference_golden.py —
    the one function of the hot path that imports without TensorFlow.  With distinct scores the oracle must give the same rows in
    the same order per user (id mapping, score-descending order, fewer than k pairs); with tied scores the rows may differ only
    inside a tie (the reference keeps pandas' sort order there, the oracle breaks ties on item id) and the scores are the same."""
    z = np.load(os.path.join(GOLDEN, 'topk_reference.npz'))
    users, items = z['users'], z['items']
    for name in ('distinct', 'ties'):
        pred = z['pred_' + name]
        tu, ti, ts = om.top_k(pred[:, 0], pred[:, 1], pred[:, 2], users, items, k)
        got = _by_user(tu, ti, ts)
        want = _by_user(z['{}_k{}_users'.format(name, k)], z['{}_k{}_items'.format(name, k)], z['{}_k{}_scores'.format(name, k)])
        assert sorted(got) == sorted(want)
        assert len(want[int(users[7])]) == 2                           # a user with fewer than k test pairs keeps what it has
        for u in want:
            assert [s for _, s in got[u]] == [s for _, s in want[u]], (name, u)       # bit-equal float64 scores, same order
            if name == 'distinct':
                assert [i for i, _ in got[u]] == [i for i, _ in want[u]], u
            else:
                strict = [i for i, s in want[u] if [t for _, t in want[u]].count(s) == 1 and s > want[u][-1][1]]
                assert [i for i, s in got[u] if i in strict] == strict, u              # everything outside a tie matches exactly


def _ref_coo(z, tag, j=0):
    return z['{}_{}_row'.format(tag, j)], z['{}_{}_col'.format(tag, j)], z['{}_{}_val'.format(tag, j)], tuple(z['{}_{}_shape'.format(tag, j)]), str(z['{}_{}_dtype'.format(tag, j)])


def _same_coo(m, ref):
    r, c, v, shape, dtype = ref
    m = m.tocoo() if not sparse.isspmatrix_coo(m) else m
    assert tuple(m.shape) == shape and str(m.dtype) == dtype
    assert np.array_equal(m.row, r) and np.array_equal(m.col, c) and np.array_equal(np.asarray(m.data), v)      # same triplets, same ORDER


@pytest.mark.parametrize('sym', [True, False])
def test_graph_construction_reproduces_the_reference_functions_own_output(sym):
    """tests/golden/graph_reference.npz holds inputs and outputs of the REFERENCE's `load_train_test_ratings`,
    `build_adjacency_matrix`, `symmetrize_matrix` and `get_user_properties` (loaders.py:11-82, preprocess.py:9-170, math.py:6-21),
    executed in the build container by tests/golden/make_graph_reference_golden.py.  The oracle's rows A1 / A1' must give the same
    contiguous ids, the same users / items arrays and bit-identical COO triplets IN THE SAME ORDER (duplicates kept) and dtype."""
    from oracle import graph as og
    z = np.load(os.path.join(GOLDEN, 'graph_reference.npz'))
    (train, test), (users, items) = og.remap_ratings(z['raw_train'], z['raw_test'])
    assert np.array_equal(train, z['train_indexed']) and np.array_equal(test, z['test_indexed'])
    assert np.array_equal(users, z['users']) and np.array_equal(items, z['items'])
    triples, props = og.remap_props(z['raw_props'], items)
    nu, ni, n_props = len(users), len(items), len(props)
    tag = 'sym' if sym else 'raw'
    _same_coo(og.adjacency_unary(train, nu, ni, sym), _ref_coo(z, 'unary_' + tag))
    _same_coo(og.adjacency_unary_uip(train, triples, nu, ni, n_props, sym), _ref_coo(z, 'unary_uip_' + tag))
    bi, kg = og.adjacency_unary_kg(train, triples, nu, ni, n_props, sym)
    _same_coo(bi, _ref_coo(z, 'unary_kg_' + tag, 0))
    _same_coo(kg, _ref_coo(z, 'unary_kg_' + tag, 1))
    if sym:
        up = og.user_properties(bi, kg, nu, ni)
        assert tuple(up.shape) == tuple(z['user_props_shape']) and str(up.dtype) == str(z['user_props_dtype'])
        assert np.array_equal(up.row, z['user_props_row']) and np.array_equal(up.col, z['user_props_col']) and np.array_equal(up.data, z['user_props_val'])


def test_weighted_sum_reduction_by_hand():
    """ReductionLayer('w-sum') = WeightedSum.call (reduction.py:54-55): reduce_sum(multiply(w * w, inputs), axis=0); weights start
    at ones (reduction.py:50), where it equals 'sum'; the torch restatement used as the training oracle agrees and gives
    d out / d w_l = 2 w_l X_l."""
    import torch
    from oracle import layers as ol
    from oracle import train as otrain
    hs = [np.array([[1.0, 2.0]]), np.array([[3.0, 4.0]])]
    assert np.array_equal(ol.reduce_layers(hs, 'w-sum', [2.0, -1.0]), np.array([[7.0, 12.0]]))     # 4 * [1, 2] + 1 * [3, 4]
    assert np.array_equal(ol.reduce_layers(hs, 'w-sum'), ol.reduce_layers(hs, 'sum'))
    adj = sparse.csr_matrix(np.array([[0.0, 1.0, 0.0], [1.0, 0.0, 1.0], [0.0, 1.0, 0.0]]))
    x0 = np.arange(6, dtype=np.float64).reshape(3, 2) / 10
    gnn = {'kind': 'lightgcn', 'embeddings': x0, 'layers': [{}, {}], 'final_node': 'w-sum', 'reduction_w': np.array([0.5, 2.0, -1.0])}
    want = om.propagate_from(adj, x0, gnn, np.float64, force_mean=False)
    w = torch.tensor(gnn['reduction_w'], requires_grad=True)
    st = {'kind': 'lightgcn', 'layers': [{}, {}], 'final_node': 'w-sum', 'reduction_w': w}
    got = otrain._torch_stack(adj, torch.tensor(x0), st, force_mean=False)
    assert np.allclose(got.detach().numpy(), want, atol=1e-12)
    got.sum().backward()
    a_hat = og.gcn_filter(adj).toarray()
    terms = [x0, a_hat @ x0, a_hat @ a_hat @ x0]
    assert np.allclose(w.grad.numpy(), [2 * wk * t.sum() for wk, t in zip(gnn['reduction_w'], terms)], atol=1e-12)
