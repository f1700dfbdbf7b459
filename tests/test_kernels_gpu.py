"""GPU parity: every C-ABI kernel against the CPU oracle on seeded inputs (pytest -m gpu).

Tolerances: fp32 kernels vs the fp32 oracle, relative to the largest reference magnitude:
2e-6 for single sums of <= a few thousand terms (different but fixed summation orders),
bit-exact for pure data movement and for top-k.
"""
import numpy as np
import pytest
import torch
from scipy import sparse

from oracle import layers as ol
from tests import helpers
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rand_csr(n, avg_deg, seed, empty_rows=True, long_row=True, dup=False):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, size=n)
    if empty_rows:
        deg[rng.integers(0, n, size=max(1, n // 10))] = 0
    if long_row:
        deg[rng.integers(0, n)] = min(n, 700)
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, size=len(rows))
    if not dup:
        key = np.unique(rows * n + cols)
        rows, cols = key // n, key % n
    vals = rng.uniform(-1, 1, size=len(rows)).astype(np.float32)
    return sparse.coo_matrix((vals, (rows, cols)), shape=(n, n))


def _dev_csr(m, **kw):
    from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR
    return DeviceCSR.from_scipy(m, **kw)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize('F', [4, 8, 16, 32, 64, 12, 24, 48, 96])       # 12 / 24 / 48 / 96 run as column chunks
@pytest.mark.parametrize('n', [1, 67, 1000])
def test_spmm_plain(hip, F, n):
    m = _rand_csr(n, 9, seed=F + n)
    a = _dev_csr(m)
    x = np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)
    y = torch.full((n, F), float('nan'), device=DEV)
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, _t(x), y)
    want = m.tocsr().astype(np.float64) @ x.astype(np.float64)
    assert rel_err(y.cpu().numpy(), want) < 2e-6


def test_spmm_binary_bias_relu_strided(hip):
    n, F = 500, 8
    m = _rand_csr(n, 12, seed=5, dup=True)
    a = _dev_csr(m, with_values=False)
    rng = np.random.default_rng(2)
    xbuf = rng.standard_normal((n, 24)).astype(np.float32)
    bias = rng.uniform(-1, 1, F).astype(np.float32)
    xd = _t(xbuf)
    out = torch.zeros((n, 24), device=DEV)
    hip.spmm_csr(a.rowptr, a.colidx, None, xd[:, 8:16], out[:, 16:24], bias=_t(bias), relu=True)
    ones = sparse.csr_matrix((np.ones(m.nnz), (m.row, m.col)), shape=m.shape)       # duplicates summed == counted
    want = np.maximum(ones @ xbuf[:, 8:16].astype(np.float64) + bias, 0)
    got = out.cpu().numpy()
    assert rel_err(got[:, 16:24], want) < 2e-6
    assert np.all(got[:, :16] == 0), "wrote outside its column slice"


@pytest.mark.parametrize('C', [24, 48, 20])
def test_column_chunked_widths(hip, C):
    """Widths the row kernels are not instantiated for (TwoStep / TwoWay 'concatenation' hand-over: 3 d) run as column
    chunks inside the C-ABI: the GCN layer, and the SpMM with its whole epilogue (bias, ReLU, running mean), on strided
    views; nothing outside the C columns is touched.  A fused next-layer product is refused for such a width."""
    n = 400
    m = _rand_csr(n, 10, seed=C)
    a = _dev_csr(m)
    rng = np.random.default_rng(C)
    hbuf, b = rng.standard_normal((n, C + 8)).astype(np.float32), rng.uniform(-1, 1, C).astype(np.float32)
    acc = rng.standard_normal((n, C)).astype(np.float32)
    hd = _t(hbuf)
    A = m.tocsr().astype(np.float64)
    want = np.maximum(A @ hbuf[:, 4:4 + C].astype(np.float64) + b, 0)
    y = torch.zeros((n, C + 8), device=DEV)
    hip.gcn_layer(a.rowptr, a.colidx, a.vals, hd[:, 4:4 + C], _t(b), y[:, 8:])
    got = y.cpu().numpy()
    assert rel_err(got[:, 8:], want) < 2e-6 and np.all(got[:, :8] == 0)
    y2, e = torch.zeros((n, C), device=DEV), torch.empty((n, C), device=DEV)
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, hd[:, 4:4 + C], y2, bias=_t(b), relu=True, acc_in=_t(acc), acc_out=e, acc_div=3)
    assert torch.equal(y2, y[:, 8:]), "chunked SpMM epilogue and chunked GCN layer must agree bit for bit"
    assert rel_err(e.cpu().numpy(), (acc.astype(np.float64) + want) / 3) < 2e-6
    with pytest.raises(Exception):
        hip.gcn_layer(a.rowptr, a.colidx, a.vals, hd[:, 4:4 + C], _t(b), y[:, 8:], Wnext=_t(rng.standard_normal((C, 8)).astype(np.float32)),
                      Hnext=torch.empty((n, 8), device=DEV))
    with pytest.raises(Exception):                           # not a multiple of 4
        hip.spmm_csr(a.rowptr, a.colidx, a.vals, _t(hbuf[:, :6].copy()), torch.empty((n, 6), device=DEV))


def test_spmm_running_mean(hip):
    """LightGCN: S1 = X0 + A.X0 ; E = (S1 + A.X1) / 3 with X1 = A.X0 (reduction.py:28-30)."""
    n, F = 300, 16
    m = _rand_csr(n, 7, seed=9)
    a = _dev_csr(m)
    x0 = np.random.default_rng(3).standard_normal((n, F)).astype(np.float32)
    x0d = _t(x0)
    x1 = torch.empty((n, F), device=DEV)
    s1 = torch.empty((n, F), device=DEV)
    e = torch.empty((n, F), device=DEV)
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, x0d, x1, acc_in=x0d, acc_out=s1)
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, x1, None, acc_in=s1, acc_out=e, acc_div=3)
    A = m.tocsr().astype(np.float32)
    h1 = ol.lightgcn_conv(x0, A)
    h2 = ol.lightgcn_conv(h1, A)
    want = ol.reduce_layers([x0, h1, h2], 'mean')
    assert rel_err(e.cpu().numpy(), want.astype(np.float64)) < 2e-6


@pytest.mark.parametrize('C,Cn', [(8, 8), (16, 32), (32, 16), (64, 8), (4, 4)])
def test_gcn_layer_fused_next(hip, C, Cn):
    n = 400
    m = _rand_csr(n, 10, seed=C)
    a = _dev_csr(m)
    rng = np.random.default_rng(C + Cn)
    h = rng.standard_normal((n, C)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, C).astype(np.float32)
    wn = rng.uniform(-0.5, 0.5, (C, Cn)).astype(np.float32)
    y = torch.empty((n, C), device=DEV)
    hn = torch.empty((n, Cn), device=DEV)
    hip.gcn_layer(a.rowptr, a.colidx, a.vals, _t(h), _t(b), y, Wnext=_t(wn), Hnext=hn)
    want_y = np.maximum(m.tocsr().astype(np.float64) @ h.astype(np.float64) + b, 0)
    assert rel_err(y.cpu().numpy(), want_y) < 2e-6
    assert rel_err(hn.cpu().numpy(), want_y @ wn.astype(np.float64)) < 3e-6
    y2 = torch.empty((n, C), device=DEV)
    hip.gcn_layer(a.rowptr, a.colidx, a.vals, _t(h), _t(b), y2)
    assert torch.equal(y, y2), "fused and unfused epilogues must agree bit for bit"


@pytest.mark.parametrize('F,C', [(8, 8), (24, 8), (5, 3), (64, 64), (16, 32), (16, 16), (8, 4), (12, 12), (32, 16), (8, 6)])   # C <= 16, aligned: one thread per row
def test_rowwise_xw(hip, F, C):
    n = 777
    rng = np.random.default_rng(F * C)
    x = rng.standard_normal((n, F)).astype(np.float32)
    w = rng.standard_normal((F, C)).astype(np.float32)
    a_s, a_n = rng.standard_normal(C).astype(np.float32), rng.standard_normal(C).astype(np.float32)
    h = torch.empty((n, C), device=DEV)
    cp = torch.empty((n, F), device=DEV)
    ss, sn = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    hip.rowwise_xw(_t(x), _t(w), h, copy_to=cp, a_self=_t(a_s), a_neigh=_t(a_n), s_self=ss, s_neigh=sn)
    want = x.astype(np.float64) @ w.astype(np.float64)
    assert rel_err(h.cpu().numpy(), want) < 2e-6
    assert np.array_equal(cp.cpu().numpy(), x)
    assert rel_err(ss.cpu().numpy(), want @ a_s) < 3e-6
    assert rel_err(sn.cpu().numpy(), want @ a_n) < 3e-6


@pytest.mark.parametrize('F', [8, 16])
def test_rowwise_xw_one_thread_per_row_on_slices(hip, F):
    """The one-thread-per-row form (square widths 8 / 16, float4-aligned operands) with its operands as column slices of wider
    buffers — the layer input inside a concatenation buffer, the slice copy into another — and an operand whose leading
    dimension is not a multiple of 4 floats, which must fall back to the generic kernel with the same results."""
    n = 1500
    rng = np.random.default_rng(F)
    x = rng.standard_normal((n, F)).astype(np.float32)
    w = rng.standard_normal((F, F)).astype(np.float32)
    want = x.astype(np.float64) @ w.astype(np.float64)
    for ld_in, off in ((3 * F, F), (3 * F + 1, 0)):
        wide = torch.zeros((n, ld_in), device=DEV)
        wide[:, off:off + F] = _t(x)
        h = torch.full((n, F), float('nan'), device=DEV)
        cat = torch.full((n, 2 * F + 4), float('nan'), device=DEV)
        hip.rowwise_xw(wide[:, off:off + F], _t(w), h, copy_to=cat[:, 4:4 + F])
        assert rel_err(h.cpu().numpy(), want) < 2e-6
        assert np.array_equal(cat[:, 4:4 + F].cpu().numpy(), x) and torch.isnan(cat[:, :4]).all() and torch.isnan(cat[:, 4 + F:]).all()


@pytest.mark.parametrize('F,C', [(8, 8), (24, 16), (5, 3)])
def test_rowwise_xw_row_scale(hip, F, C):
    """The fused d^-1/2 pre-scale of the value-free XS chain: H = diag(s).(X.W), bit-identical to X.W followed by the
    separate row_affine pass; together with the attention scalars it is refused."""
    n = 1234
    rng = np.random.default_rng(F + C)
    x, w = _t(rng.standard_normal((n, F)).astype(np.float32)), _t(rng.standard_normal((F, C)).astype(np.float32))
    s = _t(rng.uniform(0.1, 1.0, n).astype(np.float32))
    h, h2, cp = torch.empty((n, C), device=DEV), torch.empty((n, C), device=DEV), torch.empty((n, F), device=DEV)
    hip.rowwise_xw(x, w, h, copy_to=cp, row_scale=s)
    hip.rowwise_xw(x, w, h2)
    hip.row_affine(h2, s, h2)
    assert torch.equal(h, h2)
    assert torch.equal(cp, x)
    with pytest.raises(Exception):
        hip.rowwise_xw(x, w, h, a_self=_t(np.ones(C, np.float32)), a_neigh=_t(np.ones(C, np.float32)),
                       s_self=torch.empty(n, device=DEV), s_neigh=torch.empty(n, device=DEV), row_scale=s)


@pytest.mark.parametrize('F,C', [(8, 8), (16, 16), (32, 32), (4, 8), (8, 5)])
@pytest.mark.parametrize('self_loop', [True, False])
def test_sage_layer(hip, F, C, self_loop):
    n = 350
    m = _rand_csr(n, 8, seed=F + C, dup=True)
    m = sparse.coo_matrix((m.data[m.row != m.col], (m.row[m.row != m.col], m.col[m.row != m.col])), shape=m.shape)
    a = _dev_csr(m, with_values=False, drop_diagonal=True)
    rng = np.random.default_rng(F)
    x = rng.standard_normal((n, F)).astype(np.float32)
    w = rng.uniform(-0.6, 0.6, (2 * F, C)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, C).astype(np.float32)
    y = torch.empty((n, C), device=DEV)
    hip.sage_layer(a.rowptr, a.colidx, _t(x), _t(w), _t(b), y, self_loop=self_loop)
    # kernel: row i aggregates over its CSR row; oracle: targets = col -> feed the transposed edge list
    want = ol.sage_conv(x.astype(np.float64), m.col, m.row, w.astype(np.float64), b.astype(np.float64),
                        self_loops=self_loop)
    assert rel_err(y.cpu().numpy(), want) < 5e-6


@pytest.mark.parametrize('F,C', [(8, 8), (16, 16), (24, 24), (4, 32), (32, 12), (64, 64), (48, 20)])
def test_sage_tail(hip, F, C):
    """GraphSageConv after its mean aggregate: relu(l2_normalize([x || agg] . W + b)) in one pass, on strided views; a row
    whose pre-activation is all zero stays zero (the 1e-12 clamp of tf.nn.l2_normalize)."""
    n = 1000
    rng = np.random.default_rng(F * 100 + C)
    xbuf = rng.standard_normal((n + 5, F + 8)).astype(np.float32)
    g = rng.standard_normal((n, F)).astype(np.float32)
    w = rng.uniform(-0.5, 0.5, (2 * F, C)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, C).astype(np.float32)
    xbuf[7], g[7], b0 = 0, 0, b.copy()
    xd, out = _t(xbuf), torch.full((n, C + 4), -7.0, device=DEV)
    hip.sage_tail(xd[:, 4:4 + F], _t(g), _t(w), _t(b), out[:, :C])
    z = np.concatenate([xbuf[:n, 4:4 + F], g], 1).astype(np.float64) @ w.astype(np.float64) + b
    want = np.maximum(z / np.sqrt(np.maximum((z * z).sum(1, keepdims=True), 1e-12)), 0)
    got = out.cpu().numpy()
    assert rel_err(got[:, :C], want) < 2e-6 and np.all(got[:, C:] == -7.0)
    z7 = b0.astype(np.float64)
    assert np.allclose(got[7, :C], np.maximum(z7 / np.sqrt(max((z7 * z7).sum(), 1e-12)), 0), atol=1e-6)
    with pytest.raises(Exception):
        hip.sage_tail(xd[:, 4:4 + F], _t(g), _t(w[:-1].copy()), _t(b), out[:, :C])


@pytest.mark.parametrize('C', [4, 8, 16, 32, 64, 24, 48])
@pytest.mark.parametrize('self_loop', [True, False])
def test_gat_layer(hip, C, self_loop):
    n, F = 350, 8
    m = _rand_csr(n, 8, seed=C, dup=True)
    m = sparse.coo_matrix((m.data[m.row != m.col], (m.row[m.row != m.col], m.col[m.row != m.col])), shape=m.shape)
    a = _dev_csr(m, with_values=False, drop_diagonal=True)
    rng = np.random.default_rng(C)
    x = rng.standard_normal((n, F)).astype(np.float32)
    w = rng.uniform(-0.6, 0.6, (F, C)).astype(np.float32)
    a_s, a_n = rng.uniform(-1, 1, C).astype(np.float32), rng.uniform(-1, 1, C).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, C).astype(np.float32)
    h = torch.empty((n, C), device=DEV)
    ss, sn = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    hip.rowwise_xw(_t(x), _t(w), h, a_self=_t(a_s), a_neigh=_t(a_n), s_self=ss, s_neigh=sn)
    y = torch.empty((n, C), device=DEV)
    hip.gat_layer(a.rowptr, a.colidx, h, ss, sn, _t(b), y, self_loop=self_loop)
    want, alpha = ol.gat_conv(x.astype(np.float64), m.col, m.row, w.astype(np.float64), a_s.astype(np.float64),
                              a_n.astype(np.float64), b.astype(np.float64), self_loops=self_loop)
    assert rel_err(y.cpu().numpy(), want) < 1e-5


@pytest.mark.parametrize('M,K,N', [(1, 1, 1), (130, 24, 24), (257, 48, 48), (1000, 48, 1), (300, 768, 256),
                                   (64, 17, 65), (2048, 96, 48), (5, 256, 64)])
@pytest.mark.parametrize('act', ['relu', 'sigmoid', None])
def test_dense(hip, M, K, N, act):
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.uniform(-0.3, 0.3, (K, N)).astype(np.float32)
    b = rng.uniform(-0.2, 0.2, N).astype(np.float32)
    y = torch.empty((M, N), device=DEV)
    hip.dense(_t(x), _t(w), _t(b), y, act=act)
    want = ol.dense(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), act)
    assert rel_err(y.cpu().numpy(), want) < 3e-6


def test_dense_guard_free_form_with_row_gather(hip):
    """The guard-free staging forms of amar_dense_f32 (K % 16 == 0): the 128 x 64 tile (N % 64 == 0: N = 64, 192) and the 128 x 128
    tile with four accumulators per wave (N % 128 == 0: N = 128, 256 — one and two column blocks).  Row gather through ids, last
    row tiles that are mostly past M (those lanes re-read row M - 1 and are dropped), several row tiles (the XCD-affine tile
    order), all three activations."""
    rng = np.random.default_rng(5)
    table = rng.standard_normal((400, 48)).astype(np.float32)
    for N in (64, 128, 192, 256):
        w = rng.uniform(-0.3, 0.3, (48, N)).astype(np.float32)
        b = rng.uniform(-0.2, 0.2, N).astype(np.float32)
        for M in (1, 129, 333, 1100):
            ids = rng.integers(7, 400, size=M).astype(np.int32)
            for act in ('relu', 'sigmoid', None):
                y = torch.full((M, N), float('nan'), device=DEV)
                hip.dense(_t(table), _t(w), _t(b), y, act=act, ids=_t(ids))
                want = ol.dense(table[ids].astype(np.float64), w.astype(np.float64), b.astype(np.float64), act)
                assert rel_err(y.cpu().numpy(), want) < 3e-6
    # a deep product without ids or bias (the BERT tower's first layer in small): 768 -> 256
    x = rng.standard_normal((300, 768)).astype(np.float32)
    w = rng.uniform(-0.05, 0.05, (768, 256)).astype(np.float32)
    y = torch.full((300, 256), float('nan'), device=DEV)
    hip.dense(_t(x), _t(w), None, y, act=None)
    assert rel_err(y.cpu().numpy(), x.astype(np.float64) @ w.astype(np.float64)) < 3e-6


def test_dense_gather_and_concat_slices(hip):
    """embedding_lookup fused into the load (basic.py:73-74) + two producers filling one concat buffer (basic.py:35)."""
    rng = np.random.default_rng(0)
    table = rng.standard_normal((500, 24)).astype(np.float32)
    ids_u = rng.integers(0, 500, size=333).astype(np.int32)
    ids_i = rng.integers(0, 500, size=333).astype(np.int32)
    wu, wi = rng.uniform(-0.3, 0.3, (24, 20)).astype(np.float32), rng.uniform(-0.3, 0.3, (24, 20)).astype(np.float32)
    bu, bi = rng.uniform(-0.2, 0.2, 20).astype(np.float32), rng.uniform(-0.2, 0.2, 20).astype(np.float32)
    cat = torch.full((333, 40), float('nan'), device=DEV)
    td = _t(table)
    hip.dense(td, _t(wu), _t(bu), cat[:, :20], act='relu', ids=_t(ids_u))
    hip.dense(td, _t(wi), _t(bi), cat[:, 20:], act='relu', ids=_t(ids_i))
    want = np.concatenate([ol.dense(table[ids_u].astype(np.float64), wu.astype(np.float64), bu, 'relu'),
                           ol.dense(table[ids_i].astype(np.float64), wi.astype(np.float64), bi, 'relu')], axis=1)
    assert rel_err(cat.cpu().numpy(), want) < 3e-6


def test_copy_and_reduce_layers(hip):
    rng = np.random.default_rng(4)
    xs = [rng.standard_normal((123, 8)).astype(np.float32) for _ in range(3)]
    cat = torch.empty((123, 24), device=DEV)
    for k, x in enumerate(xs):
        hip.copy_columns(_t(x), cat[:, 8 * k:8 * k + 8])
    assert np.array_equal(cat.cpu().numpy(), np.concatenate(xs, axis=1))
    for mean in (False, True):
        out = torch.empty((123, 8), device=DEV)
        hip.reduce_layers(cat, 3, 8, out, mean=mean)
        want = ol.reduce_layers(xs, 'mean' if mean else 'sum')
        assert np.array_equal(out.cpu().numpy(), want), "same fp32 operations in the same order: bit-exact"


@pytest.mark.parametrize('k', [5, 10])
def test_topk_reproduces_the_reference_functions_own_output(hip, k):
    """The device top-k (amar_topk_segmented_f32 behind utilities.metrics.top_k_predictions) against the vectors the REFERENCE's
    own `top_k_predictions` produced (tests/golden/topk_reference.npz, made by tests/golden/make_topk_reference_golden.py from
    /root/reference/src/utilities/metrics.py:11-34): same users, same items in the same order, scores equal to fp32 rounding."""
    import os
    from deep_cbrs_amar_renaissance_amd.utilities.metrics import top_k_predictions
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'topk_reference.npz'))
    users, items, pred = z['users'], z['items'], z['pred_distinct']
    assert len(np.unique(pred[:, 2].astype(np.float32))) == len(pred)          # distinct in fp32 too: the order is defined
    df = top_k_predictions(pred, users, items, k=k)
    got_u, got_i, got_s = df['users'].to_numpy(), df['items'].to_numpy(), df['scores'].to_numpy()
    order = np.argsort(got_u, kind='stable')
    assert np.array_equal(got_u[order], z['distinct_k{}_users'.format(k)])
    assert np.array_equal(got_i[order], z['distinct_k{}_items'.format(k)])
    assert np.abs(got_s[order] - z['distinct_k{}_scores'.format(k)]).max() < 1e-7


def test_topk_segmented(hip):
    from oracle import models as om
    from deep_cbrs_amar_renaissance_amd.utilities.metrics import top_k_arrays
    rng = np.random.default_rng(7)
    n_users, n_items = 200, 500
    counts = rng.integers(0, 90, size=n_users)
    counts[3], counts[5], counts[7] = 0, 1, 300
    u = np.repeat(np.arange(n_users), counts)
    i = np.concatenate([rng.choice(n_items, size=c, replace=False) for c in counts]) + n_users
    s = rng.random(len(u)).astype(np.float32)
    s[rng.integers(0, len(s), size=len(s) // 3)] = 0.5                      # many exact ties
    perm = rng.permutation(len(u))
    u, i, s = u[perm], i[perm], s[perm]
    users, items = np.arange(n_users) * 2 + 10, np.arange(n_items) * 3 + 7
    for k in (5, 10):
        seg_users, top_items, top_scores = top_k_arrays(u, i, s, k)
        valid = top_items >= 0
        got_u = users[np.repeat(seg_users, k).reshape(-1, k)[valid]]
        got_i = items[top_items[valid] - n_users]
        want_u, want_i, want_s = om.top_k(u, i, s, users, items, k)
        assert np.array_equal(got_u, want_u) and np.array_equal(got_i, want_i)
        assert np.array_equal(top_scores[valid].astype(np.float64), want_s)


def test_bad_arguments_fail_loudly(hip):
    a = _dev_csr(_rand_csr(10, 3, seed=1))
    x = torch.zeros((10, 10), device=DEV)
    with pytest.raises(Exception):
        hip.spmm_csr(a.rowptr, a.colidx, a.vals, x, torch.empty((10, 10), device=DEV))      # F = 10: not a multiple of 4
    with pytest.raises(Exception):
        hip.spmm_csr(a.rowptr, a.colidx, a.vals, torch.zeros((10, 8)), torch.empty((10, 8), device=DEV))  # CPU tensor


CHAIN_CASES = [
    # (Da, Db, units, final act)          towers / classifiers of the econfigs grids
    (24, 0, [24, 24], 'relu'), (8, 0, [24, 24], 'relu'), (96, 0, [96, 48], 'relu'), (128, 0, [128, 64], 'relu'),
    (24, 24, [48, 48, 1], 'sigmoid'), (48, 48, [64, 64, 1], 'sigmoid'), (64, 64, [64, 64, 1], 'sigmoid'),
    (24, 24, [64, 64], 'relu'), (64, 64, [64, 64], 'relu'), (4, 0, [20], 'relu'), (16, 8, [12, 1], 'sigmoid'),
]


@pytest.mark.parametrize('Da,Db,units,final', CHAIN_CASES)
@pytest.mark.parametrize('P', [1, 77, 5000])
def test_chain_fused_stack(hip, Da, Db, units, final, P):
    rng = np.random.default_rng(Da + Db + P + len(units))
    na, nb = 300, 200
    A = rng.standard_normal((na + 7, Da)).astype(np.float32)
    B = rng.standard_normal((nb + 11, max(Db, 4))).astype(np.float32)[:, :Db] if Db else None
    ids_a = rng.integers(7, na + 7, size=P).astype(np.int32)
    ids_b = rng.integers(11, nb + 11, size=P).astype(np.int32)
    dims = [Da + Db] + units
    ks = [rng.uniform(-0.4, 0.4, (dims[k], dims[k + 1])).astype(np.float32) for k in range(len(units))]
    bs = [rng.uniform(-0.2, 0.2, dims[k + 1]).astype(np.float32) for k in range(len(units))]
    acts = ['relu'] * (len(units) - 1) + [final]
    assert hip.chain_supported(dims, Da, Db)
    blob, pdims = hip.chain_pack(ks, bs)
    assert pdims == dims
    out = torch.full((P, dims[-1]), float('nan'), device=DEV)
    Ad = _t(A)[7:]                                   # row base folded into the view ...
    Bd = _t(np.ascontiguousarray(B)) if Db else None
    hip.chain(Ad, _t(blob), dims, acts, out, ids_a=_t(ids_a), base_a=7, B=Bd, ids_b=_t(ids_b) if Db else None, base_b=0)
    x = A[ids_a].astype(np.float64)                  # ... so ids - base_a indexes the view: A[7:][ids - 7] == A[ids]
    if Db:
        x = np.concatenate([x, B[ids_b].astype(np.float64)], axis=1)
    for k, b, a in zip(ks, bs, acts):
        x = ol.dense(x, k.astype(np.float64), b.astype(np.float64), a)
    assert rel_err(out.cpu().numpy(), x) < 5e-6
    # no ids: rows taken in order
    if P <= na and (not Db or P <= nb):
        out2 = torch.empty((P, dims[-1]), device=DEV)
        hip.chain(_t(A), _t(blob), dims, acts, out2, B=Bd)
        x = A[:P].astype(np.float64)
        if Db:
            x = np.concatenate([x, B[:P].astype(np.float64)], axis=1)
        for k, b, a in zip(ks, bs, acts):
            x = ol.dense(x, k.astype(np.float64), b.astype(np.float64), a)
        assert rel_err(out2.cpu().numpy(), x) < 5e-6


def test_chain_rejects_unsupported_shapes(hip):
    assert not hip.chain_supported([768, 512, 256, 128], 768)
    assert not hip.chain_supported([48, 48, 1], 22, 26)
    blob, dims = hip.chain_pack([np.zeros((8, 8), np.float32)], [np.zeros(8, np.float32)])
    with pytest.raises(Exception):
        hip.chain(torch.zeros((4, 6), device=DEV), _t(blob), dims, ['relu'], torch.zeros((4, 8), device=DEV))


def test_gather_copy(hip):
    rng = np.random.default_rng(1)
    src = rng.standard_normal((50, 12)).astype(np.float32)
    ids = rng.integers(5, 50, size=31).astype(np.int32)
    dst = torch.zeros((31, 20), device=DEV)
    hip.copy_columns(_t(src)[5:], dst[:, 4:16], ids=_t(ids), base=5)
    assert np.array_equal(dst.cpu().numpy()[:, 4:16], src[ids])


@pytest.mark.parametrize('F', [4, 8, 16, 32, 64])
@pytest.mark.parametrize('n,cbits', [(1, 4), (67, 3), (1000, 6), (1000, 12), (4097, 9)])
def test_spmm_sliced_jagged(hip, F, n, cbits):
    """SJ image (lane-per-row, column slices) against the dense oracle product, all epilogues."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import SlicedJagged
    m = _rand_csr(n, 9, seed=F + n + cbits)
    a = _dev_csr(m)
    sj = SlicedJagged.from_csr(a, cbits)
    assert int(sj.wave_start[-1]) == a.nnz and sj.n_slices == max(1, (n + (1 << cbits) - 1) >> cbits)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, F)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, F).astype(np.float32)
    wn = rng.uniform(-0.5, 0.5, (F, 8)).astype(np.float32)
    A = m.tocsr().astype(np.float64)
    y = torch.full((n, F), float('nan'), device=DEV)
    hip.spmm_sj(sj, _t(x), y)
    assert rel_err(y.cpu().numpy(), A @ x.astype(np.float64)) < 2e-6
    hn = torch.full((n, 8), float('nan'), device=DEV)
    hip.spmm_sj(sj, _t(x), y, bias=_t(b), relu=True, Wnext=_t(wn), Hnext=hn)
    want = np.maximum(A @ x.astype(np.float64) + b, 0)
    assert rel_err(y.cpu().numpy(), want) < 2e-6 and rel_err(hn.cpu().numpy(), want @ wn.astype(np.float64)) < 3e-6
    s1, e = torch.empty((n, F), device=DEV), torch.empty((n, F), device=DEV)
    xd = _t(x)
    hip.spmm_sj(sj, xd, y, acc_in=xd, acc_out=s1)
    hip.spmm_sj(sj, y, None, acc_in=s1, acc_out=e, acc_div=3)
    x1 = A @ x.astype(np.float64)
    assert rel_err(e.cpu().numpy(), (x + x1 + A @ x1) / 3) < 3e-6


def test_spmm_kinds_agree_on_models(hip, monkeypatch):
    """The CSR and SJ forms of the propagation give the same node table (to rounding) for GCN and LightGCN."""
    from deep_cbrs_amar_renaissance_amd.models import basic
    from tests import helpers
    g = helpers.tiny_graph(n_users=90, n_items=70, n_ratings=2500, seed=6, n_props=40, n_links=120)
    for name in ('BasicGCN', 'BasicLightGCN'):
        model = getattr(basic, name)(g['adj'], embedding_dim=8, n_hiddens=[8, 16], n_layers=2, dense_units=[24, 24], clf_units=[48])
        helpers.randomize_biases(model, seed=2)
        monkeypatch.setenv('AMAR_SPMM_KIND', 'csr')
        e_csr = model.gnn(None).cpu().numpy()
        for kind in ('sj', 'xs'):
            monkeypatch.setenv('AMAR_SPMM_KIND', kind)
            e_k = model.gnn(None).cpu().numpy()
            assert rel_err(e_k, e_csr.astype(np.float64)) < 2e-6
    # GraphSAGE: fused row kernel vs mean aggregate on the XCD-sliced value-free image + dense + l2-normalise
    for loops in (True, False):
        model = basic.BasicGraphSage(g['adj'], embedding_dim=8, n_hiddens=[8, 16], n_layers=2, dense_units=[24, 24], clf_units=[48])
        for layer in model.gnn.gnn_layers.seq_layers:
            layer.self_loops = loops
        helpers.randomize_biases(model, seed=2)
        monkeypatch.setenv('AMAR_SPMM_KIND', 'csr')
        e_row = model.gnn(None).cpu().numpy()
        monkeypatch.setenv('AMAR_SPMM_KIND', 'xs')
        e_xs = model.gnn(None).cpu().numpy()
        assert rel_err(e_xs, e_row.astype(np.float64)) < 3e-6
    # GAT: row kernel vs XCD-sliced online-softmax form
    model = basic.BasicGAT(g['adj'], embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48])
    helpers.randomize_biases(model, seed=2)
    monkeypatch.setenv('AMAR_SPMM_KIND', 'csr')
    e_row = model.gnn(None).cpu().numpy()
    monkeypatch.setenv('AMAR_SPMM_KIND', 'xs')
    e_xs = model.gnn(None).cpu().numpy()
    assert rel_err(e_xs, e_row.astype(np.float64)) < 3e-6
    # a device-built adjacency (factors known) takes the value-free XS image and the pre-scaled fused GCN chain
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    coo = g['adj'].tocoo()
    keep = coo.row < coo.col
    a = gcn_filter_device(torch.from_numpy(coo.row[keep].astype(np.int64)).to(DEV), torch.from_numpy(coo.col[keep].astype(np.int64)).to(DEV), coo.shape[0])
    for name in ('BasicGCN', 'BasicLightGCN'):
        model = getattr(basic, name)(a, embedding_dim=8, n_hiddens=[8, 16], n_layers=2, dense_units=[24, 24], clf_units=[48])
        helpers.randomize_biases(model, seed=2)
        monkeypatch.setenv('AMAR_SPMM_KIND', 'csr')
        e_csr = model.gnn(None).cpu().numpy()
        monkeypatch.setenv('AMAR_SPMM_KIND', 'xs')
        e_xs = model.gnn(None).cpu().numpy()
        assert model.gnn.gnn_layers.adj_matrix.xcd_sliced().row_scale is not None
        assert rel_err(e_xs, e_csr.astype(np.float64)) < 2e-6


@pytest.mark.parametrize('F', [4, 8, 16, 32, 64])
@pytest.mark.parametrize('n', [1, 67, 1000, 4097])
def test_spmm_xcd_sliced(hip, F, n):
    """XS image (XCD-affine column slices: partial products + combine) against the dense oracle product."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced
    m = _rand_csr(n, 9, seed=F + n) + sparse.identity(n, dtype=np.float32, format='coo') * 0.25
    m = m.tocoo()
    a = _dev_csr(m)
    xs = XcdSliced.from_csr(a)
    assert xs.colidx.numel() + int((xs.diag != 0).sum()) >= 0 and xs.rowptr.numel() == xs.n_slices * n + 1
    rng = np.random.default_rng(1)
    x = rng.standard_normal((n, F)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, F).astype(np.float32)
    wn = rng.uniform(-0.5, 0.5, (F, 12)).astype(np.float32)
    A = m.tocsr().astype(np.float64)
    y = torch.full((n, F), float('nan'), device=DEV)
    hip.spmm_xs(xs, _t(x), y)
    assert rel_err(y.cpu().numpy(), A @ x.astype(np.float64)) < 2e-6
    hn = torch.full((n, 12), float('nan'), device=DEV)
    cat = torch.zeros((n, 2 * F + 4), device=DEV)
    hip.spmm_xs(xs, _t(x), cat[:, 4:4 + F], bias=_t(b), relu=True, Wnext=_t(wn), Hnext=hn)
    want = np.maximum(A @ x.astype(np.float64) + b, 0)
    got = cat.cpu().numpy()
    assert rel_err(got[:, 4:4 + F], want) < 2e-6 and rel_err(hn.cpu().numpy(), want @ wn.astype(np.float64)) < 3e-6
    assert np.all(got[:, :4] == 0) and np.all(got[:, 4 + F:] == 0)
    s1, e = torch.empty((n, F), device=DEV), torch.empty((n, F), device=DEV)
    xd = _t(x)
    hip.spmm_xs(xs, xd, y, acc_in=xd, acc_out=s1)
    hip.spmm_xs(xs, y, None, acc_in=s1, acc_out=e, acc_div=3)
    x1 = A @ x.astype(np.float64)
    assert rel_err(e.cpu().numpy(), (x + x1 + A @ x1) / 3) < 3e-6


@pytest.mark.parametrize('D,units', [(48, [48, 1]), (64, [64, 1]), (16, [24, 8]), (128, [64, 64, 1])])
def test_chain_sum_inputs(hip, D, units):
    """x = relu(A[ids_a] + B[ids_b]) as the chain's input: a Dense layer over a concatenation, split per entity."""
    rng = np.random.default_rng(D)
    P = 3001
    A = rng.standard_normal((200, D)).astype(np.float32)
    B = rng.standard_normal((150, D)).astype(np.float32)
    ia, ib = rng.integers(0, 200, P).astype(np.int32), rng.integers(0, 150, P).astype(np.int32)
    dims = [D] + units
    ks = [rng.uniform(-0.4, 0.4, (dims[k], dims[k + 1])).astype(np.float32) for k in range(len(units))]
    bs = [rng.uniform(-0.2, 0.2, dims[k + 1]).astype(np.float32) for k in range(len(units))]
    acts = ['relu'] * (len(units) - 1) + ['sigmoid' if units[-1] == 1 else 'relu']
    assert hip.chain_supported(dims, D, D, sum_inputs=True)
    blob, _ = hip.chain_pack(ks, bs)
    out = torch.empty((P, units[-1]), device=DEV)
    hip.chain(_t(A), _t(blob), dims, acts, out, ids_a=_t(ia), B=_t(B), ids_b=_t(ib), sum_inputs=True, in_act='relu')
    x = np.maximum(A[ia].astype(np.float64) + B[ib].astype(np.float64), 0)
    for k, b, a in zip(ks, bs, acts):
        x = ol.dense(x, k.astype(np.float64), b.astype(np.float64), a)
    assert rel_err(out.cpu().numpy(), x) < 5e-6


@pytest.mark.parametrize('F', [4, 8, 16])
@pytest.mark.parametrize('uip', [False, True])
def test_spmm_xcd_sliced_value_free(hip, F, uip, monkeypatch):
    """The value-free XS image of a gcn-filtered matrix (A_hat = S (A + I) S, entries weigh 1, multiplicities
    repeated) against the valued CSR product and the valued XS image, plain and as a pre-scaled fused GCN chain."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced, gcn_filter_device
    g = helpers.tiny_graph(n_users=300, n_items=200, n_ratings=9000, seed=F, n_props=60 if uip else 0, n_links=400 if uip else 0)
    coo = g['adj'].tocoo()
    keep = coo.row < coo.col                                    # the un-symmetrised edges, duplicates (multi-relation links) kept
    rows, cols = torch.from_numpy(coo.row[keep].astype(np.int64)).to(DEV), torch.from_numpy(coo.col[keep].astype(np.int64)).to(DEV)
    n = coo.shape[0]
    a = gcn_filter_device(rows, cols, n)
    assert a.dinv is not None and (not uip or int(a.mult.max()) > 1)
    xs = XcdSliced.from_csr(a)
    assert xs.vals is None and xs.row_scale is not None
    monkeypatch.setenv('AMAR_XS_VALUES', '1')
    xs_valued = XcdSliced.from_csr(a)
    monkeypatch.delenv('AMAR_XS_VALUES')
    assert xs_valued.vals is not None and xs.colidx.numel() >= xs_valued.colidx.numel()
    A = a.to_scipy().astype(np.float64)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((n, F)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, F).astype(np.float32)
    w2 = rng.uniform(-0.5, 0.5, (F, F)).astype(np.float32)
    y, yv = torch.empty((n, F), device=DEV), torch.empty((n, F), device=DEV)
    hip.spmm_xs(xs, _t(x), y)                                   # un-scaled input: the wrapper pre-scales
    hip.spmm_xs(xs_valued, _t(x), yv)
    want = A @ x.astype(np.float64)
    assert rel_err(y.cpu().numpy(), want) < 2e-6 and rel_err(yv.cpu().numpy(), want) < 2e-6
    # two chained GCN layers kept in the pre-scaled form: H1' = S (relu(A H0 + b) W2), then A_hat H1
    h0 = torch.empty((n, F), device=DEV)
    hip.row_affine(_t(x), xs.row_scale, h0)
    y1, h1 = torch.empty((n, F), device=DEV), torch.empty((n, F), device=DEV)
    hip.spmm_xs(xs, h0, y1, bias=_t(b), relu=True, Wnext=_t(w2), Hnext=h1, prescaled=True, scale_next=True)
    y2 = torch.empty((n, F), device=DEV)
    hip.spmm_xs(xs, h1, y2, bias=_t(b), relu=True, prescaled=True)
    w1 = np.maximum(want + b, 0)
    want2 = np.maximum(A @ (w1 @ w2.astype(np.float64)) + b, 0)
    assert rel_err(y1.cpu().numpy(), w1) < 2e-6 and rel_err(y2.cpu().numpy(), want2) < 3e-6


@pytest.mark.parametrize('F', [4, 8, 16, 32])
@pytest.mark.parametrize('n,avg_deg,seed', [(300, 160, 1), (513, 260, 2), (200, 40, 3), (1030, 520, 4)])
def test_spmm_xcd_sliced_dense_tiles(hip, F, n, avg_deg, seed):
    """Tiles far longer than one super-step (64 lanes x 8 entries), tile lengths on and around its multiples, rows that are
    empty in some slices and very long in others: the full-step path, the clamped last step and the run bookkeeping."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced
    m = _rand_csr(n, avg_deg, seed=seed, dup=(seed == 3)).tocoo()
    a = _dev_csr(m)
    xs = XcdSliced.from_csr(a)
    tile = (xs.rowptr[64::64].long() - xs.rowptr[:-64:64].long())
    assert int(tile.max()) > 256, "this case is meant to exceed one super-step"
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, F)).astype(np.float32)
    y = torch.full((n, F), float('nan'), device=DEV)
    hip.spmm_xs(xs, _t(x), y)
    assert rel_err(y.cpu().numpy(), m.tocsr().astype(np.float64) @ x.astype(np.float64)) < 3e-6
    ycsr = torch.empty((n, F), device=DEV)
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, _t(x), ycsr)
    assert rel_err(y.cpu().numpy(), ycsr.cpu().numpy().astype(np.float64)) < 3e-6


@pytest.mark.parametrize('n,avg_deg,seed,dup', [(67, 9, 1, False), (1000, 9, 2, True), (300, 160, 3, False), (1030, 400, 4, True), (4097, 20, 5, False)])
@pytest.mark.parametrize('self_loop', [True, False])
@pytest.mark.parametrize('C', [8, 16, 32])
def test_gat_xcd_sliced(hip, n, avg_deg, seed, dup, self_loop, C):
    """amar_gat_xs_f32 (XCD-sliced image, online softmax) against the row kernel and the dense float64 softmax, incl. tiles
    longer than a super-step, duplicate edges, empty rows and widely spread attention scalars."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced
    m = _rand_csr(n, avg_deg, seed=seed, dup=dup).tocoo()
    keep = m.row != m.col
    m = sparse.coo_matrix((m.data[keep], (m.row[keep], m.col[keep])), shape=m.shape)
    a = _dev_csr(m, with_values=False)
    xs = XcdSliced.from_csr(a)
    rng = np.random.default_rng(seed)
    h = rng.standard_normal((n, C)).astype(np.float32)
    ss = (rng.standard_normal(n) * 4).astype(np.float32)            # spread: exp() ranges over many decades
    sn = (rng.standard_normal(n) * 4).astype(np.float32)
    b = rng.uniform(-0.3, 0.3, C).astype(np.float32)
    y_row, y_xs = torch.empty((n, C), device=DEV), torch.full((n, C), float('nan'), device=DEV)
    hip.gat_layer(a.rowptr, a.colidx, _t(h), _t(ss), _t(sn), _t(b), y_row, self_loop=self_loop)
    hip.gat_xs(xs, _t(h), _t(ss), _t(sn), _t(b), y_xs, self_loop=self_loop)
    got = y_xs.cpu().numpy()
    assert np.isfinite(got).all()
    # dense float64 reference of the layer
    rows, cols = m.row, m.col
    if self_loop:
        rows, cols = np.concatenate([rows, np.arange(n)]), np.concatenate([cols, np.arange(n)])
    e = ss.astype(np.float64)[rows] + sn.astype(np.float64)[cols]
    e = np.where(e > 0, e, 0.2 * e)
    mx = np.full(n, -np.inf); np.maximum.at(mx, rows, e)
    ex = np.exp(e - mx[rows])
    den = np.zeros(n); np.add.at(den, rows, ex)
    num = np.zeros((n, C)); np.add.at(num, rows, ex[:, None] * h.astype(np.float64)[cols])
    want = np.maximum(num / (den + 1e-9)[:, None] + b, 0)
    assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(want).max())
    assert np.abs(got - y_row.cpu().numpy()).max() < 2e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize('n,avg_deg,seed,dup', [(67, 9, 1, False), (1000, 9, 2, True), (300, 160, 3, False), (1030, 400, 4, True), (4097, 20, 5, False)])
@pytest.mark.parametrize('self_loop', [True, False])
@pytest.mark.parametrize('C', [8, 16, 32])
def test_gat_lds_tiled(hip, n, avg_deg, seed, dup, self_loop, C):
    """amar_gat_lt_f32 (LDS-tiled walk, additive weights against the per-row bound) against the row kernel and the dense
    float64 softmax: several tiles, rows cut into virtual rows, duplicate edges, (i, i) edges of the list itself, empty rows;
    attention scalars of moderate spread (the bound holds) and spread so widely that most rows take the exact fall-back."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import _unit_entries
    m = _rand_csr(n, avg_deg, seed=seed, dup=dup).tocoo()
    a = _dev_csr(m, with_values=False)                              # self edges stay in the list
    rows, cols, diag, off = _unit_entries(a, False)
    assert seed != 2 or float(diag.sum()) > 0
    rw = lds_tiled.GAT_ROWS_PER_WAVE[C]
    assert rw == hip.load().amar_gat_lt_rows_per_wave(C)
    lt = lds_tiled.LdsTiled.build(rows, cols, n, n, C, diag, torch.ones(n, device=DEV), None, off, n_cu=3, split=64, rw=rw, split_growth=1.25)
    rng = np.random.default_rng(seed)
    h = rng.standard_normal((n, C)).astype(np.float32)
    b = rng.uniform(-0.3, 0.3, C).astype(np.float32)
    if C == 8 and seed == 1:                                        # an image cut for the plain sum's taller tiles is refused, not walked
        plain = lds_tiled.LdsTiled.build(rows, cols, n, n, C, diag, torch.ones(n, device=DEV), None, off, n_cu=3)
        with pytest.raises(Exception):
            hip.gat_lt(plain, a, _t(h), _t(h[:, 0].copy()), _t(h[:, 1].copy()), _t(b), torch.empty((n, C), device=DEV))
    for spread in (4.0, 60.0):
        ss = (rng.standard_normal(n) * spread).astype(np.float32)
        sn = (rng.standard_normal(n) * spread).astype(np.float32)
        y_row, y_lt = torch.empty((n, C), device=DEV), torch.full((n, C), float('nan'), device=DEV)
        hip.gat_layer(a.rowptr, a.colidx, _t(h), _t(ss), _t(sn), _t(b), y_row, self_loop=self_loop)
        hip.gat_lt(lt, a, _t(h), _t(ss), _t(sn), _t(b), y_lt, self_loop=self_loop)
        got = y_lt.cpu().numpy()
        assert np.isfinite(got).all()
        r, c = m.row, m.col
        if self_loop:
            r, c = np.concatenate([r, np.arange(n)]), np.concatenate([c, np.arange(n)])
        e = ss.astype(np.float64)[r] + sn.astype(np.float64)[c]
        e = np.where(e > 0, e, 0.2 * e)
        mx = np.full(n, -np.inf); np.maximum.at(mx, r, e)
        ex = np.exp(e - mx[r])
        den = np.zeros(n); np.add.at(den, r, ex)
        num = np.zeros((n, C)); np.add.at(num, r, ex[:, None] * h.astype(np.float64)[c])
        want = np.maximum(num / (den + 1e-9)[:, None] + b, 0)
        tol = 2e-5 * max(1.0, np.abs(want).max())
        assert np.abs(got - want).max() < tol
        assert np.abs(got - y_row.cpu().numpy()).max() < tol
        again = torch.empty_like(y_lt)
        hip.gat_lt(lt, a, _t(h), _t(ss), _t(sn), _t(b), again, self_loop=self_loop)
        assert torch.equal(again, y_lt)                             # fixed summation order: reproducible bit for bit


@pytest.mark.parametrize('C', [8, 16, 32])
def test_gat_layer_on_lds_tiled(hip, C, monkeypatch):
    """GATConv routed onto the LDS-tiled image (forced: the graph is far below the size rule) on a user-item-property graph
    with duplicate links, against the oracle and the row-kernel route."""
    from deep_cbrs_amar_renaissance_amd.layers.gat_conv import GATConv
    g = helpers.tiny_graph(n_users=700, n_items=400, n_ratings=30000, seed=C, n_props=120, n_links=900)
    e = _dev_csr(g['adj_uip'] if 'adj_uip' in g else g['adj'], with_values=False, drop_diagonal=True)
    n = e.shape[0]
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, 8)).astype(np.float32)
    layer = GATConv(C, dropout_rate=0.0, activation='relu')
    layer.build([(n, 8), None])
    helpers.randomize_biases(layer, seed=2)
    monkeypatch.setenv('AMAR_SPMM_KIND', 'csr')
    y_row = layer([_t(x), e])
    monkeypatch.setenv('AMAR_SPMM_KIND', 'xs')
    monkeypatch.setenv('AMAR_SPMM_LT', '1')
    calls = []
    monkeypatch.setattr(hip, 'gat_lt', lambda *a, _f=hip.gat_lt, **k: (calls.append(1), _f(*a, **k))[1])
    y_lt = layer([_t(x), e])
    assert calls, "the layer did not take the LDS-tiled route"
    assert float((y_row - y_lt).abs().max()) < 2e-5
    coo = e.to_scipy().tocoo() if hasattr(e, 'to_scipy') else None
    if coo is not None:
        w = layer.kernel.detach().cpu().numpy().reshape(8, C).astype(np.float64)
        want, _ = ol.gat_conv(x.astype(np.float64), coo.col, coo.row, w, layer.attn_kernel_self.detach().cpu().numpy().reshape(C).astype(np.float64),
                              layer.attn_kernel_neighs.detach().cpu().numpy().reshape(C).astype(np.float64),
                              layer.bias.detach().cpu().numpy().astype(np.float64), self_loops=True)
        assert rel_err(y_lt.cpu().numpy(), want) < 1e-5


@pytest.mark.parametrize('n_slices', [16, 24, 5])
def test_xcd_sliced_multi_phase(hip, n_slices, monkeypatch):
    """8 k slices processed in k phases (tables beyond the aggregate L2), and a slice count that is no multiple of 8: the
    SpMM (valued and value-free), GraphSAGE's mean aggregate and the GAT form against the row kernels."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced, gcn_filter_device
    monkeypatch.setenv('AMAR_XS_SLICES', str(n_slices))
    g = helpers.tiny_graph(n_users=400, n_items=300, n_ratings=20000, seed=n_slices)
    coo = g['adj'].tocoo()
    keep = coo.row < coo.col
    n = coo.shape[0]
    a = gcn_filter_device(torch.from_numpy(coo.row[keep].astype(np.int64)).to(DEV), torch.from_numpy(coo.col[keep].astype(np.int64)).to(DEV), n)
    xs = a.xcd_sliced()
    assert xs.n_slices == n_slices and xs.row_scale is not None
    rng = np.random.default_rng(1)
    x = _t(rng.standard_normal((n, 8)).astype(np.float32))
    y, ycsr = torch.empty((n, 8), device=DEV), torch.empty((n, 8), device=DEV)
    hip.spmm_xs(xs, x, y)
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, x, ycsr)
    assert rel_err(y.cpu().numpy(), ycsr.cpu().numpy().astype(np.float64)) < 3e-6
    monkeypatch.setenv('AMAR_XS_VALUES', '1')
    hip.spmm_xs(XcdSliced.from_csr(a), x, y)
    assert rel_err(y.cpu().numpy(), ycsr.cpu().numpy().astype(np.float64)) < 3e-6
    e = _dev_csr(g['adj'], with_values=False, drop_diagonal=True)
    h = _t(rng.standard_normal((n, 8)).astype(np.float32))
    ss, sn, b = _t(rng.standard_normal(n).astype(np.float32)), _t(rng.standard_normal(n).astype(np.float32)), _t(rng.uniform(-0.2, 0.2, 8).astype(np.float32))
    yr, yx = torch.empty((n, 8), device=DEV), torch.empty((n, 8), device=DEV)
    hip.gat_layer(e.rowptr, e.colidx, h, ss, sn, b, yr, self_loop=True)
    hip.gat_xs(e.xcd_sliced(), h, ss, sn, b, yx, self_loop=True)
    assert e.xcd_sliced().n_slices == n_slices and float((yr - yx).abs().max()) < 2e-5
    agg_x, agg_r = torch.empty((n, 8), device=DEV), torch.empty((n, 8), device=DEV)
    hip.spmm_xs(e.xcd_sliced_mean(True), x, agg_x, prescaled=True)
    hip.spmm_csr(e.rowptr, e.colidx, None, x, agg_r)
    deg = (e.rowptr[1:] - e.rowptr[:-1]).float()
    want = (agg_r + x) / (deg + 1)[:, None]
    assert float((agg_x - want).abs().max()) < 1e-5


def test_spmm_xcd_sliced_is_reproducible(hip):
    """LDS float adds inside the XS partial kernel follow a fixed order: two launches give the same bits."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced
    m = (_rand_csr(3000, 40, seed=3, dup=True) + sparse.identity(3000, dtype=np.float32, format='coo')).tocoo()
    xs = XcdSliced.from_csr(_dev_csr(m))
    x = _t(np.random.default_rng(0).standard_normal((3000, 8)).astype(np.float32))
    y1, y2 = torch.empty((3000, 8), device=DEV), torch.empty((3000, 8), device=DEV)
    hip.spmm_xs(xs, x, y1)
    hip.spmm_xs(xs, x, y2)
    assert torch.equal(y1, y2)


def test_xs_randomised_sweep(hip, monkeypatch):
    """Many random shapes through the XCD-sliced SpMM (valued, value-free, mean-aggregate) and GAT forms against the row
    kernels: sizes around the 64-row block and 256-entry super-step boundaries, skewed degrees, duplicates, slice counts."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced, gcn_filter_device
    rng = np.random.default_rng(2024 + helpers.seed_offset())
    for case in range(40):
        n = int(rng.choice([1, 2, 63, 64, 65, 127, 129, 200, 511, 777, 1500]))
        avg = float(rng.choice([0.5, 3, 20, 120]))
        slices = int(rng.choice([8, 8, 8, 16, 3]))
        F = int(rng.choice([4, 8, 16]))
        monkeypatch.setenv('AMAR_XS_SLICES', str(slices))
        m = _rand_csr(n, min(avg, n), seed=1000 + case, dup=bool(case % 3 == 0)).tocoo()
        a = _dev_csr(m)
        x = _t(rng.standard_normal((n, F)).astype(np.float32))
        y, ycsr = torch.full((n, F), float('nan'), device=DEV), torch.empty((n, F), device=DEV)
        hip.spmm_xs(XcdSliced.from_csr(a), x, y)
        hip.spmm_csr(a.rowptr, a.colidx, a.vals, x, ycsr)
        ref = ycsr.cpu().numpy().astype(np.float64)
        assert np.abs(y.cpu().numpy() - ref).max() <= 3e-6 * max(1.0, np.abs(ref).max()), (case, n, avg, slices, F)
        # edge-list forms on the off-diagonal structure of the same matrix
        keep = m.row != m.col
        e = _dev_csr(sparse.coo_matrix((m.data[keep], (m.row[keep], m.col[keep])), shape=m.shape), with_values=False)
        agg, raw = torch.full((n, F), float('nan'), device=DEV), torch.empty((n, F), device=DEV)
        hip.spmm_xs(e.xcd_sliced_mean(True), x, agg, prescaled=True)
        hip.spmm_csr(e.rowptr, e.colidx, None, x, raw)
        want = ((raw + x) / ((e.rowptr[1:] - e.rowptr[:-1]).float() + 1)[:, None]).cpu().numpy()
        assert np.abs(agg.cpu().numpy() - want).max() <= 3e-6 * max(1.0, np.abs(want).max()), (case, 'mean', n, avg, slices, F)
        if F == 8:
            ss, sn = _t((rng.standard_normal(n) * 3).astype(np.float32)), _t((rng.standard_normal(n) * 3).astype(np.float32))
            b = _t(rng.uniform(-0.3, 0.3, 8).astype(np.float32))
            yr, yx = torch.empty((n, 8), device=DEV), torch.full((n, 8), float('nan'), device=DEV)
            loop = bool(case % 2)
            hip.gat_layer(e.rowptr, e.colidx, x, ss, sn, b, yr, self_loop=loop)
            hip.gat_xs(e.xcd_sliced(), x, ss, sn, b, yx, self_loop=loop)
            assert float((yr - yx).abs().max()) <= 2e-5 * max(1.0, float(yr.abs().max())), (case, 'gat', n, avg, slices, loop)
        # value-free image of a device-built gcn filter over the same symmetric structure
        if n >= 2 and keep.any():
            lo = m.row[keep] < m.col[keep]
            if lo.any():
                r = torch.from_numpy(m.row[keep][lo].astype(np.int64)).to(DEV)
                c = torch.from_numpy(m.col[keep][lo].astype(np.int64)).to(DEV)
                ah = gcn_filter_device(r, c, n)
                yv, yc = torch.full((n, F), float('nan'), device=DEV), torch.empty((n, F), device=DEV)
                hip.spmm_xs(ah.xcd_sliced(), x, yv)
                hip.spmm_csr(ah.rowptr, ah.colidx, ah.vals, x, yc)
                assert ah.xcd_sliced().row_scale is not None
                assert float((yv - yc).abs().max()) <= 3e-6 * max(1.0, float(yc.abs().max())), (case, 'value-free', n, avg, slices, F)


@pytest.mark.parametrize('F', [4, 8, 16, 32])
@pytest.mark.parametrize('uip,n_cu,window', [(False, 256, None), (True, 3, None), (False, 1, 64), (True, 5, 4096)])
def test_spmm_lds_tiled(hip, F, uip, n_cu, window):
    """amar_spmm_lt_f32 (LDS-tiled image: column-ordered windows, tile sums in LDS, one launch) against the float64 product
    of the gcn-filtered matrix and against the value-free XS image: plain, with the layer epilogue into a concat slice,
    as a pre-scaled two-layer GCN chain and as LightGCN's running mean; identical bits run to run."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced, gcn_filter_device, _unit_entries
    g = helpers.tiny_graph(n_users=900, n_items=500, n_ratings=40000, seed=F, n_props=160 if uip else 0, n_links=1500 if uip else 0)
    coo = g['adj'].tocoo()
    keep = coo.row < coo.col
    rows, cols = torch.from_numpy(coo.row[keep].astype(np.int64)).to(DEV), torch.from_numpy(coo.col[keep].astype(np.int64)).to(DEV)
    n = coo.shape[0]
    a = gcn_filter_device(rows, cols, n)
    r, c, diag, off = _unit_entries(a, True)
    lt = lds_tiled.LdsTiled.build(r, c, n, n, F, diag, a.dinv, a.dinv, off, window_entries=window, n_cu=n_cu)
    assert lt.n_tiles >= min(n_cu, 2) or n_cu == 1
    A = a.to_scipy().astype(np.float64)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((n, F)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, F).astype(np.float32)
    w2 = rng.uniform(-0.5, 0.5, (F, F)).astype(np.float32)
    y = torch.full((n, F), float('nan'), device=DEV)
    hip.spmm_lt(lt, _t(x), y)                                   # un-scaled input: the wrapper pre-scales
    want = A @ x.astype(np.float64)
    assert rel_err(y.cpu().numpy(), want) < 2e-6
    y_again = torch.empty_like(y)
    hip.spmm_lt(lt, _t(x), y_again)
    assert torch.equal(y, y_again)
    y_xs = torch.empty_like(y)
    hip.spmm_xs(XcdSliced.from_csr(a), _t(x), y_xs)
    assert rel_err(y.cpu().numpy(), y_xs.cpu().numpy().astype(np.float64)) < 2e-6
    # fused GCN chain in the pre-scaled form, first layer written into a column slice of a wider buffer
    h0 = torch.empty((n, F), device=DEV)
    hip.row_affine(_t(x), lt.row_scale, h0)
    cat = torch.zeros((n, 2 * F + 4), device=DEV)
    h1 = torch.full((n, F), float('nan'), device=DEV)
    hip.spmm_lt(lt, h0, cat[:, 4:4 + F], bias=_t(b), relu=True, Wnext=_t(w2), Hnext=h1, prescaled=True, scale_next=True)
    y2 = torch.empty((n, F), device=DEV)
    hip.spmm_lt(lt, h1, y2, bias=_t(b), relu=True, prescaled=True)
    w1 = np.maximum(want + b, 0)
    want2 = np.maximum(A @ (w1 @ w2.astype(np.float64)) + b, 0)
    got = cat.cpu().numpy()
    assert rel_err(got[:, 4:4 + F], w1) < 2e-6 and rel_err(y2.cpu().numpy(), want2) < 3e-6
    assert np.all(got[:, :4] == 0) and np.all(got[:, 4 + F:] == 0)
    # LightGCN: running sum, then the mean over 3 terms
    xd = _t(x)
    s1, e = torch.empty((n, F), device=DEV), torch.empty((n, F), device=DEV)
    hip.spmm_lt(lt, xd, y, acc_in=xd, acc_out=s1)
    hip.spmm_lt(lt, y, None, acc_in=s1, acc_out=e, acc_div=3)
    assert rel_err(e.cpu().numpy(), (x + want + A @ want) / 3) < 3e-6


@pytest.mark.parametrize('F', [16, 32])
@pytest.mark.parametrize('pairs', [True, False])
def test_spmm_lds_tiled_wide_row_forms(hip, F, pairs):
    """The wide-row forms of the LT walk (F = 16, 32): images with and without implicit pairs (AMAR_SPMM_LT_NOPAIRS: the step without
    its pair logic), each with the table dense and as a column slice of a wider buffer, plain and with the fused layer epilogue whose
    next kernel is staged in LDS: against float64."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device, _unit_entries
    g = helpers.tiny_graph(n_users=900, n_items=500, n_ratings=40000, seed=F + int(pairs), n_props=160, n_links=1500)
    coo = g['adj'].tocoo()
    keep = coo.row < coo.col
    rows, cols = torch.from_numpy(coo.row[keep].astype(np.int64)).to(DEV), torch.from_numpy(coo.col[keep].astype(np.int64)).to(DEV)
    n = coo.shape[0]
    a = gcn_filter_device(rows, cols, n)
    r, c, diag, off = _unit_entries(a, True)
    lt = lds_tiled.LdsTiled.build(r, c, n, n, F, diag, a.dinv, a.dinv, off, n_cu=5, pairs=pairs)
    assert lt.pairs == pairs
    A = a.to_scipy().astype(np.float64)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, F)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, F).astype(np.float32)
    w2 = rng.uniform(-0.5, 0.5, (F, F)).astype(np.float32)
    want = A @ x.astype(np.float64)
    h0 = torch.empty((n, F), device=DEV)
    hip.row_affine(_t(x), lt.row_scale, h0)
    wide = torch.zeros((n, F + 8), device=DEV)
    wide[:, 4:4 + F] = h0
    for table in (h0, wide[:, 4:4 + F]):                         # dense table: 32-bit shifted offsets; slice: multiply-add offsets
        y = torch.full((n, F), float('nan'), device=DEV)
        hip.spmm_lt(lt, table, y, prescaled=True)
        assert rel_err(y.cpu().numpy(), want) < 2e-6
        y1, h1 = torch.full((n, F), float('nan'), device=DEV), torch.full((n, F), float('nan'), device=DEV)
        hip.spmm_lt(lt, table, y1, bias=_t(b), relu=True, Wnext=_t(w2), Hnext=h1, prescaled=True, scale_next=True)
        w1 = np.maximum(want + b, 0)
        assert rel_err(y1.cpu().numpy(), w1) < 2e-6
        assert rel_err(h1.cpu().numpy(), a.dinv.cpu().numpy().astype(np.float64)[:, None] * (w1 @ w2.astype(np.float64))) < 3e-6
        again = torch.empty_like(y)
        hip.spmm_lt(lt, table, again, prescaled=True)
        assert torch.equal(y, again)


def test_spmm_lds_tiled_heavy_rows(hip):
    """Rows holding a large share of all entries (long same-row runs inside a window: ranks, flagged atomic adds) and
    rows without any entry."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    n = 3000
    rng = np.random.default_rng(11)
    r = np.concatenate([np.full(20000, 5), np.full(9000, 2100), rng.integers(0, 1500, 60000)])
    c = np.concatenate([rng.integers(1500, 2900, 29000), rng.integers(1500, 2900, 60000)])
    rows = torch.from_numpy(np.concatenate([r, c]).astype(np.int64)).to(DEV)
    cols = torch.from_numpy(np.concatenate([c, r]).astype(np.int64)).to(DEV)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32)).to(DEV)
    diag = torch.ones(n, device=DEV)
    for F, n_cu, split in ((8, 4, 256), (16, 2, 4096), (8, 1, 100000), (32, 3, 64), (4, 2, 100000)):
        lt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, scale, scale, 0, n_cu=n_cu, split=split)
        assert split < 20000 or (lt.n_flagged > 0 and lt.n_pairs > 0)       # un-split heavy rows: pairs and flagged repeats
        xs_tab = rng.standard_normal((n, F)).astype(np.float32)
        y = torch.full((n, F), float('nan'), device=DEV)
        hip.spmm_lt(lt, _t(xs_tab), y, prescaled=True)
        A = sparse.coo_matrix((np.ones(rows.numel()), (rows.cpu().numpy(), cols.cpu().numpy())), shape=(n, n)).tocsr()
        want = scale.cpu().numpy()[:, None].astype(np.float64) * (xs_tab.astype(np.float64) + A @ xs_tab.astype(np.float64))
        assert rel_err(y.cpu().numpy(), want) < 1e-5             # rows of 20 000 fp32 terms


@pytest.mark.parametrize('D,units,last_act', [(48, [48, 1], 'sigmoid'), (64, [64, 1], 'sigmoid'), (16, [24, 1], 'sigmoid'), (16, [24, 8], 'relu')])
def test_chain_out_index(hip, D, units, last_act):
    """amar_chain_indexed_f32: row p of the chain lands in out row out_index[p] — pipelined pair-stage kernel (square ReLU
    stacks) and generic kernel (other shapes, vector outputs) — bit-identical to the direct call on the un-permuted list."""
    rng = np.random.default_rng(D + len(units))
    P = 5003
    A = rng.standard_normal((200, D)).astype(np.float32)
    B = rng.standard_normal((150, D)).astype(np.float32)
    ia, ib = rng.integers(0, 200, P).astype(np.int32), rng.integers(0, 150, P).astype(np.int32)
    dims = [D] + units
    ks = [rng.uniform(-0.4, 0.4, (dims[k], dims[k + 1])).astype(np.float32) for k in range(len(units))]
    bs = [rng.uniform(-0.2, 0.2, dims[k + 1]).astype(np.float32) for k in range(len(units))]
    acts = ['relu'] * (len(units) - 1) + [last_act]
    blob, _ = hip.chain_pack(ks, bs)
    ref = torch.empty((P, units[-1]), device=DEV)
    hip.chain(_t(A), _t(blob), dims, acts, ref, ids_a=_t(ia), B=_t(B), ids_b=_t(ib), sum_inputs=True, in_act='relu')
    perm = rng.permutation(P)
    out = torch.full((P, units[-1]), float('nan'), device=DEV)
    hip.chain(_t(A), _t(blob), dims, acts, out, ids_a=_t(ia[perm]), B=_t(B), ids_b=_t(ib[perm]), sum_inputs=True, in_act='relu',
              out_index=_t(perm.astype(np.int32)))
    assert torch.equal(out, ref)


@pytest.mark.parametrize('Da,units,last_act', [(24, [24, 24, 48], None), (48, [48, 48, 64], None), (24, [24, 24], 'relu'), (48, [48, 48], 'relu'),
                                               (20, [24, 24, 40], None), (36, [44, 48, 64], None), (8, [24, 24, 48], None), (16, [48, 48, 64], None)])
@pytest.mark.parametrize('P', [1, 130, 300_001])
def test_entity_towers_compile_time_shapes(hip, Da, units, last_act, P):
    """The per-entity towers (one table, ReLU layers, an optionally linear last layer — the folded half of the classifier's first
    layer, models/basic.py:_split_plan) run a kernel whose tile counts are compile-time constants and whose rows are requested one
    iteration ahead (chain_rows_kernel).  Against the float64 oracle, with and without ids, and bit for bit against the generic
    kernel (a call with an identity out_index takes the generic one).  300 001 rows: every wave loops, the last iteration is ragged;
    widths 20 / 36 / 44 / 40: partly filled last tiles on the input and the output side."""
    rng = np.random.default_rng(Da + len(units) + P)
    n = max(P, 200)
    A = rng.standard_normal((n, Da)).astype(np.float32)
    dims = [Da] + units
    ks = [rng.uniform(-0.4, 0.4, (dims[k], dims[k + 1])).astype(np.float32) for k in range(len(units))]
    bs = [rng.uniform(-0.2, 0.2, dims[k + 1]).astype(np.float32) for k in range(len(units))]
    acts = ['relu'] * (len(units) - 1) + [last_act]
    blob, _ = hip.chain_pack(ks, bs)
    Ad, bd = _t(A), _t(blob)
    x = A[:P].astype(np.float64)
    for k, b, a in zip(ks, bs, acts):
        x = ol.dense(x, k.astype(np.float64), b.astype(np.float64), a)
    out = torch.full((P, units[-1]), float('nan'), device=DEV)
    hip.chain(Ad, bd, dims, acts, out)                                          # rows in order
    assert rel_err(out.cpu().numpy(), x) < 5e-6
    generic = torch.full((P, units[-1]), float('nan'), device=DEV)
    hip.chain(Ad, bd, dims, acts, generic, out_index=torch.arange(P, device=DEV, dtype=torch.int32))
    assert torch.equal(out, generic)
    ids = rng.integers(0, n, P).astype(np.int32)                                # gathered rows
    out_ids = torch.full((P, units[-1]), float('nan'), device=DEV)
    hip.chain(Ad, bd, dims, acts, out_ids, ids_a=_t(ids))
    gen_ids = torch.full((P, units[-1]), float('nan'), device=DEV)
    hip.chain(Ad, bd, dims, acts, gen_ids, ids_a=_t(ids), out_index=torch.arange(P, device=DEV, dtype=torch.int32))
    assert torch.equal(out_ids, gen_ids)
    if P <= 130:
        x = A[ids].astype(np.float64)
        for k, b, a in zip(ks, bs, acts):
            x = ol.dense(x, k.astype(np.float64), b.astype(np.float64), a)
        assert rel_err(out_ids.cpu().numpy(), x) < 5e-6


@pytest.mark.parametrize('widths,units,last_act', [([8, 8, 8], [24, 24, 48], None), ([16, 16, 16], [48, 48, 64], None), ([8, 8, 8], [24, 24], 'relu'),
                                                   ([16, 16, 16], [48, 48], 'relu'), ([8, 12], [24, 24, 40], None), ([8], [24, 24, 48], None),
                                                   ([32, 32, 32], [96, 48, 64], None)])
@pytest.mark.parametrize('P', [1, 130, 100_003])
def test_towers_read_per_layer_tables_in_place(hip, widths, units, last_act, P):
    """amar_chain_segments_f32 (capi.ConcatTable): the towers over [X_0 || X_1 || ...] read from the per-layer tables where
    they lie — tables with their own leading dimensions, one of them a row slice of a larger buffer — must equal, BIT FOR BIT,
    the same stack on the assembled table; with and without ids.  Shapes without a segment-reading kernel (96-wide) fall back
    to assembling inside capi.chain, same result."""
    rng = np.random.default_rng(sum(widths) + len(units) + P)
    n = max(P, 300) + 8
    tabs = [rng.standard_normal((n + 7, w + (4 * j))).astype(np.float32) for j, w in enumerate(widths)]     # padded leading dimensions
    views = [_t(t)[5:5 + n, :w] if j == 1 else _t(t)[:n, :w] for j, (t, w) in enumerate(zip(tabs, widths))]
    cat = torch.cat(views, dim=1).contiguous()
    dims = [sum(widths)] + units
    ks = [rng.uniform(-0.4, 0.4, (dims[k], dims[k + 1])).astype(np.float32) for k in range(len(units))]
    bs = [rng.uniform(-0.2, 0.2, dims[k + 1]).astype(np.float32) for k in range(len(units))]
    acts = ['relu'] * (len(units) - 1) + [last_act]
    bd = _t(hip.chain_pack(ks, bs)[0])
    table = hip.ConcatTable(views)
    assert table.shape == (n, sum(widths)) and torch.equal(table.materialize(), cat)
    want, got = torch.full((P, units[-1]), float('nan'), device=DEV), torch.full((P, units[-1]), float('nan'), device=DEV)
    hip.chain(cat, bd, dims, acts, want)
    hip.chain(table, bd, dims, acts, got)
    assert torch.equal(got, want)
    x = cat[:P].cpu().numpy().astype(np.float64)
    for k, b, a in zip(ks, bs, acts):
        x = ol.dense(x, k.astype(np.float64), b.astype(np.float64), a)
    assert rel_err(got.cpu().numpy(), x) < 5e-6
    ids = _t(rng.integers(40, 40 + n, P).astype(np.int32))                      # gathered rows with a base
    want_i, got_i = torch.full_like(want, float('nan')), torch.full_like(want, float('nan'))
    hip.chain(cat, bd, dims, acts, want_i, ids_a=ids, base_a=40)
    hip.chain(table, bd, dims, acts, got_i, ids_a=ids, base_a=40)
    assert torch.equal(got_i, want_i)
    sub = table[3:3 + P]                                                        # a row range of the concatenation
    got_s = torch.full_like(want, float('nan'))
    hip.chain(sub, bd, dims, acts, got_s)
    hip.chain(cat[3:3 + P], bd, dims, acts, want)
    assert torch.equal(got_s, want)


@pytest.mark.parametrize('F,C', [(8, 8), (16, 16), (8, 16), (24, 8)])
def test_rowwise_xw_row_gather(hip, F, C):
    """amar_rowwise_xw_gather_f32: H[p] = s[p] . (X[ids[p]] . W), a negative id leaves a zero row — bit-identical to the plain
    product of the gathered rows followed by the row scale."""
    rng = np.random.default_rng(F * 7 + C)
    n, m = 900, 2500
    x, w = _t(rng.standard_normal((n, F)).astype(np.float32)), _t(rng.standard_normal((F, C)).astype(np.float32))
    ids_np = rng.integers(0, n, m).astype(np.int32)
    ids_np[rng.integers(0, m, 60)] = -1
    ids = _t(ids_np)
    s = _t(rng.uniform(0.1, 1.0, m).astype(np.float32))
    for scale in (None, s):
        h = torch.full((m, C), float('nan'), device=DEV)
        hip.rowwise_xw(x, w, h, row_ids=ids, row_scale=scale)
        gathered = x[ids.long().clamp(min=0)].contiguous()
        ref = torch.empty((m, C), device=DEV)
        hip.rowwise_xw(gathered, w, ref, row_scale=scale)
        ref[ids < 0] = 0
        assert torch.equal(h, ref)
    with pytest.raises(ValueError):
        hip.rowwise_xw(x, w, torch.empty((m, C), device=DEV), row_ids=ids, copy_to=torch.empty((m, F), device=DEV))


@pytest.mark.parametrize('n,L,width', [(1, 1, 4), (77, 3, 8), (5000, 4, 16), (1234, 8, 12)])
def test_weighted_sum_reduction_kernels(hip, n, L, width):
    """amar_reduce_layers_wsum_f32 / _bwd_f32 (WeightedSum, reduction.py:36-55): out = sum_l w_l^2 X_l over the column blocks of a
    strided buffer; reverse: d_cat = w_l^2 d_out per block, dw_l = 2 w_l sum(d_out . X_l), reproducible bit for bit."""
    rng = np.random.default_rng(n + L)
    buf = rng.standard_normal((n, L * width + 4)).astype(np.float32)
    w = rng.uniform(-1.5, 1.5, L).astype(np.float32)
    cat = _t(buf)[:, :L * width]
    out = torch.full((n, width), float('nan'), device=DEV)
    hip.reduce_layers_wsum(cat, L, width, _t(w), out)
    blocks = [buf[:, l * width:(l + 1) * width].astype(np.float64) for l in range(L)]
    want = ol.reduce_layers(blocks, 'w-sum', w.astype(np.float64))
    assert rel_err(out.cpu().numpy(), want) < 1e-6
    d_out = rng.standard_normal((n, width)).astype(np.float32)
    d_cat, dw = torch.full((n, L * width), float('nan'), device=DEV), torch.full((L,), float('nan'), device=DEV)
    hip.reduce_layers_wsum_bwd(cat, L, width, _t(w), _t(d_out), d_cat, dw)
    for l in range(L):
        assert rel_err(d_cat[:, l * width:(l + 1) * width].cpu().numpy(), (w[l].astype(np.float64) ** 2) * d_out) < 1e-6
    want_dw = np.array([2 * w[l] * float((d_out.astype(np.float64) * blocks[l]).sum()) for l in range(L)])
    assert np.abs(dw.cpu().numpy() - want_dw).max() <= 1e-5 * max(1.0, np.abs(want_dw).max())
    dw2 = torch.empty_like(dw)
    hip.reduce_layers_wsum_bwd(cat, L, width, _t(w), _t(d_out), torch.empty_like(d_cat), dw2)
    assert torch.equal(dw, dw2)
    with pytest.raises(Exception):
        hip.reduce_layers_wsum(torch.zeros((4, 9 * 4), device=DEV), 9, 4, torch.ones(9, device=DEV), torch.empty((4, 4), device=DEV))   # > 8 terms


@pytest.mark.parametrize('two_step', [False, True])
def test_pair_plan_scores_equal_direct(hip, two_step, monkeypatch):
    """models.basic.PairPlan: the XCD-affine item-range order of a pair list + out_index gives the same bits, in the caller's
    order, as scoring the list directly; positions p with (p >> 7) % 8 == x only see items of the x-th item range.  two_step: the
    scores return through the window streams and amar_scatter_f32 (long lists; forced here) instead of the direct scattered store."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.models import basic
    monkeypatch.setenv('AMAR_PAIR_WINDOW_MIN', '0' if two_step else str(1 << 30))
    engine.set_seed(3)
    nu, ni, P = 5000, 3000, 200_001
    rs = basic.BasicRS([24, 24], [48, 48])
    rs.build_head(24, 24)
    helpers.randomize_biases(rs, seed=4)
    emb = torch.randn((nu + ni, 24), device=DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    u = torch.randint(0, nu, (P,), device=DEV, generator=g, dtype=torch.int32)
    i = (torch.randint(0, ni, (P,), device=DEV, generator=g, dtype=torch.int32) + nu).to(torch.int32)
    tw = rs.towers(emb[:nu], emb[nu:])
    ref = rs.score_towers(tw, u, i, 0, nu)
    plan = basic.PairPlan(u, i)
    assert sorted(plan.out_index.cpu().tolist()) == list(range(P))
    assert torch.equal(plan.u_ids, u[plan.out_index.long()]) and torch.equal(plan.i_ids, i[plan.out_index.long()])
    xcd = (torch.arange(P, device=DEV) // 128) % 8
    hi = torch.stack([plan.i_ids[xcd == x].max() for x in range(8)])
    lo = torch.stack([plan.i_ids[xcd == x].min() for x in range(8)])
    assert bool((lo[1:] >= hi[:-1]).all())                              # item ranges of consecutive XCDs do not overlap (they may touch)
    assert (plan.mid_index is not None) == two_step
    if two_step:                                                          # mid_index o final_index == out_index, windows cover the list
        assert torch.equal(plan.final_index[plan.mid_index.long()], plan.out_index)
        off = plan.window_off.cpu().tolist()
        assert off[0] == 0 and off[-1] == P and len(off) == plan.n_windows + 1 and all(b >= a for a, b in zip(off, off[1:]))
        win = plan.final_index.long() // plan.window
        assert bool((win[1:] >= win[:-1]).all())                          # the scratch vector is ordered by window of the final position
    got = rs.score_towers(tw, u, i, 0, nu, pair_plan=plan)
    assert torch.equal(got, ref)
    with pytest.raises(ValueError):
        rs.score_towers(tw, u.clone(), i, 0, nu, pair_plan=plan)


@pytest.mark.parametrize('width', [48, 64])
def test_pair_stage_split_products_against_f32_and_f64(hip, width):
    """The pair-stage kernel takes its products on the bf16 matrix instruction with both operands split three ways (csrc/amar_chain.hip,
    SPLIT): as accurate as the f32 instruction, not bit-identical to it.  Against a float64 evaluation of the same head both forms must
    be equally close (1e-6 is 100 times tighter than the north star's 1e-4), and the two forms must agree within a few ulps of a score:
    the f32 form here is the generic kernel on pre-gathered, pre-summed rows (exact f32 MFMA chain)."""
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.models import basic
    engine.set_seed(11)
    nu, ni, P = 3000, 2000, 150_003
    rs = basic.BasicRS([width // 2, width // 2], [width, width])
    rs.build_head(24, 24)
    helpers.randomize_biases(rs, seed=12)
    g = torch.Generator(device=DEV)
    g.manual_seed(13)
    emb = torch.randn((nu + ni, 24), device=DEV, generator=g) * 3.0                     # activations up to a few tens
    u = torch.randint(0, nu, (P,), device=DEV, generator=g, dtype=torch.int32)
    i = (torch.randint(0, ni, (P,), device=DEV, generator=g, dtype=torch.int32) + nu).to(torch.int32)
    tu, ti, split = rs.towers(emb[:nu], emb[nu:])
    assert split
    got = rs.score_towers((tu, ti, True), u, i, 0, nu).view(-1)
    blob, dims, acts = rs._split_cache[1]['rest']
    x = torch.relu(tu[u.long()] + ti[(i - nu).long()]).contiguous()                      # the summed input, exactly as the kernel forms it
    f32 = torch.empty((P, 1), dtype=torch.float32, device=DEV)
    capi.chain(x, blob, dims, acts, f32)                                                 # generic kernel, rows themselves: f32 MFMA
    f32 = f32.view(-1)
    ref = x.double()
    layers = list(rs.clf.layers)[1:]
    for l in layers[:-1]:
        ref = torch.relu(ref @ l.kernel.detach().double() + l.bias.detach().double())
    ref = torch.sigmoid(ref @ layers[-1].kernel.detach().double() + layers[-1].bias.detach().double()).view(-1)
    e_split, e_f32 = (got.double() - ref).abs(), (f32.double() - ref).abs()
    assert float(e_split.max()) < 1e-6 and float(e_f32.max()) < 1e-6
    assert float(e_split.mean()) < 1.5 * float(e_f32.mean()) + 1e-9                      # no less accurate than the f32 instruction
    assert float((got - f32).abs().max()) < 5e-7


@pytest.mark.parametrize('M,K,N,act,use_ids', [(1000, 768, 256, 'relu', True), (129, 64, 128, None, False), (4097, 256, 384, 'sigmoid', False),
                                                (1, 32, 128, 'relu', True)])
def test_dense_split_products(hip, M, K, N, act, use_ids):
    """amar_dense_split_f32 (bf16 matrix instruction, both operands split three ways) against a float64 product and against
    amar_dense_f32 (exact f32 MFMA): no less accurate, rows gathered by id, output into a column slice of a wider buffer."""
    from deep_cbrs_amar_renaissance_amd import capi
    g = torch.Generator(device=DEV)
    g.manual_seed(31 + M)
    rows = M + 37 if use_ids else M
    X = torch.randn((rows, K), device=DEV, generator=g) * 0.7
    W = (torch.rand((K, N), device=DEV, generator=g) - 0.5) * (2.0 / K ** 0.5)
    b = torch.randn(N, device=DEV, generator=g) * 0.1
    ids = torch.randint(0, rows, (M,), device=DEV, generator=g, dtype=torch.int32) if use_ids else None
    assert capi.dense_split_supported(K, N)
    Wq = torch.from_numpy(capi.dense_split_pack(W.cpu().numpy())).to(DEV)
    wide = torch.full((M, N + 8), 7.0, device=DEV)
    capi.dense_split(X, Wq, K, N, b, wide[:, 4:4 + N], act=act, ids=ids)
    assert float((wide[:, :4] - 7.0).abs().max()) == 0.0 and float((wide[:, 4 + N:] - 7.0).abs().max()) == 0.0
    got = wide[:, 4:4 + N]
    f32 = torch.empty((M, N), device=DEV)
    capi.dense(X, W, b, f32, act=act, ids=ids)
    xs = X[ids.long()] if use_ids else X
    ref = xs.double() @ W.double() + b.double()
    ref = torch.relu(ref) if act == 'relu' else (torch.sigmoid(ref) if act == 'sigmoid' else ref)
    e_split, e_f32 = (got.double() - ref).abs(), (f32.double() - ref).abs()
    scale = float(ref.abs().max()) + 1.0
    assert float(e_split.max()) < 4e-6 * scale
    assert float(e_split.mean()) < 1.5 * float(e_f32.mean()) + 1e-9        # as accurate as the f32 instruction
    assert not capi.dense_split_supported(K + 4, N) and not capi.dense_split_supported(K, N + 64)


def test_round3_entry_points_reject_bad_arguments(hip):
    """Argument checks of the entry points added in round 3 (AMAR_EINVAL -> ValueError, AMAR_EUNSUPPORTED -> AmarError): nothing is
    launched on a refused call."""
    import ctypes
    from deep_cbrs_amar_renaissance_amd import capi
    lib = capi.load()
    x = torch.zeros((8, 64), device=DEV)
    y = torch.zeros((8, 128), device=DEV)
    wq = torch.zeros(64 * 128 * 6, dtype=torch.uint8, device=DEV)
    idx = torch.arange(8, dtype=torch.int32, device=DEV)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.amar_dense_split_bytes(64, 128) == 64 * 128 * 6
    assert lib.amar_dense_split_bytes(48, 128) == -2 and lib.amar_dense_split_bytes(64, 64) == -2          # K % 32, N % 128
    assert lib.amar_dense_split_f32(p(x), 64, None, p(wq), None, p(y), 128, 8, 48, 128, 0, None) == -2     # unsupported shape
    assert lib.amar_dense_split_f32(p(x), 64, None, None, None, p(y), 128, 8, 64, 128, 0, None) == -1      # no weights
    assert lib.amar_dense_split_f32(p(x), 32, None, p(wq), None, p(y), 128, 8, 64, 128, 0, None) == -1     # ldx < K
    assert lib.amar_dense_split_f32(p(x), 64, None, p(wq), None, p(y), 128, 8, 64, 128, 7, None) == -1     # unknown activation
    assert lib.amar_dense_split_f32(p(x), 64, None, p(wq), None, p(y), 128, 0, 64, 128, 0, None) == 0      # no rows: nothing to do
    assert lib.amar_dense_split_pack_f32(None, 64, 128, None) == -1
    src = torch.zeros(8, device=DEV)
    dst = torch.zeros(8, device=DEV)
    assert lib.amar_scatter_f32(p(src), p(idx), p(dst), 1, 8, None, 0, None) == -1                          # no windows
    assert lib.amar_scatter_f32(p(src), None, p(dst), 1, 8, None, 1, None) == -1
    assert lib.amar_scatter_f32(p(src), p(idx), p(dst), 0, 8, None, 1, None) == -1
    assert lib.amar_scatter_f32(p(src), p(idx), p(dst), 1, 0, None, 1, None) == 0
    with pytest.raises(ValueError):
        capi.dense_split(x, wq[:-6], 64, 128, None, y)


def test_scatter_by_windows(hip):
    """amar_scatter_f32: dst[index[t]] = src[t], visited window by window; any window table gives the same result as none."""
    from deep_cbrs_amar_renaissance_amd import capi
    g = torch.Generator(device=DEV)
    g.manual_seed(21)
    n = 300_007
    src = torch.randn(n, device=DEV, generator=g)
    index = torch.randperm(n, device=DEV, generator=g).to(torch.int32)
    want = torch.empty(n, device=DEV)
    want[index.long()] = src
    for n_win, off in ((1, None), (13, None), (5, torch.tensor([0, 10, 10, 200_000, 300_000, n], dtype=torch.int32, device=DEV))):
        dst = torch.full((n, 1), float('nan'), device=DEV)
        capi.scatter(src, index, dst, off, n_win)
        assert torch.equal(dst.view(-1), want)
    wide = torch.zeros((n, 3), device=DEV)                                               # a column of a wider destination
    capi.scatter(src, index, wide[:, 1:2], None, 7)
    assert torch.equal(wide[:, 1], want) and float(wide[:, 0].abs().max()) == 0.0 and float(wide[:, 2].abs().max()) == 0.0
    capi.scatter(src[:0], index[:0], wide[:, 1:2], None, 1)                              # nothing to do
    with pytest.raises(ValueError):
        capi.scatter(src, index, wide[:, 1:2], torch.zeros(3, dtype=torch.int32, device=DEV), 7)


@pytest.mark.parametrize('F,C', [(8, 8), (16, 16), (32, 32), (16, 8)])
@pytest.mark.parametrize('self_loops', [True, False])
def test_sage_mean_on_lds_tiled(hip, F, C, self_loops, monkeypatch):
    """GraphSAGE's mean aggregate on the LDS-tiled image (edge-list CSR with duplicate edges, optional self loop,
    rows without edges) against the row kernel's aggregate, and the whole layer through it against the fused row kernel:
    C == F takes the tail fused into the SpMM launch (AMAR_SPMM_SAGE_TAIL), C != F the separate tail kernel; the layer's
    input once as a dense table and once as a column slice of a wider buffer."""
    from deep_cbrs_amar_renaissance_amd.layers.graphsage_conv import GraphSageConv
    from deep_cbrs_amar_renaissance_amd.utilities.lds_tiled import LdsTiled
    g = helpers.tiny_graph(n_users=700, n_items=400, n_ratings=30000, seed=F, n_props=120, n_links=900)
    e = _dev_csr(g['adj'], with_values=False, drop_diagonal=True)
    n = e.shape[0]
    rng = np.random.default_rng(3)
    x = _t(rng.standard_normal((n, F)).astype(np.float32))
    monkeypatch.setenv('AMAR_SPMM_LT', '1')
    img = e.tiled_mean_image(F, self_loops)
    assert isinstance(img, LdsTiled)
    agg = torch.full((n, F), float('nan'), device=DEV)
    hip.spmm_xs(img, x, agg, prescaled=True)
    ref = torch.empty((n, F), device=DEV)
    hip.spmm_csr(e.rowptr, e.colidx, None, x, ref)
    deg = (e.rowptr[1:] - e.rowptr[:-1]).float()
    want = (ref + x) / (deg + 1)[:, None] if self_loops else torch.where(deg[:, None] > 0, ref / deg.clamp(min=1)[:, None], torch.zeros_like(ref))
    assert float((agg - want).abs().max()) < 1e-5
    layer = GraphSageConv(C, activation='relu', self_loops=self_loops)
    layer.build([(n, F), None])
    helpers.randomize_biases(layer, seed=2)
    monkeypatch.setenv('AMAR_SPMM_KIND', 'csr')
    y_row = layer([x, e])
    monkeypatch.setenv('AMAR_SPMM_KIND', 'xs')
    y_lt = layer([x, e])
    assert float((y_row - y_lt).abs().max()) < 2e-5
    wide = torch.zeros((n, F + 8), device=DEV)
    wide[:, 4:4 + F] = x
    y_slice = layer([wide[:, 4:4 + F], e], out=torch.full((n, C + 4), float('nan'), device=DEV)[:, 4:])
    assert torch.equal(y_slice, y_lt)
    dense = torch.full((n, C), float('nan'), device=DEV)           # the second, dense copy for the next layer's gathers
    y_two = layer([x, e], out=torch.full((n, C + 4), float('nan'), device=DEV)[:, 4:], dense_out=dense)
    assert torch.equal(dense, y_lt) and torch.equal(y_two, y_lt)


def test_host_gcn_filter_route_keeps_the_factors(hip, monkeypatch):
    """utilities.math.gcn_filter (host scipy, the route `experiment.py` takes from files) keeps A_hat's factors on its result,
    DeviceCSR.from_scipy hands them on: the same value-free LT / XS images as from gcn_filter_device, bit-equal values."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR, gcn_filter
    from deep_cbrs_amar_renaissance_amd.utilities.lds_tiled import LdsTiled
    g = helpers.tiny_graph(n_users=800, n_items=500, n_ratings=30000, seed=21, n_props=100, n_links=900)
    a_hat = gcn_filter(g['adj'])
    a = DeviceCSR.from_scipy(a_hat)
    assert a.dinv is not None and a.mult is not None and a.gcn_filtered and int(a.mult.max()) > 1
    deg = (a.rowptr[1:] - a.rowptr[:-1]).long()
    r = torch.repeat_interleave(torch.arange(a.shape[0], device=DEV), deg)
    assert torch.equal(a.vals, (a.dinv[r] * a.mult.float()) * a.dinv[a.colidx.long()])
    monkeypatch.setenv('AMAR_SPMM_LT', '1')
    img = a.tiled_image(8)
    assert isinstance(img, LdsTiled)
    x = np.random.default_rng(0).standard_normal((a.shape[0], 8)).astype(np.float32)
    y = torch.empty((a.shape[0], 8), device=DEV)
    hip.spmm_xs(img, _t(x), y)
    assert rel_err(y.cpu().numpy(), a_hat.astype(np.float64) @ x.astype(np.float64)) < 2e-6
    monkeypatch.setenv('AMAR_SPMM_LT', '0')
    xs = a.xcd_sliced()
    assert xs.vals is None and xs.row_scale is not None              # value-free XS image too
    hip.spmm_xs(xs, _t(x), y)
    assert rel_err(y.cpu().numpy(), a_hat.astype(np.float64) @ x.astype(np.float64)) < 2e-6


@pytest.mark.parametrize('M,K,N,act,gather', [(1024, 768, 256, 'relu', False), (85, 256, 64, None, True), (2048, 64, 64, 'sigmoid', True), (1, 768, 256, 'relu', False)])
def test_dense_batch_sized_wide_layers(hip, M, K, N, act, gather):
    """amar_dense_f32 on batch-sized operands of wide layers (round 4: 64 x 64 tiles, two k-tiles in flight — dense_mfma_small_kernel) against
    float64, and bit for bit against the same rows computed inside a larger product (the big-tile kernels): same instruction, same k order."""
    rng = np.random.default_rng(M + K + N)
    big = 6000
    x = rng.standard_normal((big, K)).astype(np.float32)
    w = (rng.standard_normal((K, N)) * 0.2).astype(np.float32)
    b = (rng.standard_normal(N) * 0.1).astype(np.float32)
    ids = rng.integers(0, big, big).astype(np.int32) if gather else None
    x_d, w_d, b_d = _t(x), _t(w), _t(b)
    ids_d = _t(ids) if gather else None
    y_small = torch.empty((M, N), device=DEV)
    hip.dense(x_d if gather else x_d[:M], w_d, b_d, y_small, act=act, ids=ids_d[:M] if gather else None)
    src = x[ids[:M]] if gather else x[:M]
    z = src.astype(np.float64) @ w.astype(np.float64) + b
    want = np.maximum(z, 0) if act == 'relu' else 1 / (1 + np.exp(-z)) if act == 'sigmoid' else z
    assert helpers.rel_err(y_small.cpu().numpy(), want) < 3e-6
    y_big = torch.empty((big, N), device=DEV)
    hip.dense(x_d, w_d, b_d, y_big, act=act, ids=ids_d)
    assert torch.equal(y_big[:M], y_small)
