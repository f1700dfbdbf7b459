"""GPU parity of the training step against oracle/train.py (pytest -m gpu)."""
import numpy as np
import pytest
import torch

from scipy import sparse

from oracle import train as otrain
from tests import helpers

pytestmark = pytest.mark.gpu
DEV = 'cuda'
CFG = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_training_kernels(hip):
    rng = np.random.default_rng(0)
    M, K, N = 1300, 40, 24
    x = rng.standard_normal((M, K)).astype(np.float32)
    yv = np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    dz = torch.empty((M, N), device=DEV)
    hip.act_bwd(_t(dy), _t(yv), dz, 'relu')
    assert np.array_equal(dz.cpu().numpy(), dy * (yv > 0))
    s = 1 / (1 + np.exp(-yv))
    hip.act_bwd(_t(dy), _t(s.astype(np.float32)), dz, 'sigmoid')
    assert helpers.rel_err(dz.cpu().numpy(), dy * s * (1 - s)) < 1e-6
    dw, db = torch.empty((K, N), device=DEV), torch.empty(N, device=DEV)
    hip.wgrad(_t(x), _t(dy), dw, db)
    assert helpers.rel_err(dw.cpu().numpy(), x.astype(np.float64).T @ dy) < 2e-6
    assert helpers.rel_err(db.cpu().numpy(), dy.astype(np.float64).sum(0)) < 2e-6
    db2 = torch.empty(N, device=DEV)
    hip.wgrad(None, _t(dy), None, db2)
    assert torch.equal(db, db2)
    dw2 = torch.empty((K, N), device=DEV)
    hip.wgrad(_t(x), _t(dy), dw2, None)
    assert torch.equal(dw, dw2), "two-stage reduction: reproducible bits"
    # BCE gradient
    p = rng.uniform(0, 1, 500).astype(np.float32); p[:3] = [0.0, 1.0, 5e-8]
    lab = rng.integers(0, 2, 500).astype(np.float32)
    dzb, terms = torch.empty((500, 1), device=DEV), torch.empty(500, device=DEV)
    hip.bce_grad(_t(p.reshape(-1, 1)), _t(lab), dzb, terms)
    # Keras evaluates this in float32: so does the expectation (1 - 1e-7 is not representable, the clip lands on 1 - 1.19e-7)
    e32, one = np.float32(1e-7), np.float32(1)
    pc = np.clip(p, e32, one - e32)
    want_terms = -(lab * np.log(pc + e32) + (one - lab) * np.log(one - pc + e32))
    inside = (p >= e32) & (p <= one - e32)
    want_dz = -(lab / (pc + e32) - (one - lab) / (one - pc + e32)) / np.float32(500) * inside * p * (one - p)
    assert helpers.rel_err(terms.cpu().numpy(), want_terms.astype(np.float64)) < 1e-5
    assert helpers.rel_err(dzb.cpu().numpy()[:, 0], want_dz.astype(np.float64)) < 1e-5
    # scatter-add, in-place add, transpose, Adam
    ids = rng.integers(5, 60, 2000).astype(np.int32)
    src = rng.standard_normal((2000, 12)).astype(np.float32)
    dst = torch.zeros((55, 20), device=DEV)
    hip.scatter_add_rows(_t(src), _t(ids), dst[:, 4:16], base=5)
    want = np.zeros((55, 12)); np.add.at(want, ids - 5, src)
    assert helpers.rel_err(dst.cpu().numpy()[:, 4:16], want) < 1e-5 and float(dst[:, :4].abs().max()) == 0
    # ... without atomics for a batch-sized id list: the first position of an id adds the rows of all its positions in position order,
    # so the float32 result is exactly the sequential sum (and the same on every run); longer lists use float atomics (tolerance only)
    seq = np.zeros((55, 12), np.float32)
    first = {}
    for q, r in enumerate(ids - 5):
        if r not in first:
            first[r] = q
            seq[r] = src[q]
        else:
            seq[r] = seq[r] + src[q]
    assert np.array_equal(dst.cpu().numpy()[:, 4:16], seq)
    dst2 = torch.zeros((55, 20), device=DEV)
    hip.scatter_add_rows(_t(src), _t(ids), dst2[:, 4:16], base=5)
    assert torch.equal(dst, dst2)
    for width in (1, 5, 8):                                          # widths that are not multiples of four
        d3 = torch.zeros((55, width), device=DEV)
        hip.scatter_add_rows(_t(np.ascontiguousarray(src[:, :width])), _t(ids), d3, base=5)
        w3 = np.zeros((55, width)); np.add.at(w3, ids - 5, src[:, :width])
        assert helpers.rel_err(d3.cpu().numpy(), w3) < 1e-5
    big_ids = rng.integers(0, 3000, 20000).astype(np.int32)
    big_src = rng.standard_normal((20000, 8)).astype(np.float32)
    big = torch.zeros((3000, 8), device=DEV)
    hip.scatter_add_rows(_t(big_src), _t(big_ids), big)
    want_big = np.zeros((3000, 8)); np.add.at(want_big, big_ids, big_src)
    assert helpers.rel_err(big.cpu().numpy(), want_big) < 1e-5
    a, b = _t(src[:50]), _t(src[50:100])
    hip.add_inplace(a, b, 0.5)
    assert helpers.rel_err(a.cpu().numpy(), src[:50] + 0.5 * src[50:100]) < 1e-6
    assert np.array_equal(hip.transpose(_t(x)).cpu().numpy(), x.T)
    w, g = rng.standard_normal(1000).astype(np.float32), rng.standard_normal(1000).astype(np.float32)
    m, v = rng.standard_normal(1000).astype(np.float32) * 0.1, rng.uniform(0, 1, 1000).astype(np.float32)
    wd, md, vd = _t(w), _t(m), _t(v)
    lr_t = 1e-3 * np.sqrt(1 - 0.999 ** 3) / (1 - 0.9 ** 3)
    hip.adam(wd, _t(g), md, vd, lr_t, 0.9, 0.999, 1e-7, l2=1e-3)
    g2 = g.astype(np.float64) + 2e-3 * w
    m2, v2 = 0.9 * m + 0.1 * g2, 0.999 * v + 0.001 * g2 * g2
    assert helpers.rel_err(wd.cpu().numpy(), w - lr_t * m2 / (np.sqrt(v2) + 1e-7)) < 1e-6
    assert helpers.rel_err(md.cpu().numpy(), m2) < 1e-6 and helpers.rel_err(vd.cpu().numpy(), v2) < 1e-6


@pytest.mark.parametrize('M,K,N,act', [(1024, 48, 48, 'relu'), (85, 96, 64, 'relu'), (1024, 64, 1, 'sigmoid'), (1024, 64, 1, None),
                                         (9228, 16, 16, None), (1, 24, 24, 'relu'), (300, 128, 128, 'relu'), (64, 5, 3, 'sigmoid'), (40000, 32, 8, 'relu'),
                                         (600001, 8, 8, 'relu'), (40000, 8, 8, 'relu'), (40000, 16, 16, None), (70000, 32, 32, 'relu'), (40000, 16, 8, 'sigmoid'),
                                         (40000, 8, 32, None), (40000, 24, 12, 'relu'), (40000, 6, 8, 'relu'), (9228, 1, 8, None), (40000, 2, 16, None)])     # (K = 1: a GAT attention vector's gradient, X read by single floats)                     # (a convolution layer's reverse pass over every node: folded partials, two tiles per workgroup)
def test_dense_bwd_fused(hip, M, K, N, act):
    """amar_dense_bwd_f32 (round 4: the reverse pass of one Dense layer in two launches instead of four — act', dX = dZ . W^T, dW = X^T . dZ, db) against
    float64 arithmetic and against the separate kernels it replaces; strided operands (column slices of wider buffers, as the
    classifier's split input gradient and the GCN stack's concat buffer are); any of the outputs left out; called twice on one
    workspace with bit-identical results."""
    rng = np.random.default_rng(M + K + N)
    xw = rng.standard_normal((M, K + 8)).astype(np.float32)            # X is a column slice of a wider buffer
    w = (rng.standard_normal((K, N)) * 0.3).astype(np.float32)
    z = (xw[:, 4:4 + K].astype(np.float64) @ w + rng.standard_normal(N) * 0.1)
    y = (np.maximum(z, 0) if act == 'relu' else 1 / (1 + np.exp(-z)) if act == 'sigmoid' else z).astype(np.float32)
    dyw = rng.standard_normal((M, N + 4)).astype(np.float32)           # ... and so is dY
    x_d, y_d, dy_d, w_d = _t(xw)[:, 4:4 + K], _t(y), _t(dyw)[:, 2:2 + N], _t(w)
    dy64, y64 = dyw[:, 2:2 + N].astype(np.float64), y.astype(np.float64)
    dz = dy64 * (y64 > 0) if act == 'relu' else dy64 * y64 * (1 - y64) if act == 'sigmoid' else dy64
    want_dx, want_dw, want_db = dz @ w.astype(np.float64).T, xw[:, 4:4 + K].astype(np.float64).T @ dz, dz.sum(0)
    assert hip.dense_bwd_supported(K, N)
    ws = hip.dense_bwd_workspace(M, K, N, DEV)
    dxw = torch.zeros((M, K + 3), device=DEV)
    dx, dw, db = dxw[:, 1:1 + K], torch.empty((K, N), device=DEV), torch.empty(N, device=DEV)
    hip.dense_bwd(x_d, y_d if act is not None else None, dy_d, w_d, act, ws, dX=dx, dW=dw, db=db)
    tol = 3e-6
    assert helpers.rel_err(dx.cpu().numpy(), want_dx) < tol and helpers.rel_err(dw.cpu().numpy(), want_dw) < tol
    assert helpers.rel_err(db.cpu().numpy(), want_db) < tol
    assert float(dxw[:, 0].abs().max()) == 0 and float(dxw[:, 1 + K:].abs().max()) == 0        # the strided store stays in its columns
    dx2, dw2, db2 = torch.empty((M, K), device=DEV), torch.empty((K, N), device=DEV), torch.empty(N, device=DEV)
    hip.dense_bwd(x_d, y_d if act is not None else None, dy_d, w_d, act, ws, dX=dx2, dW=dw2, db=db2)
    assert torch.equal(dx2, dx) and torch.equal(dw2, dw) and torch.equal(db2, db)
    # outputs left out: only the weight gradients (a first layer over constant inputs), only dX + dW (a GCN layer's X_k^T . dH and dH . W^T)
    dw3, db3 = torch.empty((K, N), device=DEV), torch.empty(N, device=DEV)
    hip.dense_bwd(x_d, y_d if act is not None else None, dy_d, None, act, ws, dW=dw3, db=db3)
    assert torch.equal(dw3, dw) and torch.equal(db3, db)
    dx4, dw4 = torch.empty((M, K), device=DEV), torch.empty((K, N), device=DEV)
    hip.dense_bwd(x_d, y_d if act is not None else None, dy_d, w_d, act, ws, dX=dx4, dW=dw4)
    assert torch.equal(dx4, dx) and torch.equal(dw4, dw)
    # dZ itself as an output (act', the bias gradient and dZ in one launch: what a GCN layer's reverse pass starts with), the input
    # gradient ACCUMULATED into a buffer that already carries one, and deferred partial sums (what the Adam launch adds itself)
    dz6, db6 = torch.empty((M, N), device=DEV), torch.empty(N, device=DEV)
    hip.dense_bwd(None, y_d if act is not None else None, dy_d, None, act, ws, db=db6, dZ=dz6, K=1)
    assert helpers.rel_err(dz6.cpu().numpy(), dz) < 1e-6 and torch.equal(db6, db)
    dx7 = torch.ones((M, K), device=DEV)
    hip.dense_bwd(x_d, y_d if act is not None else None, dy_d, w_d, act, ws, dX=dx7, accumulate_dx=True)
    assert helpers.rel_err(dx7.cpu().numpy() - 1.0, want_dx) < 1e-5
    lazy_w, lazy_b = hip.dense_bwd(x_d, y_d if act is not None else None, dy_d, w_d, act, ws, dX=dx2, dW=dw2, db=db2, defer=True)
    assert torch.equal(lazy_w.materialize(), dw) is not None and helpers.rel_err(lazy_w.materialize().cpu().numpy(), want_dw) < tol
    assert helpers.rel_err(lazy_b.materialize().cpu().numpy(), want_db) < tol and 1 <= lazy_w.groups == lazy_b.groups <= 64
    # against the kernels it replaces (other summation orders: tolerance, not bits)
    dz_d = torch.empty((M, N), device=DEV)
    if act is not None:
        hip.act_bwd(dy_d, y_d, dz_d, act)
    else:
        dz_d.copy_(dy_d)
    dw5, db5, dx5 = torch.empty((K, N), device=DEV), torch.empty(N, device=DEV), torch.empty((M, K), device=DEV)
    hip.wgrad(x_d, dz_d, dw5, db5)
    hip.dense(dz_d, w_d, None, dx5, act=None, w_transposed=True)
    assert helpers.rel_err(dw.cpu().numpy(), dw5.cpu().numpy().astype(np.float64)) < tol
    assert helpers.rel_err(dx.cpu().numpy(), dx5.cpu().numpy().astype(np.float64)) < tol
    assert not hip.dense_bwd_supported(768, 256)                       # the content towers' wide layers keep the separate kernels


@pytest.mark.parametrize('M,dims,acts,gather', [(1024, [24, 24, 24], ['relu', 'relu'], True), (85, [96, 64, 64, 1], ['relu', 'relu', 'sigmoid'], False),
                                               (1024, [48, 48, 48, 48, 1], ['relu', 'relu', 'relu', 'sigmoid'], False), (3, [5, 7, 3], ['relu', None], True),
                                               (300, [128, 128, 16], ['relu', 'relu'], False)])
def test_dense_stack_one_launch(hip, M, dims, acts, gather):
    """amar_dense_stack_f32 (round 4: a Dense stack's forward in one launch, every layer's output kept, optional row gather and a strided
    last output) against float64 and against amar_dense_f32 layer by layer."""
    rng = np.random.default_rng(M + sum(dims))
    n_src = 500
    x = rng.standard_normal((n_src if gather else M, dims[0])).astype(np.float32)
    ids = rng.integers(0, n_src, M).astype(np.int32) if gather else None
    ws = [(rng.standard_normal((dims[l], dims[l + 1])) * 0.3).astype(np.float32) for l in range(len(acts))]
    bs = [(rng.standard_normal(dims[l + 1]) * 0.1).astype(np.float32) for l in range(len(acts))]
    assert hip.dense_stack_supported(dims)
    outs = [torch.empty((M, d), device=DEV) for d in dims[1:]]
    wide = torch.zeros((M, dims[-1] + 5), device=DEV)                  # the last output is a column slice of a wider buffer
    outs[-1] = wide[:, 2:2 + dims[-1]]
    xcopy = torch.empty((M, dims[0]), device=DEV) if gather else None
    hip.dense_stack(_t(x), [_t(w) for w in ws], [_t(b) for b in bs], acts, outs, ids=_t(ids) if gather else None, xcopy=xcopy)
    cur = x[ids].astype(np.float64) if gather else x.astype(np.float64)
    if gather:
        assert np.array_equal(xcopy.cpu().numpy(), x[ids])
    ref_in = _t(x[ids] if gather else x)
    for l, act in enumerate(acts):
        z = cur @ ws[l].astype(np.float64) + bs[l]
        cur = np.maximum(z, 0) if act == 'relu' else 1 / (1 + np.exp(-z)) if act == 'sigmoid' else z
        assert helpers.rel_err(outs[l].cpu().numpy(), cur) < 3e-6, l
        ref = torch.empty((M, dims[l + 1]), device=DEV)
        hip.dense(ref_in, _t(ws[l]), _t(bs[l]), ref, act=act)
        assert helpers.rel_err(outs[l].cpu().numpy(), ref.cpu().numpy().astype(np.float64)) < 3e-6, l
        ref_in = ref
    assert float(wide[:, :2].abs().max()) == 0 and float(wide[:, 2 + dims[-1]:].abs().max()) == 0
    assert not hip.dense_stack_supported([768, 256, 64]) and not hip.dense_stack_supported([8, 8, 8, 8, 8, 8])


@pytest.mark.parametrize('M,dims,acts,top_is_dz', [(1024, [24, 24, 24], ['relu', 'relu'], False), (85, [96, 64, 64, 1], ['relu', 'relu', 'sigmoid'], True),
                                                  (1000, [48, 48, 48, 48, 1], ['relu', 'relu', 'relu', 'sigmoid'], True), (3, [5, 7, 3], ['relu', None], False),
                                                  (300, [128, 128, 16], ['relu', 'relu'], False)])
def test_dense_stack_bwd_one_launch(hip, M, dims, acts, top_is_dz):
    """amar_dense_stack_bwd_f32 (round 4: the reverse pass of a whole Dense stack in one launch) against float64: every layer's dW and
    db, the input gradient, with the top gradient given w.r.t. the last output or w.r.t. its pre-activation; strided top operands;
    deferred partial sums equal the reduced gradients."""
    rng = np.random.default_rng(M + sum(dims))
    L = len(acts)
    ws = [(rng.standard_normal((dims[l], dims[l + 1])) * 0.3).astype(np.float32) for l in range(L)]
    bs = [(rng.standard_normal(dims[l + 1]) * 0.1).astype(np.float32) for l in range(L)]
    xs = [rng.standard_normal((M, dims[0])).astype(np.float32)]
    for l, act in enumerate(acts):                                     # forward in float32 on the host: the tape's saved activations
        z = xs[-1].astype(np.float64) @ ws[l] + bs[l]
        xs.append((np.maximum(z, 0) if act == 'relu' else 1 / (1 + np.exp(-z)) if act == 'sigmoid' else z).astype(np.float32))
    wide = rng.standard_normal((M, dims[-1] + 4)).astype(np.float32)   # the top gradient is a column slice of a wider buffer
    top = wide[:, 4:]
    g = top.astype(np.float64)
    want_dw, want_db = [None] * L, [None] * L
    for l in range(L - 1, -1, -1):
        y = xs[l + 1].astype(np.float64)
        if not (top_is_dz and l == L - 1):
            g = g * (y > 0) if acts[l] == 'relu' else g * y * (1 - y) if acts[l] == 'sigmoid' else g
        want_dw[l], want_db[l] = xs[l].astype(np.float64).T @ g, g.sum(0)
        g = g @ ws[l].astype(np.float64).T
    want_dx = g
    assert hip.dense_stack_bwd_supported(dims, M)
    wk = hip.dense_stack_bwd_workspace(M, dims, DEV)
    ins = [_t(x) for x in xs[:-1]]
    dws, dbs = [torch.empty((dims[l], dims[l + 1]), device=DEV) for l in range(L)], [torch.empty(dims[l + 1], device=DEV) for l in range(L)]
    dx0 = torch.empty((M, dims[0]), device=DEV)
    top_d = _t(wide)[:, 4:]
    hip.dense_stack_bwd(top_d, None if top_is_dz else _t(xs[-1]), ins, [_t(w) for w in ws], acts, wk, dws, dbs, dX0=dx0)
    tol = 5e-6
    assert helpers.rel_err(dx0.cpu().numpy(), want_dx) < tol
    for l in range(L):
        assert helpers.rel_err(dws[l].cpu().numpy(), want_dw[l]) < tol, l
        assert helpers.rel_err(dbs[l].cpu().numpy(), want_db[l]) < tol, l
    lazy = hip.dense_stack_bwd(top_d, None if top_is_dz else _t(xs[-1]), ins, [_t(w) for w in ws], acts, wk,
                               [torch.empty_like(w) for w in dws], [torch.empty_like(b) for b in dbs], dX0=None, defer=True)
    for l in range(L):
        assert helpers.rel_err(lazy[l][0].materialize().cpu().numpy(), dws[l].cpu().numpy().astype(np.float64)) < 1e-6
        assert helpers.rel_err(lazy[l][1].materialize().cpu().numpy(), dbs[l].cpu().numpy().astype(np.float64)) < 1e-6
    assert not hip.dense_stack_bwd_supported(dims, 5000) and not hip.dense_stack_bwd_supported([768, 256, 64], 100)


def test_sage_training_kernels(hip):
    rng = np.random.default_rng(1)
    M, W = 777, 12
    a, b = rng.standard_normal((M, W)).astype(np.float32), rng.standard_normal((M, W)).astype(np.float32)
    sc = rng.uniform(0.1, 1, M).astype(np.float32)
    out = torch.zeros((M, 2 * W), device=DEV)
    hip.row_affine(_t(a), _t(sc), out[:, W:], b=_t(b))
    assert helpers.rel_err(out.cpu().numpy()[:, W:], (a + b) * sc[:, None]) < 1e-6 and float(out[:, :W].abs().max()) == 0
    hip.row_affine(_t(a), _t(sc), out[:, :W])
    assert helpers.rel_err(out.cpu().numpy()[:, :W], a * sc[:, None]) < 1e-6
    z = rng.standard_normal((M, W)).astype(np.float32)
    z[5] = 0                                           # clamped row: inv = 1e6
    z[6] *= 1e-8
    nrm, inv, y = torch.empty((M, W), device=DEV), torch.empty(M, device=DEV), torch.empty((M, W), device=DEV)
    hip.l2norm_fwd(_t(z), nrm, inv, y, act='relu')
    z64 = z.astype(np.float64)
    want_inv = 1 / np.sqrt(np.maximum((z64 * z64).sum(1), 1e-12))
    assert helpers.rel_err(inv.cpu().numpy(), want_inv) < 1e-6
    assert helpers.rel_err(nrm.cpu().numpy(), z64 * want_inv[:, None]) < 1e-6
    assert np.array_equal(y.cpu().numpy(), np.maximum(nrm.cpu().numpy(), 0))
    dy = rng.standard_normal((M, W)).astype(np.float32)
    dz = torch.empty((M, W), device=DEV)
    hip.l2norm_bwd(_t(dy), nrm, inv, dz, act='relu')
    zt = torch.tensor(z64, requires_grad=True)
    nt = zt * torch.rsqrt(torch.clamp((zt * zt).sum(1, keepdim=True), min=1e-12))
    (torch.relu(nt) * torch.tensor(dy.astype(np.float64))).sum().backward()
    assert helpers.rel_err(dz.cpu().numpy(), zt.grad.numpy()) < 1e-5


@pytest.mark.parametrize('C', [4, 16, 32, 64, 24, 48])      # 24 / 48: lane groups padded to the next power of two
@pytest.mark.parametrize('self_loop', [True, False])
def test_gat_bwd_kernel(hip, C, self_loop):
    """amar_gat_bwd_f32 for every lane layout (C/4 lanes per edge) against torch autograd of the restated forward."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import convert_to_tensor
    g = helpers.tiny_graph(n_users=50, n_items=40, n_ratings=700, seed=C, n_props=10, n_links=40)
    a = convert_to_tensor(g['adj'], with_values=False, drop_diagonal=True)
    n = g['adj'].shape[0]
    rng = np.random.default_rng(C)
    h = rng.standard_normal((n, C)).astype(np.float32) * 0.7
    a_s, a_n = rng.standard_normal(C).astype(np.float32) * 0.5, rng.standard_normal(C).astype(np.float32) * 0.5
    bias = rng.standard_normal(C).astype(np.float32) * 0.1
    dy = rng.standard_normal((n, C)).astype(np.float32)
    rowptr, colidx = a.rowptr.cpu().numpy(), a.colidx.cpu().numpy()
    tgt = np.repeat(np.arange(n), np.diff(rowptr)); src = colidx.astype(np.int64)
    if self_loop:
        tgt, src = np.concatenate([tgt, np.arange(n)]), np.concatenate([src, np.arange(n)])
    ht = torch.tensor(h.astype(np.float64), requires_grad=True)
    ast, ant = torch.tensor(a_s.astype(np.float64), requires_grad=True), torch.tensor(a_n.astype(np.float64), requires_grad=True)
    T, S = torch.as_tensor(tgt), torch.as_tensor(src)
    e = (ht @ ast)[T] + (ht @ ant)[S]
    e = torch.where(e > 0, e, 0.2 * e)
    mx = torch.full((n,), -1e30, dtype=torch.float64).scatter_reduce(0, T, e.detach(), 'amax')
    ex = torch.exp(e - mx[T])
    alpha = ex / (torch.zeros(n, dtype=torch.float64).index_add(0, T, ex) + 1e-9)[T]
    y = torch.relu(torch.zeros_like(ht).index_add(0, T, alpha[:, None] * ht[S]) + torch.tensor(bias.astype(np.float64)))
    (y * torch.tensor(dy.astype(np.float64))).sum().backward()
    hd, sd, nd = _t(h), torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    x_id = torch.eye(C, device=DEV)
    hip.rowwise_xw(hd, x_id.contiguous(), torch.empty((n, C), device=DEV), a_self=_t(a_s), a_neigh=_t(a_n), s_self=sd, s_neigh=nd)
    yd = torch.empty((n, C), device=DEV)
    hip.gat_layer(a.rowptr, a.colidx, hd, sd, nd, _t(bias), yd, self_loop=self_loop)
    assert helpers.rel_err(yd.cpu().numpy(), y.detach().numpy()) < 1e-5
    dout, ds, dt, dh = hip.gat_bwd(a.rowptr, a.colidx, hd, sd, nd, yd, _t(dy), _t(bias), _t(a_s), _t(a_n), self_loop=self_loop)
    assert np.array_equal(dout.cpu().numpy(), dy * (yd.cpu().numpy() > 0))
    assert np.abs(dh.cpu().numpy() - ht.grad.numpy()).max() <= 2e-4 * np.abs(ht.grad.numpy()).max()
    das = (hd.double() * ds.double()[:, None]).sum(0).cpu().numpy()
    dan = (hd.double() * dt.double()[:, None]).sum(0).cpu().numpy()
    scale = max(np.abs(ant.grad.numpy()).max(), np.abs(ast.grad.numpy()).max())
    assert np.abs(das - ast.grad.numpy()).max() <= 2e-4 * scale and np.abs(dan - ant.grad.numpy()).max() <= 2e-4 * scale


def _flatten_oracle_grads(model, grads):
    """Oracle gradient containers -> {product parameter: ndarray}."""
    out = {}
    seq = model.gnn.gnn_layers
    out[seq.embeddings] = grads['gnn']['embeddings']
    for layer, gl in zip(seq.seq_layers, grads['gnn']['layers']):
        for name, arr in gl.items():
            out[getattr(layer, {'attn_self': 'attn_kernel_self', 'attn_neigh': 'attn_kernel_neighs'}.get(name, name))] = arr
    if 'reduction_w' in grads['gnn']:                                # ReductionLayer('w-sum')
        out[seq.reduce.w] = grads['gnn']['reduction_w']
    for name in grads['head']:
        if name.startswith('fuse'):                                  # attention fusion weights
            for key, arr in grads['head'][name].items():
                out[getattr(getattr(model.rs, name), key)] = arr
            continue
        for layer, (gw, gb) in zip(getattr(model.rs, name).layers, grads['head'][name]):
            out[layer.kernel], out[layer.bias] = gw, gb
    return out


@pytest.mark.parametrize('fused', [False, True])
@pytest.mark.parametrize('cls', ['BasicGCN', 'BasicLightGCN'])
@pytest.mark.parametrize('graph', ['ui', 'uip'])
def test_gradients_match_oracle(hip, cls, graph, fused, monkeypatch):
    from deep_cbrs_amar_renaissance_amd import engine, training
    monkeypatch.setenv('AMAR_DENSE_BWD', '1' if fused else '0')     # the tapes on amar_dense_bwd_f32 / amar_dense_stack_f32 (the default) or on the separate kernels
    monkeypatch.setenv('AMAR_DENSE_STACK', '1' if fused else '0')
    monkeypatch.setenv('AMAR_DENSE_STACK_BWD', '1' if fused else '0')
    from deep_cbrs_amar_renaissance_amd.models import basic
    engine.set_seed(5)
    g = helpers.tiny_graph(n_users=80, n_items=60, n_ratings=1500, seed=9,
                           n_props=30 if graph == 'uip' else 0, n_links=90 if graph == 'uip' else 0)
    model = getattr(basic, cls)(g['adj'], **CFG)
    helpers.randomize_biases(model, seed=6)
    rng = np.random.default_rng(2)
    y = rng.integers(0, 2, len(g['u_ids']))
    trainer = training.Trainer(model)
    loss, grads = trainer.loss_and_grads(g['u_ids'], g['i_ids'], y)
    want_loss, want, _ = otrain.loss_and_grads(g['adj'], helpers.gnn_to_oracle(model.gnn), helpers.basic_head_to_oracle(model.rs),
                                               g['u_ids'], g['i_ids'], y, l2=1e-4)
    assert abs(loss - want_loss) < 1e-5
    flat = _flatten_oracle_grads(model, want)
    assert set(flat) == set(grads)
    for prm, gw in flat.items():
        got = grads[prm].cpu().numpy().reshape(gw.shape).astype(np.float64)
        got += 2 * trainer._l2(prm) * prm.detach().cpu().numpy().reshape(gw.shape)      # the trainer folds the L2 term into amar_adam_f32
        # (absolute floor: d/d(attn_kernel_self) vanishes where a row's softmax is shift-invariant in s_i)
        assert np.abs(got - gw).max() <= 2e-4 * np.abs(gw).max() + 1e-10, tuple(prm.shape)


@pytest.mark.parametrize('cls', ['BasicGraphSage', 'BasicGAT', 'BasicDGCF'])
@pytest.mark.parametrize('graph', ['ui', 'uip'])
def test_gradients_match_autograd_oracle(hip, cls, graph):
    """Model kinds without a manual numpy reverse pass: oracle = torch autograd of the restated forward (float64)."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import basic
    engine.set_seed(5)
    g = helpers.tiny_graph(n_users=80, n_items=60, n_ratings=1500, seed=9,
                           n_props=30 if graph == 'uip' else 0, n_links=90 if graph == 'uip' else 0)
    model = getattr(basic, cls)(g['adj'], **CFG)
    helpers.randomize_biases(model, seed=6)
    y = np.random.default_rng(2).integers(0, 2, len(g['u_ids']))
    trainer = training.Trainer(model)
    loss, grads = trainer.loss_and_grads(g['u_ids'], g['i_ids'], y)
    if cls == 'BasicDGCF':                                    # gates away from their all-ones start
        with torch.no_grad():
            for layer in model.gnn.gnn_layers.seq_layers:
                layer.w.add_(torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, tuple(layer.w.shape)).astype(np.float32)).to(layer.w.device))
        loss, grads = trainer.loss_and_grads(g['u_ids'], g['i_ids'], y)
    # the training forward (unfused, keeps intermediates) scores like the fused inference forward
    with torch.no_grad():
        e_inf = model.gnn.gnn_layers(None)
        e_trn = trainer._propagation_forward()
    assert float((e_inf - e_trn).abs().max()) < 2e-6
    want_loss, want, _ = otrain.torch_model_grads(g['adj'], helpers.gnn_to_oracle(model.gnn), helpers.basic_head_to_oracle(model.rs),
                                                  g['u_ids'], g['i_ids'], y, l2=1e-4)
    assert abs(loss - want_loss) < 1e-5
    flat = _flatten_oracle_grads(model, want)
    assert set(flat) == set(grads)
    for prm, gw in flat.items():
        got = grads[prm].cpu().numpy().reshape(gw.shape).astype(np.float64)
        got += 2 * trainer._l2(prm) * prm.detach().cpu().numpy().reshape(gw.shape)
        # (absolute floor: d/d(attn_kernel_self) vanishes where a row's softmax is shift-invariant in s_i)
        assert np.abs(got - gw).max() <= 2e-4 * np.abs(gw).max() + 1e-10, tuple(prm.shape)


@pytest.mark.parametrize('cls', ['BasicGCN', 'BasicGraphSage', 'BasicGAT'])
def test_weighted_sum_reduction_gradients_match_autograd_oracle(hip, cls):
    """final_node='w-sum' (WeightedSum, reduction.py:36-55): out = sum_l w_l^2 X_l with learnable w.  Forward against the numpy
    oracle, loss and every gradient — the reduction weights' included — against torch autograd of the restated forward (float64),
    with the weights moved away from their all-ones start."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import basic
    from oracle import models as om
    engine.set_seed(7)
    g = helpers.tiny_graph(n_users=80, n_items=60, n_ratings=1500, seed=9, n_props=30, n_links=90)
    model = getattr(basic, cls)(g['adj'], **dict(CFG, final_node='w-sum'))
    helpers.randomize_biases(model, seed=6)
    red = model.gnn.gnn_layers.reduce
    assert tuple(red.w.shape) == (3, 1, 1) and float(red.w.detach().min()) == 1.0 and red.w.regularizer is None       # reduction.py:45-52, gnn.py:62
    with torch.no_grad():
        red.w.copy_(torch.tensor([0.7, -1.3, 0.4], device=red.w.device).view(3, 1, 1))
    got = model.gnn(None).cpu().numpy()
    want_e = om.propagate(g['adj'], helpers.gnn_to_oracle(model.gnn), np.float64)
    assert got.shape == want_e.shape == (g['adj'].shape[0], 8) and helpers.rel_err(got, want_e) < 1e-5
    y = np.random.default_rng(2).integers(0, 2, len(g['u_ids']))
    trainer = training.Trainer(model)
    assert any(p is red.w for p in trainer.params)
    loss, grads = trainer.loss_and_grads(g['u_ids'], g['i_ids'], y)
    want_loss, want, _ = otrain.torch_model_grads(g['adj'], helpers.gnn_to_oracle(model.gnn), helpers.basic_head_to_oracle(model.rs),
                                                  g['u_ids'], g['i_ids'], y, l2=1e-4)
    assert abs(loss - want_loss) < 1e-5
    flat = _flatten_oracle_grads(model, want)
    assert set(flat) == set(grads)
    for prm, gw in flat.items():
        got = grads[prm].cpu().numpy().reshape(gw.shape).astype(np.float64)
        got += 2 * trainer._l2(prm) * prm.detach().cpu().numpy().reshape(gw.shape)
        assert np.abs(got - gw).max() <= 2e-4 * np.abs(gw).max() + 1e-10, tuple(prm.shape)
    loss2, grads2 = trainer.loss_and_grads(g['u_ids'], g['i_ids'], y)                  # reproducible bit for bit: no float atomics on this path
    assert loss2 == loss
    for prm in grads:
        assert torch.equal(grads2[prm], grads[prm]), tuple(prm.shape)


@pytest.mark.parametrize('cls,feature_based,fusion,residual', [
    ('HybridBertGCN', True, 'concatenate', False), ('HybridBertGCN', False, 'concatenate', False),
    ('HybridBertGraphSage', True, 'concatenate', False), ('HybridBertLightGCN', True, 'concatenate', False),
    ('HybridBertGAT', True, 'concatenate', False),
    ('HybridBertGCN', True, 'attention', False), ('HybridBertGCN', True, 'concatenate', True),        # hybrid-gnn-tweaks*.yaml
    ('HybridBertGCN', False, 'attention', False), ('HybridBertLightGCN', True, 'attention', True)])
def test_hybrid_gradients_match_autograd_oracle(hip, cls, feature_based, fusion, residual):
    """HybridBertGNN (hybrid.py:92-140): GNN + four-input head; BERT rows from the batch or from the resident table."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    engine.set_seed(11)
    g = helpers.tiny_graph(n_users=80, n_items=60, n_ratings=1500, seed=4)
    cfg = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[[24, 16], [32, 24], [16, 16]], clf_units=[24, 16],
               l2_regularizer=1e-4, feature_based=feature_based, fusion_method=fusion, residual=residual)
    model = getattr(hybrid, cls)(g['adj'], **cfg)
    rng = np.random.default_rng(3)
    table = rng.standard_normal((g['adj'].shape[0], 40)).astype(np.float32) * 0.5
    model.set_bert_table(table)
    y = rng.integers(0, 2, len(g['u_ids']))
    trainer = training.Trainer(model)
    helpers.randomize_biases(model, seed=7)
    u, i = g['u_ids'], g['i_ids']
    loss, grads = trainer.loss_and_grads(u, i, y)                                  # rows of the resident table
    loss_b, grads_b = trainer.loss_and_grads(u, i, y, bert=(table[u], table[i]))   # blocks delivered by the Sequence
    # (the embedding-row scatter uses float atomics: sums may differ in the last bits between runs)
    assert loss == loss_b and all(torch.allclose(grads[k], grads_b[k], rtol=1e-5, atol=1e-9) for k in grads)
    want_loss, want, p = otrain.torch_model_grads(g['adj'], helpers.gnn_to_oracle(model.gnn), helpers.hybrid_head_to_oracle(model.rs),
                                                  u, i, y, l2=1e-4, bert=(table[u], table[i]), feature_based=feature_based)
    with torch.no_grad():
        got_p = model((u, i, None, None)).cpu().numpy()[:, 0]
    assert np.abs(got_p - p).max() < 1e-5
    assert abs(loss - want_loss) < 1e-5
    flat = _flatten_oracle_grads(model, want)
    assert set(flat) == set(grads)
    for prm, gw in flat.items():
        got = grads[prm].cpu().numpy().reshape(gw.shape).astype(np.float64)
        got += 2 * trainer._l2(prm) * prm.detach().cpu().numpy().reshape(gw.shape)
        # (absolute floor: d/d(attn_kernel_self) vanishes where a row's softmax is shift-invariant in s_i)
        assert np.abs(got - gw).max() <= 2e-4 * np.abs(gw).max() + 1e-10, tuple(prm.shape)


def test_hybrid_fit_reduces_loss(hip):
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    engine.set_seed(2)
    g = helpers.tiny_graph(n_users=60, n_items=50, n_ratings=1200, seed=1)
    model = hybrid.HybridBertGCN(g['adj'], embedding_dim=8, n_hiddens=[8, 8], dense_units=[[16], [16], [16]], clf_units=[16],
                                 feature_based=True, l2_regularizer=1e-5)
    rng = np.random.default_rng(0)
    table = rng.standard_normal((g['adj'].shape[0], 24)).astype(np.float32)
    u, i = g['u_ids'], g['i_ids']
    y = ((table[u, 0] + table[i, 1]) > 0).astype(np.float32)          # learnable from the BERT rows alone

    class Seq:
        def __len__(self):
            return 3

        def __getitem__(self, b):
            s = slice(b * 100, (b + 1) * 100)
            return (u[s], i[s], table[u[s]], table[i[s]]), y[s]
    import types
    model.compile(optimizer=types.SimpleNamespace(learning_rate=5e-3, beta_1=0.9))
    hist = model.fit(Seq(), epochs=30, verbose=False)['loss']
    assert hist[-1] < 0.6 * hist[0], hist[::6]


def test_adam_steps_match_oracle(hip):
    """Three optimizer steps on one batch: weights follow the oracle's Adam to fp32 rounding."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import basic
    engine.set_seed(8)
    g = helpers.tiny_graph(n_users=80, n_items=60, n_ratings=1500, seed=3)
    model = basic.BasicGCN(g['adj'], **CFG)
    helpers.randomize_biases(model, seed=1)
    y = np.random.default_rng(4).integers(0, 2, len(g['u_ids']))
    gnn, head = helpers.gnn_to_oracle(model.gnn), helpers.basic_head_to_oracle(model.rs)
    gnn = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in gnn.items()}
    gnn['layers'] = [{k: v.astype(np.float64) for k, v in lw.items()} for lw in gnn['layers']]
    head = {k: [(w.astype(np.float64), b.astype(np.float64)) for w, b in net] for k, net in head.items()}
    state = {}
    trainer = training.Trainer(model)
    for t in range(1, 4):
        trainer.train_batch(g['u_ids'], g['i_ids'], y)
        _, og, _ = otrain.loss_and_grads(g['adj'], gnn, head, g['u_ids'], g['i_ids'], y, l2=1e-4)

        def upd(key, w, gr):
            m, v = state.get(key, (np.zeros_like(w), np.zeros_like(w)))
            w2, m, v = otrain.adam_update(w, gr, m, v, t)
            state[key] = (m, v)
            return w2
        gnn['embeddings'] = upd('emb', gnn['embeddings'], og['gnn']['embeddings'])
        for k, lw in enumerate(gnn['layers']):
            for nm in ('kernel', 'bias'):
                lw[nm] = upd(('l', k, nm), lw[nm], og['gnn']['layers'][k][nm])
        for name in head:
            head[name] = [(upd((name, k, 'w'), w, og['head'][name][k][0]), upd((name, k, 'b'), b, og['head'][name][k][1]))
                          for k, (w, b) in enumerate(head[name])]
    got = helpers.gnn_to_oracle(model.gnn)
    assert np.abs(got['embeddings'] - gnn['embeddings']).max() < 2e-5          # three steps of ~1e-3 each
    assert np.abs(got['layers'][0]['kernel'] - gnn['layers'][0]['kernel']).max() < 2e-5
    gh = helpers.basic_head_to_oracle(model.rs)
    assert np.abs(gh['clf'][-1][0] - head['clf'][-1][0]).max() < 2e-5


def test_fit_learns_a_separable_task(hip):
    """fit() over a Sequence: the loss falls and training accuracy rises on labels a GCN can represent."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.experiment import Adam
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraph
    engine.set_seed(11)
    g = helpers.tiny_graph(n_users=100, n_items=80, n_ratings=4000, seed=5)
    ratings = g['ratings']
    model = basic.BasicGCN(g['adj'], **dict(CFG, l2_regularizer=1e-6))
    model.compile(loss='binary_crossentropy', optimizer=Adam(learning_rate=0.01), metrics=['accuracy'])
    seq = UserItemGraph(ratings, g['users'], g['items'], g['adj'], batch_size=512, shuffle=True)
    before = model.evaluate(seq)
    hist = model.fit(seq, epochs=12, verbose=False)
    after = model.evaluate(seq)
    assert hist['loss'][-1] < hist['loss'][0] - 0.02
    assert after[0] < before[0] and after[1] > max(before[1], 0.6)


@pytest.mark.parametrize('cls', ['BasicGCN', 'BasicGraphSage', 'BasicGAT', 'BasicLightGCN', 'BasicDGCF'])
def test_graph_replayed_batches_equal_eager_batches(hip, cls):
    """train_batch_graphed (hipGraph replay incl. the device-side Adam counter) == train_batch, step for step."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import basic
    g = helpers.tiny_graph(n_users=80, n_items=60, n_ratings=1500, seed=3)
    rng = np.random.default_rng(4)
    batches = [(g['u_ids'][k * 64:(k + 1) * 64], g['i_ids'][k * 64:(k + 1) * 64], rng.integers(0, 2, 64)) for k in range(4)]
    models = []
    for _ in range(2):
        engine.set_seed(8)
        m = getattr(basic, cls)(g['adj'], **CFG)
        helpers.randomize_biases(m, seed=1)
        models.append(m)
    eager, graphed = training.Trainer(models[0]), training.Trainer(models[1])
    loss_eager = 0.0
    for epoch in range(3):
        for u, i, y in batches:
            loss_eager += eager.train_batch(u, i, y) * len(y)
            graphed.train_batch_graphed(u, i, y)
    assert graphed._g is not None and graphed.t == eager.t == 12
    assert abs(graphed.pop_loss_sum() - loss_eager) < 1e-3 * abs(loss_eager)
    for pa, pb in zip(models[0].parameters(), models[1].parameters()):
        # identical kernels, identical order; only the float atomics of the embedding scatter may differ in the last bits
        assert torch.allclose(pa, pb, rtol=1e-4, atol=1e-6), tuple(pa.shape)


@pytest.mark.parametrize('kind', ['BasicRS', 'HybridCBRS', 'HybridCBRS-attention'])
def test_head_only_training(hip, kind):
    """basic-kge.yaml / hybrid-kge.yaml: BasicRS / HybridCBRS train on pre-computed embedding rows (datasets.py:43-77).
    Gradients against the autograd oracle (the embeddings enter as a zero-layer 'GNN'), then fit() on a separable task."""
    import types
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import basic, hybrid
    engine.set_seed(4)
    rng = np.random.default_rng(6)
    n, d, b = 90, 32, 200
    table = rng.standard_normal((n, d)).astype(np.float32) * 0.5
    bert = rng.standard_normal((n, 24)).astype(np.float32) * 0.5
    u, i = rng.integers(0, 50, b), rng.integers(50, 90, b)
    y = rng.integers(0, 2, b)
    if kind == 'BasicRS':
        model = basic.BasicRS(dense_units=[24, 16], clf_units=[16])
        blocks = (table[u], table[i])
    else:
        model = hybrid.HybridCBRS(feature_based=True, dense_units=[[24, 16], [24, 16], [16, 16]], clf_units=[16],
                                  fusion_method='attention' if kind.endswith('attention') else 'concatenate')
        blocks = (table[u], table[i], bert[u], bert[i])
    model(blocks)                                             # builds the weights
    helpers.randomize_biases(model, seed=8)
    trainer = training.HeadTrainer(model)
    loss, grads = trainer.loss_and_grads(blocks, y)
    head = helpers.basic_head_to_oracle(model) if kind == 'BasicRS' else helpers.hybrid_head_to_oracle(model)
    gnn = {'kind': 'lightgcn', 'embeddings': table, 'layers': []}            # E = the table itself
    adj = sparse.coo_matrix((n, n), dtype=np.float32)
    want_loss, want, _ = otrain.torch_model_grads(adj, gnn, head, u, i, y, bert=(bert[u], bert[i]) if kind != 'BasicRS' else None)
    assert abs(loss - want_loss) < 1e-5
    flat = {}
    for name, val in want['head'].items():
        if name.startswith('fuse'):
            for key, arr in val.items():
                flat[getattr(getattr(model, name), key)] = arr
        else:
            for layer, (gw, gb) in zip(getattr(model, name).layers, val):
                flat[layer.kernel], flat[layer.bias] = gw, gb
    assert set(flat) == set(grads)
    for prm, gw in flat.items():
        got = grads[prm].cpu().numpy().reshape(gw.shape).astype(np.float64)
        assert np.abs(got - gw).max() <= 2e-4 * np.abs(gw).max() + 1e-10, tuple(prm.shape)
    # fit() through the Keras protocol: labels readable from the rows
    yy = ((table[u, 0] + table[i, 1]) > 0).astype(np.float32)

    class Seq:
        def __len__(self):
            return 2

        def __getitem__(self, k):
            s = slice(k * 100, (k + 1) * 100)
            return tuple(blk[s] for blk in blocks), yy[s]
    model.compile(optimizer=types.SimpleNamespace(learning_rate=5e-3, beta_1=0.9))
    hist = model.fit(Seq(), epochs=40, verbose=False)['loss']
    assert hist[-1] < 0.6 * hist[0], hist[::8]


def test_randomised_gradient_sweep(hip):
    """Tiny random graphs (isolated nodes, batches of 1 ... 40 pairs, repeated pairs) through every trainable Basic family:
    loss and gradients against the autograd oracle."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.data.preprocess import build_adjacency_matrix
    from deep_cbrs_amar_renaissance_amd.models import basic
    rng = np.random.default_rng(31 + helpers.seed_offset())
    kinds = ['BasicGCN', 'BasicGraphSage', 'BasicGAT', 'BasicLightGCN', 'BasicDGCF']
    for case in range(15):
        engine.set_seed(case)
        n_users, n_items = int(rng.choice([2, 7, 40])), int(rng.choice([2, 9, 30]))
        n_r = int(rng.integers(2, 5 * (n_users + n_items)))
        ratings = np.unique(np.stack([rng.integers(0, n_users, n_r), rng.integers(0, n_items, n_r) + n_users,
                                      (rng.random(n_r) < 0.6).astype(np.int64)], 1), axis=0)
        adj = build_adjacency_matrix(ratings, np.arange(n_users), np.arange(n_items))
        cls = kinds[case % len(kinds)]
        model = getattr(basic, cls)(adj, **CFG)
        helpers.randomize_biases(model, seed=case)
        b = int(rng.choice([1, 2, 17, 40]))
        pick = rng.integers(0, len(ratings), b)                       # with repetitions
        u, i, y = ratings[pick, 0], ratings[pick, 1], ratings[pick, 2]
        trainer = training.Trainer(model)
        loss, grads = trainer.loss_and_grads(u, i, y)
        want_loss, want, _ = otrain.torch_model_grads(adj, helpers.gnn_to_oracle(model.gnn), helpers.basic_head_to_oracle(model.rs),
                                                      u, i, y, l2=1e-4)
        assert abs(loss - want_loss) < 1e-5, (case, cls)
        flat = _flatten_oracle_grads(model, want)
        assert set(flat) == set(grads)
        for prm, gw in flat.items():
            got = grads[prm].cpu().numpy().reshape(gw.shape).astype(np.float64)
            got += 2 * trainer._l2(prm) * prm.detach().cpu().numpy().reshape(gw.shape)
            # (absolute floor: a bias gradient is a sum of +-0.5/B terms that may cancel to 1e-4 of their size — seen with
            # AMAR_TEST_SEED_OFFSET=3: 9.5e-8 off on a value of 2.6e-4, i.e. fp32 rounding of the un-cancelled terms)
            assert np.abs(got - gw).max() <= 3e-4 * np.abs(gw).max() + 3e-7, (case, cls, tuple(prm.shape))


def test_dense_stack_pair_equals_two_launches(hip):
    """amar_dense_stack_pair_f32 / amar_dense_stack_bwd_pair_f32 (round 4: the user and the item tower of a training batch in ONE launch each way)
    against the two separate launches: every output, gradient and deferred partial sum bit for bit — stacks of different depth, width,
    row count and input form (gathered rows / plain rows, a concat slice as the last output)."""
    rng = np.random.default_rng(11)

    def stack(dims, acts, M, gather):
        table = _t(rng.standard_normal((500, dims[0])).astype(np.float32))
        ids = _t(rng.integers(0, 500, M).astype(np.int32)) if gather else None
        x = table if gather else _t(rng.standard_normal((M, dims[0])).astype(np.float32))
        ws = [_t((rng.standard_normal((dims[l], dims[l + 1])) * 0.3).astype(np.float32)) for l in range(len(dims) - 1)]
        bs = [_t((rng.standard_normal(dims[l + 1]) * 0.1).astype(np.float32)) for l in range(len(dims) - 1)]
        return dict(x=x, ids=ids, ws=ws, bs=bs, acts=acts, dims=dims, M=M)

    def fwd_spec(s, cat, lo):
        outs = [torch.empty((s['M'], n), device=DEV) for n in s['dims'][1:]]
        outs[-1] = cat[:s['M'], lo:lo + s['dims'][-1]]
        xin = torch.empty((s['M'], s['dims'][0]), device=DEV) if s['ids'] is not None else None
        return dict(X=s['x'], weights=s['ws'], biases=s['bs'], acts=s['acts'], outs=outs, ids=s['ids'], xcopy=xin)

    for (d0, a0, m0, g0), (d1, a1, m1, g1) in ((([24, 24, 24], ['relu', 'relu'], 1024, True), ([24, 24, 24], ['relu', 'relu'], 1024, True)),
                                               (([16, 48, 48], ['relu', 'relu'], 300, True), ([96, 64, 32, 48], ['relu', 'relu', None], 85, False))):
        s0, s1 = stack(d0, a0, m0, g0), stack(d1, a1, m1, g1)
        width = d0[-1] + d1[-1]
        cat_a, cat_b = torch.zeros((max(m0, m1), width), device=DEV), torch.zeros((max(m0, m1), width), device=DEV)
        fa0, fa1 = fwd_spec(s0, cat_a, 0), fwd_spec(s1, cat_a, d0[-1])
        hip.dense_stack(**fa0)
        hip.dense_stack(**fa1)
        fb0, fb1 = fwd_spec(s0, cat_b, 0), fwd_spec(s1, cat_b, d0[-1])
        hip.dense_stack_pair(fb0, fb1)
        assert torch.equal(cat_a, cat_b)
        for ya, yb in zip(fa0['outs'][:-1] + fa1['outs'][:-1], fb0['outs'][:-1] + fb1['outs'][:-1]):
            assert torch.equal(ya, yb)
        if fa0['xcopy'] is not None:
            assert torch.equal(fa0['xcopy'], fb0['xcopy'])
        # reverse passes: from a gradient of the concatenation down to the stacks' inputs
        dcat = _t(rng.standard_normal((max(m0, m1), width)).astype(np.float32))

        def bwd_spec(s, f, lo, defer):
            inputs = [f['xcopy'] if s['ids'] is not None else s['x']] + f['outs'][:-1]
            return dict(dYtop=dcat[:s['M'], lo:lo + s['dims'][-1]], Ytop=f['outs'][-1], inputs=inputs, weights=s['ws'], acts=s['acts'],
                        workspace=hip.dense_stack_bwd_workspace(s['M'], s['dims'], DEV), dWs=[torch.empty_like(w) for w in s['ws']],
                        dbs=[torch.empty_like(b) for b in s['bs']], dX0=torch.empty((s['M'], s['dims'][0]), device=DEV), defer=defer)
        for defer in (False, True):
            ba0, ba1 = bwd_spec(s0, fa0, 0, defer), bwd_spec(s1, fa1, d0[-1], defer)
            la0, la1 = hip.dense_stack_bwd(**ba0), hip.dense_stack_bwd(**ba1)
            bb0, bb1 = bwd_spec(s0, fa0, 0, defer), bwd_spec(s1, fa1, d0[-1], defer)
            lb0, lb1 = hip.dense_stack_bwd_pair(bb0, bb1)
            assert torch.equal(ba0['dX0'], bb0['dX0']) and torch.equal(ba1['dX0'], bb1['dX0'])
            if defer:
                for la, lb in ((la0, lb0), (la1, lb1)):
                    for (wa, ba_), (wb, bb_) in zip(la, lb):
                        assert wa.groups == wb.groups and torch.equal(wa.partials, wb.partials) and torch.equal(ba_.partials, bb_.partials)
            else:
                assert lb0 is None and lb1 is None
                for a, b in ((ba0, bb0), (ba1, bb1)):
                    for x, y in zip(a['dWs'] + a['dbs'], b['dWs'] + b['dbs']):
                        assert torch.equal(x, y)


@pytest.mark.parametrize('M,K,N', [(1024, 768, 256), (1000, 256, 64), (85, 768, 256), (3000, 132, 96), (1024, 768, 100)])
def test_wgrad_wide_layers_on_the_matrix_instruction(hip, M, K, N):
    """amar_wgrad_f32 for batch-sized wide layers (round 4: one workgroup per 32 x 32 tile of dW walks all rows on the f32 matrix instruction, no
    partial sums): dW = X^T . dZ and db against float64, strided operands, called twice with identical bits, with and without db."""
    rng = np.random.default_rng(M + K + N)
    xw = rng.standard_normal((M, K + 8)).astype(np.float32)
    dzw = rng.standard_normal((M, N + 4)).astype(np.float32)
    x_d, dz_d = _t(xw)[:, 4:4 + K], _t(dzw)[:, 4:4 + N]
    want_dw = xw[:, 4:4 + K].astype(np.float64).T @ dzw[:, 4:4 + N].astype(np.float64)
    dw, db = torch.empty((K, N), device=DEV), torch.empty(N, device=DEV)
    hip.wgrad(x_d, dz_d, dw, db)
    assert helpers.rel_err(dw.cpu().numpy(), want_dw) < 3e-6
    assert helpers.rel_err(db.cpu().numpy(), dzw[:, 4:4 + N].astype(np.float64).sum(0)) < 3e-6
    dw2, db2 = torch.empty((K, N), device=DEV), torch.empty(N, device=DEV)
    hip.wgrad(x_d, dz_d, dw2, db2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    dw3 = torch.empty((K, N), device=DEV)
    hip.wgrad(x_d, dz_d, dw3, None)
    assert torch.equal(dw, dw3)


def test_hybrid_fit_registers_the_bert_table_of_the_reference_sequence(hip, monkeypatch):
    """fit() on the reference's own hybrid Sequence (ids + the BERT rows of the batch, gathered on the host from one table) registers that
    table on the device once and reads the batches as ids only (round 4): the weights after two epochs equal those of the batches taken
    as they come (AMAR_RESIDENT_BERT=0) — the same rows either way."""
    import types
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraphEmbeddings
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    g = helpers.tiny_graph(n_users=60, n_items=50, n_ratings=1200, seed=1)
    rng = np.random.default_rng(0)
    table = rng.standard_normal((110, 24)).astype(np.float32)
    ratings = np.stack([g['u_ids'], g['i_ids'], rng.integers(0, 2, len(g['u_ids']))], axis=1).astype(np.int64)
    users, items = np.arange(60), np.arange(60, 110)

    def run(resident):
        monkeypatch.setenv('AMAR_RESIDENT_BERT', '1' if resident else '0')
        engine.set_seed(2)
        model = hybrid.HybridBertGCN(g['adj'], embedding_dim=8, n_hiddens=[8, 8], dense_units=[[16], [16], [16]], clf_units=[16],
                                     feature_based=True, l2_regularizer=1e-5)
        model.compile(optimizer=types.SimpleNamespace(learning_rate=5e-3, beta_1=0.9))
        seq = UserItemGraphEmbeddings(ratings, users, items, g['adj'], table, batch_size=256, shuffle=True)
        hist = model.fit(seq, epochs=2, verbose=False)['loss']
        assert (getattr(model, 'bert_table', None) is not None) == resident
        return hist, [p.detach().clone() for p in model.parameters()]
    h0, w0 = run(False)
    h1, w1 = run(True)
    assert len(w0) == len(w1) and all(torch.allclose(a, b, rtol=0, atol=1e-7) for a, b in zip(w0, w1))
    assert np.allclose(h0, h1, rtol=1e-6)


def test_hybrid_predict_reads_the_reference_sequence_as_ids(hip, monkeypatch):
    """predict() on the reference's hybrid Sequence: ids against the table registered on the device (towers once per entity in the hoisted
    pass) give the scores of the batches taken as they come (BERT rows gathered on the host, towers per pair) to fp32 rounding."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraphEmbeddings
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    g = helpers.tiny_graph(n_users=60, n_items=50, n_ratings=1200, seed=1)
    rng = np.random.default_rng(0)
    table = rng.standard_normal((110, 24)).astype(np.float32)
    ratings = np.stack([g['u_ids'], g['i_ids'], rng.integers(0, 2, len(g['u_ids']))], axis=1).astype(np.int64)
    users, items = np.arange(60), np.arange(60, 110)
    engine.set_seed(3)
    model = hybrid.HybridBertGCN(g['adj'], embedding_dim=8, n_hiddens=[8, 8], dense_units=[[16], [16], [16]], clf_units=[16],
                                 feature_based=False, l2_regularizer=1e-5)
    seq = UserItemGraphEmbeddings(ratings, users, items, g['adj'], table, batch_size=128, shuffle=False)
    monkeypatch.setenv('AMAR_RESIDENT_BERT', '0')
    plain = model.predict(seq)
    assert getattr(model, 'bert_table', None) is None
    helpers.randomize_biases(model, seed=2)
    plain = model.predict(seq)
    monkeypatch.setenv('AMAR_RESIDENT_BERT', '1')
    resident = model.predict(seq)
    assert model.bert_table is not None and resident.shape == plain.shape == (len(ratings), 1)
    assert float(np.abs(resident - plain).max()) < 2e-6
    assert float(np.abs(model.predict(seq, hoist=False) - plain).max()) < 2e-6


@pytest.mark.parametrize('kind', ['basic', 'hybrid'])
def test_head_only_models_train_from_resident_tables(hip, monkeypatch, kind):
    """BasicRS / HybridCBRS (the reference's baselines on pre-computed embedding rows): fit() on the reference's Sequence classes keeps
    their table(s) on the device and replays batches of ids from a hipGraph (round 4) — the weights after two epochs equal those of the
    batches taken as they come (rows gathered on the host, eager), and the loss falls."""
    import types
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.data.datasets import HybridUserItemEmbeddings, UserItemEmbeddings
    from deep_cbrs_amar_renaissance_amd.models import basic, hybrid
    rng = np.random.default_rng(5)
    n_users, n_items, n = 70, 50, 120
    graph_table = rng.standard_normal((n, 16)).astype(np.float32)
    bert_table = rng.standard_normal((n, 40)).astype(np.float32)
    u = rng.integers(0, n_users, 900)
    i = rng.integers(n_users, n, 900)
    y = ((graph_table[u, 0] + graph_table[i, 1]) > 0).astype(np.int64)
    ratings = np.stack([u, i, y], axis=1).astype(np.int64)
    users, items = np.arange(n_users), np.arange(n_users, n)

    def run(resident):
        monkeypatch.setenv('AMAR_RESIDENT_ROWS', '1' if resident else '0')
        engine.set_seed(4)
        if kind == 'basic':
            model = basic.BasicRS(dense_units=[16, 8], clf_units=[8])
            seq = UserItemEmbeddings(ratings, users, items, graph_table, batch_size=128, shuffle=True)
        else:
            model = hybrid.HybridCBRS(dense_units=[[16, 8], [16, 8], [8, 8]], clf_units=[8], feature_based=True)
            seq = HybridUserItemEmbeddings(ratings, users, items, graph_table, bert_table, batch_size=128, shuffle=True)
        model.compile(optimizer=types.SimpleNamespace(learning_rate=5e-3, beta_1=0.9))
        hist = model.fit(seq, epochs=3, verbose=False)['loss']
        assert (getattr(model._trainer, 'tables', None) is not None) == resident
        return hist, [p.detach().clone() for p in model.parameters()]
    h0, w0 = run(False)
    h1, w1 = run(True)
    assert h1[-1] < h1[0]
    # (eager batches reduce their weight-gradient partials in a launch of their own, replayed ones inside the Adam launch, and the two
    # routes pick different Dense kernels: fp32 rounding that 20 Adam steps at lr 5e-3 spread to a few 1e-5)
    # Adam's m / sqrt(v) turns those into up to ~lr per step on weights whose gradient is nearly zero: the weights agree to a few lr)
    assert np.allclose(h0, h1, rtol=5e-4), (h0, h1)
    assert len(w0) == len(w1) and all(torch.allclose(a, b, rtol=0, atol=1e-2) for a, b in zip(w0, w1))
    # one batch, same weights: the gradients of ids against the resident tables equal those of the rows gathered on the host
    monkeypatch.setenv('AMAR_RESIDENT_ROWS', '1')
    engine.set_seed(4)
    if kind == 'basic':
        model = basic.BasicRS(dense_units=[16, 8], clf_units=[8])
        blocks, tables = (graph_table[u[:100]], graph_table[i[:100]]), [graph_table]
    else:
        model = hybrid.HybridCBRS(dense_units=[[16, 8], [16, 8], [8, 8]], clf_units=[8], feature_based=True)
        blocks, tables = (graph_table[u[:100]], graph_table[i[:100]], bert_table[u[:100]], bert_table[i[:100]]), [graph_table, bert_table]
    from deep_cbrs_amar_renaissance_amd import training
    model(blocks)
    tr = training.HeadTrainer(model)
    loss_rows, g_rows = tr.loss_and_grads(blocks, y[:100])
    tr.set_tables(tables)
    with torch.no_grad():
        terms, g_ids = tr._forward_backward(_t(u[:100].astype(np.int32)), _t(i[:100].astype(np.int32)), _t(y[:100].astype(np.float32)))
    assert abs(float(terms.sum()) / 100 - loss_rows) < 1e-6
    for prm in tr.params:
        assert torch.equal(g_rows[prm], g_ids[prm])
