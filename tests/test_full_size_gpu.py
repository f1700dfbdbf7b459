"""GPU: the bench's full-size workload, ml1m(s=64) — 590 592 nodes, 55.9 M non-zeros, 12.1 M pairs — checked through
size-independent properties (the oracle cannot run this size in seconds) plus oracle arithmetic on sampled rows.

  * A_hat = D^-1/2 (A + I) D^-1/2 has sqrt(deg) as an eigenvector with eigenvalue 1 (a known-answer vector at any size);
  * linearity and symmetry of the product;
  * the three kernels (row-streaming CSR, XCD-sliced with values, XCD-sliced value-free) agree;
  * sampled output rows equal the float64 row sums computed on the host from the CSR arrays;
  * scores do not depend on the order of the pair list; the fused (hoisted) and the per-batch (faithful) call agree;
  * top-k is idempotent.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GRID1 = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4)


@pytest.fixture(scope='module')
def big(hip):
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(64, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    return {'a': a, 'n': n, 'n_users': data['n_users'], 'n_items': data['n_items'], 'test': data['test']}


def _rel(got, want):
    return float((got.double() - want.double()).abs().max() / want.double().abs().max())


def test_full_size_spmm_properties(hip, big, monkeypatch):
    from deep_cbrs_amar_renaissance_amd.utilities.math import XcdSliced
    a, n = big['a'], big['n']
    assert n == 590592 and a.nnz > 5.5e7
    dev = a.rowptr.device
    g = torch.Generator(device=dev); g.manual_seed(5)
    x = torch.randn((n, 8), device=dev, generator=g)
    y = torch.randn((n, 8), device=dev, generator=g)
    xs = a.xcd_sliced()
    assert xs.vals is None and xs.row_scale is not None
    monkeypatch.setenv('AMAR_XS_VALUES', '1')
    xs_valued = XcdSliced.from_csr(a)
    monkeypatch.delenv('AMAR_XS_VALUES')
    ax_csr, ax_xs, ax_xsv = (torch.empty((n, 8), device=dev) for _ in range(3))
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, x, ax_csr)
    hip.spmm_xs(xs, x, ax_xs)
    hip.spmm_xs(xs_valued, x, ax_xsv)
    assert _rel(ax_xs, ax_csr) < 2e-6 and _rel(ax_xsv, ax_csr) < 2e-6
    # eigenvector: A_hat sqrt(deg) = sqrt(deg), deg = row sums of A + I = 1 / dinv^2
    v = (1.0 / a.dinv).view(-1, 1).repeat(1, 8).contiguous()
    av = torch.empty_like(v)
    hip.spmm_xs(xs, v, av)
    assert _rel(av, v) < 1e-5
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, v, av)
    assert _rel(av, v) < 1e-5
    # linearity and symmetry
    ay, comb = torch.empty_like(x), torch.empty_like(x)
    hip.spmm_xs(xs, y, ay)
    hip.spmm_xs(xs, (2 * x - 3 * y).contiguous(), comb)
    assert _rel(comb, 2 * ax_xs - 3 * ay) < 1e-5
    lhs, rhs = float((x.double() * ay.double()).sum()), float((y.double() * ax_xs.double()).sum())
    assert abs(lhs - rhs) < 1e-6 * max(abs(lhs), abs(rhs), 1.0)
    # sampled rows against float64 host arithmetic on the CSR arrays
    rows = torch.from_numpy(np.random.default_rng(1).integers(0, n, 300)).to(dev)
    rp = a.rowptr.long()
    xd = x.double().cpu().numpy()
    for r in rows.tolist():
        lo, hi = int(rp[r]), int(rp[r + 1])
        cols = a.colidx[lo:hi].long().cpu().numpy()
        vals = a.vals[lo:hi].double().cpu().numpy()
        want = (vals[:, None] * xd[cols]).sum(0)
        assert np.abs(ax_xs[r].double().cpu().numpy() - want).max() <= 1e-5 * max(1e-3, np.abs(want).max())


def test_full_size_scoring_properties(hip, big):
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities import metrics
    engine.set_seed(42)
    model = basic.BasicGCN(big['a'], **GRID1)
    model.n_users, model.n_items = big['n_users'], big['n_items']
    test = big['test']
    p = int(test.shape[0])
    assert p > 1.2e7
    u, i = test[:, 0].to(torch.int32).contiguous(), test[:, 1].to(torch.int32).contiguous()
    nu = big['n_users']
    emb = model.gnn(None)
    towers = model.rs.towers(emb[:nu], emb[nu:nu + big['n_items']])
    scores = model.rs.score_towers(towers, u, i, 0, nu)
    assert scores.shape == (p, 1) and bool(torch.isfinite(scores).all()) and float(scores.min()) >= 0 and float(scores.max()) <= 1
    # order independence: any permutation of the pair list permutes the scores, bit for bit
    gen = torch.Generator(device=u.device); gen.manual_seed(9)
    perm = torch.randperm(p, device=u.device, generator=gen)
    shuffled = model.rs.score_towers(towers, u[perm].contiguous(), i[perm].contiguous(), 0, nu)
    assert torch.equal(shuffled, scores[perm])
    # the per-batch call of the reference (propagation inside, gather per pair) on a sample
    sample = perm[:4096]
    faithful = model((u[sample].contiguous(), i[sample].contiguous()))
    assert torch.equal(faithful, scores[sample])
    # top-k idempotence on a slice of users
    sel = (u < 2000)
    pred = np.stack([u[sel].cpu().numpy(), i[sel].cpu().numpy(), scores[sel, 0].cpu().numpy()], axis=1)
    users, top_items, top_scores = metrics.top_k_arrays(pred[:, 0], pred[:, 1], pred[:, 2], 10)
    valid = top_items >= 0
    again_u = np.repeat(users, 10).reshape(-1, 10)[valid]
    users2, top_items2, top_scores2 = metrics.top_k_arrays(again_u, top_items[valid], top_scores[valid], 10)
    assert np.array_equal(users, users2) and np.array_equal(top_items, top_items2) and np.array_equal(top_scores, top_scores2)
    assert (np.diff(np.where(valid, top_scores, -1.0), axis=1) <= 0).all()             # every list is sorted (scores lie in [0, 1])
