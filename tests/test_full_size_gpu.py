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


def test_full_size_headline_layer_on_the_lds_tiled_walk(hip, big):
    """The HEADLINE kernel at the headline size: `spmm_lt_kernel<8>` on the image `tiled_image(8)` dispatches to at ml1m(s=64) — the
    F = 8 layers of econfigs/basic-gnn.yaml grid1, what bench.py's `roofline` object times — as the plain product and as the fused
    GCN layer (bias, ReLU, concat-slice store, next layer's X.W pre-scaled), against the row-streaming CSR kernel, the eigenvector
    of A_hat, 300 sampled rows in float64, run-to-run bit reproducibility, and the un-fused sequence of the epilogue."""
    a, n = big['a'], big['n']
    dev = a.rowptr.device
    F = 8
    lt = a.tiled_image(F)
    assert hasattr(lt, 'words'), "ml1m(s=64), F = 8 must dispatch to the LDS-tiled image"
    g = torch.Generator(device=dev); g.manual_seed(80)
    x = torch.randn((n, F), device=dev, generator=g)
    y_lt, y_csr, y_again = (torch.empty((n, F), device=dev) for _ in range(3))
    hip.spmm_xs(lt, x, y_lt)
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, x, y_csr)
    assert _rel(y_lt, y_csr) < 2e-6
    hip.spmm_xs(lt, x, y_again)
    assert torch.equal(y_lt, y_again), "the LT walk must be bitwise reproducible run to run"
    v = (1.0 / a.dinv).view(-1, 1).repeat(1, F).contiguous()
    av = torch.empty_like(v)
    hip.spmm_xs(lt, v, av)
    assert _rel(av, v) < 1e-5
    rows = np.random.default_rng(8).integers(0, n, 300)
    rp = a.rowptr.long()
    xd = x.double().cpu().numpy()
    for r in rows.tolist():
        lo, hi = int(rp[r]), int(rp[r + 1])
        cols = a.colidx[lo:hi].long().cpu().numpy()
        vals = a.vals[lo:hi].double().cpu().numpy()
        want = (vals[:, None] * xd[cols]).sum(0)
        assert np.abs(y_lt[r].double().cpu().numpy() - want).max() <= 1e-5 * max(1e-3, np.abs(want).max())
    # the fused GCN layer as the chain makes it: Y = relu(A_hat H + b) into a slice of the [N, 24] concat buffer, H_next = S (Y W)
    b = torch.rand(F, device=dev, generator=g) - 0.5
    w = (torch.rand((F, F), device=dev, generator=g) - 0.5).contiguous()
    h0 = torch.empty_like(x)
    hip.row_affine(x, lt.col_scale, h0)
    cat = torch.zeros((n, 3 * F), device=dev)
    h1 = torch.empty((n, F), device=dev)
    hip.spmm_xs(lt, h0, cat[:, F:2 * F], bias=b, relu=True, Wnext=w, Hnext=h1, prescaled=True, scale_next=True)
    y1 = cat[:, F:2 * F].contiguous()
    assert _rel(y1, torch.clamp(y_lt + b, min=0)) < 1e-6
    assert float(cat[:, :F].abs().max()) == 0.0 and float(cat[:, 2 * F:].abs().max()) == 0.0     # the slice store stays in its columns
    h_ref = torch.empty_like(h1)
    hip.rowwise_xw(y1, w, h_ref, row_scale=lt.row_scale)
    assert _rel(h1, h_ref) < 1e-6


@pytest.mark.parametrize('F', [16, 32])
def test_full_size_wide_layers_on_the_lds_tiled_walk(hip, big, F):
    """The F = 16 / 32 layers of econfigs/basic-gnn.yaml grid2 / grid3 at the bench's full size, on the image they dispatch to (the
    LDS-tiled walk without implicit pairs, round 3): against the row-streaming CSR kernel, the eigenvector of A_hat, sampled rows
    in float64, and the fused layer epilogue (bias, ReLU, next layer's kernel staged in LDS, pre-scaled next table) against the
    un-fused sequence."""
    a, n = big['a'], big['n']
    dev = a.rowptr.device
    lt = a.tiled_image(F)
    assert hasattr(lt, 'words') and not lt.pairs and lt.n_pairs == 0
    g = torch.Generator(device=dev); g.manual_seed(F)
    x = torch.randn((n, F), device=dev, generator=g)
    y_lt, y_csr = torch.empty((n, F), device=dev), torch.empty((n, F), device=dev)
    hip.spmm_xs(lt, x, y_lt)
    hip.spmm_csr(a.rowptr, a.colidx, a.vals, x, y_csr)
    assert _rel(y_lt, y_csr) < 2e-6
    v = (1.0 / a.dinv).view(-1, 1).repeat(1, F).contiguous()
    av = torch.empty_like(v)
    hip.spmm_xs(lt, v, av)
    assert _rel(av, v) < 1e-5
    rows = np.random.default_rng(F).integers(0, n, 100)
    rp = a.rowptr.long()
    xd = x.double().cpu().numpy()
    for r in rows.tolist():
        lo, hi = int(rp[r]), int(rp[r + 1])
        cols = a.colidx[lo:hi].long().cpu().numpy()
        vals = a.vals[lo:hi].double().cpu().numpy()
        want = (vals[:, None] * xd[cols]).sum(0)
        assert np.abs(y_lt[r].double().cpu().numpy() - want).max() <= 1e-5 * max(1e-3, np.abs(want).max())
    # fused GCN layer: Y = relu(A_hat H + b), H_next = S (Y W) — against relu / bias / X.W / row scale applied one by one
    b = torch.rand(F, device=dev, generator=g) - 0.5
    w = (torch.rand((F, F), device=dev, generator=g) - 0.5).contiguous()
    h0 = torch.empty_like(x)
    hip.row_affine(x, lt.col_scale, h0)
    y1, h1 = torch.empty((n, F), device=dev), torch.empty((n, F), device=dev)
    hip.spmm_xs(lt, h0, y1, bias=b, relu=True, Wnext=w, Hnext=h1, prescaled=True, scale_next=True)
    ref = torch.clamp(y_lt + b, min=0)
    assert _rel(y1, ref) < 1e-6
    h_ref = torch.empty_like(h1)
    hip.rowwise_xw(y1, w, h_ref, row_scale=lt.row_scale)
    assert _rel(h1, h_ref) < 1e-6


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


def _edge_csr(big):
    """The raw edge list GraphSAGE / GAT take at full size: A_hat's pattern without its diagonal, no values."""
    from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR
    a, n = big['a'], big['n']
    dev = a.rowptr.device
    rp = a.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
    keep = rows != a.colidx.long()
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(rows[keep], minlength=n), 0)
    return DeviceCSR(rowptr.to(torch.int32), a.colidx[keep].contiguous(), None, (n, n))


def test_full_size_gat_and_sage_on_the_lds_tiled_walk(hip, big, monkeypatch):
    """ml1m(s=64): the layers' default route (LDS-tiled walk: amar_gat_lt_f32, fused GraphSAGE tail; node-type boundary inferred
    from the entries) against the row-kernel route, 200 sampled rows of the GAT layer against float64 host arithmetic, and the
    softmax invariance the GAT form rests on: shifting every s_neigh by a constant changes the bound, not the result."""
    from deep_cbrs_amar_renaissance_amd.layers.gat_conv import GATConv
    from deep_cbrs_amar_renaissance_amd.layers.graphsage_conv import GraphSageConv
    from deep_cbrs_amar_renaissance_amd.utilities.lds_tiled import LdsTiled
    from tests import helpers
    e, n = _edge_csr(big), big['n']
    dev = e.rowptr.device
    g = torch.Generator(device=dev); g.manual_seed(11)
    x = torch.randn((n, 8), device=dev, generator=g)
    calls = []
    for name in ('gat_lt', 'spmm_lt'):
        monkeypatch.setattr(hip, name, lambda *a, _f=getattr(hip, name), _n=name, **k: (calls.append(_n), _f(*a, **k))[1])
    gat = GATConv(8, dropout_rate=0.0, activation='relu')
    gat.build([(n, 8), None])
    sage = GraphSageConv(8, activation='relu')
    sage.build([(n, 8), None])
    helpers.randomize_biases(gat, seed=3)
    helpers.randomize_biases(sage, seed=4)
    y_gat, y_sage = gat([x, e]), sage([x, e])
    assert 'gat_lt' in calls and 'spmm_lt' in calls
    assert e.row_breaks == (big['n_users'],) and isinstance(e.tiled_gat_image(8), LdsTiled)
    monkeypatch.setenv('AMAR_SPMM_KIND', 'csr')
    r_gat, r_sage = gat([x, e]), sage([x, e])
    monkeypatch.delenv('AMAR_SPMM_KIND')
    assert float((y_gat - r_gat).abs().max()) < 2e-5 and float((y_sage - r_sage).abs().max()) < 2e-5
    # sampled GAT rows in float64 on the host
    c = 8
    w = gat.kernel.detach().view(8, c).double().cpu().numpy()
    a_s, a_n = gat.attn_kernel_self.detach().view(c).double().cpu().numpy(), gat.attn_kernel_neighs.detach().view(c).double().cpu().numpy()
    b = gat.bias.detach().double().cpu().numpy()
    xd = x.double().cpu().numpy()
    rp = e.rowptr.long()
    for r in np.random.default_rng(2).integers(0, n, 200).tolist():
        cols = np.concatenate([e.colidx[int(rp[r]):int(rp[r + 1])].long().cpu().numpy(), [r]])
        h = xd[cols] @ w
        z = (xd[r] @ w) @ a_s + h @ a_n
        z = np.where(z > 0, z, 0.2 * z)
        p = np.exp(z - z.max())
        want = np.maximum((p[:, None] * h).sum(0) / (p.sum() + 1e-9) + b, 0)
        assert np.abs(y_gat[r].double().cpu().numpy() - want).max() < 1e-5 * max(1.0, np.abs(want).max())
    # invariance: the same layer with s_neigh shifted by +5 (another bound M_i), straight through the C-ABI
    h = torch.empty((n, c), device=dev)
    ss, sn = torch.empty(n, device=dev), torch.empty(n, device=dev)
    hip.rowwise_xw(x, gat.kernel.view(-1, c), h, a_self=gat.attn_kernel_self.view(c), a_neigh=gat.attn_kernel_neighs.view(c), s_self=ss, s_neigh=sn)
    lt = e.tiled_gat_image(c)
    y0, y1 = torch.empty((n, c), device=dev), torch.empty((n, c), device=dev)
    hip.gat_lt(lt, e, h, ss, sn, gat.bias, y0)
    assert torch.equal(y0, y_gat)
    bound = lt._gat_bound.clone() + 5.0
    code = hip.load().amar_gat_lt_f32(
        lt.words.data_ptr(), lt.stream_start.data_ptr(), lt.wsteps.data_ptr(), lt.tile_row0.data_ptr(), lt.n_win.data_ptr(),
        lt.vstart.data_ptr(), lt.vcount.data_ptr(), lt.n_tiles, lt.maxwin1, lt.pace_every, lt.rw, lt.diag.data_ptr(), e.rowptr.data_ptr(),
        e.colidx.data_ptr(), h.data_ptr(), c, c, ss.data_ptr(), sn.data_ptr(), bound.data_ptr(), gat.bias.data_ptr(),
        y1.data_ptr(), c, 1, n, n, 0, None)
    torch.cuda.synchronize()
    assert code == 0 and float((y1 - y0).abs().max()) < 2e-6


def test_full_size_hybrid_head_properties(hip, big):
    """The hybrid head at the bench's size (12.1 M pairs, 768-d content rows): scores on the prepared pair list (XCD-affine order, two-step
    way back) equal the direct call bit for bit; any permutation of the pair list permutes the scores; a sample agrees with a float64
    evaluation of the pair stage from the same tower tables within 1e-6 (the north star's bar is 1e-4)."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.models import hybrid, basic
    engine.set_seed(43)
    dev = big['a'].rowptr.device
    nu, ni, n = big['n_users'], big['n_items'], big['n']
    model = hybrid.HybridBertGCN(big['a'], embedding_dim=8, n_hiddens=[8, 8], dense_units=[[24, 24], [256, 64], [64, 64]], clf_units=[64, 64],
                                 feature_based=True)
    g = torch.Generator(device=dev); g.manual_seed(44)
    bert = torch.randn((n, 768), device=dev, generator=g) * 0.5
    model.rs.build_head(model.gnn.output_dim(), 768)
    test = big['test']
    p = int(test.shape[0])
    perm = torch.randperm(p, device=dev, generator=g)
    u = test[perm, 0].to(torch.int32).contiguous()
    i = test[perm, 1].to(torch.int32).contiguous()
    emb = model.gnn(None)
    rs = model.rs
    tw = rs.towers(emb[:nu], emb[nu:], bert[:nu], bert[nu:])
    assert tw[4]
    direct = rs.score_towers(tw, u, i, 0, nu)
    assert direct.shape == (p, 1) and bool(torch.isfinite(direct).all()) and float(direct.min()) >= 0 and float(direct.max()) <= 1
    plan = basic.PairPlan(u, i)
    assert plan.mid_index is not None                                   # a list this long returns through the window streams
    assert torch.equal(rs.score_towers(tw, u, i, 0, nu, pair_plan=plan), direct)
    perm2 = torch.randperm(p, device=dev, generator=g)
    assert torch.equal(rs.score_towers(tw, u[perm2].contiguous(), i[perm2].contiguous(), 0, nu), direct[perm2])
    # float64 pair stage on a sample
    sel = perm2[:20000]
    tug, tig, tub, tib = [t.double() for t in tw[:4]]
    ul, il = u[sel].long(), (i[sel] - nu).long()
    x1, x2 = torch.relu(tug[ul] + tig[il]), torch.relu(tub[ul] + tib[il])
    kb = lambda l: (l.kernel.detach().double(), l.bias.detach().double())
    for l in list(rs.dense3a.layers)[1:]:
        k, b = kb(l); x1 = torch.relu(x1 @ k + b)
    for l in list(rs.dense3b.layers)[1:]:
        k, b = kb(l); x2 = torch.relu(x2 @ k + b)
    x = torch.cat([x1, x2], 1)
    layers = list(rs.clf.layers)
    for l in layers[:-1]:
        k, b = kb(l); x = torch.relu(x @ k + b)
    k, b = kb(layers[-1])
    ref = torch.sigmoid(x @ k + b)
    assert float((direct[sel].double() - ref).abs().max()) < 1e-6
