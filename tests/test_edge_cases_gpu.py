"""GPU edge cases of the hot path (pytest -m gpu): empty and one-element inputs, isolated nodes, edgeless graphs,
users with fewer than k scored items — the degenerate shapes a drop-in has to survive."""
import numpy as np
import pytest
import torch
from scipy import sparse

from oracle import models as om
from tests import helpers

pytestmark = pytest.mark.gpu
CFG = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4)


class _Seq:
    def __init__(self, batches):
        self.batches = batches

    def __len__(self):
        return len(self.batches)

    def __getitem__(self, b):
        return self.batches[b]


@pytest.mark.parametrize('cls', ['BasicGCN', 'BasicGraphSage', 'BasicGAT', 'BasicLightGCN'])
def test_isolated_nodes_and_single_pairs(hip, cls):
    """Nodes without any edge (degree 0: only the filter's self loop / the layers' own self loop), batches of one pair,
    ragged last batch, hoisted and faithful predict."""
    from deep_cbrs_amar_renaissance_amd.models import basic
    n_users, n_items = 37, 29
    rng = np.random.default_rng(1)
    u = rng.integers(0, n_users - 5, 150)                      # the last 5 users and 4 items never appear: isolated nodes
    i = rng.integers(0, n_items - 4, 150) + n_users
    n = n_users + n_items
    adj = sparse.coo_matrix((np.ones(300, dtype=np.float32), (np.concatenate([u, i]), np.concatenate([i, u]))), shape=(n, n))
    model = getattr(basic, cls)(adj, **CFG)
    helpers.randomize_biases(model, seed=2)
    pu = np.arange(n_users, dtype=np.int64)
    pi = (np.arange(n_users) % n_items + n_users).astype(np.int64)          # touches the isolated items and users too
    want = om.basic_gnn_scores(adj, helpers.gnn_to_oracle(model.gnn), helpers.basic_head_to_oracle(model.rs), pu, pi, dtype=np.float64)
    seq = _Seq([((pu[:1], pi[:1]), None), ((pu[1:20], pi[1:20]), None), ((pu[20:], pi[20:]), None)])     # 1, 19, 17 pairs
    for hoist in (True, False):
        got = model.predict(seq, hoist=hoist)
        assert got.shape == (n_users, 1) and np.isfinite(got).all()
        assert np.abs(got - want).max() < 1e-5


def test_empty_inputs(hip):
    """No batches, and a graph without any edge (A_hat = I): predict returns [0, 1] / finite scores; top-k of nothing."""
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities import metrics
    n = 12
    adj = sparse.coo_matrix((n, n), dtype=np.float32)
    model = basic.BasicGCN(adj, **CFG)
    helpers.randomize_biases(model, seed=3)
    assert model.predict(_Seq([])).shape == (0, 1)
    pu, pi = np.array([0, 1, 2], dtype=np.int64), np.array([6, 7, 8], dtype=np.int64)
    got = model.predict(_Seq([((pu, pi), None)]))
    want = om.basic_gnn_scores(adj, helpers.gnn_to_oracle(model.gnn), helpers.basic_head_to_oracle(model.rs), pu, pi, dtype=np.float64)
    assert np.abs(got - want).max() < 1e-5
    empty = metrics.top_k_predictions(np.zeros((0, 3)), np.arange(6), np.arange(6), k=5)
    assert len(empty) == 0


def test_topk_with_fewer_than_k_items_and_ties(hip):
    """metrics.py:11-34: users with fewer than k scored items keep what they have; equal scores break on the item id."""
    from deep_cbrs_amar_renaissance_amd.utilities import metrics
    users, items = np.array([10, 20, 30]), np.array([7, 8, 9, 11])
    pred = np.array([[0, 3, 0.5], [0, 4, 0.9], [0, 5, 0.5], [0, 6, 0.1],        # user 10: 4 items, a tie
                     [1, 3, 0.2],                                                # user 20: one item
                     [2, 6, 0.7], [2, 5, 0.7]])                                  # user 30: two tied items
    top = metrics.top_k_predictions(pred, users, items, k=3)
    got = [(int(a), int(b)) for a, b, _ in top.to_numpy().tolist()]
    assert got == [(10, 8), (10, 7), (10, 9), (20, 7), (30, 9), (30, 11)]
    assert np.allclose(top['scores'].to_numpy(), [0.9, 0.5, 0.5, 0.2, 0.7, 0.7], atol=1e-7)       # scores travel as fp32
    ou, oi, _ = om.top_k(pred[:, 0].astype(int), pred[:, 1].astype(int), pred[:, 2], users, items, 3)
    assert [(int(a), int(b)) for a, b in zip(ou, oi)] == got


def test_training_batch_of_one_and_all_negative_labels(hip):
    """fit() shapes the reference can produce: a last batch of a single rating, a batch whose labels are all 0."""
    from deep_cbrs_amar_renaissance_amd import training
    from deep_cbrs_amar_renaissance_amd.models import basic
    g = helpers.tiny_graph(n_users=30, n_items=20, n_ratings=300, seed=5)
    model = basic.BasicGCN(g['adj'], **CFG)
    trainer = training.Trainer(model)
    loss1 = trainer.train_batch(g['u_ids'][:1], g['i_ids'][:1], np.array([1.0]))
    loss0 = trainer.train_batch(g['u_ids'][:50], g['i_ids'][:50], np.zeros(50))
    assert np.isfinite(loss1) and np.isfinite(loss0)
    assert all(torch.isfinite(p).all() for p in model.parameters())


def test_randomised_model_sweep(hip):
    """Small random rating sets — down to one user and one item, or no positive rating at all — through every Basic model
    family against the oracle (scores within 1e-4, as the parity bar asks)."""
    from deep_cbrs_amar_renaissance_amd.data.preprocess import build_adjacency_matrix
    from deep_cbrs_amar_renaissance_amd.models import basic
    rng = np.random.default_rng(77 + helpers.seed_offset())
    kinds = ['BasicGCN', 'BasicGraphSage', 'BasicGAT', 'BasicLightGCN', 'BasicDGCF']
    for case in range(20):
        n_users, n_items = int(rng.choice([1, 2, 5, 33, 70])), int(rng.choice([1, 3, 17, 64]))
        n_r = int(rng.integers(1, 4 * (n_users + n_items)))
        u = rng.integers(0, n_users, n_r)
        i = rng.integers(0, n_items, n_r) + n_users
        lab = (rng.random(n_r) < (0.0 if case == 3 else 0.6)).astype(np.int64)          # case 3: no positive rating -> no edge
        ratings = np.unique(np.stack([u, i, lab], 1), axis=0)
        adj = build_adjacency_matrix(ratings, np.arange(n_users), np.arange(n_items))
        cls = kinds[case % len(kinds)]
        model = getattr(basic, cls)(adj, **CFG)
        helpers.randomize_biases(model, seed=case)
        pu, pi = ratings[:, 0], ratings[:, 1]
        got = model((pu, pi)).cpu().numpy()
        want = om.basic_gnn_scores(adj, helpers.gnn_to_oracle(model.gnn), helpers.basic_head_to_oracle(model.rs), pu, pi, dtype=np.float64)
        assert np.isfinite(got).all() and np.abs(got - want).max() < 1e-4, (case, cls, n_users, n_items, len(ratings))


@pytest.mark.parametrize('F', [8, 16, 32])
def test_lds_tiled_edge_graphs(hip, F):
    """amar_spmm_lt_f32 on degenerate inputs: a graph without any off-diagonal entry (only the diag term survives), a single
    node, one lone edge among many isolated nodes, and a tile count far above the row count."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    dev = 'cuda'

    def run(rows, cols, n, n_cu=256):
        r = torch.tensor(rows, dtype=torch.int64, device=dev)
        c = torch.tensor(cols, dtype=torch.int64, device=dev)
        diag = torch.ones(n, device=dev)
        scale = torch.linspace(0.5, 1.5, n, device=dev)
        lt = lds_tiled.LdsTiled.build(r, c, n, n, F, diag, scale, scale, 0, n_cu=n_cu)
        x = torch.randn((n, F), device=dev)
        y = torch.full((n, F), float('nan'), device=dev)
        hip.spmm_lt(lt, x, y, prescaled=True)
        want = x.clone().double()
        for a, b in zip(rows, cols):
            want[a] += x[b].double()
        want = want * scale.double()[:, None]
        assert torch.isfinite(y).all() and float((y.double() - want).abs().max()) < 1e-5
        return lt

    run([], [], 37)                                   # no entries at all: one padded word, every tile empty
    run([], [], 1)
    lt = run([5, 900], [900, 5], 1000)                # one edge, 998 isolated nodes
    assert lt.n_entries == 2
    run([0, 1, 1, 2], [1, 0, 2, 1], 3, n_cu=64)       # more tiles asked for than rows exist


@pytest.mark.parametrize('C', [8, 16, 32])
@pytest.mark.parametrize('self_loop', [True, False])
def test_gat_lds_tiled_edge_graphs(hip, C, self_loop):
    """amar_gat_lt_f32 on degenerate inputs, against the row kernel: no edge at all (with the self loop every row attends to
    itself; without it every row is empty and leaves as relu(bias)), a single node, one lone edge among isolated nodes, more
    tiles than rows, and a list made of (i, i) edges only (they enter through `diag`)."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR, _unit_entries
    dev = 'cuda'
    rw = lds_tiled.GAT_ROWS_PER_WAVE[C]

    def run(rows, cols, n, n_cu=256):
        order = np.lexsort((cols, rows)) if len(rows) else np.zeros(0, np.int64)
        r, c = np.asarray(rows, np.int64)[order], np.asarray(cols, np.int64)[order]
        rowptr = np.zeros(n + 1, np.int64)
        np.add.at(rowptr, r + 1, 1)
        a = DeviceCSR(torch.from_numpy(np.cumsum(rowptr).astype(np.int32)).to(dev), torch.from_numpy(c.astype(np.int32)).to(dev), None, (n, n))
        er, ec, diag, off = _unit_entries(a, False)
        lt = lds_tiled.LdsTiled.build(er, ec, n, n, C, diag, torch.ones(n, device=dev), None, off, n_cu=n_cu, rw=rw)
        g = torch.Generator(device=dev); g.manual_seed(n + C)
        h = torch.randn((n, C), device=dev, generator=g)
        ss, sn = torch.randn(n, device=dev, generator=g), torch.randn(n, device=dev, generator=g)
        b = torch.randn(C, device=dev, generator=g) * 0.3
        want, got = torch.empty((n, C), device=dev), torch.full((n, C), float('nan'), device=dev)
        hip.gat_layer(a.rowptr, a.colidx, h, ss, sn, b, want, self_loop=self_loop)
        hip.gat_lt(lt, a, h, ss, sn, b, got, self_loop=self_loop)
        assert torch.isfinite(got).all() and float((got - want).abs().max()) < 1e-5
        if not len(rows) and not self_loop:
            assert torch.equal(got, torch.relu(b).expand(n, C))
        return lt

    run([], [], 37)
    run([], [], 1)
    assert run([5, 900], [900, 5], 1000).n_entries == 2
    run([0, 1, 1, 2], [1, 0, 2, 1], 3, n_cu=64)
    assert run([0, 1, 2, 2], [0, 1, 2, 2], 3).n_entries == 0            # self edges only (one of them twice)
