"""Pins oracle/train.py (the training-step oracle) on the CPU: forward == numpy oracle, the manual
reverse pass == autograd, and autograd == central finite differences of the loss for every model kind."""
import numpy as np
import pytest

from oracle import models as om, train as otrain, weights as ow
from tests import helpers


def _case(kind, hybrid=False, seed=0, n_props=0):
    g = helpers.tiny_graph(n_users=14, n_items=11, n_ratings=70, seed=seed, n_props=n_props, n_links=3 * n_props)
    rng = np.random.default_rng(seed + 1)
    n = g['adj'].shape[0]
    gnn = ow.gnn(rng, kind, n, embedding_dim=4, n_hiddens=(4, 4), n_layers=2, bias_range=0.05)
    gnn['embeddings'] = (gnn['embeddings'] * 4).astype(np.float32)          # wider spread than glorot over N: gradients not tiny
    d = ow.gnn_out_dim(gnn)
    b = 24
    u, i = g['u_ids'][:b], g['i_ids'][:b]
    y = (rng.random(b) < 0.5).astype(np.float32)
    if hybrid:
        head = ow.hybrid_head(rng, d, 6, ([6], [5], [6]), [7], bias_range=0.05)
        bert = (rng.normal(size=(b, 6)).astype(np.float32), rng.normal(size=(b, 6)).astype(np.float32))
    else:
        head = ow.basic_head(rng, d, [6, 6], [8], bias_range=0.05)
        bert = None
    return g, gnn, head, u, i, y, bert


@pytest.mark.parametrize('kind', ['gcn', 'lightgcn', 'sage', 'gat'])
@pytest.mark.parametrize('hybrid', [False, True])
def test_autograd_forward_equals_numpy_oracle(kind, hybrid):
    g, gnn, head, u, i, y, bert = _case(kind, hybrid)
    _, _, p = otrain.torch_model_grads(g['adj'], gnn, head, u, i, y, bert=bert)
    if hybrid:
        e = om.propagate(g['adj'], gnn, dtype=np.float64)
        want = om.hybrid_cbrs(e[u], e[i], bert[0].astype(np.float64), bert[1].astype(np.float64), head)[:, 0]
    else:
        want = om.basic_gnn_scores(g['adj'], gnn, head, u, i, dtype=np.float64)[:, 0]
    np.testing.assert_allclose(p, want, rtol=0, atol=1e-12)


@pytest.mark.parametrize('feature_based,fusion,residual', [(True, 'attention', False), (True, 'concatenate', True),
                                                           (False, 'attention', False), (True, 'attention', True)])
def test_tweaked_heads_forward_and_finite_differences(feature_based, fusion, residual):
    """Attention fusion / residual classifier (econfigs/hybrid-gnn-tweaks*.yaml): autograd forward == numpy oracle, and a
    few gradient entries of the fusion / residual weights == central finite differences."""
    g = helpers.tiny_graph(n_users=14, n_items=11, n_ratings=70, seed=0)
    rng = np.random.default_rng(1)
    n = g['adj'].shape[0]
    gnn = ow.gnn(rng, 'gcn', n, embedding_dim=4, n_hiddens=(4, 4), bias_range=0.05)
    head = ow.hybrid_head_tweaked(rng, ow.gnn_out_dim(gnn), 6, ([6], [5], [6]), [7, 6], bias_range=0.05, fusion_method=fusion,
                                  residual=residual, feature_based=feature_based)
    b = 20
    u, i = g['u_ids'][:b], g['i_ids'][:b]
    y = (rng.random(b) < 0.5).astype(np.float32)
    bert = (rng.normal(size=(b, 6)).astype(np.float32), rng.normal(size=(b, 6)).astype(np.float32))
    run = lambda: otrain.torch_model_grads(g['adj'], gnn, head, u, i, y, bert=bert, feature_based=feature_based)
    _, grads, p = run()
    e = om.propagate(g['adj'], gnn, dtype=np.float64)
    want = om.hybrid_cbrs(e[u], e[i], bert[0].astype(np.float64), bert[1].astype(np.float64), head, feature_based=feature_based)[:, 0]
    np.testing.assert_allclose(p, want, rtol=0, atol=1e-12)
    probes = [(head[name], key, grads['head'][name][key]) for name in head if name.startswith('fuse') for key in head[name]]
    for container, key, grad in probes:
        orig = container[key]
        arr = orig.astype(np.float64)
        container[key] = arr
        idx = tuple(int(rng.integers(0, s)) for s in arr.shape)
        keep, h = arr[idx], 1e-5
        arr[idx] = keep + h
        up = run()[0]
        arr[idx] = keep - h
        dn = run()[0]
        container[key] = orig
        fd = (up - dn) / (2 * h)
        assert abs(fd - grad[idx]) <= 1e-6 * max(1.0, abs(fd)) + 1e-8, (key, fd, grad[idx])


@pytest.mark.parametrize('kind', ['gcn', 'lightgcn'])
def test_manual_reverse_pass_equals_autograd(kind):
    g, gnn, head, u, i, y, _ = _case(kind, n_props=5)
    loss_m, gm, _ = otrain.loss_and_grads(g['adj'], gnn, head, u, i, y, l2=1e-3)
    loss_t, gt, _ = otrain.torch_model_grads(g['adj'], gnn, head, u, i, y, l2=1e-3)
    assert abs(loss_m - loss_t) < 1e-12
    np.testing.assert_allclose(gm['gnn']['embeddings'], gt['gnn']['embeddings'], atol=1e-13)
    for a, b in zip(gm['gnn']['layers'], gt['gnn']['layers']):
        for name in a:
            np.testing.assert_allclose(a[name], b[name], atol=1e-13)
    for name in gm['head']:
        for (wa, ba), (wb, bb) in zip(gm['head'][name], gt['head'][name]):
            np.testing.assert_allclose(wa, wb, atol=1e-13)
            np.testing.assert_allclose(ba, bb, atol=1e-13)


@pytest.mark.parametrize('kind,hybrid', [('sage', False), ('gat', False), ('gcn', True), ('sage', True)])
def test_autograd_equals_finite_differences(kind, hybrid):
    g, gnn, head, u, i, y, bert = _case(kind, hybrid, seed=3)
    l2 = 1e-3
    _, grads, _ = otrain.torch_model_grads(g['adj'], gnn, head, u, i, y, l2=l2, bert=bert)
    rng = np.random.default_rng(9)

    def loss_at():
        return otrain.torch_model_grads(g['adj'], gnn, head, u, i, y, l2=l2, bert=bert)[0]

    def check(arr, grad, n_probe=3):
        arr64 = arr.astype(np.float64)
        for _ in range(n_probe):
            idx = tuple(rng.integers(0, s) for s in arr.shape)
            keep = arr64[idx]
            h = 1e-5
            arr64[idx] = keep + h
            up = _with(arr, arr64, loss_at)
            arr64[idx] = keep - h
            dn = _with(arr, arr64, loss_at)
            arr64[idx] = keep
            fd = (up - dn) / (2 * h)
            assert abs(fd - grad[idx]) <= 1e-6 * max(1.0, abs(fd)) + 1e-8, (idx, fd, grad[idx])

    def _with(orig, arr64, fn):
        # torch_model_grads reads np.asarray(..., float64): temporarily swap the container's array for the float64 copy
        return fn_swapped[id(orig)](arr64, fn)

    fn_swapped = {}

    def register(container, key):
        orig = container[key]

        def run(arr64, fn):
            container[key] = arr64
            try:
                return fn()
            finally:
                container[key] = orig
        fn_swapped[id(orig)] = run
        return orig

    check(register(gnn, 'embeddings'), grads['gnn']['embeddings'], 4)
    for lw, gl in zip(gnn['layers'], grads['gnn']['layers']):
        for name in list(lw):
            check(register(lw, name), gl[name], 2)
    for name in head:
        for k, (w, b) in enumerate(head[name]):
            holder = {'w': w, 'b': b}

            def reg(which, k=k, name=name, holder=holder):
                orig = holder[which]

                def run(arr64, fn):
                    pair = list(head[name][k])
                    pair[0 if which == 'w' else 1] = arr64
                    saved = head[name][k]
                    head[name][k] = tuple(pair)
                    try:
                        return fn()
                    finally:
                        head[name][k] = saved
                fn_swapped[id(orig)] = run
                return orig
            check(reg('w'), grads['head'][name][k][0], 1)
            check(reg('b'), grads['head'][name][k][1], 1)
