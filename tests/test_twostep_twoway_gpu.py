"""GPU parity of the TwoStep / TwoWay stacks (tsgnn.py, twgnn.py) against the oracle (pytest -m gpu).

Same bar as the single-graph models: node representations within 1e-5 relative, scores within 1e-4 of the fp64 oracle.
"""
import numpy as np
import pytest
import torch

from oracle import models as om
from tests import helpers

pytestmark = pytest.mark.gpu

CFG = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48],
           l2_regularizer=1e-4, final_node='concatenation', aggregate='mean', dropout_rate=0.0, activation='relu')
KINDS = ['GCN', 'GraphSage', 'GAT', 'LightGCN', 'DGCF']


def _perturb(model, seed):
    """Zero biases and unit DGCF gates hide bugs: draw both at random."""
    helpers.randomize_biases(model, seed=seed)
    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('.w'):
                p.add_(torch.from_numpy(rng.uniform(-1.5, 1.5, tuple(p.shape)).astype(np.float32)).to(p.device))


def _ml1m_kg(ml1m_s1, n_train=None, n_links=None):
    from deep_cbrs_amar_renaissance_amd.data.preprocess import build_adjacency_matrix, get_user_properties
    tr = ml1m_s1['train'] if n_train is None else ml1m_s1['train'][:n_train]
    trip = ml1m_s1['triples'] if n_links is None else ml1m_s1['triples'][:n_links]
    users, items, props = ml1m_s1['users'], ml1m_s1['items'], ml1m_s1['props']
    ui, ip = build_adjacency_matrix(tr, users, items, trip, props, 'unary-kg')
    return ui, ip, get_user_properties(ui, ip, len(users), len(items))


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('item_node', ['mean', 'concatenation'])
def test_two_step_small(hip, kind, item_node):
    from deep_cbrs_amar_renaissance_amd.models import basic
    if item_node == 'concatenation' and kind in ('LightGCN', 'DGCF'):
        pytest.skip("weight-free stacks cannot widen the user table (the reference fails in tf.concat as well)")
    g = helpers.kg_graph(seed=3)
    model = getattr(basic, 'BasicTS' + kind)(g['n_users'], g['n_items'], (g['adj_ui'], g['adj_ip']), **dict(CFG, item_node=item_node))
    _perturb(model, 7)
    ts = helpers.two_step_to_oracle(model.gnn)
    want_e = om.two_step((g['adj_ui'], g['adj_ip']), ts, g['n_users'], g['n_items'], np.float64)
    got_e = model.gnn(None).cpu().numpy()
    assert got_e.shape == want_e.shape == (g['n_users'] + g['n_items'], model.gnn.output_dim())
    assert helpers.rel_err(got_e, want_e) < 1e-5
    got = model((g['u_ids'], g['i_ids'])).cpu().numpy()
    want = om.basic_rs(want_e[g['u_ids']], want_e[g['i_ids']], {k: [(w.astype(np.float64), b.astype(np.float64)) for w, b in v]
                                                              for k, v in helpers.basic_head_to_oracle(model.rs).items()})
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('user_item_node', ['mean', 'concatenation'])
def test_two_way_small(hip, kind, user_item_node):
    from deep_cbrs_amar_renaissance_amd.models import basic
    g = helpers.kg_graph(seed=4)
    adjs = (g['adj_ui'], g['adj_ip'], g['adj_up'])
    model = getattr(basic, 'BasicTW' + kind)(g['n_users'], g['n_items'], adjs, **dict(CFG, user_item_node=user_item_node))
    _perturb(model, 9)
    tw = helpers.two_way_to_oracle(model.gnn)
    want_e = om.two_way(adjs, tw, g['n_users'], g['n_items'], np.float64)
    got_e = model.gnn(None).cpu().numpy()
    assert got_e.shape == want_e.shape == (g['n_users'] + g['n_items'], model.gnn.output_dim())
    assert helpers.rel_err(got_e, want_e) < 1e-5
    got = model((g['u_ids'], g['i_ids'])).cpu().numpy()
    want = om.basic_rs(want_e[g['u_ids']], want_e[g['i_ids']], {k: [(w.astype(np.float64), b.astype(np.float64)) for w, b in v]
                                                              for k, v in helpers.basic_head_to_oracle(model.rs).items()})
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize('kind', ['GCN', 'GraphSage', 'GAT', 'LightGCN'])
def test_two_step_ml1m(hip, ml1m_s1, kind):
    """ML-1M-shape graphs (9 228 + 20 746 nodes): hoisted predict over the whole test file equals the oracle."""
    from deep_cbrs_amar_renaissance_amd.models import basic
    ui, ip, _ = _ml1m_kg(ml1m_s1)
    nu, ni = len(ml1m_s1['users']), len(ml1m_s1['items'])
    model = getattr(basic, 'BasicTS' + kind)(nu, ni, (ui, ip), **CFG)
    _perturb(model, 11)
    helpers.spread_scores(model)
    data = ml1m_s1['test'][:20000]
    u, i = data[:, 0], data[:, 1]
    want_e = om.two_step((ui, ip), helpers.two_step_to_oracle(model.gnn), nu, ni, np.float64)
    assert helpers.rel_err(model.gnn(None).cpu().numpy(), want_e) < 1e-5
    head = {k: [(w.astype(np.float64), b.astype(np.float64)) for w, b in v] for k, v in helpers.basic_head_to_oracle(model.rs).items()}
    want = om.basic_rs(want_e[u], want_e[i], head)
    got = model((u, i)).cpu().numpy()
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize('kind', ['GCN', 'GraphSage', 'GAT', 'LightGCN'])
def test_two_way_ml1m(hip, ml1m_s1, kind):
    from deep_cbrs_amar_renaissance_amd.models import basic
    ui, ip, up = _ml1m_kg(ml1m_s1)
    nu, ni = len(ml1m_s1['users']), len(ml1m_s1['items'])
    model = getattr(basic, 'BasicTW' + kind)(nu, ni, (ui, ip, up), **CFG)
    _perturb(model, 13)
    helpers.spread_scores(model)
    data = ml1m_s1['test'][:20000]
    u, i = data[:, 0], data[:, 1]
    want_e = om.two_way((ui, ip, up), helpers.two_way_to_oracle(model.gnn), nu, ni, np.float64)
    assert helpers.rel_err(model.gnn(None).cpu().numpy(), want_e) < 1e-5
    head = {k: [(w.astype(np.float64), b.astype(np.float64)) for w, b in v] for k, v in helpers.basic_head_to_oracle(model.rs).items()}
    want = om.basic_rs(want_e[u], want_e[i], head)
    got = model((u, i)).cpu().numpy()
    assert np.abs(got - want).max() < 1e-4


def test_two_way_dgcf_subgraph(hip, ml1m_s1):
    """DGCF's cross-hop product on a sub-sampled ML-1M graph (the full one has ~45 M entries), all three graphs."""
    from deep_cbrs_amar_renaissance_amd.models import basic
    ui, ip, up = _ml1m_kg(ml1m_s1, n_train=40000, n_links=3000)
    nu, ni = len(ml1m_s1['users']), len(ml1m_s1['items'])
    model = basic.BasicTWDGCF(nu, ni, (ui, ip, up), **CFG)
    _perturb(model, 17)
    want_e = om.two_way((ui, ip, up), helpers.two_way_to_oracle(model.gnn), nu, ni, np.float64)
    assert helpers.rel_err(model.gnn(None).cpu().numpy(), want_e) < 1e-5


def test_hybrid_two_step(hip):
    """HybridBertTSGCN (hybrid.py:160-181): the TwoStep stack under the hybrid head, BERT rows travelling with the batch."""
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    g = helpers.kg_graph(seed=5)
    cfg = dict(CFG, dense_units=[[24, 24], [32, 16], [16, 16]], clf_units=[16, 16], feature_based=True)
    model = hybrid.HybridBertTSGCN(g['n_users'], g['n_items'], (g['adj_ui'], g['adj_ip']), **cfg)
    rng = np.random.default_rng(0)
    bert = (rng.standard_normal((g['n_users'] + g['n_items'], 48)) * 0.5).astype(np.float32)
    u, i = g['u_ids'], g['i_ids']
    got = model((u, i, bert[u], bert[i])).cpu().numpy()
    _perturb(model, 19)
    got = model((u, i, bert[u], bert[i])).cpu().numpy()
    want_e = om.two_step((g['adj_ui'], g['adj_ip']), helpers.two_step_to_oracle(model.gnn), g['n_users'], g['n_items'], np.float64)
    b64 = bert.astype(np.float64)
    want = om.hybrid_cbrs(want_e[u], want_e[i], b64[u], b64[i], helpers.hybrid_head_to_oracle(model.rs))
    assert np.abs(got - want).max() < 1e-4


def test_hoisted_predict_and_errors(hip):
    """predict() runs each stack once per weight state; malformed adjacency tuples and width mismatches raise ValueError
    like the reference (tsgnn.py:50-51, twgnn.py:50-51)."""
    from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraph
    from deep_cbrs_amar_renaissance_amd.models import basic
    g = helpers.kg_graph(seed=6)
    adjs = (g['adj_ui'], g['adj_ip'], g['adj_up'])
    model = basic.BasicTWGCN(g['n_users'], g['n_items'], adjs, **CFG)
    ratings = np.stack([g['u_ids'], g['i_ids'], np.zeros_like(g['u_ids'])], axis=1)
    seq = UserItemGraph(ratings, g['users'], g['items'], adjs, batch_size=64)
    pred = model.predict(seq)
    direct = model((g['u_ids'], g['i_ids'])).cpu().numpy()
    assert pred.shape == (len(ratings), 1) and np.abs(pred - direct).max() < 1e-6
    with pytest.raises(ValueError):
        basic.BasicTSGCN(g['n_users'], g['n_items'], adjs, **CFG)
    with pytest.raises(ValueError):
        basic.BasicTWGCN(g['n_users'], g['n_items'], adjs[:2], **CFG)
    with pytest.raises(ValueError):                           # 'last' hands over 4-wide items to an 8-wide user table
        basic.BasicTSGCN(g['n_users'], g['n_items'], adjs[:2], **dict(CFG, n_hiddens=[8, 4], item_node='last'))


# ---- training (SURVEY.md §8f N1 for the TwoStep / TwoWay stacks) -----------------------------------------------------------------

def _flat_stack(seq, gs, out):
    if 'embeddings' in gs:
        out[seq.embeddings] = gs['embeddings']
    for layer, gl in zip(seq.seq_layers, gs['layers']):
        for name, arr in gl.items():
            out[getattr(layer, {'attn_self': 'attn_kernel_self', 'attn_neigh': 'attn_kernel_neighs'}.get(name, name))] = arr


def _flatten(model, grads, layout):
    out = {}
    names = {'two_step': ('step_one', 'step_two'), 'two_way': ('way_one', 'way_two', 'step_two')}[layout]
    for name in names:
        _flat_stack(getattr(model.gnn, name + '_gnn_layers'), grads['gnn'][name], out)
    for name in grads['head']:
        for layer, (gw, gb) in zip(getattr(model.rs, name).layers, grads['head'][name]):
            out[layer.kernel], out[layer.bias] = gw, gb
    return out


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('layout,node', [('two_step', 'mean'), ('two_step', 'concatenation'), ('two_way', 'mean'),
                                         ('two_way', 'concatenation'), ('two_way', 'last')])
def test_gradients_match_autograd_oracle(hip, kind, layout, node):
    """One training batch: loss and every gradient (all tables, all layers of all stacks, the head) against torch
    autograd of the oracle's restated forward (float64), including the reductions 'mean' / 'last' between the stacks."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import basic
    from oracle import train as otrain
    if layout == 'two_step' and node == 'concatenation' and kind in ('LightGCN', 'DGCF'):
        pytest.skip("weight-free stacks cannot widen the user table")
    engine.set_seed(5)
    g = helpers.kg_graph(n_users=60, n_items=45, n_props=30, n_ratings=900, n_links=120, seed=11)
    if layout == 'two_step':
        adjs = (g['adj_ui'], g['adj_ip'])
        model = getattr(basic, 'BasicTS' + kind)(g['n_users'], g['n_items'], adjs, **dict(CFG, item_node=node))
        ow = helpers.two_step_to_oracle
    else:
        adjs = (g['adj_ui'], g['adj_ip'], g['adj_up'])
        model = getattr(basic, 'BasicTW' + kind)(g['n_users'], g['n_items'], adjs, **dict(CFG, user_item_node=node))
        ow = helpers.two_way_to_oracle
    # (seed 23 puts one classifier pre-activation of the two_way / 'last' / GAT case at +2.5e-9: its ReLU mask then differs
    # between fp32 and the float64 oracle and moves that unit's gradient by one pair's worth, 8 %)
    _perturb(model, 29)
    y = np.random.default_rng(2).integers(0, 2, len(g['u_ids']))
    trainer = training.Trainer(model)
    assert trainer.layout == layout
    loss, grads = trainer.loss_and_grads(g['u_ids'], g['i_ids'], y)
    with torch.no_grad():                                     # the taped forward scores like the inference forward
        e_inf = model.gnn(None)
        assert float((e_inf - trainer._propagation_forward()).abs().max()) <= 2e-6 * float(e_inf.abs().max())
    want_loss, want, _ = otrain.torch_model_grads(adjs, ow(model.gnn), helpers.basic_head_to_oracle(model.rs), g['u_ids'], g['i_ids'], y,
                                                  l2=1e-4, n_users=g['n_users'], n_items=g['n_items'])
    assert abs(loss - want_loss) < 1e-5
    flat = _flatten(model, want, layout)
    assert set(flat) == set(grads)
    for prm, gw in flat.items():
        got = grads[prm].cpu().numpy().reshape(gw.shape).astype(np.float64)
        got += 2 * trainer._l2(prm) * prm.detach().cpu().numpy().reshape(gw.shape)      # the trainer folds the L2 term into the Adam kernel
        assert np.abs(got - gw).max() <= 2e-4 * np.abs(gw).max() + 1e-10, tuple(prm.shape)
    if layout == 'two_step':                                  # tsgnn.py:77-81: the user table carries no regulariser, step one's does
        assert trainer._l2(model.gnn.step_two_gnn_layers.embeddings) == 0.0
        assert trainer._l2(model.gnn.step_one_gnn_layers.embeddings) == 1e-4


@pytest.mark.parametrize('cls', ['BasicTSGCN', 'BasicTWGraphSage', 'BasicTWLightGCN'])
def test_fit_reduces_loss_and_replays_graphs(hip, cls):
    """model.fit over a Sequence: hipGraph-replayed batches, loss going down, same weights as eager batches."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import basic
    g = helpers.kg_graph(n_users=60, n_items=45, n_props=30, n_ratings=900, n_links=120, seed=12)
    adjs = (g['adj_ui'], g['adj_ip']) if 'TS' in cls else (g['adj_ui'], g['adj_ip'], g['adj_up'])
    rng = np.random.default_rng(4)
    labels = ((g['u_ids'] + g['i_ids']) % 2).astype(np.int64)
    batches = [(g['u_ids'][k * 64:(k + 1) * 64], g['i_ids'][k * 64:(k + 1) * 64], labels[k * 64:(k + 1) * 64]) for k in range(4)]
    models = []
    for _ in range(2):
        engine.set_seed(8)
        m = getattr(basic, cls)(g['n_users'], g['n_items'], adjs, **CFG)
        helpers.randomize_biases(m, seed=1)
        models.append(m)
    eager, graphed = training.Trainer(models[0], learning_rate=1e-2), training.Trainer(models[1], learning_rate=1e-2)
    epoch_loss = []
    for epoch in range(6):
        tot = 0.0
        for u, i, y in batches:
            tot += eager.train_batch(u, i, y) * len(y)
            graphed.train_batch_graphed(u, i, y)
        epoch_loss.append(tot)
    assert graphed._g is not None and graphed.t == eager.t == 24
    assert epoch_loss[-1] < epoch_loss[0]
    assert abs(graphed.pop_loss_sum() - sum(epoch_loss)) < 1e-3 * sum(epoch_loss)
    for pa, pb in zip(models[0].parameters(), models[1].parameters()):
        assert torch.allclose(pa, pb, rtol=1e-3, atol=1e-5), tuple(pa.shape)


@pytest.mark.parametrize('cls,layout', [('HybridBertTSGCN', 'two_step'), ('HybridBertTWGraphSage', 'two_way')])
def test_hybrid_gradients_match_autograd_oracle(hip, cls, layout):
    """HybridBertTS* / HybridBertTW* (hybrid.py:160-181): the chained stacks under the four-input head with attention fusion."""
    from deep_cbrs_amar_renaissance_amd import engine, training
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    from oracle import train as otrain
    engine.set_seed(11)
    g = helpers.kg_graph(n_users=60, n_items=45, n_props=30, n_ratings=900, n_links=120, seed=13)
    adjs = (g['adj_ui'], g['adj_ip']) if layout == 'two_step' else (g['adj_ui'], g['adj_ip'], g['adj_up'])
    cfg = dict(CFG, dense_units=[[24, 16], [32, 24], [16, 16]], clf_units=[24, 16], feature_based=True, fusion_method='attention')
    model = getattr(hybrid, cls)(g['n_users'], g['n_items'], adjs, **cfg)
    rng = np.random.default_rng(3)
    table = rng.standard_normal((g['n_users'] + g['n_items'], 40)).astype(np.float32) * 0.5
    model.set_bert_table(table)
    y = rng.integers(0, 2, len(g['u_ids']))
    trainer = training.Trainer(model)
    _perturb(model, 31)
    u, i = g['u_ids'], g['i_ids']
    loss, grads = trainer.loss_and_grads(u, i, y)
    ow = helpers.two_step_to_oracle if layout == 'two_step' else helpers.two_way_to_oracle
    want_loss, want, p = otrain.torch_model_grads(adjs, ow(model.gnn), helpers.hybrid_head_to_oracle(model.rs), u, i, y, l2=1e-4,
                                                  bert=(table[u], table[i]), feature_based=True, n_users=g['n_users'], n_items=g['n_items'])
    with torch.no_grad():
        assert np.abs(model((u, i, None, None)).cpu().numpy()[:, 0] - p).max() < 1e-5
    assert abs(loss - want_loss) < 1e-5
    flat = {}
    for name in (('step_one', 'step_two') if layout == 'two_step' else ('way_one', 'way_two', 'step_two')):
        _flat_stack(getattr(model.gnn, name + '_gnn_layers'), want['gnn'][name], flat)
    for name in want['head']:
        if name.startswith('fuse'):
            for key, arr in want['head'][name].items():
                flat[getattr(getattr(model.rs, name), key)] = arr
            continue
        for layer, (gw, gb) in zip(getattr(model.rs, name).layers, want['head'][name]):
            flat[layer.kernel], flat[layer.bias] = gw, gb
    assert set(flat) == set(grads)
    for prm, gw in flat.items():
        got = grads[prm].cpu().numpy().reshape(gw.shape).astype(np.float64)
        got += 2 * trainer._l2(prm) * prm.detach().cpu().numpy().reshape(gw.shape)
        assert np.abs(got - gw).max() <= 2e-4 * np.abs(gw).max() + 1e-10, tuple(prm.shape)


def test_randomised_shapes(hip):
    """Seeded sweep over graph sizes, widths, depths and reductions of both layouts (inference parity with the oracle)."""
    from deep_cbrs_amar_renaissance_amd.models import basic
    rng = np.random.default_rng(2024 + helpers.seed_offset())
    for trial in range(12):
        nu, ni, n_props = int(rng.integers(5, 70)), int(rng.integers(5, 60)), int(rng.integers(3, 40))
        g = helpers.kg_graph(n_users=nu, n_items=ni, n_props=n_props, n_ratings=int(rng.integers(10, 600)), n_links=int(rng.integers(5, 150)),
                             seed=int(rng.integers(0, 1000)))
        kind = KINDS[trial % len(KINDS)]
        d = int(rng.choice([4, 8, 16]))
        hops = int(rng.integers(1, 4))
        node = str(rng.choice(['mean', 'sum', 'last'] if kind in ('LightGCN', 'DGCF') else ['mean', 'sum', 'last', 'concatenation']))
        final = str(rng.choice(['concatenation', 'mean', 'last']))
        cfg = dict(CFG, embedding_dim=d, n_hiddens=[d] * hops, n_layers=hops, final_node=final)
        if trial % 2 == 0:
            adjs = (g['adj_ui'], g['adj_ip'])
            model = getattr(basic, 'BasicTS' + kind)(nu, ni, adjs, **dict(cfg, item_node=node))
            want_fn, ow = om.two_step, helpers.two_step_to_oracle
        else:
            adjs = (g['adj_ui'], g['adj_ip'], g['adj_up'])
            model = getattr(basic, 'BasicTW' + kind)(nu, ni, adjs, **dict(cfg, user_item_node=node))
            want_fn, ow = om.two_way, helpers.two_way_to_oracle
        _perturb(model, trial)
        want_e = want_fn(adjs, ow(model.gnn), nu, ni, np.float64)
        got_e = model.gnn(None).cpu().numpy()
        label = '{} {} d={} hops={} node={} final={} ({} users, {} items, {} props)'.format(type(model).__name__, kind, d, hops, node, final, nu, ni, n_props)
        assert got_e.shape == want_e.shape, label
        assert helpers.rel_err(got_e, want_e) < 1e-5, label
