"""CPU: host-side logic of the drop-in boundary (no kernels run here)."""
import os
import re

import numpy as np
import pytest
import torch
from scipy import sparse

from oracle import graph as og

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_capi_library_exports_every_declared_symbol():
    from deep_cbrs_amar_renaissance_amd import capi
    header = open(os.path.join(ROOT, 'include', 'amar_hip.h')).read()
    declared = set(re.findall(r'\b(amar_[a-z0-9_]+)\s*\(', header))
    assert declared and declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)
    lib = capi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.amar_version() == 100
    assert b'invalid' in lib.amar_error_string(-1)


def test_no_fallback_without_gpu_tensors():
    from deep_cbrs_amar_renaissance_amd import capi
    x = torch.zeros((4, 8))
    with pytest.raises(capi.AmarError):
        capi.dense(x, torch.zeros((8, 8)), torch.zeros(8), torch.zeros((4, 8)))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'deep_cbrs_amar_renaissance_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dirpath, f)


def test_adjacency_and_filter_match_oracle(ml1m_s1):
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter, DeviceCSR
    g = ml1m_s1
    nu, ni, npr = len(g['users']), len(g['items']), len(g['props'])
    assert (nu, ni) == (6036, 3192)
    for adj, want in ((g['adj_ui'], og.adjacency_unary(g['train'], nu, ni)),
                      (g['adj_uip'], og.adjacency_unary_uip(g['train'], g['triples'], nu, ni, npr))):
        assert adj.dtype == np.float32
        assert np.array_equal(adj.row, want.row) and np.array_equal(adj.col, want.col) and np.array_equal(adj.data, want.data)
        a, b = gcn_filter(adj), og.gcn_filter(want)
        assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.array_equal(a.data, b.data)
    # CSR conversion keeps duplicates, row-major order (tf.sparse.reorder)
    csr = DeviceCSR.from_scipy(g['adj_uip'], with_values=False, drop_diagonal=True, device='cpu')
    row, col, _ = og.reordered_coo(g['adj_uip'])
    assert csr.nnz == g['adj_uip'].nnz == len(row)
    assert np.array_equal(csr.colidx.numpy(), col) and np.array_equal(np.diff(csr.rowptr.numpy()), np.bincount(row, minlength=csr.shape[0]))
    dup = len(row) - len(np.unique(row * csr.shape[0] + col))
    assert dup > 0, "the UIP fixture must exercise duplicate edges"


def test_build_adjacency_errors():
    from deep_cbrs_amar_renaissance_amd.data.preprocess import build_adjacency_matrix
    r = np.array([[0, 2, 1], [1, 3, 0]])
    with pytest.raises(ValueError):
        build_adjacency_matrix(r, [0, 1], [0, 1], type_adjacency='nope')
    with pytest.raises(ValueError):
        build_adjacency_matrix(r, [0, 1], [0, 1], type_adjacency='unary-uip')
    a = build_adjacency_matrix(r, [0, 1], [0, 1], symmetric_adjacency=False)
    assert a.nnz == 1 and a.shape == (4, 4)


def test_loaders_roundtrip_through_reference_file_formats(tmp_path):
    from deep_cbrs_amar_renaissance_amd.data import loaders, synthetic
    ds = synthetic.ml1m(1)
    ds.train, ds.test = ds.train[:60000], ds.test[np.isin(ds.test[:, 0], ds.train[:60000, 0]) & np.isin(ds.test[:, 1], ds.train[:60000, 1])][:5000]
    ds.props = ds.props[np.isin(ds.props[:, 0], ds.train[:, 1])][:3000]
    paths = synthetic.write_dataset(ds, str(tmp_path), bert_dim=16, kge_dim=8)
    tr, te = loaders.load_user_item_graph(paths['train_ratings_filepath'], paths['test_ratings_filepath'],
                                          paths['props_triples_filepath'], type_adjacency='unary-uip')
    (otr, ote), (users, items) = og.remap_ratings(ds.train, ds.test)
    assert np.array_equal(tr.ratings, otr) and np.array_equal(te.ratings, ote)
    assert np.array_equal(tr.users, users) and np.array_equal(tr.items, items)
    assert len(tr) == int(np.ceil(len(otr) / 1024)) and len(te) == int(np.ceil(len(ote) / 2048))
    (u, i), y = te[len(te) - 1]
    assert u.dtype == np.int64 and len(u) == len(ote) - 2048 * (len(te) - 1) and i.min() >= len(users)
    # shuffling: RandomState(42) permutation, redrawn per epoch
    idx0 = tr.indexes.copy()
    expect = np.arange(len(otr)); np.random.RandomState(42).shuffle(expect)
    assert np.array_equal(idx0, expect)
    tr.on_epoch_end()
    assert not np.array_equal(tr.indexes, idx0)
    # hybrid loader: BERT rows follow users then items
    htr, hte = loaders.load_user_item_graph_bert_embeddings(
        paths['train_ratings_filepath'], paths['test_ratings_filepath'], paths['bert_user_filepath'], paths['bert_item_filepath'])
    (hu, hi, hub, hib), _ = hte[0]
    assert hub.shape == (len(hu), 16) and hub.dtype == np.float32
    want_u = synthetic.entity_embeddings(len(users), 16, 'bert')
    assert np.allclose(hub, want_u[hu], atol=1e-6)
    # KGE loader
    ktr, kte = loaders.load_graph_embeddings(paths['train_ratings_filepath'], paths['test_ratings_filepath'], paths['graph_filepath'])
    (ku, ki), _ = kte[0]
    assert ku.shape == (min(2048, len(ote)), 8)
    with pytest.raises(ValueError):
        loaders.index_ratings(ds.train, np.array([[10 ** 9, ds.train[0, 1], 1]]))


def test_grid_and_yaml_semantics(tmp_path):
    from deep_cbrs_amar_renaissance_amd.utilities.utils import make_grid, nested_dict_update, mlflow_linearize
    from deep_cbrs_amar_renaissance_amd.experiment import load_yaml, AttrDict
    p = tmp_path / 'c.yaml'
    p.write_text("model:\n  l2_regularizer: 1e-4\n  name: basic.BasicGCN\ngrid:\n  g1:\n    model:\n      l2_regularizer: [1e-5, 1e-3]\n      n_hiddens: [[8, 8]]\n    dataset:\n      x: [a, b, c]\n")
    cfg = load_yaml(str(p))
    assert isinstance(cfg['model']['l2_regularizer'], float) and cfg['model']['l2_regularizer'] == 1e-4   # YAML 1.2 float
    grid = make_grid(cfg['grid']['g1'])
    assert len(grid) == 6 and grid[0] == {'model': {'l2_regularizer': 1e-5, 'n_hiddens': [8, 8]}, 'dataset': {'x': 'a'}}
    with pytest.raises(ValueError):
        make_grid({'a': 3})
    base = {'model': {'name': 'x', 'k': 1}, 'seed': 1}
    assert nested_dict_update(base, {'model': {'k': 2}}) == {'model': {'name': 'x', 'k': 2}, 'seed': 1}
    assert mlflow_linearize({'a': {'b': {'c': 1}}, 'd': 2}) == {'a.b.c': 1, 'd': 2}
    ad = AttrDict({'a': {'b': 3}})
    assert ad.a.b == 3 and dict(**ad.a) == {'b': 3}


def test_model_classes_resolve_and_param_counts():
    from deep_cbrs_amar_renaissance_amd.models import basic, hybrid
    N = 9228
    adj = sparse.coo_matrix((np.ones(2, dtype=np.float32), ([0, 1], [1, 0])), shape=(N, N))
    cfg = dict(name='basic.BasicGCN', embedding_dim=8, n_hiddens=[8, 8], l2_regularizer=1e-4, final_node='concatenation',
               item_node='mean', user_item_node='mean', aggregate='mean', dropout_rate=0.0, n_layers=2,
               dense_units=[24, 24], clf_units=[48, 48], activation='relu', feature_based=True,
               fusion_method='concatenate', residual=False)              # the WHOLE model: section is passed (experiment.py:146-153)
    count = lambda m: sum(p.numel() for p in m.parameters())
    assert count(basic.BasicGCN(adj, **cfg)) == 81121
    assert count(basic.BasicGraphSage(adj, **cfg)) == 81249
    assert count(basic.BasicGAT(adj, **cfg)) == 81153
    assert count(basic.BasicLightGCN(adj, **cfg)) == 80209
    h = dict(cfg, embedding_dim=16, n_hiddens=[16, 16], dense_units=[[48, 48], [256, 64], [64, 64]], clf_units=[64, 64])
    m = hybrid.HybridBertGCN(adj, **h)
    m.rs.build_head(m.gnn.output_dim(), 768)
    assert count(m) == 619489
    for name in ['BasicTSGCN', 'BasicTWGAT', 'BasicDGCF', 'BasicKnowledgeGCN', 'BasicTSGNN', 'BasicTWGNN']:
        assert hasattr(basic, name)
    for name in ['HybridBertTSGCN', 'HybridBertTWLightGCN', 'HybridBertDGCF', 'HybridCBRS']:
        assert hasattr(hybrid, name)
    assert count(basic.BasicDGCF(adj, **cfg)) == 80209 + 2 * 9228      # LightGCN's table + one gate per node and layer (n_layers=2)
    with pytest.raises(TypeError):                            # the TwoStep classes take (n_users, n_items, adjacencies), experiment.py:143-148
        basic.BasicTSGCN(adj, **cfg)
    with pytest.raises(ValueError):
        basic.BasicGCN(adj, **dict(cfg, final_node='bogus'))
    with pytest.raises(ValueError):
        hybrid.HybridCBRS(fusion_method='bogus')
    from deep_cbrs_amar_renaissance_amd import training
    assert training.Trainer(basic.BasicGCN(adj, **dict(cfg, final_node='last'))).tapes[0].kind == 'gcn'     # every reduction has a reverse pass
    # 'w-sum' (WeightedSum, reduction.py:36-55): one learnable weight per term X_0 .. X_L, ones, no regulariser (gnn.py:62 passes none);
    # the output is one layer wide (the heads are built for it) and the weights train with everything else
    ws = basic.BasicGCN(adj, **dict(cfg, final_node='w-sum'))
    red = ws.gnn.gnn_layers.reduce
    assert tuple(red.w.shape) == (3, 1, 1) and bool((red.w == 1).all()) and red.w.regularizer is None
    assert ws.gnn.output_dim() == 8
    assert count(ws) == 8 * N + 2 * (8 * 8 + 8) + 2 * ((8 * 24 + 24) + (24 * 24 + 24)) + (48 * 48 + 48) * 2 + 49 + 3
    trainer = training.Trainer(ws)
    assert trainer.tapes[0].kind == 'gcn' and any(p is red.w for p in trainer.params)


def test_seed_reproducibility_and_glorot_limits():
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.models import basic
    adj = sparse.coo_matrix((np.ones(2, dtype=np.float32), ([0, 1], [1, 0])), shape=(500, 500))
    cfg = dict(embedding_dim=8, n_hiddens=[8, 8], dense_units=[24, 24], clf_units=[48, 48])
    engine.set_seed(42)
    a = basic.BasicGCN(adj, **cfg)
    engine.set_seed(42)
    b = basic.BasicGCN(adj, **cfg)
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb and torch.equal(pa, pb)
    emb = a.gnn.gnn_layers.embeddings
    assert float(emb.abs().max()) <= np.sqrt(6 / (500 + 8)) and float(emb.abs().max()) > 0.9 * np.sqrt(6 / 508)
    assert all(float(p.abs().max()) == 0 for n, p in a.named_parameters() if n.endswith('bias'))


def test_streaming_embedding_readers(tmp_path):
    """data/jsonstream.py against json.load on the reference's two file formats (loaders.py:85-105), with chunk sizes small
    enough that keys, rows and records straddle chunk boundaries."""
    import json
    from deep_cbrs_amar_renaissance_amd.data import jsonstream, loaders
    rng = np.random.default_rng(0)
    kge = rng.standard_normal((37, 12)).astype(np.float32)
    kge[3, 4] = 1e-9
    kge[5, 0] = -123456.75
    doc = {'rel_embeddings': [[0.5, 1.5], [2.5, 3.5]], 'ent_embeddings': kge.astype(np.float64).tolist(), 'zz_after': [[9.0]]}
    p_kge = tmp_path / 'kge.json'
    p_kge.write_text(json.dumps(doc, indent=1))                       # newlines / indentation inside the rows
    for chunk in (7, 64, 1000, 1 << 20):
        got = jsonstream.stream_ent_embeddings(str(p_kge), chunk=chunk)
        assert got.dtype == np.float32 and np.array_equal(got, kge)
    (tmp_path / 'compact.json').write_text(json.dumps(doc, separators=(',', ':')))
    assert np.array_equal(jsonstream.stream_ent_embeddings(str(tmp_path / 'compact.json'), chunk=13), kge)
    with pytest.raises(KeyError):
        jsonstream.stream_ent_embeddings(str(p_kge), key='missing')
    (tmp_path / 'cut.json').write_text(json.dumps(doc)[:400])
    with pytest.raises(ValueError):
        jsonstream.stream_ent_embeddings(str(tmp_path / 'cut.json'))
    # BERT records: unordered ids, extra fields, both column names
    ids = rng.permutation(50)[:20] * 3 + 1
    emb = rng.standard_normal((20, 9)).astype(np.float32)
    records = [{'title': 'x [y] {z}', 'ID_OpenKE': int(i), 'embedding': e.astype(np.float64).tolist(), 'n': 1} for i, e in zip(ids, emb)]
    p_bert = tmp_path / 'items.json'
    p_bert.write_text(json.dumps(records, indent=2))
    for chunk in (11, 300, 1 << 20):
        got_ids, got = jsonstream.stream_bert_records(str(p_bert), 'embedding', chunk=chunk)
        assert np.array_equal(got_ids, ids) and np.array_equal(got, emb)
    users = [{'ID_OpenKE': int(i), 'profile_embedding': e.astype(np.float64).tolist()} for i, e in zip(ids, emb[::-1])]
    (tmp_path / 'users.json').write_text(json.dumps(users))
    want_ids = np.sort(ids)[:5]
    table = loaders.load_bert_user_item_embeddings(str(tmp_path / 'users.json'), str(p_bert), want_ids, want_ids)
    lookup = {int(i): k for k, i in enumerate(ids)}
    assert np.array_equal(table[:5], emb[::-1][[lookup[int(i)] for i in want_ids]])
    assert np.array_equal(table[5:], emb[[lookup[int(i)] for i in want_ids]])
    assert np.array_equal(loaders.load_graph_user_item_embeddings(str(p_kge), [1, 2], [30, 36]), kge[[1, 2, 30, 36]])
    (tmp_path / 'empty.json').write_text('[]')
    e_ids, e_tab = jsonstream.stream_bert_records(str(tmp_path / 'empty.json'), 'embedding')
    assert e_ids.size == 0 and e_tab.size == 0


def test_host_ids_are_range_checked():
    """engine.ids_to_device refuses ids past the table (TensorFlow's embedding_lookup would raise; the kernels do not check)."""
    from deep_cbrs_amar_renaissance_amd import engine
    ids = np.array([0, 5, 9])
    with pytest.raises(IndexError):
        engine.ids_to_device(ids, n_rows=9)
    with pytest.raises(ValueError):
        engine.ids_to_device(np.array([-1, 2]), n_rows=9)


@pytest.mark.parametrize('phases', [1, 2])
def test_pair_plan_is_a_permutation_with_xcd_affine_item_ranges(phases):
    """models.basic.PairPlan (plain torch: runs on CPU tensors): a permutation of the list; positions with equal
    (p >> 7) % 8 — the pairs one XCD's workgroups score — hold one contiguous item range per phase; out_index sends every
    score back to its place."""
    from deep_cbrs_amar_renaissance_amd.models.basic import PairPlan
    g = torch.Generator().manual_seed(3)
    p = 40_013
    u = torch.randint(0, 900, (p,), generator=g, dtype=torch.int32)
    i = (torch.randint(0, 700, (p,), generator=g) + 900).to(torch.int32)
    plan = PairPlan(u, i, phases=phases)
    assert sorted(plan.out_index.tolist()) == list(range(p))
    assert torch.equal(plan.u_ids, u[plan.out_index.long()]) and torch.equal(plan.i_ids, i[plan.out_index.long()])
    pos = torch.arange(p)
    per_phase = -(-p // phases)
    cls = (pos // per_phase) * 8 + (pos // 128) % 8
    lo = [int(plan.i_ids[cls == c].min()) for c in range(8 * phases)]
    hi = [int(plan.i_ids[cls == c].max()) for c in range(8 * phases)]
    order = [ph * 8 + x for x in range(8) for ph in range(phases)]          # item ranges ascend in (XCD, phase) order
    assert all(lo[b] >= hi[a] for a, b in zip(order[:-1], order[1:]))
    scores_in_plan_order = (plan.u_ids.double() * 1000 + plan.i_ids.double())
    out = torch.empty(p, dtype=torch.float64)
    out[plan.out_index.long()] = scores_in_plan_order                        # what the kernel's indexed store does
    assert torch.equal(out, u.double() * 1000 + i.double())
    with pytest.raises(ValueError):
        plan.check(u.clone(), i)


def test_infer_row_breaks():
    """Node-type boundaries read off the entries: users | items, users | items | properties, none for an ordinary graph."""
    import torch
    from deep_cbrs_amar_renaissance_amd.utilities.math import infer_row_breaks
    rng = np.random.default_rng(0)
    nu, ni, npr = 300, 200, 150
    u, i = rng.integers(0, nu, 4000), rng.integers(0, ni, 4000) + nu
    i[0], i[1] = nu, nu + ni - 1
    rows, cols = np.concatenate([u, i]), np.concatenate([i, u])
    t = lambda a: torch.from_numpy(a.astype(np.int64))
    assert infer_row_breaks(t(rows), t(cols), nu + ni) == (nu,)
    it, pr = rng.integers(0, ni, 900) + nu, rng.integers(0, npr, 900) + nu + ni
    pr[0] = nu + ni
    rows3, cols3 = np.concatenate([rows, it, pr]), np.concatenate([cols, pr, it])
    assert infer_row_breaks(t(rows3), t(cols3), nu + ni + npr) == (nu, nu + ni)
    # self edges do not hide the structure; an ordinary graph has none
    assert infer_row_breaks(t(np.concatenate([rows, [5]])), t(np.concatenate([cols, [5]])), nu + ni) == (nu,)
    r, c = rng.integers(0, 500, 5000), rng.integers(0, 500, 5000)
    assert infer_row_breaks(t(np.concatenate([r, c])), t(np.concatenate([c, r])), 500) == ()
    assert infer_row_breaks(t(np.zeros(0)), t(np.zeros(0)), 10) == ()


def test_spmm_kind_size_and_density_rules(monkeypatch):
    """Which propagation form a graph takes (utilities.math.spmm_kind), on stand-in shapes: row kernels while the gathered table
    is small, the tiled route (LT where eligible) from a 2 MB table on for gcn-filtered matrices with known
    factors AND for edge-list graphs (GraphSAGE / GAT) that are dense enough per tile and column, XS by the older 8 / 16 MB rule
    otherwise; AMAR_SPMM_KIND / AMAR_SPMM_LT override."""
    from deep_cbrs_amar_renaissance_amd.utilities import math as m

    class G:
        def __init__(self, n, nnz, factors=False, vals='v'):
            self.shape, self.nnz, self.vals = (n, n), nnz, vals
            self.dinv = self.mult = ('x' if factors else None)
    for k in ('AMAR_SPMM_KIND', 'AMAR_SPMM_LT', 'AMAR_XS_VALUES'):
        monkeypatch.delenv(k, raising=False)
    n16 = 9228 * 16                                                  # 4.7 MB at F = 8
    assert m.spmm_kind(G(9228, 876_000, True), 8) == 'csr'           # ML-1M itself: row streaming
    assert m.spmm_kind(G(n16, 14_000_000, True), 8) == 'xs'          # factors known, dense enough: tiled (LT) from 2 MB on
    assert m.spmm_kind(G(9228 * 8, 6_900_000, True), 8) == 'xs' and m.spmm_kind(G(9228 * 4, 3_400_000, True), 8) == 'csr'
    assert m.spmm_kind(G(n16, 14_000_000, False), 8) == 'csr'        # valued matrix without factors: below the 8 MB XS rule
    assert m.spmm_kind(G(n16, 14_000_000, False, vals=None), 8) == 'xs'      # edge list (GraphSAGE / GAT): LT walk from 2 MB on
    assert m.spmm_kind(G(n16, 1_000_000, False, vals=None), 8) == 'csr'      # ... unless too sparse per tile and column
    assert m.spmm_kind(G(9228 * 64, 1_000_000, False, vals=None), 8) == 'xs' # a large sparse one: the XS forms (18.9 MB table)
    assert m.spmm_kind(G(n16, 14_000_000, False, vals=None), 32) == 'xs'     # C = 32 walks LT too
    monkeypatch.setenv('AMAR_SPMM_LT', '0')
    assert m.spmm_kind(G(n16, 14_000_000, False, vals=None), 8) == 'csr'
    monkeypatch.setenv('AMAR_SPMM_KIND', 'sj')
    assert m.spmm_kind(G(n16, 14_000_000, True), 8) == 'sj'


def test_integration_md_stubs_match_the_binding_table():
    """Every `lib.<symbol>.argtypes = [...]` line of INTEGRATION.md lists the same ctypes, in the same order, as capi.SIGNATURES
    (which test_capi_library_exports_every_declared_symbol holds against include/amar_hip.h): the documented binding cannot drift from the library."""
    import ctypes
    import os
    import re
    from deep_cbrs_amar_renaissance_amd import capi
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'INTEGRATION.md')).read()
    names = {'P': ctypes.c_void_p, 'I32': ctypes.c_int32, 'I64': ctypes.c_int64, 'U32': ctypes.c_uint32, 'F32': ctypes.c_float}
    found = re.findall(r'lib\.(amar_\w+)\.argtypes\s*=\s*\[([^\]]*)\]', text, flags=re.S)
    assert len(found) >= 4
    for sym, body in found:
        doc = [names[t.strip()] for t in body.replace('\n', ' ').split(',') if t.strip()]
        assert sym in capi.SIGNATURES, sym
        assert doc == list(capi.SIGNATURES[sym][1]), sym


@pytest.mark.parametrize('sym', [True, False])
def test_loaders_reproduce_the_reference_functions_own_output(tmp_path, sym):
    """The product's `data.loaders.load_train_test_ratings` (+ `data.preprocess`) on the files of tests/golden/graph_reference.npz
    against what the REFERENCE's own functions returned for them (made by tests/golden/make_graph_reference_golden.py): indexed
    ratings, users / items, and the adjacency matrices — same triplets in the same order, shape and dtype — for 'unary',
    'unary-uip', 'unary-kg' and the user-property graph."""
    import os
    from scipy import sparse
    from deep_cbrs_amar_renaissance_amd.data import loaders, preprocess
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'graph_reference.npz'))
    paths = {}
    for k in ('train', 'test', 'props'):
        paths[k] = str(tmp_path / (k + '.tsv'))
        np.savetxt(paths[k], z['raw_' + k], fmt='%d', delimiter='\t')
    (tr, te), (users, items) = loaders.load_train_test_ratings(paths['train'], paths['test'])
    assert np.array_equal(tr, z['train_indexed']) and np.array_equal(te, z['test_indexed'])
    assert np.array_equal(users, z['users']) and np.array_equal(items, z['items'])

    def same(m, tag, j=0):
        m = m.tocoo() if not sparse.isspmatrix_coo(m) else m
        assert tuple(m.shape) == tuple(z['{}_{}_shape'.format(tag, j)]) and str(m.dtype) == str(z['{}_{}_dtype'.format(tag, j)])
        assert np.array_equal(m.row, z['{}_{}_row'.format(tag, j)]) and np.array_equal(m.col, z['{}_{}_col'.format(tag, j)])
        assert np.array_equal(np.asarray(m.data), z['{}_{}_val'.format(tag, j)])
    tag = 'sym' if sym else 'raw'
    for kind in ('unary', 'unary-uip', 'unary-kg'):
        _, _, adj = loaders.load_train_test_ratings(paths['train'], paths['test'], paths['props'] if kind != 'unary' else None,
                                                    return_adjacency=True, type_adjacency=kind, symmetric_adjacency=sym)
        mats = adj if isinstance(adj, tuple) else (adj,)
        for j, m in enumerate(mats):
            same(m, '{}_{}'.format(kind.replace('-', '_'), tag), j)
        if kind == 'unary-kg' and sym:
            up = preprocess.get_user_properties(mats[0], mats[1], len(users), len(items)).tocoo()
            ref = sparse.coo_matrix((z['user_props_val'], (z['user_props_row'], z['user_props_col'])), shape=tuple(z['user_props_shape']))
            assert tuple(up.shape) == ref.shape and str(up.dtype) == str(z['user_props_dtype'])
            assert (abs(up - ref)).nnz == 0                           # the sparse build: same matrix (its triplet order is its own)


def test_embedding_loaders_and_property_filter_reproduce_the_reference_functions_own_output(tmp_path):
    """SURVEY 8f N2 against the reference itself: tests/golden/loaders_reference.npz holds what the REFERENCE's
    `load_graph_user_item_embeddings`, `load_bert_user_item_embeddings` (loaders.py:85-144) and `process_item_properties_graph`
    (preprocess.py:173-198) returned for small files of its on-disk formats (made by tests/golden/make_graph_reference_golden.py);
    the product's loaders must return the same arrays bit for bit and write the same filtered property file."""
    import json
    import os
    from deep_cbrs_amar_renaissance_amd.data import loaders, preprocess
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loaders_reference.npz'))
    users, items = z['users'], z['items']
    kge = str(tmp_path / 'kge.json')
    json.dump({'ent_embeddings': z['kge_table'].tolist()}, open(kge, 'w'))
    got = loaders.load_graph_user_item_embeddings(kge, users, items)
    assert got.dtype == np.float32 and np.array_equal(got, z['kge_rows'])
    ub, ib = str(tmp_path / 'u.json'), str(tmp_path / 'i.json')
    json.dump([{'ID_OpenKE': int(users[k]), 'profile_embedding': z['bert_users'][k].tolist()} for k in z['bert_user_file_order']], open(ub, 'w'))
    json.dump([{'ID_OpenKE': int(items[k]), 'embedding': z['bert_items'][k].tolist()} for k in z['bert_item_file_order']], open(ib, 'w'))
    got = loaders.load_bert_user_item_embeddings(ub, ib, users, items)
    assert got.dtype == np.float32 and np.array_equal(got, z['bert_rows'])
    ratings, graph, out = str(tmp_path / 'train.tsv'), str(tmp_path / 'graph.tsv'), str(tmp_path / 'kg.tsv')
    np.savetxt(ratings, z['filter_ratings'], fmt='%d', delimiter='\t')
    with open(graph, 'w') as fp:
        fp.write('head\ttail\trel\n')
        np.savetxt(fp, np.concatenate([z['filter_ratings'], z['filter_kg_rows']]), fmt='%d', delimiter='\t')
    preprocess.process_item_properties_graph(ratings, graph, out)
    assert np.array_equal(np.loadtxt(out, dtype=np.int64, delimiter='\t').reshape(-1, 3), z['filter_output'])


def test_experiment_grid_expansion_reproduces_the_reference_functions_own_output():
    """tests/golden/grid_reference.json: grids shaped like the reference's econfigs and what the REFERENCE's `make_grid` /
    `nested_dict_update` (utils.py:19-99, executed by tests/golden/make_grid_reference_golden.py) made of them — the product's
    `utilities.utils` must expand to the same experiments in the same order and merge nested sections the same way."""
    import copy
    import json
    import os
    from deep_cbrs_amar_renaissance_amd.utilities import utils
    z = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'grid_reference.json')))
    for name, case in z['grids'].items():
        assert utils.make_grid(copy.deepcopy(case['input'])) == case['output'], name
    for case in z['updates']:
        assert utils.nested_dict_update(copy.deepcopy(case['d']), copy.deepcopy(case['u'])) == case['output']
    for case in z['log_keys']:                                       # the parameter names a run is logged under (experiment.py:149)
        got = utils.mlflow_linearize(copy.deepcopy(case['input']))
        assert got == case['output'] and list(got) == list(case['output'])


def test_weights_roundtrip_through_npz(tmp_path):
    """Model.save_weights / load_weights (what mlflow.tensorflow.autolog keeps for the reference, utils.py:108): an .npz keyed by
    Keras-style variable names; a rebuilt model of the same architecture takes the values back, bit for bit, and every weight's
    version counter moves (hoisted tables and captured graphs are invalidated); a different architecture is refused."""
    import torch
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.models import basic
    from tests import helpers
    g = helpers.tiny_graph()
    cfg = dict(embedding_dim=8, n_hiddens=[8, 8], dense_units=[24, 24], clf_units=[48, 48])
    engine.set_seed(1)
    a = basic.BasicGCN(g['adj'], **cfg)
    helpers.randomize_biases(a, seed=2)
    names = [n for n, _ in a.keras_variable_names()]
    assert len(names) == len(set(names)) == len(list(a.parameters()))
    assert 'sequential_gnn/embeddings:0' in names and 'gcn_conv_1/kernel:0' in names and 'dense/kernel:0' in names and 'dense_6/bias:0' in names
    path = str(tmp_path / 'weights')
    a.save_weights(path)
    engine.set_seed(99)
    b = basic.BasicGCN(g['adj'], **cfg)
    assert not all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
    before = b.weights_version
    b.load_weights(path)
    assert all(torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters())) and b.weights_version != before
    other = basic.BasicGCN(g['adj'], **dict(cfg, n_hiddens=[8, 8, 8]))
    with pytest.raises(ValueError):
        other.load_weights(path)


def test_bench_profile_bookkeeping(tmp_path, monkeypatch):
    """bench.py applies counter-derived numbers (roofline.traffic, the L2 request floor, the spread over boxes) only when they belong to
    the CURRENT kernel sources, the same kernel form and the same scale — never silently."""
    import json
    import importlib
    bench = importlib.import_module('bench')
    sha = bench.csrc_sha()
    assert len(sha) == 16 and sha == bench.csrc_sha()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    os.makedirs(tmp_path / 'profiles')
    monkeypatch.setattr(bench, 'csrc_sha', lambda: sha)
    assert bench.pmc_profile(64, 'lt')[0] is None and 'missing' in bench.pmc_profile(64, 'lt')[1]
    good = {'scale': 64, 'kind': 'lt', 'csrc_sha': sha, 'traffic_bytes_per_launch': 1.0, 'TCC': {'REQ': 2.0}}
    json.dump(good, open(tmp_path / 'profiles' / 'spmm_pmc_latest.json', 'w'))
    pmc, src = bench.pmc_profile(64, 'lt')
    assert pmc == good and sha in src
    assert bench.pmc_profile(256, 'lt')[0] is None and bench.pmc_profile(64, 'xs')[0] is None          # other scale / kernel form
    json.dump(dict(good, csrc_sha='0' * 16), open(tmp_path / 'profiles' / 'spmm_pmc_latest.json', 'w'))
    pmc, src = bench.pmc_profile(64, 'lt')
    assert pmc is None and 'stale' in src                                                              # kernels changed since the profile
    assert bench.value_spread(64) is None
    json.dump({'scale': 64, 'csrc_sha': sha, 'boxes': 3, 'min_ms_per_step': 1.0, 'max_ms_per_step': 1.1}, open(tmp_path / 'profiles' / 'r3_bench_repeats.json', 'w'))
    assert bench.value_spread(64)['stale'] is False and bench.value_spread(256) is None
    json.dump({'scale': 64, 'csrc_sha': 'x', 'boxes': 3}, open(tmp_path / 'profiles' / 'r3_bench_repeats.json', 'w'))
    assert bench.value_spread(64)['stale'] is True


def test_dense_bwd_plan_host_functions():
    """amar_dense_bwd_groups / amar_dense_bwd_workspace_floats are host-only: how a reverse-pass call over M rows is cut (no GPU needed).
    At most 64 partials reach the consumer, one per 64-row tile up to 4 096 rows; the workspace holds them and, where a fold launch
    follows, the raw partials of every workgroup behind them; both grow monotonically with M."""
    import ctypes
    from deep_cbrs_amar_renaissance_amd import capi
    lib = capi.load()
    prev_g, prev_w = 0, 0
    for M in (1, 63, 64, 65, 1024, 4096, 4097, 9228, 16384, 40000, 590592, 600001, 5_000_000):
        g = lib.amar_dense_bwd_groups(M)
        tiles = -(-M // 64)
        assert 1 <= g <= 64 and (g == tiles if tiles <= 64 else g > 32), (M, g)
        w = lib.amar_dense_bwd_workspace_floats(M, 16, 8)
        assert w >= 4 + g * (16 * 8 + 8) and (tiles <= 64) == (w == 4 + g * (16 * 8 + 8)), (M, w)
        assert g >= prev_g or tiles > 64                                # (past 64 tiles the count is ceil(launch groups / fold): not monotone, still <= 64)
        assert w >= prev_w
        prev_g, prev_w = g, w
    assert lib.amar_dense_bwd_groups(-1) < 0 and lib.amar_dense_bwd_workspace_floats(10, 4, 0) < 0
    # the stack reverse pass: one partial per 64-row tile and layer
    dims = (ctypes.c_int32 * 3)(24, 16, 8)
    g = lib.amar_dense_stack_bwd_groups(1024)
    assert g in (16, 64) and lib.amar_dense_stack_bwd_groups(4096) == 64 and lib.amar_dense_stack_bwd_groups(1) == 1
    assert lib.amar_dense_stack_bwd_workspace_floats(1024, 2, dims) == 4 + g * (24 * 16 + 16 + 16 * 8 + 8)
