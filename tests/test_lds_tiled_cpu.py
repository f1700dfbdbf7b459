"""LDS-tiled (LT) SpMM image (utilities/lds_tiled.py): format invariants and a numpy walk of the image that follows
spmm_lt_kernel step by step (stream order, per-wave LDS rows, flags, window pacing table) against scipy A_hat @ X.
Runs on CPU tensors: the builder is plain torch."""
import numpy as np
import pytest
import torch
from scipy import sparse

from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR, _unit_entries, gcn_filter


def _gcn_csr(n_users, n_items, n_ratings, seed, dup=True):
    """A gcn-filtered bipartite DeviceCSR on CPU tensors with its factors (dinv, mult), like gcn_filter_device makes."""
    rng = np.random.default_rng(seed)
    u = rng.integers(0, n_users, n_ratings)
    i = rng.integers(0, n_items, n_ratings) + n_users          # duplicates on purpose: multiplicities > 1
    if not dup:
        key = np.unique(u * (n_users + n_items) + i)
        u, i = key // (n_users + n_items), key % (n_users + n_items)
    n = n_users + n_items
    a = sparse.coo_matrix((np.ones(len(u), np.float32), (u, i)), shape=(n, n))
    sym = (a + a.T).tocsr()
    sym.sum_duplicates()
    c = (sym + sparse.identity(n, dtype=np.float32, format='csr')).tocsr()
    c.sort_indices()
    deg = np.asarray(c.sum(1)).ravel()
    dinv = np.power(deg, -0.5).astype(np.float32)
    csr = DeviceCSR(torch.from_numpy(c.indptr.astype(np.int32)), torch.from_numpy(c.indices.astype(np.int32)),
                    torch.from_numpy(((dinv[np.repeat(np.arange(n), np.diff(c.indptr))] * c.data) * dinv[c.indices]).astype(np.float32)),
                    (n, n), gcn_filtered=True, dinv=torch.from_numpy(dinv), mult=torch.from_numpy(c.data.astype(np.int32)))
    return csr, gcn_filter(sym)


def walk(lt, xs, rw=None):
    """What spmm_lt_kernel computes, in its order: per tile, per wave, steps of EPS words; implicit pairs folded into the
    previous slot, plain read-add-writes (which must hit distinct LDS rows), then the flagged adds; epilogue over the
    row's virtual rows."""
    F = lt.F
    eps, rw, cbits = lds_tiled.geometry(F, rw)
    lmask = (1 << (rw - 1).bit_length()) - 1
    W = lds_tiled.WAVES
    spr = max(1, 16 // (F // 4))
    words = lt.words.numpy().astype(np.int64) & 0xffffffff
    n_rows = lt.shape[0]
    y = np.zeros((n_rows, F), np.float32)
    tb, vstart, vcount = lt.tile_row0.numpy(), lt.vstart.numpy(), lt.vcount.numpy()
    stats = {'pairs': 0, 'flagged': 0}
    for t in range(lt.n_tiles):
        r0, nr = tb[t], tb[t + 1] - tb[t]
        assert vcount[t] <= W * (rw - 1) and vstart[r0] == 0
        tile = np.zeros((W * rw, F), np.float32)
        for w in range(W):
            tab = lt.wsteps[t, w].numpy()
            nwin = int(lt.n_win[t])
            total = tab[nwin]
            assert (np.diff(tab[:nwin + 1]) >= 0).all() and tab[0] == 0      # (total: the steps that hold entries; the stream is padded to whole chunks)
            beg = int(lt.stream_start[t * W + w])
            assert beg % lds_tiled.CHUNK == 0
            for k in range(total):
                ws = words[beg + k * eps: beg + (k + 1) * eps]
                lrow = (ws >> cbits) & lmask
                col = ws & ((1 << cbits) - 1)
                flag = (ws >> 31).astype(bool)
                vals = xs[col].copy()
                prev_same = np.zeros(eps, bool)
                prev_same[1:] = lrow[1:] == lrow[:-1]
                prev_same[np.arange(eps) % spr == 0] = False
                paired = prev_same & ~flag
                real = lrow != rw - 1                                # PAD words (scratch row) may chain: harmless
                if not getattr(lt, 'pairs', True):                   # AMAR_SPMM_LT_NOPAIRS: the kernel runs no pair logic at all,
                    assert not (paired & real).any(), "an image built without pairs must flag every repeat of a step"
                    paired &= False                                  # ... so PAD words are plain adds to the scratch row
                for j in np.where(paired)[0]:
                    assert not real[j] or (not paired[j - 1] and not flag[j - 1]), "a pair folds into a plain first entry"
                    vals[j - 1] += xs[col[j]]
                plain = ~flag & ~paired
                assert len(set(lrow[plain & real])) == (plain & real).sum(), "plain entries of a step must hit distinct rows"
                ldsrow = w * rw + lrow                               # wave-major LDS rows (csrc lt_lds_row)
                for j in np.where(plain)[0]:
                    tile[ldsrow[j]] += vals[j]
                for j in np.where(flag)[0]:
                    assert lrow[j] in lrow[plain], "a flagged entry follows a plain entry of the same row in its step"
                    tile[ldsrow[j]] += vals[j]
                stats['pairs'] += int((paired & real).sum())
                stats['flagged'] += int(flag.sum())
        for lr in range(nr):
            v0 = vstart[r0 + lr]
            v1 = vstart[r0 + lr + 1] if lr + 1 < nr else vcount[t]
            for v in range(v0, v1):
                y[r0 + lr] += tile[(v % W) * rw + v // W]
    assert stats['pairs'] == lt.n_pairs and stats['flagged'] == lt.n_flagged
    d, sc, off = lt.diag.numpy(), lt.row_scale.numpy(), lt.diag_offset
    return sc[:, None] * (d[:, None] * xs[off:off + n_rows] + y)


@pytest.mark.parametrize('F,n_cu,window', [(8, 4, None), (8, 1, 64), (16, 3, None), (32, 2, 16), (4, 2, None)])
def test_lt_image_walk_matches_scipy(F, n_cu, window):
    csr, a_hat = _gcn_csr(700, 300, 30000, seed=F + n_cu)
    rows, cols, diag, off = _unit_entries(csr, True)
    n = csr.shape[0]
    lt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, csr.dinv, csr.dinv, off, window_entries=window, n_cu=n_cu)
    assert lt.n_tiles >= n_cu and int(lt.tile_row0[-1]) == n
    eps, rw, _ = lds_tiled.geometry(F)
    assert int(lt.vcount.max()) <= lds_tiled.WAVES * (rw - 1) and int(lt.vcount.sum()) >= n
    x = np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)
    xs = (csr.dinv.numpy()[:, None] * x).astype(np.float32)
    got = walk(lt, xs)
    want = a_hat.astype(np.float64) @ x.astype(np.float64)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-6)
    # every entry is stored once per unit of multiplicity
    assert lt.n_entries == int(csr.mult.sum()) - int(diag.sum())


@pytest.mark.parametrize('F,pairs', [(16, False), (32, False), (16, True), (32, True), (8, False)])
def test_lt_image_without_pairs(F, pairs):
    """The wide-row form of the image: no implicit pairs (every repeat of a step flagged: AMAR_SPMM_LT_NOPAIRS, the default from
    F = 16 on) — same product, same invariants."""
    csr, a_hat = _gcn_csr(500, 200, 20000, seed=F + int(pairs))
    rows, cols, diag, off = _unit_entries(csr, True)
    n = csr.shape[0]
    lt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, csr.dinv, csr.dinv, off, n_cu=3, pairs=pairs)
    assert lt.pairs == pairs and (pairs or lt.n_pairs == 0)
    x = np.random.default_rng(2).standard_normal((n, F)).astype(np.float32)
    xs = (csr.dinv.numpy()[:, None] * x).astype(np.float32)
    np.testing.assert_allclose(walk(lt, xs), a_hat.astype(np.float64) @ x.astype(np.float64), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('F,n_cu,window,sub', [(8, 4, 256, 64), (8, 1, 2048, 512), (8, 3, None, None), (16, 2, 512, 128)])
def test_lt_image_deferred_repeats(F, n_cu, window, sub):
    """layout='defer' (round 4): a window's list keeps column order by sub-window and moves only the repeats of a virtual row behind
    the first occurrences; built without pairs.  Same product, same invariants; fewer flagged entries than the dealt image of
    single-step windows, which is what the form is for."""
    csr, a_hat = _gcn_csr(700, 300, 30000, seed=F + n_cu)
    rows, cols, diag, off = _unit_entries(csr, True)
    n = csr.shape[0]
    lt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, csr.dinv, csr.dinv, off, window_entries=window, n_cu=n_cu, pairs=False,
                                  layout='defer', sub_window=sub)
    assert not lt.pairs and lt.n_pairs == 0
    x = np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)
    xs = (csr.dinv.numpy()[:, None] * x).astype(np.float32)
    np.testing.assert_allclose(walk(lt, xs), a_hat.astype(np.float64) @ x.astype(np.float64), rtol=2e-5, atol=2e-6)
    if window is not None and window >= 4 * (sub or 0):
        dealt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, csr.dinv, csr.dinv, off, window_entries=sub, n_cu=n_cu, pairs=False)
        assert lt.n_flagged < dealt.n_flagged


@pytest.mark.parametrize('F,n_cu,window,layout', [(8, 4, 64, 'deal'), (8, 2, None, 'deal'), (8, 3, 256, 'defer'), (16, 2, 64, 'deal')])
def test_lt_image_spread_repeats(F, n_cu, window, layout):
    """spread=3 (round 4): repeats of a row inside a step trade places with entries a few steps down the stream; the image stays
    a permutation of the same entries (same product), and fewer entries are left for the LDS-atomic path."""
    csr, a_hat = _gcn_csr(700, 300, 30000, seed=F + n_cu)
    rows, cols, diag, off = _unit_entries(csr, True)
    n = csr.shape[0]
    kw = dict(window_entries=window, n_cu=n_cu, pairs=False, layout=layout, sub_window=64 if layout == 'defer' else None)
    plain = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, csr.dinv, csr.dinv, off, **kw)
    lt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, csr.dinv, csr.dinv, off, spread=3, **kw)
    assert lt.n_entries == plain.n_entries and lt.words.numel() == plain.words.numel()
    assert lt.n_flagged < plain.n_flagged or plain.n_flagged == 0
    x = np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)
    xs = (csr.dinv.numpy()[:, None] * x).astype(np.float32)
    np.testing.assert_allclose(walk(lt, xs), a_hat.astype(np.float64) @ x.astype(np.float64), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('F,n_cu,window', [(8, 4, 1024), (8, 2, 512), (16, 3, 512), (8, 1, 4096)])
def test_lt_image_windows_by_count(F, n_cu, window):
    """layout='count' (round 4): every wave runs the same number of steps between two barriers — a window is a fixed number of
    the wave's own entries, not a column range of the tile.  Same product; the window table is k * S steps for every wave."""
    csr, a_hat = _gcn_csr(700, 300, 30000, seed=F + n_cu)
    rows, cols, diag, off = _unit_entries(csr, True)
    n = csr.shape[0]
    lt = lds_tiled.LdsTiled.build(rows, cols, n, n, F, diag, csr.dinv, csr.dinv, off, window_entries=window, n_cu=n_cu, pairs=False,
                                  layout='count', spread=2)
    eps = lds_tiled.geometry(F)[0]
    S = max(1, window // (lds_tiled.WAVES * eps))
    for t in range(lt.n_tiles):
        nwin = int(lt.n_win[t])
        for w in range(lds_tiled.WAVES):
            tab = lt.wsteps[t, w].numpy()
            total = int(tab[nwin])
            full = [k * S for k in range(nwin + 1) if k * S <= total - S]
            assert list(tab[:len(full)]) == full                    # (the last windows of a stream stop at its own length)
    x = np.random.default_rng(1).standard_normal((n, F)).astype(np.float32)
    xs = (csr.dinv.numpy()[:, None] * x).astype(np.float32)
    np.testing.assert_allclose(walk(lt, xs), a_hat.astype(np.float64) @ x.astype(np.float64), rtol=2e-5, atol=2e-6)


def test_lt_image_heavy_row_and_empty_rows():
    """One row holding a third of all entries (many occurrences per window -> ranks, flags) and trailing empty rows."""
    n = 600
    rng = np.random.default_rng(7)
    r = np.concatenate([np.full(4000, 5), rng.integers(0, 300, 8000)])
    c = np.concatenate([rng.integers(300, 500, 4000), rng.integers(300, 500, 8000)])
    rows = torch.from_numpy(np.concatenate([r, c]).astype(np.int64))
    cols = torch.from_numpy(np.concatenate([c, r]).astype(np.int64))
    diag = torch.ones(n)
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32))
    lt = lds_tiled.LdsTiled.build(rows, cols, n, n, 8, diag, scale, scale, 0, n_cu=2, split=64)
    assert int(lt.vcount.sum()) > n                                 # the heavy row was cut into virtual rows
    xs = rng.standard_normal((n, 8)).astype(np.float32)
    got = walk(lt, xs)
    a = sparse.coo_matrix((np.ones(len(rows)), (rows.numpy(), cols.numpy())), shape=(n, n)).tocsr()
    want = scale.numpy()[:, None].astype(np.float64) * (xs.astype(np.float64) + a @ xs.astype(np.float64))
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-5)


def test_lt_rejects_too_many_columns():
    assert not lds_tiled.supported(8, (1 << 23) + 1)
    assert lds_tiled.supported(8, 1 << 23) and lds_tiled.supported(16, 1 << 24)
    with pytest.raises(ValueError):
        lds_tiled.LdsTiled.build(torch.zeros(1, dtype=torch.int64), torch.zeros(1, dtype=torch.int64), 4, (1 << 23) + 1, 8,
                                 torch.ones(4), torch.ones(4), torch.ones(4))


def _uip_entries(seed):
    """users | items | many low-degree property rows (duplicate links kept), symmetrised unit entries."""
    rng = np.random.default_rng(seed)
    nu, ni, npr = 90, 60, 400
    u, i = rng.integers(0, nu, 2500), rng.integers(0, ni, 2500) + nu
    it, pr = rng.integers(0, ni, 900) + nu, rng.integers(0, npr, 900) + nu + ni
    r, c = np.concatenate([u, it]), np.concatenate([i, pr])
    return np.concatenate([r, c]), np.concatenate([c, r]), nu + ni + npr, (nu, nu + ni)


@pytest.mark.parametrize('F,rw', [(8, 12), (8, None), (16, 10)])
def test_lt_image_three_node_types_and_small_tiles(F, rw):
    """A user-item-property graph with forced breaks at the type boundaries, also with a tile far smaller than the plain
    sum's (the GAT mode's geometry argument, here tiny so that the LDS capacity binds): the walk still gives the product,
    no tile straddles a boundary, and the property rows — whose COUNT needs more tiles than their share of the entries —
    get those tiles without the split being raised for everyone."""
    rows, cols, n, breaks = _uip_entries(F)
    rng = np.random.default_rng(1)
    diag = torch.from_numpy(rng.integers(0, 3, n).astype(np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32))
    lt = lds_tiled.LdsTiled.build(torch.from_numpy(rows), torch.from_numpy(cols), n, n, F, diag, scale, None, 0, n_cu=4, split=32,
                                  row_breaks=breaks, rw=rw, split_growth=1.25)
    xs = rng.standard_normal((n, F)).astype(np.float32)
    got = walk(lt, xs, rw)
    a = sparse.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n)).tocsr()
    want = scale.numpy()[:, None].astype(np.float64) * (diag.numpy()[:, None] * xs.astype(np.float64) + a @ xs.astype(np.float64))
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-5)
    tb = lt.tile_row0.numpy()
    assert all(b in tb for b in breaks), "a tile straddles a node-type boundary"
    if rw is not None:
        vmax = lds_tiled.WAVES * (rw - 1)
        assert int((tb[:-1] >= breaks[1]).sum()) >= -(-400 // vmax)
        deg = np.bincount(rows, minlength=n)
        heavy = int(np.argmax(deg[:breaks[1]]))                     # the heaviest user / item row is cut at least every 32 entries (never coarser)
        t = int(np.searchsorted(tb, heavy, side='right') - 1)
        vs = lt.vstart.numpy()
        k = (vs[heavy + 1] if heavy + 1 < tb[t + 1] else int(lt.vcount[t])) - vs[heavy]
        assert -(-deg[heavy] // 32) <= k <= -(-deg[heavy] // 16)


def test_lt_image_row_block_with_column_offset():
    """A row block of a larger matrix (multi-GPU partition): rows [lo, hi) against all columns, own rows at column offset lo."""
    rows, cols, n, _ = _uip_entries(3)
    lo, hi = 40, 130
    sel = (rows >= lo) & (rows < hi)
    rng = np.random.default_rng(2)
    diag = torch.from_numpy(rng.integers(0, 2, hi - lo).astype(np.float32))
    scale = torch.from_numpy(rng.uniform(0.5, 1.5, hi - lo).astype(np.float32))
    lt = lds_tiled.LdsTiled.build(torch.from_numpy(rows[sel] - lo), torch.from_numpy(cols[sel]), hi - lo, n, 8, diag, scale, None, lo, n_cu=2)
    xs = rng.standard_normal((n, 8)).astype(np.float32)
    got = walk(lt, xs)
    a = sparse.coo_matrix((np.ones(int(sel.sum())), (rows[sel] - lo, cols[sel])), shape=(hi - lo, n)).tocsr()
    want = scale.numpy()[:, None].astype(np.float64) * (diag.numpy()[:, None] * xs[lo:hi].astype(np.float64) + a @ xs.astype(np.float64))
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-5)


def test_lt_geometry_tables():
    assert lds_tiled.geometry(8) == (32, 256, 23) and lds_tiled.geometry(16) == (16, 128, 24) and lds_tiled.geometry(32) == (8, 64, 25)
    assert lds_tiled.geometry(8, 216) == (32, 216, 23) and lds_tiled.geometry(16, 124) == (16, 124, 24)
    for C, rw in lds_tiled.GAT_ROWS_PER_WAVE.items():               # one workgroup's LDS: rows + (sum w, s_self) + the index ring
        assert lds_tiled.WAVES * rw * (C + 2) * 4 + lds_tiled.WAVES * lds_tiled.CHUNK * 4 + 32 <= 160 * 1024
