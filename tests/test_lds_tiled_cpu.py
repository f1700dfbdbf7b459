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


def walk(lt, xs):
    """What spmm_lt_kernel computes, in its order: per tile, per wave, steps of EPS words; implicit pairs folded into the
    previous slot, plain read-add-writes (which must hit distinct LDS rows), then the flagged adds; epilogue over the
    row's virtual rows."""
    F = lt.F
    eps, rw, cbits = lds_tiled.geometry(F)
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
            assert total % (lds_tiled.CHUNK // eps) == 0 and (np.diff(tab[:nwin + 1]) >= 0).all() and tab[0] == 0
            beg = int(lt.stream_start[t * W + w])
            assert beg % lds_tiled.CHUNK == 0
            for k in range(total):
                ws = words[beg + k * eps: beg + (k + 1) * eps]
                lrow = (ws >> cbits) & (rw - 1)
                col = ws & ((1 << cbits) - 1)
                flag = (ws >> 31).astype(bool)
                vals = xs[col].copy()
                prev_same = np.zeros(eps, bool)
                prev_same[1:] = lrow[1:] == lrow[:-1]
                prev_same[np.arange(eps) % spr == 0] = False
                paired = prev_same & ~flag
                real = lrow != rw - 1                                # PAD words (scratch row) may chain: harmless
                for j in np.where(paired)[0]:
                    assert not real[j] or (not paired[j - 1] and not flag[j - 1]), "a pair folds into a plain first entry"
                    vals[j - 1] += xs[col[j]]
                plain = ~flag & ~paired
                assert len(set(lrow[plain & real])) == (plain & real).sum(), "plain entries of a step must hit distinct rows"
                ldsrow = lrow * W + (w + lrow) % W
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
                y[r0 + lr] += tile[(v // W) * W + (v % W + v // W) % W]
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
