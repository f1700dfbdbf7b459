"""LDS-tiled ("LT") image of a value-free sparse matrix for `amar_spmm_lt_f32` (include/amar_hip.h).

Why.  The XCD-sliced kernels (utilities/math.py:XcdSliced) pay ONE L2 request per gathered row: entries are
sorted by row, so the columns a wave touches are scattered and a 32-byte row (F = 8) costs a whole 128-byte line
fill.  A CU sustains ~0.43 such requests per clock (profiles/r1_exp_gather_frontend.txt): 0.207 ms for the
55.9 M gathers of one ml1m(s=64) layer, whatever the kernel does around them.

Here a workgroup owns a TILE of up to W.(RW-1) consecutive rows of Y (one tile per CU: the whole [rows, F] fp32
tile sits in 128 KB of LDS) and walks ALL of the tile's non-zeros in COLUMN order, cut into "windows" of
`window_entries` consecutive (by column) entries.  The tile's 8 waves own disjoint row ranges of it (so the LDS
accumulation needs no atomics) and step through the windows together (one s_barrier per window): the few hundred
neighbouring columns of a window are fetched into the CU's L1 once and then hit by every further entry of the
window — 2-4 entries share a 128-byte line on ml1m(s=64) — instead of one L2 request per entry.  Each tile is
finished inside its workgroup (diag term, row scale, bias / ReLU / next X.W epilogue): no partial-sum round trip
and no second launch.

Layout (all int32, device):
    words         one 32-bit word per entry:  flag << 31 | lrow << cbits | column
                  (lrow = row inside the owning wave's range, cbits = 31 - log2(RW)); each (tile, wave) stream is
                  contiguous, starts at a multiple of 256 entries and is padded to one with PAD words
                  (lrow = RW-1: a scratch row of the LDS tile, column 0).  Inside a (tile, wave, window) list the
                  entries are ordered (occurrence of the row in the list, row): the EPS = 64/(F/4) entries of one
                  wave-instruction ("step") then hit distinct LDS rows, and the read-add-write needs no conflict
                  handling; the rare entry whose row already occurs earlier in its step carries flag = 1 and is
                  added with an LDS float atomic after the step's plain read-add-write.
    stream_start  [T*W]              first word of every (tile, wave) stream
    wsteps        [T, W, maxwin+1]   step index (stream-relative) at which window w of the tile begins for the
                                     wave; entry n_win[t] holds the stream's padded step count
    tile_row0     [T+1]              row range of every tile;  n_win [T]
Entries are value-free (weight 1; an entry of multiplicity c is stored c times): A = S C S exactly as the value-free
XS image, with `diag`, `row_scale`, `col_scale`, `diag_offset` of the same meaning.
"""
import numpy as np
import torch

WAVES = 8                      # waves per workgroup of spmm_lt_kernel (one workgroup per CU)
TILE_BYTES = 128 << 10         # LDS bytes of the Y tile
CHUNK = 256                    # entries per index-stream chunk (64 lanes x 16 bytes)
N_CU = 256


def geometry(F):
    """(entries per step, rows per wave in the LDS tile, column bits) for feature width F."""
    eps = 64 // (F // 4)
    rw = TILE_BYTES // (4 * F * WAVES)
    lbits = rw.bit_length() - 1
    return eps, rw, 31 - lbits


def supported(F, n_cols):
    return F in (4, 8, 16, 32) and n_cols <= (1 << geometry(F)[2])


class LdsTiled:
    def __init__(self, F, words, stream_start, wsteps, tile_row0, n_win, maxwin1, diag, row_scale, col_scale, diag_offset, shape,
                 window_entries, n_entries, n_flagged):
        self.F, self.words, self.stream_start, self.wsteps = F, words, stream_start, wsteps
        self.tile_row0, self.n_win, self.maxwin1 = tile_row0, n_win, int(maxwin1)
        self.diag, self.row_scale, self.col_scale, self.diag_offset = diag, row_scale, col_scale, int(diag_offset)
        self.shape = tuple(shape)
        self.n_tiles = int(n_win.numel())
        self.window_entries, self.n_entries, self.n_flagged = int(window_entries), int(n_entries), int(n_flagged)

    @classmethod
    def build(cls, rows, cols, n_rows, n_cols, F, diag, row_scale, col_scale, diag_offset=0, window_entries=None, n_cu=N_CU):
        """`rows`/`cols`: int64 device tensors of the unit-weight off-diagonal entries (multiplicities expanded)."""
        dev = rows.device
        W = WAVES
        eps, rw, cbits = geometry(F)
        if not supported(F, n_cols):
            raise ValueError("LT image: F = {} with {} columns is outside the packed word's range".format(F, n_cols))
        if window_entries is None:
            window_entries = 2 * W * eps
        rmax = W * (rw - 1)
        m = int(rows.numel())
        # a. row tiles: contiguous, (nearly) equal entry counts, at most rmax rows
        deg = torch.bincount(rows, minlength=n_rows) if m else torch.zeros(n_rows, dtype=torch.int64, device=dev)
        cum = np.concatenate([[0], np.cumsum(deg.cpu().numpy().astype(np.int64))])
        n_rounds = max(1, -(-n_rows // (n_cu * rmax)))
        e_t = max(1, -(-m // (n_cu * n_rounds)))
        tb = [0]
        while tb[-1] < n_rows:
            r0 = tb[-1]
            r1 = int(np.searchsorted(cum, cum[r0] + e_t, side='right')) - 1
            tb.append(min(max(r1, r0 + 1), r0 + rmax, n_rows))
        T = len(tb) - 1
        tb_t = torch.tensor(tb, dtype=torch.int64, device=dev)
        nr_t = tb_t[1:] - tb_t[:-1]
        blk_t = (nr_t + W - 1) // W                                   # rows per wave of each tile (<= rw - 1)
        tile = torch.searchsorted(tb_t, rows, right=True) - 1
        lr = rows - tb_t[tile]
        wave = lr // blk_t[tile]
        lrow = lr - wave * blk_t[tile]
        # b. windows: position of the entry in its tile's column-sorted list
        order = torch.argsort(tile * n_cols + cols)
        tile_cnt = torch.bincount(tile, minlength=T)
        tile_start = torch.cumsum(tile_cnt, 0) - tile_cnt
        win = torch.empty(m, dtype=torch.int64, device=dev)
        win[order] = (torch.arange(m, device=dev) - tile_start[tile[order]]) // window_entries
        del order
        n_win = (tile_cnt + window_entries - 1) // window_entries
        maxwin = max(1, int(n_win.max())) if T else 1
        # c. occurrence rank of the entry among the entries of its (tile, window, row)
        tw = tile * W + wave
        key = (tw * maxwin + win) * rw + lrow
        order = torch.argsort(key, stable=True)
        ks = key[order]
        idx = torch.arange(m, device=dev)
        first = torch.ones(m, dtype=torch.bool, device=dev)
        if m > 1:
            first[1:] = ks[1:] != ks[:-1]
        run_start = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values
        rank = torch.empty(m, dtype=torch.int64, device=dev)
        rank[order] = idx - run_start
        del order, ks, first, run_start
        max_rank = int(rank.max()) + 1 if m else 1
        if T * W * maxwin * max_rank * rw >= (1 << 62):
            raise ValueError("LT image: sort key overflow")
        # d. final order: (tile, wave, window, rank, lrow); streams padded to whole chunks
        key = ((tw * maxwin + win) * max_rank + rank) * rw + lrow
        order = torch.argsort(key)
        del key, rank
        cnt_tw = torch.bincount(tw, minlength=T * W)
        len_tw = (cnt_tw + CHUNK - 1) // CHUNK * CHUNK
        stream_start = torch.cumsum(len_tw, 0) - len_tw
        total = int(len_tw.sum())
        if total >= (1 << 31):
            raise ValueError("LT image: more than 2^31 entries")
        cnt_start = torch.cumsum(cnt_tw, 0) - cnt_tw
        tw_s = tw[order]
        dest = stream_start[tw_s] + (idx - cnt_start[tw_s])
        word = (lrow[order] << cbits) | cols[order]
        # e. flag the entries whose LDS row already occurs earlier in their step (the kernel adds them atomically)
        lrow_s = lrow[order]
        k4 = (dest // eps) * rw + lrow_s
        o4 = torch.argsort(k4, stable=True)
        k4s = k4[o4]
        dup = torch.zeros(m, dtype=torch.bool, device=dev)
        if m > 1:
            dup[o4[1:]] = k4s[1:] == k4s[:-1]
        n_flagged = int(dup.sum())
        word = torch.where(dup, word - (1 << 31), word)              # bit 31 as two's complement
        words = torch.full((max(total, 1),), (rw - 1) << cbits, dtype=torch.int32, device=dev)
        words[dest] = word.to(torch.int32)
        del o4, k4, k4s, dup, word, dest, lrow_s, tw_s
        # f. step at which each window begins for each wave
        cnt = torch.bincount(tw * maxwin + win, minlength=T * W * maxwin).view(T, W, maxwin)
        e0 = torch.cumsum(cnt, 2) - cnt                               # entries of the stream before the window
        wsteps = torch.empty((T, W, maxwin + 1), dtype=torch.int64, device=dev)
        wsteps[:, :, :maxwin] = e0 // eps
        end_steps = (len_tw // eps).view(T, W, 1)
        beyond = torch.arange(maxwin + 1, device=dev).view(1, 1, -1) >= n_win.view(T, 1, 1)
        wsteps = torch.where(beyond, end_steps.expand(T, W, maxwin + 1), wsteps)
        return cls(F, words, stream_start.to(torch.int32), wsteps.to(torch.int32).contiguous(),
                   tb_t.to(torch.int32), n_win.to(torch.int32), maxwin + 1, diag, row_scale, col_scale, diag_offset,
                   (n_rows, n_cols), window_entries, m, n_flagged)
