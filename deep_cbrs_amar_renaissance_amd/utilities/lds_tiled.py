"""LDS-tiled ("LT") image of a value-free sparse matrix for `amar_spmm_lt_f32` (include/amar_hip.h).

Why.  The XCD-sliced kernels (utilities/math.py:XcdSliced) pay ONE L2 request per gathered row: entries are
sorted by row, so the columns a wave touches are scattered and a 32-byte row (F = 8) costs a whole 128-byte line
fill.  A CU sustains ~0.43 such requests per clock (profiles/r1_exp_gather_frontend.txt): 0.207 ms for the
55.9 M gathers of one ml1m(s=64) layer, whatever the kernel does around them.

Here a workgroup owns a TILE of consecutive rows of Y (one tile per CU: the tile's sums sit in 128 KB of LDS) and
walks ALL of the tile's non-zeros in COLUMN order, cut into "windows" of `window_entries` consecutive (by column)
entries.  The tile's W = 16 waves step through the windows together (one s_barrier per window): the few hundred
neighbouring columns of a window are fetched into the CU's L1 once and then hit by every further entry of the
window — 2-4 entries share a 128-byte line on ml1m(s=64) — instead of one L2 request per entry.  Each tile is
finished inside its workgroup (diag term, row scale, bias / ReLU / next X.W epilogue): no partial-sum round trip
and no second launch.

Accumulation without atomics (LDS float atomics run at ~3 clocks per LANE on gfx950).  The unit of ownership is
the VIRTUAL ROW: a row with more than `split` entries is cut into ceil(d / split) virtual rows (its entries dealt
round-robin in column order), virtual row v of a tile belongs to wave v % W and sits in LDS row
(v // W) * W + (v % W + v // W) % W.  A wave only ever touches its own LDS rows, so it adds with a plain
ds_read_b128 / add / ds_write_b128.  Within a wave-instruction ("step": EPS = 64/(F/4) entries) the same virtual
row must not be read-modified-written twice; the image therefore spreads the entries of a (wave, window) list over
the steps the list covers (same-row entries go to different steps first), and what still collides is either
    * an implicit PAIR: the entry in the slot right after the row's first entry of the step (same 16-lane DPP row):
      the kernel recognises it (same lrow as the previous slot, flag clear), adds its gathered values to the first
      entry's registers with one DPP shift and skips its LDS update; or
    * FLAGGED (bit 31): added with ds_add_f32 after the step's plain updates (rare: < 1 % on ml1m graphs).

Layout (all int32, device):
    words         one 32-bit word per entry:  flag << 31 | lrow << cbits | column   (lrow = v // W,
                  cbits = 31 - log2(RW), RW = rows per wave in the LDS tile); each (tile, wave) stream is contiguous,
                  starts at a multiple of 256 entries and is padded to one with PAD words (lrow = RW-1, a scratch
                  row of the wave; column 0)
    stream_start  [T*W]              first word of every (tile, wave) stream
    wsteps        [T, W, maxwin+1]   step index (stream-relative) at which window w of the tile begins for the
                                     wave; entry n_win[t] holds the number of steps that hold entries of the stream
    tile_row0     [T+1]              row range of every tile;  n_win [T]
    vstart        [n_rows+1]         first virtual row of every row, numbered from 0 inside its tile
                                     (vstart[tile_row0[t]] = 0; the last row of a tile ends at vstart_end[t])
    vcount        [T]                virtual rows of every tile
Entries are value-free (weight 1; an entry of multiplicity c is stored c times): A = S C S exactly as the value-free
XS image, with `diag`, `row_scale`, `col_scale`, `diag_offset` of the same meaning.
"""
import os

import numpy as np
import torch

WAVES = 16                     # waves per workgroup of spmm_lt_kernel (one workgroup per CU)
TILE_BYTES = 128 << 10         # LDS bytes of the Y tile
CHUNK = 256                    # entries per index-stream chunk (64 lanes x 16 bytes)
N_CU = 256
SPLIT = 128                    # rows longer than this are cut into virtual rows of at most this many entries
BALANCE_PASSES = 3
COST_ENTRY, COST_LINE = 1.68, 1.71   # cycles per entry / per 128-byte line of X a tile touches (fit on ml1m(s=64), F = 8)


GAT_ROWS_PER_WAVE = {8: 216, 16: 124, 32: 64}      # amar_gat_lt_f32: the LDS row also holds (sum of weights, s_self) — csrc lt_gat_rw


def geometry(F, rw=None):
    """(entries per step, rows per wave in the LDS tile, column bits) for feature width F (rw: a smaller tile, GAT mode)."""
    eps = 64 // (F // 4)
    rw = TILE_BYTES // (4 * F * WAVES) if rw is None else int(rw)
    lbits = (rw - 1).bit_length()
    return eps, rw, 31 - lbits


def _run_starts(first, idx):
    """Index of the first element of every element's run (`first` marks run starts of a sorted key sequence)."""
    if idx.numel() == 0:
        return idx
    starts = torch.nonzero(first).view(-1)
    return starts[torch.searchsorted(starts, idx, right=True) - 1]


def _spread_repeats(dest, lrow_s, tw_s, stream_start, cnt_tw, eps, rw, total, passes):
    """Round 4: move the REPEATS of a virtual row inside a step (what an image without pairs sends to the LDS-atomic path: 4 % of
    the entries of ml1m(s=64) with single-step windows) one to three steps down the wave's stream, each trading places with the
    entry in its slot there — if neither then repeats a row of its new step.  The walk stays in column order to within three
    steps of a wave for a few per cent of its entries; whatever still collides afterwards is flagged as before (the flags are
    taken from the final positions, so the result never depends on this heuristic)."""
    m = int(dest.numel())
    if m < 2 or passes <= 0:
        return dest
    dev = dest.device
    idx = torch.arange(m, device=dev)
    stream_end = (stream_start + cnt_tw)[tw_s]                         # first position past the real entries of the entry's stream
    for p in range(passes):
        key = (dest // eps) * rw + lrow_s
        o = torch.argsort(key * eps + dest % eps)
        ko = key[o]
        first = torch.ones(m, dtype=torch.bool, device=dev)
        first[1:] = ko[1:] != ko[:-1]
        rank_o = idx - _run_starts(first, idx)
        sel = torch.nonzero(rank_o >= 1).view(-1)
        if sel.numel() == 0:
            break
        e, r = o[sel], rank_o[sel]
        is_mover = torch.zeros(m, dtype=torch.bool, device=dev)
        is_mover[e] = True
        pp = dest[e] + ((r - 1 + p) % 3 + 1) * eps
        ok = pp < stream_end[e]
        e, pp = e[ok], pp[ok]
        inv = torch.full((total,), -1, dtype=torch.int64, device=dev)
        inv[dest] = idx
        pe = inv[pp]

        def present(k):
            i = torch.searchsorted(ko, k).clamp_(max=m - 1)
            return ko[i] == k
        ok = (pe >= 0) & ~is_mover[pe.clamp(min=0)]
        ok &= ~present((pp // eps) * rw + lrow_s[e]) & ~present((dest[e] // eps) * rw + lrow_s[pe.clamp(min=0)])
        e, pp, pe = e[ok], pp[ok], pe[ok]
        if e.numel() == 0:
            continue
        so = torch.argsort(pp)                                        # one mover per target position
        pps = pp[so]
        keep = torch.ones(pps.numel(), dtype=torch.bool, device=dev)
        keep[1:] = pps[1:] != pps[:-1]
        e, pp, pe = e[so][keep], pps[keep], pe[so][keep]
        back = dest[e].clone()
        dest[e] = pp
        dest[pe] = back
    return dest


def supported(F, n_cols, rw=None):
    return F in (4, 8, 16, 32) and n_cols <= (1 << geometry(F, rw)[2])


class LdsTiled:
    def __init__(self, F, words, stream_start, wsteps, tile_row0, n_win, maxwin1, vstart, vcount, diag, row_scale, col_scale,
                 diag_offset, shape, window_entries, n_entries, n_flagged, n_pairs, rw=None, pairs=True):
        self.F, self.words, self.stream_start, self.wsteps = F, words, stream_start, wsteps
        self.tile_row0, self.n_win, self.maxwin1 = tile_row0, n_win, int(maxwin1)
        self.vstart, self.vcount = vstart, vcount
        self.diag, self.row_scale, self.col_scale, self.diag_offset = diag, row_scale, col_scale, int(diag_offset)
        self.shape = tuple(shape)
        self.rw = int(rw) if rw is not None else geometry(F)[1]        # LDS rows per wave the image was cut for
        self.pairs = bool(pairs)                                       # False: every repeat inside a step is flagged (AMAR_SPMM_LT_NOPAIRS)
        self.n_tiles = int(n_win.numel())
        self.window_entries, self.n_entries, self.n_flagged, self.n_pairs = int(window_entries), int(n_entries), int(n_flagged), int(n_pairs)
        # the waves of a tile meet at a barrier every `pace_every` windows: windows of ONE step per wave keep the dealing span (and
        # with it the L1 footprint) small — 24 M L2 requests per ml1m(s=64) layer against 28 M for two-step windows — while a barrier
        # per four of them costs no more synchronisation than before (0.2136 against 0.2200 ms)
        steps = max(1, self.window_entries // (WAVES * (64 // (F // 4))))
        # F = 32 (a gathered row is a whole 128-byte line: nothing to share in the L1, the barrier only keeps the tile's waves near
        # each other in the L2s): one barrier per four 2 048-entry windows, 0.468 -> 0.430 ms per ml1m(s=64) product (every window 0.468,
        # every second 0.452, every eighth 0.456; F = 16 is flat: profiles/r3_exp_lt_pace_wide.txt)
        wide = F >= 32 and self.rw == geometry(F)[1]
        self.pace_every = int(os.environ.get('AMAR_LT_PACE', 4 if steps == 1 or wide else (2 if steps == 2 else 1)))

    @classmethod
    def build(cls, rows, cols, n_rows, n_cols, F, diag, row_scale, col_scale, diag_offset=0, window_entries=None, n_cu=N_CU,
              split=SPLIT, balance=True, row_breaks=(), rw=None, split_growth=2.0, pairs=None, layout=None, sub_window=None, spread=None, colsort=None):
        """`rows`/`cols`: int64 device tensors of the unit-weight off-diagonal entries (multiplicities expanded).
        `rw`: LDS rows per wave when the tile is smaller than the plain sum's (GAT mode); `split_growth`: factor by which the
        virtual-row length grows while the tiles do not fit the LDS (longer virtual rows repeat more often inside a step).
        `pairs`: whether the repeat in the slot right after a row's first entry of a step is left to the kernel's in-register
        pair (default: F < 16 — at 8 entries per step such repeats are 0.1 % of the entries and the pair logic a third of a step's
        instructions; AMAR_LT_PAIRS=0|1 overrides).  An image without pairs runs with AMAR_SPMM_LT_NOPAIRS.
        `layout`: how a (wave, window) list is laid out over its steps.  'deal' (rounds 2-3): sorted by virtual row and dealt
        slot-major over the steps the list covers, so that a row's repeats land in different steps.  'defer' (round 4): the list
        keeps COLUMN order at the granularity of `sub_window` entries of the tile, and only the REPEATS of a virtual row inside
        the list are moved — the k-th occurrence of a row goes behind all (k-1)-th occurrences — so that a window can span several
        steps per wave (one barrier per window, few repeats inside a step: no implicit pairs needed) while the entries of a step
        still come from one sub-window's column range (what keeps the L1 footprint of a step small).
        `spread`: passes of `_spread_repeats` over the finished layout (images without pairs).
        `row_breaks`: rows at which a tile must end (node-type boundaries of a bipartite / tripartite graph with grouped ids:
        a tile that straddles one walks two column ranges at half the density each and runs ~25 % longer than its peers)."""
        dev = rows.device
        W = WAVES
        rw_arg = rw
        eps, rw, cbits = geometry(F, rw)
        if not supported(F, n_cols, rw):
            raise ValueError("LT image: F = {} with {} columns is outside the packed word's range".format(F, n_cols))
        vmax = W * (rw - 1)                                           # virtual rows a tile can hold
        m = int(rows.numel())
        idx = torch.arange(m, device=dev)
        # a. virtual rows per row, then row tiles: contiguous, (nearly) equal COST, at most vmax virtual rows.  Cost of an entry
        #    = COST_ENTRY + COST_LINE * (128-byte lines of X the tile touches) / (entries of the tile): tiles whose entries share
        #    fewer lines (item rows of a bipartite rating graph: more columns, fewer entries per column) run longer per entry
        #    (measured on ml1m(s=64), F = 8: 2.08 vs 2.44 cycles per entry).  First pass: equal entry counts; second pass: row
        #    weights = entries x the cost per entry of the row's first-pass tile.
        deg = torch.bincount(rows, minlength=n_rows) if m else torch.zeros(n_rows, dtype=torch.int64, device=dev)
        deg_np = deg.cpu().numpy().astype(np.int64)

        breaks = sorted(int(b) for b in row_breaks if 0 < int(b) < n_rows)

        def make_tiles(weight_cum, split_):
            edges = [0] + breaks + [n_rows]
            n_seg = len(edges) - 1
            seg_w = np.array([weight_cum[edges[i + 1]] - weight_cum[edges[i]] for i in range(n_seg)])
            seg_e = np.array([int(deg_np[edges[i]:edges[i + 1]].sum()) for i in range(n_seg)], dtype=np.int64)
            k_row_ = np.maximum((deg_np + split_ - 1) // split_, 1)

            def shares(k_row__):
                # the segments between forced breaks share the tiles in proportion to their weight (at least one each); a segment
                # whose ROW count alone needs more tiles than that share (the four-entry property rows of a user-item-property
                # graph: 275 tiles of 4 080 rows at ml1m(s=64) for 7 % of the entries) takes what the LDS capacity dictates and
                # leaves the rounds of the others alone — its tiles are short and fill in behind the tall ones
                vcum__ = np.concatenate([[0], np.cumsum(k_row__)])
                seg_v = np.array([vcum__[edges[i + 1]] - vcum__[edges[i]] for i in range(n_seg)], dtype=np.int64)
                cap = -(-seg_v // vmax)
                share_ = np.ones(n_seg, dtype=np.int64)
                bound_ = np.zeros(n_seg, dtype=bool)
                free = np.ones(n_seg, dtype=bool)
                while free.any():
                    wanted_free = n_cu * max(1, -(-int(seg_v[free].sum()) // (n_cu * vmax)))
                    w_free = np.where(free, seg_w, 0.0)
                    sh = np.maximum(1, np.floor(w_free / max(w_free.sum(), 1e-30) * wanted_free)).astype(np.int64)
                    while sh[free].sum() < max(wanted_free, int(free.sum())):   # hand the remaining tiles to the segments with the most weight per tile
                        sh[int(np.argmax(np.where(free, seg_w / sh, -1.0)))] += 1
                    bound = free & (cap > sh)
                    share_[free] = sh[free]
                    if not bound.any():
                        break
                    share_[bound] = cap[bound]
                    bound_ |= bound
                    free &= ~bound
                return vcum__, share_, bound_

            vcum_, share, cap_bound = shares(k_row_)
            # Small tiles need fine virtual rows.  A wave's step touches 32 entries; they fall on distinct LDS rows only if the tile
            # offers the wave several times that many virtual rows (~1 000 per tile).  Tall tiles get them from their row count; a
            # segment cut into SMALL tiles of few heavy rows does not (the head of the property rows of a user-item-property graph:
            # tiles of 10-200 rows, 75-190 virtual rows and 19 k entries had 37-66 % of their entries on the LDS-atomic path and took
            # 137 us each, at the END of the launch).  Where a segment's tiles hold E entries, rows are cut into pieces of E / 1 024
            # entries (at least 16, never coarser than the split; no more pieces than one tile holds).
            changed = False
            for i in range(n_seg):
                e_tile = seg_e[i] / max(int(share[i]), 1)
                fine = int(min(split_, max(16, e_tile // 1024)))
                if fine >= split_:
                    continue
                lo, hi = edges[i], edges[i + 1]
                k_fine = np.maximum(k_row_[lo:hi], np.minimum(np.maximum(-(-deg_np[lo:hi] // fine), 1), vmax))
                # ... only where the extra virtual rows cannot push the segment past its tiles (ONE tile more than a round of CUs
                # doubles the launch: GAT C = 8 at ml1m(s=64) went from 256 to 257 tiles and from 0.36 to 0.64 ms): segments whose
                # tile count the LDS capacity dictates anyway, or with a fifth of their capacity to spare afterwards
                if (k_fine > k_row_[lo:hi]).any() and (cap_bound[i] or int(k_fine.sum()) <= 0.8 * int(share[i]) * vmax):
                    k_row_[lo:hi] = k_fine
                    changed = True
            if changed:
                vcum_, share, _ = shares(k_row_)
            wanted_ = int(share.sum())
            tb_ = [0]
            for i in range(len(edges) - 1):
                seg_end, target = edges[i + 1], seg_w[i] / share[i]
                while tb_[-1] < seg_end:
                    r0 = tb_[-1]
                    r1 = int(np.searchsorted(weight_cum, weight_cum[r0] + target * (1 - 1e-9), side='left')) if target > 0 else seg_end   # first row end reaching the target
                    rv = int(np.searchsorted(vcum_, vcum_[r0] + vmax, side='right')) - 1  # last row end within the LDS capacity
                    if rv <= r0:
                        raise ValueError("LT image: a single row needs more than {} virtual rows".format(vmax))
                    tb_.append(min(max(r1, r0 + 1), rv, seg_end))
            return tb_, k_row_, vcum_, wanted_, bool(cap_bound.any())

        # The per-line term holds while the gathered table lives in the L2s (32 MB in all): there a tile whose entries spread over
        # more lines pays for more L1 fills.  Past that size every tile's gathers go to the Infinity Cache and the entries cost the
        # same whatever their spread (in-kernel cycle stamps, tools/exp_lt_stamps.py: F = 16 at ml1m(s=64), 37.8 MB: 3.22 / 3.20
        # cycles per entry for user / item tiles): tiles are then cut by entry count alone (ml1m(s=256), 75 MB: 1.26 -> 1.17 ms per
        # layer) — unless a capacity-bound segment supplies short fill-in tiles: then the staggered finish the line term gives
        # the tall tiles is what lets the short ones start early (user-item-property graph, 55 MB: 0.334 ms with the term, 0.349
        # without, same box).
        big_table = n_cols * F * 4 > (32 << 20)
        cum = np.concatenate([[0.0], np.cumsum(deg_np.astype(np.float64))])
        while True:
            tb, k_row_np, vcum, wanted, fill_in = make_tiles(cum, split)
            line_cost = 0.0 if big_table and not fill_in else COST_LINE
            if os.environ.get('AMAR_LT_LINE_COST'):                   # development switch (A/B of the rule above)
                line_cost = float(os.environ['AMAR_LT_LINE_COST'])
            for _ in range(BALANCE_PASSES if balance and m else 0):
                tb0 = torch.tensor(tb, dtype=torch.int64, device=dev)
                tile0 = torch.searchsorted(tb0, rows, right=True) - 1
                cpl = max(1, 128 // (4 * F))                          # columns per 128-byte line
                n_lines = (n_cols + cpl - 1) // cpl
                lines_t = torch.bincount(torch.unique(tile0 * n_lines + cols // cpl) // n_lines, minlength=len(tb) - 1).cpu().numpy()
                ent_t = torch.bincount(tile0, minlength=len(tb) - 1).cpu().numpy()
                cost_t = COST_ENTRY + line_cost * lines_t / np.maximum(ent_t, 1)
                row_tile = np.searchsorted(np.asarray(tb), np.arange(n_rows), side='right') - 1
                wcum = np.concatenate([[0.0], np.cumsum(deg_np * cost_t[row_tile])])
                tb, k_row_np, vcum, wanted, _ = make_tiles(wcum, split)
                del tb0, tile0
            # the LDS capacity forced extra tiles: cut long rows less finely — where that can help.  When the extra tiles come
            # from the sheer number of ROWS (the 1.1 M four-entry property rows of a user-item-property graph need 275 tiles
            # whatever the split) longer virtual rows only add repeats inside a step: keep the split, take the extra tiles.
            if len(tb) - 1 <= wanted or split >= 4096 or int(vcum[-1]) - n_rows < 0.05 * int(vcum[-1]):
                break
            split = max(split + 1, int(split * split_growth))
        k_row = torch.from_numpy(k_row_np).to(dev)
        T = len(tb) - 1
        tb_t = torch.tensor(tb, dtype=torch.int64, device=dev)
        vcum_t = torch.from_numpy(vcum).to(dev)
        tile = torch.searchsorted(tb_t, rows, right=True) - 1
        vbase_tile = vcum_t[tb_t[:-1]]                                # first virtual row (global numbering) of each tile
        vstart = vcum_t[:-1] - vbase_tile[torch.searchsorted(tb_t, torch.arange(n_rows, device=dev), right=True) - 1] \
            if n_rows else torch.zeros(0, dtype=torch.int64, device=dev)
        vstart = torch.cat([vstart, torch.zeros(1, dtype=torch.int64, device=dev)])
        vcount = vcum_t[tb_t[1:]] - vbase_tile
        if window_entries is None:
            # steps per wave and window by tile height: short tiles (few LDS rows per wave) need longer lists to spread the
            # repeats of a row over several steps, tall tiles prefer the smallest L1 footprint.  ml1m(s), F = 8, ms per launch
            # for 1 / 2 / 4 steps per window: s=16 (~600 rows per tile) .092 / .073 / .067, s=32 (~1 150) .129 / .117 / .124,
            # s=64 (~2 300) .218 / .222 / .238, s=128 .540 / .546 / .570
            avg_v = float(vcount.double().mean()) if T else 0.0
            steps = 1 if avg_v >= 0.4 * vmax and F == 8 else (2 if avg_v >= 0.2 * vmax else 4)
            # F = 16, 32 at s=64: 1 024 entries (.33 / .58 ms) beat 512 / 256 (.35 / .70); round 3, the step without pair logic: F = 32
            # 1 024 / 2 048 / 4 096 entries .493 / .445 / .445 ms, F = 16 .297 / .302 (profiles/r3_exp_lt_nopairs_windows.txt)
            window_entries = max(steps * W * eps, 2048 if F >= 32 and rw_arg is None else (1024 if F >= 16 else 0))   # (GAT geometry: as tuned)
        # b. windows + the virtual row of every entry: both from the tile's column-sorted order
        order = torch.argsort(tile * n_cols + cols)
        tile_cnt = torch.bincount(tile, minlength=T)
        tile_start = torch.cumsum(tile_cnt, 0) - tile_cnt
        win = torch.empty(m, dtype=torch.int64, device=dev)
        win[order] = (idx - tile_start[tile[order]]) // window_entries
        del order
        n_win = (tile_cnt + window_entries - 1) // window_entries
        maxwin = max(1, int(n_win.max())) if T else 1
        order = torch.argsort(rows * n_cols + cols)                   # j-th entry of its row in column order -> virtual row j % k
        row_start = torch.cumsum(deg, 0) - deg
        sub = torch.empty(m, dtype=torch.int64, device=dev)
        sub[order] = (idx - row_start[rows[order]]) % k_row[rows[order]]
        del order
        v = vstart[rows] + sub                                        # virtual row inside the tile
        wave, lrow = v % W, v // W
        # c. order inside a (tile, wave, window) list: by virtual row; then deal the list's entries to its stream positions
        #    slot-major over the steps the list covers, so that neighbours (same virtual row) land in different steps
        tw = tile * W + wave
        if layout is None:
            layout = os.environ.get('AMAR_LT_LAYOUT', 'deal')
        if layout == 'count':
            # windows by COUNT (round 4): window k of a wave = entries [k, k + 1) * S * EPS of the wave's OWN stream in column order,
            # S = window_entries / (W * EPS) steps — every wave runs exactly S steps between two barriers.  With windows cut by
            # column (the other layouts) a wave holds 32 S +- 6 sqrt(S) entries of a window, i.e. S - 1, S or S + 1 steps, and the
            # barrier waits for the slowest: with the gathers compiled out that alone took the walk from 0.126 to 0.180 ms per
            # ml1m(s=64) layer.  The price: the waves' column positions drift apart like a random walk (+- sqrt(k) * 6 entries after k
            # steps: a few windows over a 2 300-row tile), which the L1 sharing has to bear.
            spw = max(1, window_entries // (W * eps)) * eps          # entries per wave and window
            o = torch.argsort(tw * n_cols + cols)
            cnt_s = torch.bincount(tw, minlength=T * W)
            start_s = torch.cumsum(cnt_s, 0) - cnt_s
            win = torch.empty(m, dtype=torch.int64, device=dev)
            win[o] = (idx - start_s[tw[o]]) // spw
            del o
            n_win = ((cnt_s + spw - 1) // spw).view(T, W).max(1).values if T else n_win
            maxwin = max(1, int(n_win.max())) if T else 1
            layout = 'deal'
        lst = tw * maxwin + win
        if T * W * maxwin * rw >= (1 << 62):
            raise ValueError("LT image: sort key overflow")
        defer = layout == 'defer'
        if defer:
            sub_window = int(sub_window or os.environ.get('AMAR_LT_SUB_WINDOW', 0) or W * eps)
            n_sub = max(1, -(-window_entries // sub_window))
            o1 = torch.argsort(tile * n_cols + cols)                   # (column position inside the tile) // sub_window
            sw = torch.empty(m, dtype=torch.int64, device=dev)
            sw[o1] = ((idx - tile_start[tile[o1]]) % window_entries) // sub_window
            del o1
            o2 = torch.argsort((lst * rw + lrow) * n_sub + sw)         # occurrence number of the virtual row inside its list
            k2 = (lst * rw + lrow)[o2]
            first2 = torch.ones(m, dtype=torch.bool, device=dev)
            if m > 1:
                first2[1:] = k2[1:] != k2[:-1]
            occ = torch.empty(m, dtype=torch.int64, device=dev)
            occ[o2] = (idx - _run_starts(first2, idx)).clamp_(max=1023)
            del o2, k2, first2
            order = torch.argsort(((lst * 1024 + occ) * n_sub + sw) * rw + lrow)
            del occ, sw
        else:
            order = torch.argsort(lst * rw + lrow)
        cnt_tw = torch.bincount(tw, minlength=T * W)
        len_tw = (cnt_tw + CHUNK - 1) // CHUNK * CHUNK
        stream_start = torch.cumsum(len_tw, 0) - len_tw
        total = int(len_tw.sum())
        if total >= (1 << 31):
            raise ValueError("LT image: more than 2^31 entries")
        cnt_start = torch.cumsum(cnt_tw, 0) - cnt_tw
        tw_s = tw[order]
        rel = idx - cnt_start[tw_s]                                   # position inside the (tile, wave) stream, list after list
        lst_s = lst[order]
        if defer:
            dest = stream_start[tw_s] + rel                           # the sorted order IS the stream order
        else:
            deal = torch.argsort((lst_s * eps + rel % eps) * (int(len_tw.max()) // eps + 1) + rel // eps)
            dest = stream_start[tw_s] + rel[deal]                     # i-th entry of the sorted order takes the i-th dealt position
            del deal
        del rel
        lrow_s = lrow[order]
        if spread is None:
            spread = int(os.environ.get('AMAR_LT_SPREAD', 0))
        if spread:
            dest = _spread_repeats(dest, lrow_s, tw_s, stream_start, cnt_tw, eps, rw, total, int(spread))
        if colsort is None:
            colsort = os.environ.get('AMAR_LT_COLSORT') == '1'
        if colsort and pairs is False and m:
            # inside a step the slots are interchangeable once no pair logic reads them: order a step's entries by COLUMN, so that
            # entries of one 128-byte line of X sit in neighbouring lanes of the gather (development: does the L1's tag path care?)
            step_ = dest // eps
            o5 = torch.argsort(step_ * n_cols + cols[order])
            s5 = step_[o5]
            first5 = torch.ones(m, dtype=torch.bool, device=dev)
            first5[1:] = s5[1:] != s5[:-1]
            dest = dest.clone()
            dest[o5] = s5 * eps + (idx - _run_starts(first5, idx))
            del o5, s5, first5, step_
        word = (lrow_s << cbits) | cols[order]
        # d. inside every step: the first entry of a virtual row is plain; the one in the next slot (same DPP row) is an implicit
        #    pair; every other repeat is flagged
        step, slot = dest // eps, dest % eps
        spr = max(1, 16 // (F // 4))                                   # entry slots per 16-lane DPP row
        o4 = torch.argsort((step * rw + lrow_s) * eps + slot)
        g4 = (step * rw + lrow_s)[o4]
        s4 = slot[o4]
        first = torch.ones(m, dtype=torch.bool, device=dev)
        if m > 1:
            first[1:] = g4[1:] != g4[:-1]
        run_start = _run_starts(first, idx)
        rank = idx - run_start
        if pairs is None:
            pairs = F < 16 or rw_arg is not None                       # (the GAT geometry keeps them: its kernel form is not instantiated without)
            if os.environ.get('AMAR_LT_PAIRS') in ('0', '1'):
                pairs = os.environ['AMAR_LT_PAIRS'] == '1' or rw_arg is not None
        pair = (rank == 1) & (s4 == s4[run_start] + 1) & (s4 % spr != 0)
        if not pairs:
            pair = torch.zeros_like(pair)
        flagged = (rank >= 1) & ~pair
        n_flagged, n_pairs = int(flagged.sum()), int(pair.sum())
        flag = torch.zeros(m, dtype=torch.bool, device=dev)
        flag[o4] = flagged
        word = torch.where(flag, word - (1 << 31), word)              # bit 31 as two's complement
        words = torch.full((max(total, 1),), (rw - 1) << cbits, dtype=torch.int32, device=dev)
        words[dest] = word.to(torch.int32)
        del o4, g4, s4, first, run_start, rank, pair, flagged, flag, word, dest, lrow_s, tw_s, lst_s
        # e. step at which each window begins for each wave
        cnt = torch.bincount(lst, minlength=T * W * maxwin).view(T, W, maxwin)
        e0 = torch.cumsum(cnt, 2) - cnt                               # entries of the stream before the window
        wsteps = torch.empty((T, W, maxwin + 1), dtype=torch.int64, device=dev)
        wsteps[:, :, :maxwin] = e0 // eps
        end_steps = ((cnt_tw + eps - 1) // eps).view(T, W, 1)          # the steps that hold entries (the padding's steps are not walked)
        beyond = torch.arange(maxwin + 1, device=dev).view(1, 1, -1) >= n_win.view(T, 1, 1)
        wsteps = torch.where(beyond, end_steps.expand(T, W, maxwin + 1), wsteps)
        return cls(F, words, stream_start.to(torch.int32), wsteps.to(torch.int32).contiguous(),
                   tb_t.to(torch.int32), n_win.to(torch.int32), maxwin + 1, vstart.to(torch.int32).contiguous(),
                   vcount.to(torch.int32).contiguous(), diag, row_scale, col_scale, diag_offset,
                   (n_rows, n_cols), window_entries, m, n_flagged, n_pairs, rw, pairs=pairs)
