"""Adjacency helpers of the hot path: symmetrisation, GCN filter, scipy -> device CSR.

Mirrors `/root/reference/src/utilities/math.py:6-56` (``symmetrize_matrix``, ``convert_to_tensor``,
``sparse_matrix_to_tensor``) and Spektral's ``utils.gcn_filter`` (call sites
`src/models/gnn.py:283,381`, `src/layers/lightgcn_conv.py:56-58`).  Where the reference builds a
``tf.SparseTensor`` (COO, int64 indices, row-major reorder) this builds a :class:`DeviceCSR`
(int32 rowptr/colidx + fp32 values resident in HBM): the same non-zeros in the same row-major
order — duplicates included — in 8 bytes per non-zero instead of 20.
"""
import os

import numpy as np
import torch
from scipy import sparse

from deep_cbrs_amar_renaissance_amd.engine import default_device


def symmetrize_matrix(x):
    """Symmetrise a matrix; a sparse one by appending the transposed triplets (no dedupe)."""
    if sparse.issparse(x):
        x = x.tocoo()
        rows = np.concatenate([x.row, x.col])
        cols = np.concatenate([x.col, x.row])
        data = np.concatenate([x.data, x.data])
        return sparse.coo_matrix((data, (rows, cols)), shape=x.shape, dtype=x.dtype)
    return np.maximum(x, x.T)


def gcn_filter(a, symmetric=True):
    """D^-1/2 (A + I) D^-1/2 (or D^-1 (A + I)) in the matrix dtype, as Spektral's ``gcn_filter``.

    Sparse input: duplicates are summed (CSR), 1 is added on the diagonal, infinite inverse
    degrees become 0, column indices come out sorted.
    """
    if isinstance(a, DeviceCSR):
        if not a.gcn_filtered:
            raise ValueError("a DeviceCSR handed to gcn_filter must come from gcn_filter_device")
        return a
    if sparse.issparse(a):
        m = sparse.csr_matrix(a, copy=True)
        m.sum_duplicates()
        m = (m + sparse.identity(m.shape[0], dtype=m.dtype, format='csr')).tocsr()
        m.sort_indices()
        with np.errstate(divide='ignore'):
            dinv = np.power(np.asarray(m.sum(1)).ravel(), -0.5 if symmetric else -1.0).astype(m.dtype)
        dinv[np.isinf(dinv)] = 0.0
        coo = m.tocoo()
        data = (dinv[coo.row] * coo.data).astype(m.dtype)          # D . A first,
        if symmetric:
            data = (data * dinv[coo.col]).astype(m.dtype)          # ... then (D . A) . D: two fp32 roundings
        out = sparse.csr_matrix((data, m.indices.copy(), m.indptr.copy()), shape=m.shape)   # m's structure: sorted, duplicate-free
        # the factors, for the value-free device images (A_hat = S (A + I) S with small integer entries): kept on the result so
        # that DeviceCSR.from_scipy can hand them to the XS / LT image builders, as gcn_filter_device does
        if symmetric and m.dtype == np.float32 and m.nnz and float(np.abs(m.data - np.rint(m.data)).max()) == 0.0 and m.data.max() < (1 << 20):
            out.amar_factors = (dinv.astype(np.float32), np.rint(m.data).astype(np.int32))
        return out
    m = np.array(a, copy=True)
    m[np.diag_indices_from(m)] += 1
    with np.errstate(divide='ignore'):
        dinv = np.power(m.sum(1), -0.5 if symmetric else -1.0)
    dinv[np.isinf(dinv)] = 0.0
    return (dinv[:, None] * m) * dinv[None, :] if symmetric else dinv[:, None] * m


class DeviceCSR:
    """Canonical CSR in HBM: int32 ``rowptr`` [n+1], int32 ``colidx`` [nnz], fp32 ``vals`` [nnz] or None.

    ``vals is None`` means an all-ones matrix (GraphSAGE / GAT ignore edge values).  Duplicate
    (row, col) entries are kept as parallel non-zeros.
    """

    def __init__(self, rowptr, colidx, vals, shape, gcn_filtered=False, dinv=None, mult=None):
        self.rowptr, self.colidx, self.vals, self.shape = rowptr, colidx, vals, tuple(shape)
        self.gcn_filtered = gcn_filtered          # already D^-1/2 (A+I) D^-1/2 (built by gcn_filter_device)
        # factors of a gcn-filtered matrix, vals = (dinv[row] * mult) * dinv[col] with integer multiplicities `mult`:
        # lets the XS image drop the values (XcdSliced.from_csr)
        self.dinv, self.mult = dinv, mult

    @property
    def nnz(self):
        return int(self.colidx.numel())

    @classmethod
    def from_scipy(cls, x, with_values=True, drop_diagonal=False, device=None):
        device = device or default_device()
        coo = x.tocoo()
        row, col, data = coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data
        if drop_diagonal:
            keep = row != col
            row, col, data = row[keep], col[keep], data[keep]
        if coo.shape[0] >= 2 ** 31 or len(row) >= 2 ** 31:
            raise ValueError("graph too large for int32 CSR")
        order = np.lexsort((col, row))                      # tf.sparse.reorder: row-major, stable on duplicates
        row, col = row[order], col[order]
        rowptr = np.zeros(coo.shape[0] + 1, dtype=np.int64)
        np.cumsum(np.bincount(row, minlength=coo.shape[0]), out=rowptr[1:])
        vals = torch.from_numpy(data[order].astype(np.float32)).to(device) if with_values else None
        out = cls(torch.from_numpy(rowptr.astype(np.int32)).to(device),
                  torch.from_numpy(col.astype(np.int32)).to(device), vals, coo.shape)
        factors = getattr(x, 'amar_factors', None)                  # set by gcn_filter: A_hat = S C S
        if factors is not None and with_values and not drop_diagonal and len(factors[1]) == len(order):
            out.gcn_filtered = True
            out.dinv = torch.from_numpy(factors[0]).to(device)
            out.mult = torch.from_numpy(factors[1][order]).to(device)
        return out

    def to_scipy(self):
        rowptr = self.rowptr.cpu().numpy()
        colidx = self.colidx.cpu().numpy()
        vals = self.vals.cpu().numpy() if self.vals is not None else np.ones(len(colidx), dtype=np.float32)
        return sparse.csr_matrix((vals, colidx, rowptr), shape=self.shape)


def sparse_matrix_to_tensor(x, dtype=torch.float32, **kwargs):
    """scipy sparse -> :class:`DeviceCSR` (row-major order, duplicates kept)."""
    assert sparse.issparse(x), "The input matrix should be sparse"
    if dtype != torch.float32:
        raise ValueError("the HIP path computes in float32 only")
    return DeviceCSR.from_scipy(x, **kwargs)


def convert_to_tensor(x, dtype=torch.float32, **kwargs):
    """Array (dense or sparse) -> device tensor / :class:`DeviceCSR`."""
    if isinstance(x, DeviceCSR):
        return x
    if sparse.issparse(x):
        return sparse_matrix_to_tensor(x, dtype=dtype, **kwargs)
    raise ValueError("dense adjacency matrices are not supported by the HIP path (sparse_adjacency: True in every config)")


def gcn_filter_device(rows, cols, n_nodes):
    """``symmetrize_matrix`` + ``gcn_filter`` + CSR conversion done on the GPU with torch sorts.

    `rows`/`cols` are the device int64 endpoints of the UN-symmetrised unit-weight edges
    (positive ratings, item-property links).  Produces the same :class:`DeviceCSR`, bit for bit,
    as ``DeviceCSR.from_scipy(gcn_filter(symmetrize_matrix(coo)))`` — duplicates summed, unit
    diagonal added, fp32 ``(d_i^-1/2 * a_ij) * d_j^-1/2`` — without the host round trip, which is
    what makes s=64 graphs (112 M non-zeros) practical to build.
    """
    dev = rows.device
    loops = torch.arange(n_nodes, device=dev, dtype=torch.int64)
    keys = torch.cat([rows * n_nodes + cols, cols * n_nodes + rows, loops * n_nodes + loops])
    keys, counts = torch.unique(keys, return_counts=True)            # sorted row-major; counts = summed duplicates
    r, c = keys // n_nodes, keys % n_nodes
    a = counts.to(torch.float32)
    deg = torch.zeros(n_nodes, dtype=torch.float32, device=dev).index_add_(0, r, a)
    # per-row sums of small integers are exact in fp32 in any order (< 2^24)
    dinv = torch.from_numpy(np.power(deg.cpu().numpy(), np.float32(-0.5)).astype(np.float32)).to(dev)
    dinv[torch.isinf(dinv)] = 0
    vals = (dinv[r] * a) * dinv[c]
    rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(r, minlength=n_nodes), 0)
    out = DeviceCSR(rowptr.to(torch.int32), c.to(torch.int32), vals, (n_nodes, n_nodes), gcn_filtered=True,
                    dinv=dinv.contiguous(), mult=counts.to(torch.int32))
    if rows.numel() and int(rows.max()) < int(cols.min()):           # bipartite with grouped ids (users, then items): type boundary
        out.row_breaks = (int(cols.min()),)
    return out


class SlicedJagged:
    """Sliced-jagged ("SJ") image of a CSR matrix for `amar_spmm_sj_f32` (see include/amar_hip.h).

    Rows are taken 64 at a time (one wavefront, ONE LANE PER ROW); columns are cut into slices of
    2^cbits columns (chosen so a slice of X, 2^cbits * F * 4 bytes, stays resident in the 4 MB
    per-XCD L2).  For wave w and slice k the non-zeros are stored in jagged-diagonal order: first
    the 1st non-zero of every row that has one in this slice (ascending lane), then the 2nd, ...
    — so at step j the active lanes read consecutive 8-byte (col, val) entries and nothing is
    padded.  Blocks follow each other in (wave, slice) order, so every wave sweeps the slices in
    the same order (phase-major) and reads one contiguous range of `entries`.

        entries    int32 [nnz, 2]   (global column, fp32 value bits), 8 bytes per non-zero as in CSR
        counts     int16 [n_waves * n_slices * 64]   non-zeros of lane's row in (wave, slice)
        wave_start int32 [n_waves + 1]               first entry of each wave
    """

    def __init__(self, entries, counts, wave_start, n_slices, cbits, shape):
        self.entries, self.counts, self.wave_start = entries, counts, wave_start
        self.n_slices, self.cbits, self.shape = n_slices, cbits, tuple(shape)

    @classmethod
    def from_csr(cls, a, cbits):
        dev = a.rowptr.device
        n_rows, n_cols = a.shape
        nnz = a.nnz
        n_waves = (n_rows + 63) // 64
        n_slices = max(1, (n_cols + (1 << cbits) - 1) >> cbits)
        deg = (a.rowptr[1:] - a.rowptr[:-1]).long()
        if nnz and int(deg.max()) > 32767:
            raise ValueError("rows longer than 32767 non-zeros are not supported by the SJ format")
        rows = torch.repeat_interleave(torch.arange(n_rows, device=dev), deg)
        cols = a.colidx.long()
        sl = cols >> cbits
        run = rows * n_slices + sl                                   # CSR order = (row, col) => runs are contiguous
        counts_rs = torch.bincount(run, minlength=n_rows * n_slices)  # [row, slice]
        run_start = torch.cumsum(counts_rs, 0) - counts_rs
        j = torch.arange(nnz, device=dev) - run_start[run]           # rank inside the (row, slice) run
        jmax = int(j.max()) + 1 if nnz else 1
        key = (((rows >> 6) * n_slices + sl) * jmax + j) * 64 + (rows & 63)
        order = torch.argsort(key)
        vals = a.vals if a.vals is not None else torch.ones(nnz, dtype=torch.float32, device=dev)
        entries = torch.stack([a.colidx[order], vals[order].view(torch.int32)], dim=1).contiguous()
        pad_rows = n_waves * 64
        c = torch.zeros((pad_rows, n_slices), dtype=torch.int64, device=dev)
        c[:n_rows] = counts_rs.view(n_rows, n_slices)
        counts = c.view(n_waves, 64, n_slices).permute(0, 2, 1).contiguous().view(-1).to(torch.int16)
        per_wave = c.view(n_waves, -1).sum(1)
        wave_start = torch.zeros(n_waves + 1, dtype=torch.int64, device=dev)
        wave_start[1:] = torch.cumsum(per_wave, 0)
        return cls(entries, counts, wave_start.to(torch.int32), n_slices, cbits, a.shape)


def sj_column_bits(F, slice_bytes=2 << 20):
    """log2(columns per slice) so that one slice of X ([2^cbits, F] fp32) is `slice_bytes`."""
    cbits = 0
    while (2 << cbits) * F * 4 <= slice_bytes:
        cbits += 1
    return cbits


def spmm_kind(a, F):
    """Which image of A the propagation uses: 'xs' (XCD-affine column slices + combine) when the gathered node
    table ([n_cols, F] fp32) exceeds what stays resident in one 4 MB per-XCD L2, else 'csr' (row streaming).
    AMAR_SPMM_KIND=csr|xs|sj overrides (A/B timing; 'sj' is the sliced-jagged form)."""
    import os
    forced = os.environ.get('AMAR_SPMM_KIND')
    if forced in ('csr', 'sj', 'xs'):
        return forced
    # measured (tools/profile_step.py, value-free image): F=8: row form 0.094 / XS 0.117 ms at s=16 (4.7 MB table), 0.224 / 0.192 ms
    # at s=32 (9.4 MB), 0.62 / 0.38 ms at s=64; F=16: 0.127 / 0.169 ms at 9.4 MB, 0.319 / 0.281 ms at 18.9 MB; XS loses at F=32
    # (its scan is amortised over only 64/(F/4) lanes' worth of entries)
    table_bytes = a.shape[1] * F * 4
    big = table_bytes >= ((8 << 20) if F <= 8 else (16 << 20))
    if a.shape[0] == a.shape[1] and table_bytes >= (2 << 20) and lt_eligible(a, F):
        # the LDS-tiled image (DeviceCSR.tiled_image) wins from a 2 MB table on: F = 8, ms per launch LT / row streaming / XS:
        # ml1m(s=8) 0.036 / 0.052 / 0.062, ml1m(s=16) 0.061 / 0.093 / 0.110 (ml1m(s=4), 1.2 MB: 0.025 / 0.028; s=2: 0.018 / 0.016);
        # s=64: F = 16 0.33 / 0.78 / 0.54, F = 32 0.60 / 0.91 / 1.12
        return 'xs'
    if a.shape[0] == a.shape[1] and a.vals is None and table_bytes >= (2 << 20) and F in (8, 16, 32) and _edge_list_lt_density(a, F):
        # edge-list graphs (GraphSAGE's mean aggregate, GAT) walk the LT image from the same size on: ml1m(s=16) GAT C = 8 0.104 ms
        # against 0.193 (row kernel), C = 16 0.133 / 0.246; GraphSAGE aggregate 0.067 against 0.103 for the whole row-kernel layer
        return 'xs'
    return 'xs' if (a.shape[0] == a.shape[1] and F in (4, 8, 16) and big) else 'csr'


def _edge_list_lt_density(a, F):
    """The density rule of lt_eligible for an edge-list CSR (no factors needed: its LT images are value-free by construction)."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    forced = os.environ.get('AMAR_SPMM_LT')
    if forced == '0' or not lds_tiled.supported(F, a.shape[1]):
        return False
    return forced == '1' or a.nnz >= LT_MIN_DENSITY * lds_tiled.N_CU * getattr(a, 'active_cols', a.shape[1])


def _csr_sliced(self, F):
    """Cached SJ image for feature width F (built on first use; the graph is constant)."""
    cache = self.__dict__.setdefault('_sj_cache', {})
    cbits = sj_column_bits(F)
    if cbits not in cache:
        cache[cbits] = SlicedJagged.from_csr(self, cbits)
    return cache[cbits]


DeviceCSR.sliced = _csr_sliced


class XcdSliced:
    """XCD-sliced ("XS") image of a square CSR matrix for `amar_spmm_xs_f32` (see include/amar_hip.h).

    MI355X has eight XCDs with a private 4 MB L2 each, and workgroups are dealt to XCDs round-robin.
    A row-gather SpMM whose node table exceeds 4 MB therefore keeps eight thrashing copies of it
    (measured on ml1m(s=64): L2 hit rate 45 %, 4.3 GB of fabric reads per launch for 0.49 GB of
    algorithmic bytes).  Here the COLUMNS are cut into S = 8 contiguous slices of equal non-zero
    count and workgroup b only touches slice b % 8: every XCD gathers from one eighth of the table,
    which then lives exactly once in the 32 MB of aggregate L2 (measured: 95 % hits, 0.55 GB).

        diag     fp32 [n]            the diagonal of A (summed duplicates), applied by the combine kernel
        rowptr   int32 [S*n + 1]     start of (slice k, row r) at index k*n + r in the reordered arrays
        colidx   int32 [nnz_offdiag] (row & 63) << 26 | global column, sorted by (slice, row, column)
        vals     fp32  [nnz_offdiag] or None
        bounds   int64 [S + 1]       column range of every slice (host list)

    Value-free form (vals None, row_scale = d^-1/2): for a gcn-filtered matrix whose factors are known
    (gcn_filter_device), A = S C S with C = A + I in small integers, so every entry weighs 1 (an entry c > 1 is
    stored c times), diag holds C_ii, the kernels gather from S.X and scale the row sum by S: the read-once index
    stream is 4 bytes per non-zero instead of 8 (about -10 % per layer on ml1m(s=64)).  AMAR_XS_VALUES=1 keeps values.
    """

    N_SLICES = 8

    @classmethod
    def slices_for(cls, n_cols):
        """8 slices, one per XCD.  AMAR_XS_SLICES=8k makes the kernels work through 8k slices in k phases (one slice per XCD
        live at a time; supported and tested, but at ml1m(s=256) — a 75 MB table — 32 slices gained only 4 % over 8 and cost
        4x the partial-sum scratch, so it is not the default)."""
        forced = os.environ.get('AMAR_XS_SLICES')
        return int(forced) if forced else cls.N_SLICES

    def __init__(self, diag, rowptr, colidx, vals, bounds, shape, row_scale=None, col_scale=None, diag_offset=0):
        self.diag, self.rowptr, self.colidx, self.vals, self.bounds, self.shape = diag, rowptr, colidx, vals, bounds, tuple(shape)
        # value-free form: row_scale [n_rows] scales the row sums, col_scale [n_cols] pre-scales the gathered table
        # (the same vector for a square matrix); diag_offset: column of row 0's own entry (row block of a larger matrix)
        self.row_scale, self.col_scale, self.diag_offset = row_scale, col_scale, int(diag_offset)
        self.n_slices = len(bounds) - 1
        self._partials = {}

    def partials(self, F):
        """Scratch [S, n, F] for the per-slice partial sums (allocated once per width)."""
        if F not in self._partials:
            self._partials[F] = torch.empty((self.n_slices, self.shape[0], F), dtype=torch.float32, device=self.rowptr.device)
        return self._partials[F]

    @classmethod
    def from_csr(cls, a, n_slices=None):
        """`a`: a square DeviceCSR, or a row block of one (multi-GPU partition) carrying `diag_offset` = the column of
        its first row's own entry, and for the value-free form `dinv` over its COLUMNS plus `mult`."""
        dev = a.rowptr.device
        n, n_cols = a.shape
        if n_slices is None:
            n_slices = cls.slices_for(n_cols)
        diag_offset = int(getattr(a, 'diag_offset', 0))
        if n != n_cols and not hasattr(a, 'diag_offset'):
            raise ValueError("the XS image is defined for square matrices and for row blocks that say where their diagonal is")
        deg = (a.rowptr[1:] - a.rowptr[:-1]).long()
        rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
        cols = a.colidx.long()
        value_free = getattr(a, 'dinv', None) is not None and getattr(a, 'mult', None) is not None and \
            os.environ.get('AMAR_XS_VALUES') != '1'
        if value_free:
            vals = a.mult.to(torch.float32)                      # C = A + I: the integer multiplicities
        else:
            vals = a.vals if a.vals is not None else torch.ones(a.nnz, dtype=torch.float32, device=dev)
        on_diag = cols == rows + diag_offset
        diag = torch.zeros(n, dtype=torch.float32, device=dev).index_add_(0, rows[on_diag], vals[on_diag])
        rows, cols, vals = rows[~on_diag], cols[~on_diag], vals[~on_diag]
        if value_free and cols.numel() and int(vals.max()) > 1:  # an entry c > 1 becomes c unit entries
            rep = vals.long()
            rows, cols = torch.repeat_interleave(rows, rep), torch.repeat_interleave(cols, rep)
        m = int(cols.numel())
        if m:
            sc = torch.sort(cols).values
            cuts = [int(sc[(m * k) // n_slices]) for k in range(1, n_slices)]
        else:
            cuts = [(n_cols * k) // n_slices for k in range(1, n_slices)]
        bounds = [0] + cuts + [n_cols]
        for k in range(1, len(bounds)):
            bounds[k] = max(bounds[k], bounds[k - 1])
        b = torch.tensor(bounds, dtype=torch.int64, device=dev)
        sl = (torch.searchsorted(b, cols, right=True) - 1).clamp_(0, n_slices - 1)
        seg = sl * n + rows
        order = torch.argsort(seg * n_cols + cols)
        rowptr = torch.zeros(n_slices * n + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(seg, minlength=n_slices * n), 0)
        if n_cols > (1 << 26):
            raise ValueError("the XS image packs the row (6 bits) above a 26-bit column: n must be <= 2^26")
        packed = ((rows[order] & 63) << 26) | cols[order]
        packed = torch.where(packed >= (1 << 31), packed - (1 << 32), packed)      # two's-complement int32
        if value_free:
            col_scale = a.dinv.to(torch.float32).contiguous()                   # over the columns
            return cls(diag, rowptr.to(torch.int32), packed.to(torch.int32).contiguous(), None, bounds, a.shape,
                       row_scale=col_scale[diag_offset:diag_offset + n].contiguous(), col_scale=col_scale, diag_offset=diag_offset)
        return cls(diag, rowptr.to(torch.int32), packed.to(torch.int32).contiguous(), vals[order].contiguous(),
                   bounds, a.shape, diag_offset=diag_offset)


def _csr_xcd_sliced(self):
    if '_xs_cache' not in self.__dict__:
        self.__dict__['_xs_cache'] = XcdSliced.from_csr(self)
    return self.__dict__['_xs_cache']


def _csr_xcd_sliced_mean(self, self_loops=True):
    """XS image of an edge-list CSR (no values, duplicates kept, no diagonal) as GraphSAGE's mean aggregate:
    value-free entries, diag = 1 for the added self loop, row_scale = 1 / count (0 for an empty segment); the gathered
    table is NOT pre-scaled (call capi.spmm_xs with prescaled=True)."""
    key = '_xs_mean_cache_{}'.format(int(bool(self_loops)))
    if key not in self.__dict__:
        if self.vals is not None:
            raise ValueError("the mean-aggregate image is built from an edge-list CSR (vals None)")
        xs = XcdSliced.from_csr(self)                              # valued image of ones, zero diagonal
        deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32)
        if self_loops:
            xs.diag = xs.diag + 1.0
            inv = 1.0 / (deg + 1.0)
        else:
            inv = torch.where(deg > 0, 1.0 / deg.clamp(min=1.0), torch.zeros_like(deg))
        xs.vals, xs.row_scale, xs.col_scale = None, inv.contiguous(), None
        self.__dict__[key] = xs
    return self.__dict__[key]


DeviceCSR.xcd_sliced = _csr_xcd_sliced
DeviceCSR.xcd_sliced_mean = _csr_xcd_sliced_mean


def _unit_entries(a, use_mult):
    """Off-diagonal entries of a DeviceCSR (or a row block carrying `diag_offset`) as unit-weight (row, col) pairs —
    an entry of integer weight c is repeated c times — plus the summed diagonal.  use_mult: weights are `a.mult`
    (C = A + I of a gcn-filtered matrix), else 1 per stored entry (edge-list CSR)."""
    dev = a.rowptr.device
    n = a.shape[0]
    diag_offset = int(getattr(a, 'diag_offset', 0))
    deg = (a.rowptr[1:] - a.rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    cols = a.colidx.long()
    w = a.mult.to(torch.float32) if use_mult else torch.ones(a.nnz, dtype=torch.float32, device=dev)
    on_diag = cols == rows + diag_offset
    diag = torch.zeros(n, dtype=torch.float32, device=dev).index_add_(0, rows[on_diag], w[on_diag])
    rows, cols, w = rows[~on_diag], cols[~on_diag], w[~on_diag]
    if cols.numel() and int(w.max()) > 1:
        rep = w.long()
        rows, cols = torch.repeat_interleave(rows, rep), torch.repeat_interleave(cols, rep)
    return rows, cols, diag, diag_offset


def _csr_lds_tiled(self, F):
    """Cached LDS-tiled image (utilities/lds_tiled.py) of a gcn-filtered matrix with known factors, per width F."""
    from deep_cbrs_amar_renaissance_amd.utilities.lds_tiled import LdsTiled
    cache = self.__dict__.setdefault('_lt_cache', {})
    if F not in cache:
        if getattr(self, 'dinv', None) is None or getattr(self, 'mult', None) is None:
            raise ValueError("the LT image needs the factors of a gcn-filtered matrix (gcn_filter_device)")
        rows, cols, diag, diag_offset = _unit_entries(self, True)
        col_scale = self.dinv.to(torch.float32).contiguous()
        n = self.shape[0]
        # (a row block of a partition carries its type boundaries as LOCAL row numbers: parallel.TypedPartition.local_block)
        breaks = _row_breaks_of(self, rows, cols) if not hasattr(self, 'diag_offset') else tuple(getattr(self, 'row_breaks', None) or ())
        cache[F] = LdsTiled.build(rows, cols, n, self.shape[1], F, diag, col_scale[diag_offset:diag_offset + n].contiguous(),
                                  col_scale, diag_offset, row_breaks=breaks,
                                  window_entries=int(os.environ['AMAR_LT_WINDOW']) if os.environ.get('AMAR_LT_WINDOW') else None)
    return cache[F]


DeviceCSR.lds_tiled = _csr_lds_tiled

LT_MIN_DENSITY = 0.03      # entries per (tile, column): the sparsest case measured in LT's favour (the row blocks of an 8-rank partition of ml1m(s=64))


def lt_eligible(a, F):
    """Whether the large-graph product of `a` (a DeviceCSR or a row block of one) at width F runs on the LDS-tiled image
    rather than the XCD-sliced one: the value-free factors must be known, the packed word must hold the column, and a
    tile must see enough entries per column for the column-ordered walk to pay.  Measured on ml1m(s=64), F = 8
    (tools/exp_lt_blocks.py, LT / XS per launch): whole matrix (0.37 entries per tile and column) 0.22 / 0.36 ms; the row
    blocks of a 2-rank partition (0.14) 0.12-0.17 / 0.18-0.23; 4 ranks (0.07) 0.072-0.077 / 0.10-0.12; 8 ranks (0.035) 0.044-0.048 /
    0.059-0.066 — since small tiles get fine virtual rows (lds_tiled.py) the blocks of an 8-rank partition walk LT too (0.08 before).
    AMAR_SPMM_LT=0|1 overrides the density rule."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    forced = os.environ.get('AMAR_SPMM_LT')
    if forced == '0' or os.environ.get('AMAR_XS_VALUES') == '1':
        return False
    if getattr(a, 'dinv', None) is None or getattr(a, 'mult', None) is None or not lds_tiled.supported(F, a.shape[1]):
        return False
    if forced == '1':
        return True
    return F in (8, 16, 32) and a.nnz >= LT_MIN_DENSITY * lds_tiled.N_CU * getattr(a, 'active_cols', a.shape[1])


def infer_row_breaks(rows, cols, n):
    """Node-type boundaries of a graph with grouped ids (users | items [| properties], loaders.py:43-68), read off the
    entries: rows [lo, b) form a type when none of their columns falls inside [lo, b) and b is the smallest column above
    the diagonal.  The LDS-tiled builders end a tile there (a tile that straddles two types walks two column ranges at half
    the density each and runs ~25 % longer than its peers).  Blocks narrower than n/16 are not taken for node types."""
    breaks, lo = [], 0
    off = rows != cols
    for _ in range(4):
        upper = off & (rows >= lo) & (cols > rows)
        if not bool(upper.any()):
            break
        b = int(cols[upper].min())
        if b - lo < max(1, n // 16) or n - b < max(1, n // 16):
            break
        if bool((off & (rows >= lo) & (rows < b) & (cols >= lo) & (cols < b)).any()):
            break
        breaks.append(b)
        lo = b
    return tuple(breaks)


def _row_breaks_of(a, rows, cols):
    """`a.row_breaks` where the builder of `a` recorded them, else inferred once from the entries (square matrices only)."""
    if getattr(a, 'row_breaks', None) is None:
        a.row_breaks = infer_row_breaks(rows, cols, a.shape[0]) if a.shape[0] == a.shape[1] and not getattr(a, 'diag_offset', 0) else ()
    return tuple(a.row_breaks)


def _csr_tiled_mean_image(self, F, self_loops=True):
    """GraphSAGE's mean aggregate of an edge-list CSR (vals None, duplicates kept) on whichever tiled image pays: the
    LDS-tiled one (value-free entries, diag = 1 for the added self loop, row_scale = 1 / count, X gathered as is: call
    capi.spmm_xs with prescaled=True) under the same density rule as lt_eligible, else xcd_sliced_mean."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    forced = os.environ.get('AMAR_SPMM_LT')
    ok = forced != '0' and self.vals is None and lds_tiled.supported(F, self.shape[1]) and \
        (forced == '1' or (F in (8, 16, 32) and self.nnz >= LT_MIN_DENSITY * lds_tiled.N_CU * getattr(self, 'active_cols', self.shape[1])))
    if not ok:
        return self.xcd_sliced_mean(self_loops)
    cache = self.__dict__.setdefault('_lt_mean_cache', {})
    key = (F, bool(self_loops))
    if key not in cache:
        rows, cols, diag, diag_offset = _unit_entries(self, False)
        deg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32)
        if self_loops:
            diag = diag + 1.0
            inv = 1.0 / (deg + 1.0)
        else:
            inv = torch.where(deg > 0, 1.0 / deg.clamp(min=1.0), torch.zeros_like(deg))
        breaks = _row_breaks_of(self, rows, cols) if not diag_offset else ()
        cache[key] = lds_tiled.LdsTiled.build(rows, cols, self.shape[0], self.shape[1], F, diag, inv.contiguous(), None, diag_offset,
                                              row_breaks=breaks)
    return cache[key]


DeviceCSR.tiled_mean_image = _csr_tiled_mean_image


def _csr_tiled_gat_image(self, C):
    """The LDS-tiled image amar_gat_lt_f32 walks (capi.gat_lt) for an edge-list CSR (vals None, duplicates kept), or None where
    the XCD-sliced / row forms stay: same density rule as lt_eligible, the tile geometry of the GAT mode (the LDS row also holds
    the weight sum and s_self: lds_tiled.GAT_ROWS_PER_WAVE).  `diag` counts the (i, i) edges of the list itself."""
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    forced = os.environ.get('AMAR_SPMM_LT')
    rw = lds_tiled.GAT_ROWS_PER_WAVE.get(C)
    ok = forced != '0' and self.vals is None and rw is not None and lds_tiled.supported(C, self.shape[1], rw) and \
        (forced == '1' or self.nnz >= LT_MIN_DENSITY * lds_tiled.N_CU * getattr(self, 'active_cols', self.shape[1]))
    if not ok:
        return None
    cache = self.__dict__.setdefault('_lt_gat_cache', {})
    if C not in cache:
        rows, cols, diag, diag_offset = _unit_entries(self, False)
        ones = torch.ones(self.shape[0], dtype=torch.float32, device=diag.device)
        breaks = _row_breaks_of(self, rows, cols) if not diag_offset else ()
        cache[C] = lds_tiled.LdsTiled.build(rows, cols, self.shape[0], self.shape[1], C, diag, ones, None, diag_offset,
                                            row_breaks=breaks, rw=rw, split_growth=1.25,   # ml1m(s=64), C = 8: 0.364 ms (x2: 0.398)
                                            window_entries=int(os.environ['AMAR_LT_WINDOW']) if os.environ.get('AMAR_LT_WINDOW') else None)
    return cache[C]


DeviceCSR.tiled_gat_image = _csr_tiled_gat_image


def _csr_tiled_image(self, F):
    """The image the large-graph SpMM of width F runs on: LdsTiled where eligible (capi.spmm_xs accepts both), else XcdSliced."""
    return self.lds_tiled(F) if lt_eligible(self, F) else self.xcd_sliced()


DeviceCSR.tiled_image = _csr_tiled_image
