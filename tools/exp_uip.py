#!/usr/bin/env python
"""The fused GCN layer on the user-item-property graph at ml1m(s) (development aid): LT image statistics and time per launch."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.exp_xs_floor import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev, with_props=True)
    n = data['n_users'] + data['n_items'] + data['n_props']
    rows = torch.cat([data['train_pos'][:, 0], data['item_prop'][:, 0]])
    cols = torch.cat([data['train_pos'][:, 1], data['item_prop'][:, 1]])
    a = gcn_filter_device(rows, cols, n)
    x = torch.randn((n, F), device=dev)
    y = torch.empty((n, F), device=dev)
    for lt_on in ('1', '0'):
        os.environ['AMAR_SPMM_LT'] = lt_on
        img = a.tiled_image(F)
        xs_tab = torch.empty_like(x)
        capi.row_affine(x, img.col_scale if img.col_scale is not None else img.row_scale, xs_tab)
        t = timeit(lambda: capi.spmm_xs(img, xs_tab, y, prescaled=True))
        extra = ''
        if hasattr(img, 'n_tiles'):
            rt = img.tile_row0[1:] - img.tile_row0[:-1]
            extra = ' tiles %d (rows %d..%d) vrows <= %d window %d flagged %.2f %% pairs %.2f %% breaks %s' % (
                img.n_tiles, int(rt.min()), int(rt.max()), int(img.vcount.max()), img.window_entries,
                100.0 * img.n_flagged / max(1, img.n_entries), 100.0 * img.n_pairs / max(1, img.n_entries), getattr(a, 'row_breaks', None))
        print('uip s=%d F=%d N=%d nnz=%d: %s %.4f ms%s' % (scale, F, n, a.nnz, 'LT' if lt_on == '1' else 'XS', t, extra), flush=True)
        if hasattr(img, 'n_tiles'):
            ss = torch.cat([img.stream_start.long(), torch.tensor([img.words.numel()], device=dev)])
            per_tile = (ss[1:] - ss[:-1]).view(-1, 16).sum(1).cpu().numpy()
            r0 = img.tile_row0.cpu().numpy()
            br = [0] + list(a.row_breaks) + [n]
            for k in range(len(br) - 1):
                sel = (r0[:-1] >= br[k]) & (r0[:-1] < br[k + 1])
                pt, rr = per_tile[sel], (r0[1:] - r0[:-1])[sel]
                print('   segment %d: %d tiles, entries per tile min %d median %d max %d, rows per tile min %d median %d max %d, windows max %d' %
                      (k, sel.sum(), pt.min(), int(sorted(pt)[len(pt) // 2]), pt.max(), rr.min(), int(sorted(rr)[len(rr) // 2]), rr.max(),
                       int(img.n_win.cpu().numpy()[sel].max())), flush=True)
        y_ref = y.clone() if lt_on == '1' else y_ref
    print('max |LT - XS| %.2e' % float((y - y_ref).abs().max()))
    # the three node types on their own: which rows cost what
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import _unit_entries
    rows_e, cols_e, diag, _ = _unit_entries(a, True)
    cs = a.dinv.to(torch.float32).contiguous()
    xs_tab = torch.empty_like(x)
    capi.row_affine(x, cs, xs_tab)
    br = [0] + list(a.row_breaks) + [n]
    for k in range(len(br) - 1):
        r0, r1 = br[k], br[k + 1]
        sel = (rows_e >= r0) & (rows_e < r1)
        lt = lds_tiled.LdsTiled.build(rows_e[sel] - r0, cols_e[sel], r1 - r0, n, F, diag[r0:r1].contiguous(), cs[r0:r1].contiguous(), cs, r0)
        yb = torch.empty((r1 - r0, F), device=dev)
        t = timeit(lambda: capi.spmm_lt(lt, xs_tab, yb, prescaled=True))
        print('   rows of type %d alone (%d rows, %d entries): %d tiles, window %d, %.4f ms' % (k, r1 - r0, int(sel.sum()), lt.n_tiles, lt.window_entries, t), flush=True)
        del lt
        # the same rows on the CSR row-streaming kernel: values s_i . c_ij on the pre-scaled table (the diagonal is an ordinary entry)
        rp = a.rowptr[r0:r1 + 1]
        rows_all = torch.repeat_interleave(torch.arange(n, device=dev), (a.rowptr[1:] - a.rowptr[:-1]).long())
        vals2 = (a.mult.to(torch.float32) * cs[rows_all]).contiguous()
        yc = torch.empty((r1 - r0, F), device=dev)
        t = timeit(lambda: capi.spmm_csr(rp, a.colidx, vals2, xs_tab, yc))
        print('      ... on the CSR row kernel: %.4f ms, max |diff| %.2e' % (t, float((yc - yb).abs().max())), flush=True)
        del rows_all, vals2


if __name__ == '__main__':
    main()
