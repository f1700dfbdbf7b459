#!/usr/bin/env python
"""Round 4: the per-type row blocks of one rank of a world-W typed partition of ml1m(s) on the LT walk — plain product per launch
under window size / barrier cadence (pace 0 = the waves run free: lt.pace_every = 2^30) / image layout (development aid).
`python tools/exp_rank_block.py <scale> <world> [rank]`; EXP_NCU=tiles per launch (default 256)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    scale, world = int(sys.argv[1]), int(sys.argv[2])
    rank = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    F = 8
    from deep_cbrs_amar_renaissance_amd import capi, parallel
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device, _unit_entries
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    nu, ni = data['n_users'], data['n_items']
    n = nu + ni
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    tp = parallel.TypedPartition([0, nu, n], world)
    n_tab = world * tp.R
    x = torch.randn((n_tab, F), device=dev)
    n_cu = int(os.environ.get('EXP_NCU', 256))
    print('ml1m(s=%d) world %d rank %d: R %d rows per rank, table %d rows, %d tiles per launch' % (scale, world, rank, tp.R, n_tab, n_cu), flush=True)
    for t, name in ((0, 'users'), (1, 'items')):
        blk = tp.local_block(a, rank, t)
        rows, cols, diag, off = _unit_entries(blk, True)
        cs = blk.dinv.to(torch.float32).contiguous()
        y = torch.empty((blk.shape[0], F), device=dev)
        print(' type %s: %d rows, %d entries, active columns %d' % (name, blk.shape[0], int(rows.numel()), blk.active_cols), flush=True)
        for pairs, spread in ((True, 0), (False, 3)):
            for window in (512, 1024, 2048, 4096):
                lt = lds_tiled.LdsTiled.build(rows, cols, blk.shape[0], n_tab, F, diag, cs[off:off + blk.shape[0]].contiguous(), cs, off,
                                              window_entries=window, n_cu=n_cu, pairs=pairs, spread=spread or None)
                res = []
                for pace in (1, 2, 4, 0):
                    lt.pace_every = pace if pace else (1 << 30)
                    res.append('pace %d %.4f' % (pace, timeit(lambda: capi.spmm_lt(lt, x, y, prescaled=True))))
                rows_t = lt.tile_row0[1:] - lt.tile_row0[:-1]
                print('   pairs %d spread %d window %4d: %s ms | tiles %d (rows %d..%d) flagged %.2f%% pairs %.2f%%' % (
                    pairs, spread, lt.window_entries, ', '.join(res), lt.n_tiles, int(rows_t.min()), int(rows_t.max()),
                    100.0 * lt.n_flagged / max(1, lt.n_entries), 100.0 * lt.n_pairs / max(1, lt.n_entries)), flush=True)
                del lt


if __name__ == '__main__':
    main()
