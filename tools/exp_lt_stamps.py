#!/usr/bin/env python
"""Per-tile cycle stamps of the LDS-tiled kernels (development aid).  Builds tools/libamar_hip_stamps.so from the library's
sources with -DAMAR_LT_STAMPS (the shipped library carries no stamps), runs one image and prints, per node-type segment, the
tiles' zeroing / walk / epilogue / total cycles and the slowest tiles.
    python tools/exp_lt_stamps.py build                      (no GPU needed: hipcc cross-compiles)
    python tools/exp_lt_stamps.py <ui|uip|gat> [scale] [F]   (on the GPU box)
    python tools/exp_lt_stamps.py block [scale] [F] [world] [rank]      one rank's row block of the typed partition"""
import ctypes
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'libamar_hip_stamps.so')
CSRC = os.path.join(ROOT, 'deep_cbrs_amar_renaissance_amd', 'csrc')


def build():
    srcs = ['amar_capi.hip', 'amar_propagate.hip', 'amar_dense.hip', 'amar_layout.hip', 'amar_chain.hip', 'amar_train.hip']
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-mllvm', '-amdgpu-mfma-vgpr-form=1',
           '-DAMAR_LT_STAMPS', '-shared'] + [os.path.join(CSRC, s) for s in srcs] + ['-o', LIB]
    subprocess.run(cmd, check=True)
    print('built', LIB)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        return build()
    what = sys.argv[1] if len(sys.argv) > 1 else 'ui'
    scale = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    import numpy as np
    import torch
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.LIB_PATH = LIB
    lib = capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev, with_props=(what == 'uip'))
    n = data['n_users'] + data['n_items'] + (data['n_props'] if what == 'uip' else 0)
    rows, cols = data['train_pos'][:, 0], data['train_pos'][:, 1]
    if what == 'uip':
        rows, cols = torch.cat([rows, data['item_prop'][:, 0]]), torch.cat([cols, data['item_prop'][:, 1]])
    a = gcn_filter_device(rows, cols, n)
    x = torch.randn((n, F), device=dev)
    y = torch.empty((n, F), device=dev)
    if what == 'block':
        from deep_cbrs_amar_renaissance_amd.parallel import TypedPartition
        world = int(sys.argv[4]) if len(sys.argv) > 4 else 8
        rank = int(sys.argv[5]) if len(sys.argv) > 5 else 0
        tp = TypedPartition([0, data['n_users'], n], world)
        blk = tp.local_block(a, rank, int(os.environ.get('EXP_TYPE', 0)))       # (round 4: one row block per node type)
        os.environ['AMAR_SPMM_LT'] = '1'
        img = blk.tiled_image(F)
        x = torch.randn((world * tp.R, F), device=dev)
        y = torch.empty((blk.shape[0], F), device=dev)
        run = lambda: capi.spmm_xs(img, x, y, prescaled=True)
        breaks = ()
        n = blk.shape[0]
    elif what == 'gat':
        from tools.exp_gat_lt import edge_csr
        os.environ['AMAR_SPMM_LT'] = '1'
        e = edge_csr(data, dev)
        img = e.tiled_gat_image(F)
        ss_, sn_, b = torch.randn(n, device=dev), torch.randn(n, device=dev), torch.zeros(F, device=dev)
        run = lambda: capi.gat_lt(img, e, x, ss_, sn_, b, y)
        breaks = e.row_breaks
    else:
        img = a.tiled_image(F)
        xs_tab = torch.empty_like(x)
        capi.row_affine(x, img.col_scale, xs_tab)
        run = lambda: capi.spmm_xs(img, xs_tab, y, prescaled=True)
        breaks = a.row_breaks
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    T = img.n_tiles
    buf = (ctypes.c_ulonglong * (4 * T))()
    lib.amar_lt_debug_copy.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.amar_lt_debug_copy(buf, 4 * T) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(T, 4).astype(np.int64)
    tb = img.tile_row0.cpu().numpy()
    ss = torch.cat([img.stream_start.long(), torch.tensor([img.words.numel()], device=dev)])
    ent = (ss[1:] - ss[:-1]).view(-1, 16).sum(1).cpu().numpy()
    t0 = st[:, 0].min()
    print('%s ml1m(s=%d) F=%d: %d tiles, launch span %.0f k cycles (= %.4f ms at 2.4 GHz)' % (what, scale, F, T, (st[:, 3].max() - t0) / 1e3, (st[:, 3].max() - t0) / 2.4e6))
    br = [0] + list(breaks) + [n]
    for k in range(len(br) - 1):
        sel = (tb[:-1] >= br[k]) & (tb[:-1] < br[k + 1])
        if not sel.any():
            continue
        d = st[sel]
        tot = d[:, 3] - d[:, 0]
        print(' segment %d: %4d tiles | start %.0f..%.0f k | walk min %.0f median %.0f max %.0f k | epilogue median %.1f k | total median %.0f max %.0f k | '
              'end median %.0f max %.0f k | entries median %d -> %.2f cycles/entry' % (
                  k, sel.sum(), (d[:, 0].min() - t0) / 1e3, (d[:, 0].max() - t0) / 1e3, (d[:, 2] - d[:, 1]).min() / 1e3,
                  np.median(d[:, 2] - d[:, 1]) / 1e3, (d[:, 2] - d[:, 1]).max() / 1e3, np.median(d[:, 3] - d[:, 2]) / 1e3, np.median(tot) / 1e3, tot.max() / 1e3,
                  np.median(d[:, 3] - t0) / 1e3, (d[:, 3] - t0).max() / 1e3, int(np.median(ent[sel])), np.median(tot) / max(1, np.median(ent[sel]))))
        idx = np.where(sel)[0]
        order = np.argsort(-tot)
        tile_of_word = torch.repeat_interleave(torch.arange(T * 16, device=dev), ss[1:] - ss[:-1]) // 16
        flagged = torch.bincount(tile_of_word[img.words < 0], minlength=T).cpu().numpy()
        vc, rr = img.vcount.cpu().numpy(), tb[1:] - tb[:-1]
        for lab, pick in (('slowest', order[:4]), ('fastest', order[-4:])):
            for o in pick:
                t = idx[o]
                print('    %s tile %4d: total %.0f k | entries %d rows %d vrows %d flagged %.2f %% windows %d' % (
                    lab, t, tot[o] / 1e3, ent[t], rr[t], vc[t], 100.0 * flagged[t] / max(1, ent[t]), int(img.n_win[t])))


if __name__ == '__main__':
    main()
