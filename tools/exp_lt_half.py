#!/usr/bin/env python
"""The LDS-tiled SpMM with HALF-height tiles, two workgroups resident per CU (development experiment).  Builds
tools/libamar_hip_half.so from the library's sources with -DAMAR_LT_TILE_BYTES=61440 -DAMAR_LT_MIN_WAVES=8 (Y tile of 120 rows per
wave: 2 x (60 KB + 16 KB ring) fit the 160 KB of a CU; <= 64 VGPRs so that 32 waves fit) and times the plain product on ml1m(s)
against the shipped geometry (128 KB tile, one workgroup of 16 waves per CU), each in a child process.
    python tools/exp_lt_half.py build            (no GPU needed)
    python tools/exp_lt_half.py [scale] [F]      (on the GPU box)"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'libamar_hip_half.so')
CSRC = os.path.join(ROOT, 'deep_cbrs_amar_renaissance_amd', 'csrc')
HALF_BYTES = 61440


def build():
    srcs = ['amar_capi.hip', 'amar_propagate.hip', 'amar_dense.hip', 'amar_layout.hip', 'amar_chain.hip', 'amar_train.hip']
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-mllvm', '-amdgpu-mfma-vgpr-form=1',
           '-DAMAR_LT_TILE_BYTES=%d' % HALF_BYTES, '-DAMAR_LT_MIN_WAVES=8', '-shared'] + [os.path.join(CSRC, s) for s in srcs] + ['-o', LIB]
    subprocess.run(cmd, check=True)
    print('built', LIB)


def child(half, scale, F):
    import torch
    from deep_cbrs_amar_renaissance_amd import capi
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities import lds_tiled
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    from tools.profile_step import timeit
    if half:
        capi.LIB_PATH = LIB
        lds_tiled.TILE_BYTES = HALF_BYTES
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    x = torch.randn((n, F), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    os.environ['AMAR_SPMM_LT'] = '1'
    img = a.tiled_image(F)
    xs_tab = torch.empty_like(x)
    capi.row_affine(x, img.col_scale, xs_tab)
    y = torch.empty((n, F), device=dev)
    for _ in range(30):
        capi.spmm_xs(img, xs_tab, y, prescaled=True)
    t, tmin = timeit(lambda: capi.spmm_xs(img, xs_tab, y, prescaled=True), reps=40)
    ref = torch.zeros((n, F), device=dev, dtype=torch.float64)
    print('%s tile: rows per wave %d, tiles %d, vrows <= %d, window %d, flagged %.2f %%: %.4f ms (min %.4f)  checksum %.9e' % (
        'HALF' if half else 'full', img.rw, img.n_tiles, int(img.vcount.max()), img.window_entries, 100.0 * img.n_flagged / max(1, img.n_entries),
        t, tmin, float(y.double().sum())), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        build()
    elif len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(sys.argv[2] == '1', int(sys.argv[3]), int(sys.argv[4]))
    else:
        scale = sys.argv[1] if len(sys.argv) > 1 else '64'
        F = sys.argv[2] if len(sys.argv) > 2 else '8'
        for half in ('0', '1', '0', '1'):
            subprocess.run([sys.executable, os.path.abspath(__file__), '--child', half, scale, F], check=True)
