#!/usr/bin/env python
"""Same-box A/B of two builds of libamar_hip.so on the fused GCN layer (development aid).
usage: python tools/exp_ab.py tools/libamar_hip_old.so   (B = the in-tree build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == '--child':
    import torch
    from deep_cbrs_amar_renaissance_amd import capi
    if sys.argv[2] != '-':
        capi.LIB_PATH = os.path.abspath(sys.argv[2])
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(64, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    xs = a.xcd_sliced()
    x = torch.randn((n, 8), device=dev); y = torch.empty((n, 8), device=dev)
    b = torch.randn(8, device=dev); w = torch.randn((8, 8), device=dev).contiguous(); h = torch.empty((n, 8), device=dev)
    def run():
        capi.spmm_xs(xs, x, y, bias=b, relu=True, Wnext=w, Hnext=h)
    for _ in range(5): run()
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20)
    print('%s: fused XS layer %.4f ms (min of 5 x 20), checksum %.6e' % (sys.argv[2], min(ts), float(y.double().sum())), flush=True)
else:
    for lib in (sys.argv[1], '-', sys.argv[1], '-'):
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], check=True)
