#!/usr/bin/env python
"""Same-box A/B of two builds of libamar_hip.so on amar_dense_f32 (BERT tower shapes).
usage: python tools/exp_dense_ab.py tools/libamar_hip_old.so   (B = the in-tree build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == '--child':
    import torch
    from deep_cbrs_amar_renaissance_amd import capi
    if sys.argv[2] != '-':
        capi.LIB_PATH = os.path.abspath(sys.argv[2])
    capi.load()
    dev = torch.device('cuda')
    for M, K, N in ((386304, 768, 256), (386304, 256, 64), (204288, 768, 256), (590592, 24, 24)):
        x = torch.randn((M, K), device=dev); w = torch.randn((K, N), device=dev) * 0.05; b = torch.randn(N, device=dev)
        y = torch.empty((M, N), device=dev)
        for _ in range(3): capi.dense(x, w, b, y, act='relu')
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): capi.dense(x, w, b, y, act='relu')
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        t = min(ts)
        print('%s: [%d x %d] . [%d x %d]: %.3f ms = %.1f TFLOP/s  checksum %.6e' % (sys.argv[2], M, K, K, N, t, 2.0 * M * K * N / t / 1e9, float(y.double().sum())), flush=True)
        del x, y
else:
    for lib in (sys.argv[1], '-', sys.argv[1], '-'):
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], check=True)
