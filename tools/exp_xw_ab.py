#!/usr/bin/env python
"""Same-box A/B of two builds of libamar_hip.so on amar_rowwise_xw_f32 at ml1m(s=64) shapes.
usage: python tools/exp_xw_ab.py tools/libamar_hip_old.so   (B = the in-tree build)"""
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
    n = 590592
    for F, C, copy, attn in ((8, 8, True, False), (8, 8, False, False), (8, 8, False, True), (16, 16, True, False)):
        x = torch.randn((n, F), device=dev); w = torch.randn((F, C), device=dev).contiguous(); h = torch.empty((n, C), device=dev)
        cat = torch.empty((n, 3 * F), device=dev)
        kw = {}
        if copy:
            kw['copy_to'] = cat[:, :F]
        if attn:
            kw.update(a_self=torch.randn(C, device=dev), a_neigh=torch.randn(C, device=dev), s_self=torch.empty(n, device=dev), s_neigh=torch.empty(n, device=dev))
        else:
            kw['row_scale'] = torch.rand(n, device=dev)
        for _ in range(3): capi.rowwise_xw(x, w, h, **kw)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): capi.rowwise_xw(x, w, h, **kw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20)
        print('%s: %d -> %d copy=%s attn=%s: %.4f ms  checksum %.6e' % (sys.argv[2], F, C, copy, attn, min(ts), float(h.double().sum())), flush=True)
else:
    for lib in (sys.argv[1], '-'):
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], check=True)
