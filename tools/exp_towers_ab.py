#!/usr/bin/env python
"""Same-box A/B of two builds of libamar_hip.so on the per-entity towers and the per-batch (faithful) head call.
usage: python tools/exp_towers_ab.py tools/libamar_hip_old.so   (B = the in-tree build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == '--child':
    import torch
    from deep_cbrs_amar_renaissance_amd import capi, engine
    if sys.argv[2] != '-':
        capi.LIB_PATH = os.path.abspath(sys.argv[2])
    from deep_cbrs_amar_renaissance_amd.models import basic
    from tools.profile_step import timeit
    capi.load()
    dev = torch.device('cuda')
    nu, ni = 386304, 204288
    engine.set_seed(1)
    rs = basic.BasicRS([24, 24], [48, 48])
    rs.build_head(24, 24)
    emb = torch.randn((nu + ni, 24), device=dev)
    t, _ = timeit(lambda: rs.towers(emb[:nu], emb[nu:]), reps=20)
    u = torch.randint(0, nu, (2048,), device=dev).to(torch.int32); i = (torch.randint(0, ni, (2048,), device=dev) + nu).to(torch.int32)
    t2, _ = timeit(lambda: rs([emb, emb], u_ids=u, i_ids=i), reps=50)
    tw = rs.towers(emb[:nu], emb[nu:])
    print('%s: towers over %d + %d rows %.4f ms | per-batch head call (2048 pairs) %.4f ms | checksum %.6e' % (sys.argv[2], nu, ni, t, t2, float(tw[0].double().sum() + tw[1].double().sum())), flush=True)
else:
    for lib in (sys.argv[1], '-', sys.argv[1], '-'):
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib], check=True)
