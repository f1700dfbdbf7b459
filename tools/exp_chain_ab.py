#!/usr/bin/env python
"""Same-box A/B of two builds of libamar_hip.so on the scoring head of bench.py's workload (ml1m(s) grid1): the per-entity towers
(24 -> 24 -> 24 -> 48, last layer linear) and the pair stage on the prepared pair list.  Each build runs twice, in child processes.
usage: python tools/exp_chain_ab.py tools/libamar_hip_old.so [scale]   (B = the in-tree build)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == '--child':
    import torch
    from deep_cbrs_amar_renaissance_amd import capi, engine
    if sys.argv[2] != '-':
        capi.LIB_PATH = os.path.abspath(sys.argv[2])
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from tools.profile_step import timeit
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(int(sys.argv[3]), device=dev)
    nu, ni = data['n_users'], data['n_items']
    u = data['test'][:, 0].to(torch.int32).contiguous(); i = data['test'][:, 1].to(torch.int32).contiguous()
    plan = basic.PairPlan(u, i)
    for units, clf in (([24, 24], [48, 48]), ([48, 48], [64, 64])):
        engine.set_seed(1)
        rs = basic.BasicRS(units, clf)
        d = units[0]
        rs.build_head(d, d)
        emb = torch.randn((nu + ni, d), device=dev)
        for _ in range(30):                                         # settle the clocks
            tw = rs.towers(emb[:nu], emb[nu:])
            out = rs.score_towers(tw, u, i, 0, nu, pair_plan=plan)
        t_tow, tmin_tow = timeit(lambda: rs.towers(emb[:nu], emb[nu:]), reps=40)
        t_pair, tmin_pair = timeit(lambda: rs.score_towers(tw, u, i, 0, nu, pair_plan=plan), reps=40)
        print('%s dense %s clf %s: towers %.4f ms (min %.4f) | pair stage %.4f ms (min %.4f) | checksums towers %.9e scores %.9f' % (
            sys.argv[2], units, clf, t_tow, tmin_tow, t_pair, tmin_pair, float(tw[0].double().sum() + tw[1].double().sum()),
            float(out.double().sum())), flush=True)
else:
    scale = sys.argv[2] if len(sys.argv) > 2 else '64'
    for lib in (sys.argv[1], '-', sys.argv[1], '-'):
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child', lib, scale], check=True)
