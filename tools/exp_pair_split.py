#!/usr/bin/env python
"""Pair stage of bench.py's workload in its two forms — f32 MFMA (AMAR_PAIR_MFMA=f32), and the default split-bf16 MFMA —: one process per form (the switch is read once), scores saved and compared against a float64 evaluation of the same
weights.  python tools/exp_pair_split.py [scale]   (parent: runs all, compares)"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(scale, tag, width):
    import numpy as np
    import torch
    from tools.profile_step import timeit
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    nu, ni = data['n_users'], data['n_items']
    g = torch.Generator(device=dev); g.manual_seed(42)
    perm = torch.randperm(data['test'].shape[0], device=dev, generator=g)
    u = data['test'][perm, 0].to(torch.int32).contiguous()
    i = data['test'][perm, 1].to(torch.int32).contiguous()
    engine.set_seed(1)
    rs = basic.BasicRS([width // 2, width // 2], [width, width])
    rs.build_head(24, 24)
    emb = torch.randn((nu + ni, 24), device=dev, generator=g)
    tw = rs.towers(emb[:nu], emb[nu:])
    plan = basic.PairPlan(u, i)
    run = lambda: rs.score_towers(tw, u, i, 0, nu, pair_plan=plan)
    t, tmin = timeit(run, reps=30)
    one = (lambda: rs.score_towers(tw, torch.zeros_like(u), torch.full_like(i, nu), 0, nu))
    t1, t1min = timeit(one, reps=20)
    s = run()
    if tag == 'f32':                                               # float64 reference of the same head on a sample of the pairs
        import numpy as np
        sel = torch.arange(0, u.numel(), 97, device=dev)
        w = lambda l: (l.kernel.detach().double(), l.bias.detach().double())
        xu, xi = emb[:nu].double()[u[sel].long()], emb[nu:].double()[i[sel].long() - nu]
        for l in rs.unet.layers:
            k, b = w(l); xu = torch.relu(xu @ k + b)
        for l in rs.inet.layers:
            k, b = w(l); xi = torch.relu(xi @ k + b)
        x = torch.cat([xu, xi], 1)
        for l in list(rs.clf.layers)[:-1]:
            k, b = w(l); x = torch.relu(x @ k + b)
        k, b = w(list(rs.clf.layers)[-1])
        np.save('/tmp/pair_scores_ref_%d.npy' % width, torch.sigmoid(x @ k + b).cpu().numpy())
    np.save('/tmp/pair_scores_%s_%d.npy' % (tag, width), s.cpu().numpy())
    print('%-6s width %d: %.4f ms (min %.4f) for %d pairs; one row each (no memory system) %.4f ms' % (tag, width, t, tmin, u.numel(), t1), flush=True)


def main():
    if len(sys.argv) > 3:
        return child(int(sys.argv[1]), sys.argv[2], int(sys.argv[3]))
    import numpy as np
    scale = sys.argv[1] if len(sys.argv) > 1 else '64'
    for width in (48, 64):
        for tag in ('f32', 'split'):
            env = dict(os.environ)
            env.pop('AMAR_PAIR_MFMA', None); env.pop('AMAR_PAIR_PROJ', None)
            if tag == 'f32':
                env['AMAR_PAIR_MFMA'] = 'f32'
            subprocess.run([sys.executable, os.path.abspath(__file__), scale, tag, str(width)], env=env, check=True)
        ref = np.load('/tmp/pair_scores_ref_%d.npy' % width).ravel()
        a = np.load('/tmp/pair_scores_f32_%d.npy' % width).ravel()
        for tag in ('f32', 'split'):
            b = np.load('/tmp/pair_scores_%s_%d.npy' % (tag, width)).ravel()
            d = np.abs(a.astype(np.float64) - b)
            e = np.abs(b[::97].astype(np.float64) - ref)
            print('width %d %-5s: against float64 max |err| %.3e mean %.3e | against the f32 form max |diff| %.3e, differing %d of %d' %
                  (width, tag, e.max(), e.mean(), d.max(), int((a != b).sum()), a.size), flush=True)


if __name__ == '__main__':
    main()
