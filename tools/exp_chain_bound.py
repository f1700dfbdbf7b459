#!/usr/bin/env python
"""What bounds the pair stage?  The same launch with (a) the real shuffled ids, (b) every pair reading the same two rows (no
memory system: MFMA + VALU + LDS only), (c) a handful of rows (L1/L2 hits only).  python tools/exp_chain_bound.py [scale]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
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
    P = u.numel()
    engine.set_seed(1)
    rs = basic.BasicRS([24, 24], [48, 48])
    rs.build_head(24, 24)
    emb = torch.randn((nu + ni, 24), device=dev)
    tw = rs.towers(emb[:nu], emb[nu:])
    cases = {
        'shuffled ids (real)': (u, i),
        'one row each (no memory system)': (torch.zeros_like(u), torch.full_like(i, nu)),
        '1 K rows each (L1 / L2 hits)': ((u % 1024).contiguous(), (i - nu) % 1024 + nu),
        '32 K rows each (6 MB: L2 / MALL)': ((u % 32768).contiguous(), (i - nu) % 32768 + nu),
    }
    for name, (uu, ii) in cases.items():
        ii = ii.to(torch.int32).contiguous()
        t, tmin = timeit(lambda: rs.score_towers(tw, uu, ii, 0, nu), reps=20)
        print('{:40s} {:.3f} ms (min {:.3f})'.format(name, t, tmin), flush=True)


if __name__ == '__main__':
    main()
