#!/usr/bin/env python
"""A/B of the chain kernel's pair-tile count (AMAR_CHAIN_PT, read at first launch) on the bench's pair stage."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.models import basic
    capi.load()
    dev = torch.device('cuda')
    nu, ni = 6036 * scale, 3192 * scale
    P = 189000 * scale
    g = torch.Generator(device=dev); g.manual_seed(0)
    for cfg_name, F, dense, clf in (('grid1', 24, [24, 24], [48, 48]), ('grid2', 48, [48, 48], [64, 64]), ('grid6', 128, [128, 64], [64, 64])):
        engine.set_seed(1)
        rs = basic.BasicRS(dense, clf)
        rs.build_head(F, F)
        emb = torch.randn((nu + ni, F), device=dev)
        u = torch.randint(0, nu, (P,), device=dev, generator=g).to(torch.int32)
        i = (torch.randint(0, ni, (P,), device=dev, generator=g) + nu).to(torch.int32)
        tw = rs.towers(emb[:nu], emb[nu:])
        med_t, _ = timeit(lambda: rs.towers(emb[:nu], emb[nu:]), reps=10)
        med_c, _ = timeit(lambda: rs.score_towers(tw, u, i, 0, nu), reps=10)
        d = dense[-1]
        flops = P * (2 * d * clf[0] + clf[0] * clf[1]) * 2.0
        us = torch.sort(u).values
        med_s, _ = timeit(lambda: rs.score_towers(tw, us, i, 0, nu), reps=10)
        print('{} PT={}: towers {:.3f} ms, pair clf {:.3f} ms ({:.2f} G pairs/s, {:.1f} TFLOP/s), user-sorted pairs {:.3f} ms'.format(
            cfg_name, os.environ.get('AMAR_CHAIN_PT', 'default'), med_t, med_c, P / med_c / 1e6, flops / med_c / 1e9, med_s), flush=True)


if __name__ == '__main__':
    main()
