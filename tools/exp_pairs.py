#!/usr/bin/env python
"""Pair stage A/B (development aid): classifier's first layer folded into the per-entity towers (48-wide rows, 2 lines
per gathered row) vs un-folded 24-wide tower rows (one 128-B line per row when padded to ld=32, but 2x the MFMA work)."""
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
    engine.set_seed(1)
    rs = basic.BasicRS([24, 24], [48, 48])
    rs.build_head(24, 24)
    emb = torch.randn((nu + ni, 24), device=dev)
    u = torch.randint(0, nu, (P,), device=dev, generator=g).to(torch.int32)
    i = (torch.randint(0, ni, (P,), device=dev, generator=g) + nu).to(torch.int32)
    tw = rs.towers(emb[:nu], emb[nu:])
    t_fold, _ = timeit(lambda: rs.score_towers(tw, u, i, 0, nu), reps=10)
    ref = rs.score_towers(tw, u, i, 0, nu)
    for ld in (24, 32):
        tu_buf, ti_buf = torch.zeros((nu, ld), device=dev), torch.zeros((ni, ld), device=dev)
        tu, ti = tu_buf[:, :24], ti_buf[:, :24]
        capi.copy_columns(rs.unet.apply2(emb[:nu]), tu)
        capi.copy_columns(rs.inet.apply2(emb[nu:]), ti)
        out = rs.clf.apply2(tu, ti, ids_a=u, base_a=0, ids_b=i, base_b=nu)
        err = float((out - ref).abs().max())
        t, _ = timeit(lambda: rs.clf.apply2(tu, ti, ids_a=u, base_a=0, ids_b=i, base_b=nu), reps=10)
        print('unfolded ld={}: {:.3f} ms  (max diff vs folded {:.2e})'.format(ld, t, err), flush=True)
    print('folded (48-wide rows): {:.3f} ms'.format(t_fold), flush=True)


if __name__ == '__main__':
    main()
