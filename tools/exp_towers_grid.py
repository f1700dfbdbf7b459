#!/usr/bin/env python
"""Per-entity towers of bench.py's workload (386 304 + 204 288 rows, 24 -> 24 -> 24 -> 48) under the generic chain kernel's launch
switches: AMAR_CHAIN_GRID (workgroup cap) x AMAR_CHAIN_PT (16-row tiles per wave).  Development aid; each setting in a child process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == '--child':
    import torch
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.models import basic
    from tools.profile_step import timeit
    capi.load()
    dev = torch.device('cuda')
    nu, ni = 386304, 204288
    for units, clf in (([24, 24], [48, 48]), ([48, 48], [64, 64])):
        engine.set_seed(1)
        rs = basic.BasicRS(units, clf)
        rs.build_head(units[0], units[0])
        emb = torch.randn((nu + ni, units[0]), device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        for _ in range(50):
            rs.towers(emb[:nu], emb[nu:])
        t, tmin = timeit(lambda: rs.towers(emb[:nu], emb[nu:]), reps=40)
        tw = rs.towers(emb[:nu], emb[nu:])
        print('grid cap %5s PT %s dense %s: towers %.4f ms (min %.4f) checksum %.9e' % (os.environ.get('AMAR_CHAIN_GRID', '4096'), os.environ.get('AMAR_CHAIN_PT', '2'), units, t, tmin,
              float(tw[0].double().sum() + tw[1].double().sum())), flush=True)
else:
    # the compile-time-shaped tower kernel (chain_rows_kernel) against the generic one, then its workgroup cap
    for rows, cap in (('0', '4096'), ('1', '4096'), ('0', '4096'), ('1', '4096'), ('1', '2048'), ('1', '1536'), ('1', '1024'), ('1', '768')):
        print('AMAR_CHAIN_ROWS=' + rows, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=dict(os.environ, AMAR_CHAIN_GRID=cap, AMAR_CHAIN_ROWS=rows), check=True)
