#!/usr/bin/env python
"""A/B inside one process: the pair stage on the plain chain kernel vs the software-pipelined one (AMAR_CHAIN_PIPE is read
once per process, so each variant runs in a child process).  python tools/exp_chain_pipe.py [scale]"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from tools.profile_step import timeit
from deep_cbrs_amar_renaissance_amd import capi, engine
from deep_cbrs_amar_renaissance_amd.data import synthetic
from deep_cbrs_amar_renaissance_amd.models import basic
capi.load()
dev = torch.device('cuda')
scale = int(sys.argv[1])
data = synthetic.ml1m_device(scale, device=dev)
nu, ni = data['n_users'], data['n_items']
g = torch.Generator(device=dev); g.manual_seed(42)
perm = torch.randperm(data['test'].shape[0], device=dev, generator=g)
u = data['test'][perm, 0].to(torch.int32).contiguous(); i = data['test'][perm, 1].to(torch.int32).contiguous()
for units, clf in (([24, 24], [48, 48]), ([48, 48], [64, 64])):
    engine.set_seed(1)
    rs = basic.BasicRS(units, clf)
    d = 24 if units[0] == 24 else 48
    rs.build_head(d, d)
    emb = torch.randn((nu + ni, d), device=dev)
    tw = rs.towers(emb[:nu], emb[nu:])
    out = rs.score_towers(tw, u, i, 0, nu)
    t, tmin = timeit(lambda: rs.score_towers(tw, u, i, 0, nu), reps=20)
    print('dense %%s clf %%s: %%.3f ms (min %%.3f)  checksum %%.9f' %% (units, clf, t, tmin, float(out.double().sum())), flush=True)
''' % ROOT


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else '64'
    for pipe in ('0', '1', '0', '1'):
        env = dict(os.environ, AMAR_CHAIN_PIPE=pipe)
        print('AMAR_CHAIN_PIPE=' + pipe, flush=True)
        subprocess.run([sys.executable, '-c', CHILD, scale], env=env, check=True)


if __name__ == '__main__':
    main()
