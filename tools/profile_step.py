#!/usr/bin/env python
"""Per-kernel timing of the hot path with HIP events (development aid, not part of the product).

usage: python tools/profile_step.py [--scales 1,16,64] [--widths 8,16,32]
Prints, per scale: SpMM time / algorithmic GB/s for each width (plain and fused-GCN form),
the dense-head layers, and the full step.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scales', default='1,16,64')
    ap.add_argument('--widths', default='8,16,32')
    args = ap.parse_args()
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    for s in [int(x) for x in args.scales.split(',')]:
        data = synthetic.ml1m_device(s, device=dev)
        n = data['n_users'] + data['n_items']
        a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
        nnz = a.nnz
        deg = (a.rowptr[1:] - a.rowptr[:-1]).float()
        print('== scale {}: N={} nnz={} pairs={} deg mean {:.1f} max {:.0f}'.format(
            s, n, nnz, data['test'].shape[0], float(deg.mean()), float(deg.max())), flush=True)
        for F in [int(x) for x in args.widths.split(',')]:
            x = torch.randn((n, F), device=dev)
            y = torch.empty((n, F), device=dev)
            b = torch.randn(F, device=dev)
            w = torch.randn((F, F), device=dev)
            hn = torch.empty((n, F), device=dev)
            alg = nnz * 8 + (n + 1) * 4 + 2 * n * F * 4
            med, best = timeit(lambda: capi.spmm_csr(a.rowptr, a.colidx, a.vals, x, y))
            print('  spmm F={:2d}: {:8.3f} ms (best {:8.3f})  alg {:7.1f} MB -> {:7.1f} GB/s ({:4.1f}% of 8 TB/s)'.format(
                F, med, best, alg / 1e6, alg / med / 1e6, 100 * alg / med / 1e6 / 8000), flush=True)
            med, best = timeit(lambda: capi.gcn_layer(a.rowptr, a.colidx, a.vals, x, b, y, Wnext=w, Hnext=hn))
            print('  gcn  F={:2d}: {:8.3f} ms (best {:8.3f})  fused bias+relu+next XW -> {:7.1f} GB/s'.format(
                F, med, best, alg / med / 1e6), flush=True)
            xs = a.xcd_sliced()
            med, best = timeit(lambda: capi.spmm_xs(xs, x, y))
            print('  XS   F={:2d}: {:8.3f} ms (best {:8.3f})  -> {:7.1f} GB/s ({:4.1f}% of 8 TB/s)'.format(F, med, best, alg / med / 1e6, 100 * alg / med / 1e6 / 8000), flush=True)
            med, best = timeit(lambda: capi.spmm_xs(xs, x, y, bias=b, relu=True, Wnext=w, Hnext=hn))
            print('  XS gcn F={:2d}: {:8.3f} ms (best {:8.3f})  -> {:7.1f} GB/s'.format(F, med, best, alg / med / 1e6), flush=True)
            sj = a.sliced(F)
            med, best = timeit(lambda: capi.spmm_sj(sj, x, y))
            print('  SJ   F={:2d}: {:8.3f} ms (best {:8.3f})  {} slices -> {:7.1f} GB/s ({:4.1f}% of 8 TB/s)'.format(
                F, med, best, sj.n_slices, alg / med / 1e6, 100 * alg / med / 1e6 / 8000), flush=True)
            med, best = timeit(lambda: capi.spmm_sj(sj, x, y, bias=b, relu=True, Wnext=w, Hnext=hn))
            print('  SJ gcn F={:2d}: {:8.3f} ms (best {:8.3f})  -> {:7.1f} GB/s'.format(F, med, best, alg / med / 1e6), flush=True)
        engine.set_seed(42)
        model = basic.BasicGCN(a, embedding_dim=8, n_hiddens=[8, 8], dense_units=[24, 24], clf_units=[48, 48])
        u = data['test'][:, 0].to(torch.int32).contiguous()
        i = data['test'][:, 1].to(torch.int32).contiguous()
        P = u.numel()
        med, _ = timeit(lambda: model.gnn(None), reps=10)
        print('  propagation (xw + 2 fused layers): {:8.3f} ms'.format(med))
        emb = model.gnn(None)
        nu, ni = data['n_users'], data['n_items']
        def head():
            return model.rs.score_towers(model.rs.towers(emb[:nu], emb[nu:nu + ni]), u, i, 0, nu)
        med, _ = timeit(head, reps=5)
        tw = model.rs.towers(emb[:nu], emb[nu:nu + ni])
        med_t, _ = timeit(lambda: model.rs.towers(emb[:nu], emb[nu:nu + ni]), reps=5)
        med_c, _ = timeit(lambda: model.rs.score_towers(tw, u, i, 0, nu), reps=5)
        print('  towers (per entity): {:8.3f} ms   pair clf (gather + 48-48-48-1): {:8.3f} ms -> {:6.2f} G pairs/s'.format(med_t, med_c, P / med_c / 1e6))
        med_f, _ = timeit(lambda: model.rs([emb, emb], u_ids=u, i_ids=i), reps=5)
        print('  per-pair towers + clf (faithful form): {:8.3f} ms'.format(med_f))
        flops = P * 13920.0
        print('  head over {} pairs: {:8.3f} ms -> {:6.2f} G pairs/s, {:6.2f} TFLOP/s'.format(P, med, P / med / 1e6, flops / med / 1e9))
        t = torch.empty((P, 24), device=dev)
        lay = model.rs.unet.layers[0]
        med, _ = timeit(lambda: capi.dense(emb, lay.kernel, lay.bias, t, act='relu', ids=u), reps=5)
        print('    tower L1 (gather 24->24): {:8.3f} ms'.format(med))
        x48 = torch.randn((P, 48), device=dev)
        y48 = torch.empty((P, 48), device=dev)
        lay = model.rs.clf.layers[1]
        med, _ = timeit(lambda: capi.dense(x48, lay.kernel, lay.bias, y48, act='relu'), reps=5)
        print('    clf 48->48: {:8.3f} ms ({:5.1f} TFLOP/s, {:6.1f} GB/s)'.format(med, P * 48 * 48 * 2 / med / 1e9, P * 96 * 4 / med / 1e6))
        lay = model.rs.clf.layers[2]
        y1 = torch.empty((P, 1), device=dev)
        med, _ = timeit(lambda: capi.dense(x48, lay.kernel, lay.bias, y1, act='sigmoid'), reps=5)
        print('    clf 48->1: {:8.3f} ms'.format(med))
        del data, a, model, emb, t, x48, y48
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
