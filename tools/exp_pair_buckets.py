#!/usr/bin/env python
"""Pair stage experiment: does giving every XCD its own slice of the ITEM tower table pay?

The chain kernel gathers T_u[u] and T_i[i] (192-B rows) for 12 M shuffled pairs; both tables (74 + 39 MB at s=64) exceed
the 4 MB per-XCD L2s, so nearly every gather is an L2 miss served by the Infinity Cache (5.8 GB of fabric traffic per
launch).  Pair p is processed by XCD (p / 128) % 8 (wave = p / 32, block = wave / 4, XCD = block % 8): permuting the pair
list so that XCD x only sees items of slice x (equal pair counts per slice) keeps each slice (~4.9 MB, or 2.4 MB with 16
slices in two phases) resident in that XCD's L2.  Measures the unchanged kernel on the permuted list, plus what the
permutation itself costs when done with torch ops (upper bound for a dedicated bucket kernel).
    python tools/exp_pair_buckets.py [scale]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def xcd_layout(order_by_bucket, n_buckets, P, dev):
    """Positions p with (p >> 7) % 8 == x, ascending, receive bucket x's pairs (phase-major for n_buckets = 8 k)."""
    pos = torch.arange(P, device=dev)
    xcd = (pos >> 7) & 7
    perm = torch.empty(P, dtype=torch.int64, device=dev)
    per = P // n_buckets
    for x in range(8):
        slots = pos[xcd == x]
        src = torch.cat([order_by_bucket[b * per:(b + 1) * per] for b in range(x, n_buckets, 8)])
        m = min(slots.numel(), src.numel())
        perm[slots[:m]] = src[:m]
        if slots.numel() > m:                                    # ragged tail: reuse the first pairs (timing only)
            perm[slots[m:]] = src[:slots.numel() - m]
    return perm


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
    perm0 = torch.randperm(data['test'].shape[0], device=dev, generator=g)
    u = data['test'][perm0, 0].to(torch.int32).contiguous()
    i = data['test'][perm0, 1].to(torch.int32).contiguous()
    P = u.numel()
    engine.set_seed(1)
    rs = basic.BasicRS([24, 24], [48, 48])
    rs.build_head(24, 24)
    emb = torch.randn((nu + ni, 24), device=dev)
    tw = rs.towers(emb[:nu], emb[nu:])
    ref = rs.score_towers(tw, u, i, 0, nu)
    t0, _ = timeit(lambda: rs.score_towers(tw, u, i, 0, nu), reps=10)
    print('P = {}, shuffled order: {:.3f} ms'.format(P, t0), flush=True)
    for name, key in (('item', i), ('user', u)):
        order = torch.argsort(key.to(torch.int64), stable=True)
        for n_buckets in (8, 16, 32):
            perm = xcd_layout(order, n_buckets, P, dev)
            u2, i2 = u[perm].contiguous(), i[perm].contiguous()
            t, _ = timeit(lambda: rs.score_towers(tw, u2, i2, 0, nu), reps=10)
            out = rs.score_towers(tw, u2, i2, 0, nu)
            ok = bool(torch.equal(out, ref[perm]))
            print('{}-sliced, {:2d} buckets (sorted inside a bucket): {:.3f} ms   same scores: {}'.format(name, n_buckets, t, ok), flush=True)
    # buckets WITHOUT sorting inside them: what a counting pass by slice id delivers
    bounds_src = torch.sort(i.to(torch.int64)).values
    for n_buckets in (8, 16):
        per = P // n_buckets
        bounds = bounds_src[torch.arange(1, n_buckets, device=dev) * per]
        slice_id = torch.bucketize(i.to(torch.int64), bounds, right=True)
        order = torch.argsort(slice_id, stable=True)
        t_sort, _ = timeit(lambda: torch.argsort(torch.bucketize(i.to(torch.int64), bounds, right=True), stable=True), reps=5)
        counts = torch.bincount(slice_id, minlength=n_buckets)
        perm = xcd_layout(order, n_buckets, P, dev)
        u2, i2 = u[perm].contiguous(), i[perm].contiguous()
        t, _ = timeit(lambda: rs.score_towers(tw, u2, i2, 0, nu), reps=10)
        print('item-sliced, {:2d} buckets, shuffled inside: {:.3f} ms  (bucket sizes {}..{}; torch bucketize+argsort {:.3f} ms)'.format(
            n_buckets, t, int(counts.min()), int(counts.max()), t_sort), flush=True)
    # scattered 4-byte score writes (the price of restoring the caller's order)
    out = torch.empty(P, device=dev)
    src = torch.randn(P, device=dev)
    t_sc, _ = timeit(lambda: out.index_copy_(0, perm, src), reps=10)
    print('scatter of {} scores back to the caller order (torch index_copy_): {:.3f} ms'.format(P, t_sc), flush=True)


if __name__ == '__main__':
    main()
