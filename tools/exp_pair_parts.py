#!/usr/bin/env python
"""Which side of the pair stage costs what: the launch on the prepared pair list with (a) everything, (b) no scattered store (scores
written in list order), (c) the item side only (every pair reads user row 0), (d) the user side only (every pair reads item row 0),
(e) neither.  python tools/exp_pair_parts.py [scale] [width]   (AMAR_PAIR_MFMA=f32: the f32 instruction)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 48
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
    tu, ti, _ = rs.towers(emb[:nu], emb[nu:])
    plan = basic.PairPlan(u, i)
    sp = rs._split_cache[1]
    blob, dims, acts = sp['rest']
    mode = 1
    out = torch.empty((u.numel(), 1), dtype=torch.float32, device=dev)
    pu, pi, po = plan.u_ids, plan.i_ids, plan.out_index
    zu, zi = torch.zeros_like(pu), torch.full_like(pi, nu)

    def run(ua, ia, oi):
        return lambda: capi.chain(tu, blob, dims, acts, out, ids_a=ua, base_a=0, B=ti, ids_b=ia, base_b=nu, sum_inputs=mode,
                                  in_act=sp['in_act'], out_index=oi)
    print('form: %s, tower tables %s / %s (ld %d / %d)' % (os.environ.get('AMAR_PAIR_MFMA', 'split'),
                                                          tuple(tu.shape), tuple(ti.shape), tu.stride(0), ti.stride(0)))
    for name, fn in (('(a) prepared list, scattered store', run(pu, pi, po)),
                     ('(b) prepared list, stored in list order', run(pu, pi, None)),
                     ('(c) item side only (user row 0)', run(zu, pi, None)),
                     ('(d) user side only (item row 0)', run(pu, zi, None)),
                     ('(e) one row each', run(zu, zi, None)),
                     ('(f) one row each, scattered store', run(zu, zi, po))):
        t, tmin = timeit(fn, reps=30)
        print('  %-44s %.4f ms (min %.4f)' % (name, t, tmin), flush=True)
    # two-level way back: the kernel scatters into (window of the destination, XCD) streams, amar_scatter_f32 finishes inside windows
    direct = rs.score_towers((tu, ti, True), u, i, 0, nu, pair_plan=plan).clone() if plan.mid_index is None else None
    for win in (1 << 14, 1 << 15, 1 << 16, 1 << 17):
        os.environ['AMAR_PAIR_WINDOW'] = str(win)
        pl = basic.PairPlan(u, i)
        t, tmin = timeit(lambda: rs.score_towers((tu, ti, True), u, i, 0, nu, pair_plan=pl), reps=30)
        res2 = torch.empty((u.numel(), 1), dtype=torch.float32, device=dev)
        t2, t2min = timeit(lambda: capi.scatter(pl.mid, pl.final_index, res2, pl.window_off, pl.n_windows), reps=30)
        t3, t3min = timeit(lambda: capi.scatter(pl.mid, pl.final_index, res2, None, 1), reps=30)
        got = rs.score_towers((tu, ti, True), u, i, 0, nu, pair_plan=pl)
        os.environ['AMAR_PAIR_WINDOW'] = '0'
        ref = rs.score_towers((tu, ti, True), u, i, 0, nu, pair_plan=basic.PairPlan(u, i))
        print('  window %7d scores (%3d windows): both launches %.4f ms (min %.4f); the second alone %.4f ms (unwindowed walk %.4f); equal to the direct store: %s'
              % (win, pl.n_windows, t, tmin, t2, t3, bool(torch.equal(got, ref))), flush=True)
    # the un-permute as its own pass (torch's indexing kernels as a first estimate)
    tmp = out.clone().view(-1)
    res = torch.empty_like(tmp)
    inv = torch.empty_like(po)
    inv[po.long()] = torch.arange(po.numel(), dtype=torch.int32, device=dev)
    pol, invl = po.long(), inv.long()
    for name, fn in (('gather form  res = tmp[inv]', lambda: torch.index_select(tmp, 0, invl, out=res)),
                     ('scatter form res[out_index] = tmp', lambda: res.index_copy_(0, pol, tmp)),
                     ('plain copy of 48 MB', lambda: res.copy_(tmp))):
        t, tmin = timeit(fn, reps=30)
        print('  %-44s %.4f ms (min %.4f)' % (name, t, tmin), flush=True)


if __name__ == '__main__':
    main()
