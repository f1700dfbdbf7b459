#!/usr/bin/env python
"""Timing of the hybrid (GNN + BERT) head at ml1m(s) scale: per-entity towers (incl. 768->256->64 BERT) and pair stage."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tools.profile_step import timeit


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    data = synthetic.ml1m_device(scale, device=dev)
    nu, ni = data['n_users'], data['n_items']
    n = nu + ni
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    engine.set_seed(42)
    model = hybrid.HybridBertGCN(a, embedding_dim=8, n_hiddens=[8, 8], dense_units=[[24, 24], [256, 64], [64, 64]],
                                 clf_units=[64, 64], feature_based=True)
    model.n_users, model.n_items = nu, ni
    bert = torch.randn((n, 768), device=dev) * 0.5
    model.set_bert_table(bert)
    model.rs.build_head(model.gnn.output_dim(), 768)
    u = data['test'][:, 0].to(torch.int32).contiguous()
    i = data['test'][:, 1].to(torch.int32).contiguous()
    P = u.numel()
    emb = model.gnn(None)
    rs = model.rs
    med, _ = timeit(lambda: rs.dense2a.apply2(bert[:nu]), reps=5)
    fl = nu * (768 * 256 + 256 * 64) * 2.0
    print('scale {}: BERT user tower ({} rows, 768->256->64): {:.3f} ms = {:.1f} TFLOP/s'.format(scale, nu, med, fl / med / 1e9), flush=True)
    tw = rs.towers(emb[:nu], emb[nu:], bert[:nu], bert[nu:])
    med_t, _ = timeit(lambda: rs.towers(emb[:nu], emb[nu:], bert[:nu], bert[nu:]), reps=5)
    med_p, _ = timeit(lambda: rs.score_towers(tw, u, i, 0, nu), reps=5)
    print('  all four entity towers: {:.3f} ms; pair stage ({} pairs): {:.3f} ms = {:.2f} G pairs/s'.format(med_t, P, med_p, P / med_p / 1e6), flush=True)
    z_u, z_i = torch.zeros_like(u), torch.full_like(i, nu)
    med_z, _ = timeit(lambda: rs.score_towers(tw, z_u, z_i, 0, nu), reps=5)
    print('  pair stage with every pair reading row 0 of each table (no memory system): {:.3f} ms'.format(med_z), flush=True)
    from deep_cbrs_amar_renaissance_amd.models.basic import PairPlan
    g = torch.Generator(device=dev); g.manual_seed(42)
    perm = torch.randperm(P, device=dev, generator=g)
    us, it = u[perm].contiguous(), i[perm].contiguous()
    med_s, _ = timeit(lambda: rs.score_towers(tw, us, it, 0, nu), reps=5)
    plan = PairPlan(us, it)
    med_q, _ = timeit(lambda: rs.score_towers(tw, us, it, 0, nu, pair_plan=plan), reps=5)
    same = torch.equal(rs.score_towers(tw, us, it, 0, nu), rs.score_towers(tw, us, it, 0, nu, pair_plan=plan))
    print('  shuffled pair list: {:.3f} ms; with the prepared list (PairPlan, two-step way back): {:.3f} ms; same bits: {}'.format(med_s, med_q, same), flush=True)
    model.gnn.hoist = True
    med_all, _ = timeit(lambda: model((u, i, None, None)), reps=5)
    print('  hoisted call (cached propagation + towers): {:.3f} ms'.format(med_all))


if __name__ == '__main__':
    main()
