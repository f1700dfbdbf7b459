#!/usr/bin/env python
"""Device time of ONE rank's step in a world-W node-range partition of bench.py's workload, on one GPU (development aid for the
scaling estimate of DESIGN.md §6): the rank's own kernels run for real — its row block on LT / XS, the replicated X.W, towers,
its pair shard — while the two all-gathers are replaced by a local copy of the rank's own block (wrong neighbours' data, right
sizes), so the number is the step WITHOUT the exchange.  `python tools/exp_rank_of_n.py [scale] [world ...]`."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

GRID1 = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4,
             final_node='concatenation', activation='relu')


class LocalCopy:
    """all_gather_into_tensor stand-in: only this rank's block lands (at its place); the other blocks keep last step's bytes.
    EXP_NO_COPY=1: not even that (the first call of every buffer still copies, so the tables hold finite values) — the step
    without ANY exchange work, three ~5 us device copies less."""
    def __init__(self, rank):
        self.rank, self.seen = rank, set()

    def all_gather_into_tensor(self, out, inp):
        r = inp.shape[0]
        if os.environ.get('EXP_NO_COPY') == '1' and out.data_ptr() in self.seen:
            return
        self.seen.add(out.data_ptr())
        out[self.rank * r:(self.rank + 1) * r].copy_(inp)


def main():
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    worlds = [int(v) for v in sys.argv[2:]] or [2, 4, 8]
    from deep_cbrs_amar_renaissance_amd import capi, engine, parallel
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    capi.load()
    dev = torch.device('cuda')
    name = os.environ.get('EXP_MODEL', 'BasicGCN')                  # BasicGCN | BasicLightGCN | BasicGraphSage | BasicGAT | HybridBertGCN
    hybrid_uip = name == 'HybridBertGCN'                            # BASELINE config 5: hybrid head on the user-item-property graph
    data = synthetic.ml1m_device(scale, device=dev, with_props=hybrid_uip)
    n = data['n_users'] + data['n_items'] + (data['n_props'] if hybrid_uip else 0)
    rows_, cols_ = data['train_pos'][:, 0], data['train_pos'][:, 1]
    if hybrid_uip:
        rows_, cols_ = torch.cat([rows_, data['item_prop'][:, 0]]), torch.cat([cols_, data['item_prop'][:, 1]])
    a = gcn_filter_device(rows_, cols_, n)
    engine.set_seed(42)
    if name in ('BasicGraphSage', 'BasicGAT'):                      # edge-list graphs: the raw symmetric adjacency, no values
        from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR
        tp_ = data['train_pos']
        keys = torch.unique(torch.cat([tp_[:, 0] * n + tp_[:, 1], tp_[:, 1] * n + tp_[:, 0]]))
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(keys // n, minlength=n), 0)
        a = DeviceCSR(rowptr.to(torch.int32), (keys % n).to(torch.int32), None, (n, n))
        a.row_breaks = (data['n_users'],)
    if hybrid_uip:
        from deep_cbrs_amar_renaissance_amd.models import hybrid
        model = hybrid.HybridBertGCN(a, embedding_dim=8, n_hiddens=[8, 8], dense_units=[[24, 24], [256, 64], [64, 64]], clf_units=[64, 64], feature_based=True)
        gb = torch.Generator(device=dev); gb.manual_seed(7)
        model.set_bert_table(torch.randn((data['n_users'] + data['n_items'], 768), device=dev, generator=gb) * 0.5)
        model.rs.build_head(model.gnn.output_dim(), 768)
    else:
        model = getattr(basic, name)(a, **GRID1)
    model.n_users, model.n_items = data['n_users'], data['n_items']
    g = torch.Generator(device=dev); g.manual_seed(42)
    perm = torch.randperm(data['test'].shape[0], device=dev, generator=g)
    u = data['test'][perm, 0].to(torch.int32).contiguous()
    i = data['test'][perm, 1].to(torch.int32).contiguous()
    if os.environ.get('EXP_SINGLE_MS'):                    # (profiling runs: skip the single-GPU leg, EXP_RANKS picks the ranks)
        t1 = float(os.environ['EXP_SINGLE_MS']) * 1e-3
    elif hybrid_uip:                                       # one GPU: propagation + the four per-entity networks + every pair (eager: a 7 ms step)
        nu_, ni_, bert_ = data['n_users'], data['n_items'], model.bert_table

        def single_step():
            emb = model.gnn(None)
            tw = model.rs.towers(emb[:nu_], emb[nu_:nu_ + ni_], bert_[:nu_], bert_[nu_:nu_ + ni_])
            return model.rs.score_towers(tw, u, i, 0, nu_)
        for _ in range(3):
            single_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            single_step()
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t0) / 10
    else:
        single = parallel.SingleRunner(model, u, i)
        for _ in range(40):
            single.step_graphed()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            single.step_graphed()
        torch.cuda.synchronize()
        t1 = (time.perf_counter() - t0) / 20
        del single
    print('%s ml1m(s=%d): single GPU %.4f ms per step (graph-replayed)' % (name, scale, 1e3 * t1), flush=True)
    for world in worlds:
        ranks = [int(r) for r in os.environ['EXP_RANKS'].split(',') if int(r) < world] if os.environ.get('EXP_RANKS') else sorted({0, world // 2, world - 1})
        for rank in ranks:
            runner = parallel.PartitionedGCNRunner(model, u, i, rank, world, dist=LocalCopy(rank), timing=False)
            for _ in range(40):
                runner.step_graphed()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                runner.step_graphed()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            tp = runner.tpart                      # per layer: the next gathered table (all rows) + the item rows of X_l
            gathered = (sum(runner.widths[2:]) * world * tp.R + sum(runner.widths[1:]) * world * tp.h[1]) * 4 / 1e6
            print('  world %d rank %d: rows %d nnz %d pairs %d | %.4f ms per step without the exchange (%.1f MB gathered per step) -> '
                  'bound on the speed-up %.2fx' % (world, rank, runner.local_rows, runner.local_nnz, runner.u_ids.numel(), 1e3 * dt, gathered, t1 / dt), flush=True)
            runner.timing = True                   # phases of one eager step (HIP events)
            for _ in range(3):
                runner.step()
            ph = runner.phase_times()
            print('      eager step by phase: ' + ', '.join('%s %.4f' % (k[:-3], v) for k, v in ph.items()), flush=True)
            del runner


if __name__ == '__main__':
    main()
