"""One rank of the real-RCCL partitioned-runner test (started by tests/test_parallel_nccl_gpu.py through
`python -m torch.distributed.run`, one fresh process per GPU).  Every rank builds the same seeded graph and model, runs
parallel.PartitionedGCNRunner on the HIP kernels with `nccl` collectives, and rank 0 checks the gathered scores against
the numpy oracle (test infrastructure) — not merely against the single-GPU HIP path.  Exit code 0 = parity."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    case = sys.argv[1]
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    rehearse = os.environ.get('AMAR_REHEARSE_ONE_GPU') == '1'      # several ranks on ONE GPU: gloo + parallel.SharedDeviceCollectives
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if rehearse:
        dist.init_process_group('gloo')
    else:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    from deep_cbrs_amar_renaissance_amd import capi, engine, parallel
    from deep_cbrs_amar_renaissance_amd.models import basic, hybrid
    from oracle import models as om
    from tests import helpers
    capi.load()
    dev = torch.device('cuda', torch.cuda.current_device())
    engine.set_seed(42)
    uip = case == 'hybrid_uip'
    g = helpers.tiny_graph(n_users=400, n_items=260, n_ratings=20000, seed=8, n_props=90 if uip else 0, n_links=700 if uip else 0)
    n = g['adj'].shape[0]
    if case == 'basic_ui':
        model = basic.BasicGCN(g['adj'], embedding_dim=8, n_hiddens=[8, 8], dense_units=[24, 24], clf_units=[48, 48])
        bert = None
    else:
        model = hybrid.HybridBertGCN(g['adj'], embedding_dim=8, n_hiddens=[8, 8], dense_units=[[24, 24], [256, 64], [64, 64]],
                                     clf_units=[64, 64], feature_based=True)
        bert = (0.5 * np.random.default_rng(5).standard_normal((g['n_users'] + g['n_items'], 768))).astype(np.float32)
        model.set_bert_table(torch.from_numpy(bert).to(dev))
        model.rs.build_head(model.gnn.output_dim(), 768)
    model.n_users, model.n_items = g['n_users'], g['n_items']
    helpers.randomize_biases(model, seed=3)
    rng = np.random.default_rng(11)                                  # a pair list long enough for every rank
    P = 6000
    u_np = rng.integers(0, g['n_users'], P)
    i_np = rng.integers(0, g['n_items'], P) + g['n_users']
    u = torch.from_numpy(u_np.astype(np.int32)).to(dev)
    i = torch.from_numpy(i_np.astype(np.int32)).to(dev)
    runner = parallel.PartitionedGCNRunner(model, u, i, rank, world, dist=parallel.SharedDeviceCollectives(rank, world) if rehearse else None)
    for _ in range(2):                                               # twice: persistent buffers are reused by the second step
        scores = runner.step()
    graph_ok = True
    if not rehearse:
        # the step replayed from a hipGraph, RCCL all-gathers included, must give the eager step's bits; a weight update must
        # invalidate the captured graph (its Dense weights are packed on the host at capture time)
        eager = scores.clone()
        for _ in range(2):
            replayed = runner.step_graphed()
        graph_ok = bool(torch.equal(replayed, eager))
        with torch.no_grad():
            model.rs.clf.layers[-1].bias.add_(0.25)
        moved = runner.step().clone()
        graph_ok = graph_ok and bool(torch.equal(runner.step_graphed(), moved)) and not bool(torch.equal(moved, eager))
        with torch.no_grad():
            model.rs.clf.layers[-1].bias.sub_(0.25)
        scores = runner.step()
    # every rank scored its shard; collect (pair position, score) on all ranks
    cdev = torch.device('cpu') if rehearse else dev                 # gloo gathers host tensors
    counts = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([runner.pair_index.numel()], dtype=torch.int64, device=cdev))
    m = int(max(c.item() for c in counts))
    pad_idx = torch.full((m,), -1, dtype=torch.int64, device=cdev)
    pad_idx[:runner.pair_index.numel()] = runner.pair_index.to(cdev)
    pad_sc = torch.zeros(m, dtype=torch.float32, device=cdev)
    pad_sc[:scores.numel()] = scores.view(-1).to(cdev)
    all_idx = [torch.empty_like(pad_idx) for _ in range(world)]
    all_sc = [torch.empty_like(pad_sc) for _ in range(world)]
    dist.all_gather(all_idx, pad_idx)
    dist.all_gather(all_sc, pad_sc)
    ok = True
    if rank == 0:
        got = np.full(P, np.nan, np.float32)
        seen = np.zeros(P, np.int64)
        for ix, sc in zip(all_idx, all_sc):
            ix, sc = ix.cpu().numpy(), sc.cpu().numpy()
            keep = ix >= 0
            got[ix[keep]] = sc[keep]
            seen[ix[keep]] += 1
        gnn_w = helpers.gnn_to_oracle(model.gnn)
        if case == 'basic_ui':
            want = om.basic_gnn_scores(g['adj'], gnn_w, helpers.basic_head_to_oracle(model.rs), u_np, i_np, dtype=np.float64)
        else:
            full = np.zeros((n, 768), np.float32)
            full[:bert.shape[0]] = bert
            want = om.hybrid_gnn_scores(g['adj'], gnn_w, helpers.hybrid_head_to_oracle(model.rs), u_np, i_np, full, dtype=np.float64)
        err = float(np.abs(got - want.reshape(-1)).max())
        ok = bool((seen == 1).all()) and err < 1e-4
        print('{} world {} case {}: every pair scored once: {}, max |score - oracle| = {:.2e}, {}'.format(
            'one-GPU rehearsal (gloo)' if rehearse else 'nccl', world, case, bool((seen == 1).all()), err, runner.describe()), flush=True)
    if not graph_ok:
        print('rank {}: the graph-replayed step differs from the eager one (or survived a weight update)'.format(rank), flush=True)
    flag = torch.tensor([1 if (ok and graph_ok) else 0], device=cdev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == '__main__':
    main()
