#!/usr/bin/env python
"""Headline benchmark: (user, item) pairs scored per second, ML-1M-shape basic-gnn (2-layer GCN).

One step = one pass of the hot path over one batch of synthetic input:
full-graph propagation (X_0.W_1 prologue + one fused CSR-SpMM kernel per GCN layer) followed by
scoring every test pair (embedding gather + BasicRS towers + classifier) — the "hoisted" mode of
SURVEY.md §8d: one propagation per weight state, then all pairs.  Inputs (CSR graph, weights,
pair ids) are resident in HBM before the timed region.

Workload: ml1m(s) graph of MovieLens-1M shape scaled by s (default 64: |U| = 386 304,
|I| = 204 288, ~56 M non-zeros in A_hat, ~12 M test pairs), econfigs/basic-gnn.yaml grid1 model
(GCN d=8, n_hiddens [8, 8], concat -> 24, dense [24, 24], clf [48, 48]), fp32.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel = the fused GCN
SpMM layer, algorithmic bytes nnz*8 + (N+1)*4 + 2*N*F*4 per launch over its HIP-event time)
and `cpu_baseline` (the oracle timed on one host core on a bounded ml1m(s=1) sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

GRID1 = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48],
             l2_regularizer=1e-4, final_node='concatenation', activation='relu')
HBM_PEAK_GBPS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--scale', type=int, default=64, help='ml1m(s) scale factor of the synthetic graph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    return ap.parse_args()


def cpu_baseline():
    """Oracle (numpy/scipy port of the reference arithmetic) on ml1m(s=1), one host thread, hoisted mode."""
    from threadpoolctl import threadpool_limits
    from oracle import models as om, weights as ow
    from tests import helpers
    g = helpers.ml1m_indexed(1)
    rng = np.random.default_rng(42)
    n = g['adj_ui'].shape[0]
    gnn = ow.gnn(rng, 'gcn', n, 8, (8, 8))
    head = ow.basic_head(rng, 24, [24, 24], [48, 48])
    u, i = g['test'][:, 0], g['test'][:, 1]
    with threadpool_limits(limits=1):
        om.basic_gnn_scores(g['adj_ui'], gnn, head, u[:1000], i[:1000])          # warm
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0 or reps < 3:
            om.basic_gnn_scores(g['adj_ui'], gnn, head, u, i)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
    # the same with the BLAS pool on every host core (the sparse products of scipy stay single-threaded)
    reps_all, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 5.0 or reps_all < 3:
        om.basic_gnn_scores(g['adj_ui'], gnn, head, u, i)
        reps_all += 1
    dt_all = (time.perf_counter() - t0) / reps_all
    return {'value': len(u) / dt, 'unit': 'pairs/s', 'cores': 1, 'kind': 'port',
            'sample': 'ml1m(s=1): gcn_filter + 2-layer GCN propagation + {} test pairs, hoisted, {} reps of {:.3f} s'
                      .format(len(u), reps, dt),
            'all_cores': {'value': len(u) / dt_all, 'unit': 'pairs/s', 'cores': os.cpu_count(),
                          'sample': '{} reps of {:.3f} s, BLAS threads unrestricted'.format(reps_all, dt_all)}}


def ml1m_true_size(dev):
    """configs[1] at its real size, ml1m(s=1) (launch-latency-bound): hoisted and faithful pairs/s."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    data = synthetic.ml1m_device(1, device=dev)
    n = data['n_users'] + data['n_items']
    a_hat = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    engine.set_seed(42)
    model = basic.BasicGCN(a_hat, **GRID1)
    model.n_users, model.n_items = data['n_users'], data['n_items']
    u = data['test'][:, 0].to(torch.int32).contiguous()
    i = data['test'][:, 1].to(torch.int32).contiguous()
    p = int(u.numel())

    def hoisted():
        emb = model.gnn(None)
        return model.rs.score_towers(model.rs.towers(emb[:model.n_users], emb[model.n_users:]), u, i, 0, model.n_users)

    def faithful():                                          # basic.py:61-63: propagation + scoring per 2048-pair batch
        for lo in range(0, p, 2048):
            model((u[lo:lo + 2048], i[lo:lo + 2048]))

    out = {}
    for name, fn, reps in (('hoisted', hoisted, 20), ('faithful', faithful, 3)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name + '_pairs_per_s'] = p / dt
        out[name + '_ms'] = 1e3 * dt
    # the same hoisted step replayed from a hipGraph: at this size the eight launches are mostly gaps
    hoisted()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        scores = hoisted()
    graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    out['hoisted_graph_pairs_per_s'], out['hoisted_graph_ms'] = p / dt, 1e3 * dt
    assert scores.shape[0] == p
    # full ranking: every (user, item) combination, P_all = |U| x |I| (SURVEY.md 8d "pair sets"), hoisted
    nu, ni = model.n_users, model.n_items
    u_all = torch.arange(nu, device=dev, dtype=torch.int32).repeat_interleave(ni).contiguous()
    i_all = (torch.arange(ni, device=dev, dtype=torch.int32) + nu).repeat(nu).contiguous()

    def full_ranking():
        emb = model.gnn(None)
        return model.rs.score_towers(model.rs.towers(emb[:nu], emb[nu:]), u_all, i_all, 0, nu)
    full_ranking()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        full_ranking()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    out['full_ranking_pairs'], out['full_ranking_pairs_per_s'], out['full_ranking_ms'] = nu * ni, nu * ni / dt, 1e3 * dt
    out.update({'pairs': p, 'nodes': n, 'nnz': a_hat.nnz, 'note': 'latency / launch-bound at this size'})
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    torch.cuda.set_device(local_rank)
    force_dist = bool(os.environ.get('AMAR_FORCE_DIST'))      # rehearse the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.init_process_group('nccl')
    assert world == args.gpus, "--gpus {} but WORLD_SIZE={}".format(args.gpus, world)

    from deep_cbrs_amar_renaissance_amd import capi, engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    from deep_cbrs_amar_renaissance_amd import parallel

    capi.load()
    engine.set_seed(42)
    dev = torch.device('cuda', local_rank)
    data = synthetic.ml1m_device(args.scale, device=dev)
    n_nodes = data['n_users'] + data['n_items']
    a_hat = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n_nodes)
    nnz = a_hat.nnz
    model = basic.BasicGCN(a_hat, **GRID1)
    model.n_users, model.n_items = data['n_users'], data['n_items']
    # test-file order is arbitrary in the reference (datasets.py:199-203): shuffle, so no gather locality is assumed
    perm_gen = torch.Generator(device=dev)
    perm_gen.manual_seed(42)
    perm = torch.randperm(data['test'].shape[0], device=dev, generator=perm_gen)
    u_all = data['test'][perm, 0].to(torch.int32).contiguous()
    i_all = data['test'][perm, 1].to(torch.int32).contiguous()
    n_pairs = int(u_all.numel())
    del perm
    del data
    torch.cuda.empty_cache()

    runner = parallel.make_runner(model, u_all, i_all, rank, world) if not force_dist else \
        parallel.PartitionedGCNRunner(model, u_all, i_all, rank, world)

    spmm_events = []
    raw_gcn_layer, raw_spmm_sj, raw_spmm_xs = capi.gcn_layer, capi.spmm_sj, capi.spmm_xs

    def timed(fn):
        def wrapper(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*a, **k)
            e1.record()
            spmm_events.append((e0, e1))
        return wrapper
    capi.gcn_layer, capi.spmm_sj, capi.spmm_xs = timed(raw_gcn_layer), timed(raw_spmm_sj), timed(raw_spmm_xs)   # whichever form the layer dispatches to
    pair_events, raw_chain = [], capi.chain

    def timed_chain(*a, **k):                                  # the pair stage = the sum-input chain launch
        if not k.get('sum_inputs'):
            return raw_chain(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        raw_chain(*a, **k)
        e1.record()
        pair_events.append((e0, e1))
    capi.chain = timed_chain

    def barrier():
        if world > 1 or force_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.step()
    barrier()
    spmm_events.clear()
    pair_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step()
    host_dt = time.perf_counter() - t0                        # time to ENQUEUE the steps (host-side launch cost)
    barrier()
    dt = time.perf_counter() - t0
    capi.gcn_layer, capi.spmm_sj, capi.spmm_xs = raw_gcn_layer, raw_spmm_sj, raw_spmm_xs
    capi.chain = raw_chain
    if world > 1 or force_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    spmm_ms = [e0.elapsed_time(e1) for e0, e1 in spmm_events]
    rows_local = runner.local_rows
    nnz_local = runner.local_nnz
    f = GRID1['n_hiddens'][0]
    from deep_cbrs_amar_renaissance_amd.utilities.math import spmm_kind
    if world == 1 and not force_dist:
        kind = spmm_kind(model.gnn.gnn_layers.adj_matrix, f)
    else:
        kind = 'xs' if runner._use_xs(f) else 'csr'
    alg_bytes = nnz_local * 8 + (rows_local + 1) * 4 + (n_nodes + rows_local) * f * 4
    avg_ms = float(np.mean(spmm_ms)) if spmm_ms else float('nan')
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9

    traffic = None
    pmc_path = os.path.join(ROOT, 'profiles', 'spmm_pmc_latest.json')
    if world == 1 and os.path.exists(pmc_path):
        # HBM/fabric bytes per launch from a separate rocprofv3 --pmc run of this same command (tools/profile_bench.sh),
        # corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950) — only valid for the profiled scale
        pmc = json.load(open(pmc_path))
        if pmc.get('scale') == args.scale and kind == 'xs' and not force_dist:
            traffic = pmc['traffic_bytes_per_launch']

    if rank == 0:
        out = {
            'metric': '(user,item) pairs scored/sec, ML-1M basic-gnn 2-layer, 1/2/4/8 MI355X',
            'value': n_pairs * args.steps / dt, 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'ml1m(s={}) user-item graph: N={} nodes, nnz(A_hat)={}, {} test pairs; '
                                   'econfigs/basic-gnn.yaml grid1 BasicGCN d=8 L=2 concat, dense [24,24], clf [48,48]; '
                                   'one propagation + per-entity towers + all pairs (shuffled order) per step (hoisted)'.format(args.scale, n_nodes, nnz, n_pairs),
                       'scale': args.scale, 'parallelism': runner.describe()},
            'roofline': {'bound': 'hbm', 'kernel': {'sj': 'spmm_sj_kernel<8>', 'xs': 'spmm_xs_partial_kernel<8> + spmm_xs_combine_kernel<8>', 'csr': 'spmm_stream_kernel<8>'}[kind] + ' (fused GCN layer: SpMM + bias + ReLU + next X.W)', 'achieved': achieved,
                         'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_ms': avg_ms,
                         'launches_timed': len(spmm_ms)},
            'propagation_ms': runner.last_propagation_ms(),
            'host_enqueue_ms_per_step': 1e3 * host_dt / args.steps,
        }
        pair_ms = [e0.elapsed_time(e1) for e0, e1 in pair_events]
        if pair_ms:
            # the other large kernel of a step: chain_kernel (sum-input form) gathers two 48-float per-entity rows per pair
            c1 = GRID1['clf_units'][0]
            pairs_local = int(runner.u_ids.numel())
            pair_alg = pairs_local * (2 * c1 * 4 + 2 * 4 + 4)
            pms = float(np.mean(pair_ms))
            pair_traffic = None
            if world == 1 and os.path.exists(pmc_path) and json.load(open(pmc_path)).get('scale') == args.scale and \
                    'pair_stage_traffic_bytes_per_launch' in json.load(open(pmc_path)):
                pmc = json.load(open(pmc_path))
                pair_traffic = pmc['pair_stage_traffic_bytes_per_launch']
            out['pair_stage'] = {'kernel': 'chain_pipe_kernel<3,2> (relu(T_u[u] + T_i[i]) -> Dense 48 -> Dense 1, sigmoid)',
                                 'avg_launch_ms': pms, 'pairs_per_launch': pairs_local, 'bound': 'hbm',
                                 'algorithmic_bytes_per_launch': pair_alg, 'achieved': pair_alg / (pms * 1e-3) / 1e9,
                                 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': pair_alg / (pms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                 'traffic': pair_traffic,
                                 # the same launch against the fp32 MFMA peak (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense)
                                 'mfma': {'flop_per_pair': 2 * (c1 * GRID1['clf_units'][1] + GRID1['clf_units'][1]),
                                          'achieved_tflops': pairs_local * 2 * (c1 * GRID1['clf_units'][1] + GRID1['clf_units'][1]) / (pms * 1e-3) / 1e12,
                                          'peak_tflops': 157.3,
                                          'frac': pairs_local * 2 * (c1 * GRID1['clf_units'][1] + GRID1['clf_units'][1]) / (pms * 1e-3) / 1e12 / 157.3,
                                          'pmc_busy_frac': pmc.get('pair_stage_mfma_busy_frac') if pair_traffic is not None else None}}
        if world == 1 and not args.no_cpu_baseline:
            out['ml1m_s1'] = ml1m_true_size(dev)
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
