#!/usr/bin/env python
"""Headline benchmark: (user, item) pairs scored per second, ML-1M-shape basic-gnn (2-layer GCN).

One step = one pass of the hot path over one batch of synthetic input:
full-graph propagation (X_0.W_1 prologue + one fused SpMM kernel per GCN layer) followed by
scoring every test pair (embedding gather + BasicRS towers + classifier) — the "hoisted" mode of
SURVEY.md §8d: one propagation per weight state, then all pairs.  Inputs (graph image, weights,
pair ids) are resident in HBM before the timed region.

Workload: ml1m(s) graph of MovieLens-1M shape scaled by s (default 64: |U| = 386 304,
|I| = 204 288, ~56 M non-zeros in A_hat, ~12 M test pairs), econfigs/basic-gnn.yaml grid1 model
(GCN d=8, n_hiddens [8, 8], concat -> 24, dense [24, 24], clf [48, 48]), fp32.

`python bench.py --gpus N` with no WORLD_SIZE in the environment starts its own N ranks
(`python -m torch.distributed.run`, fresh child processes, before this process touches the GPU);
under torchrun (WORLD_SIZE set) it is one rank.  Rank 0 prints ONE JSON line.  Extra objects:
`roofline` (dominant kernel = the fused GCN SpMM layer, algorithmic bytes nnz*8 + (N+1)*4 + 2*N*F*4
per launch over its HIP-event time), `roofline_onchip` (the same launch against the DISPATCHED kernel's own on-chip
floors: its L2 request count from this build's counters and its measured no-gather time), `s256` (the same step on ml1m(s=256):
the larger scaling workload, single-GPU and multi-rank), `wider_layers` (the fused layer of econfigs/basic-gnn.yaml grid2 / grid3:
F = 16 / 32 at the same scale), `pair_stage`, `hybrid_head` (econfigs/hybrid-gnn.yaml grid1 head at the same scale: MFMA utilisation
as EXECUTED instructions over the dense peak of that instruction), `uip_graph` (econfigs/basic-gnn-uip-2relconf.yaml grid1: the same
model on the user-item-property graph), `model_families` (BasicLightGCN / BasicGraphSage / BasicGAT steps at the same scale),
`train_s1` (one `fit()` epoch at the reference's real size), `ml1m_s1` (the reference's real size) and `cpu_baseline` — the oracle on
the host: `value` / `cores` / `sample` are ONE hoisted repetition ON THE HEADLINE WORKLOAD (threads stated, score difference against
the GPU leg), the ml1m(s=1) legs (one core, faithful, the BLAS pool sweep) nest under `ml1m_s1`.  With N > 1 ranks: `per_rank` (every
rank's local_spmm_ms / exposed_exchange_ms / replicated_ms of an eager step and its score difference against the single-GPU step)
and `multi_rank_parity`.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

GRID1 = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48],
             l2_regularizer=1e-4, final_node='concatenation', activation='relu')
GRID2 = dict(GRID1, embedding_dim=16, n_hiddens=[16, 16], dense_units=[48, 48], clf_units=[64, 64])      # basic-gnn.yaml:13-22
GRID3 = dict(GRID1, embedding_dim=32, n_hiddens=[32, 32], dense_units=[96, 48], clf_units=[64, 64])      # basic-gnn.yaml:24-33
HYBRID_GRID1 = dict(embedding_dim=8, n_hiddens=[8, 8], dense_units=[[24, 24], [256, 64], [64, 64]], clf_units=[64, 64],
                    feature_based=True)
HBM_PEAK_GBPS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0         # dense bf16 (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3
L2_REQUESTS_PER_S = 270e9          # 256 CUs x 0.44 line requests per clock at 2.4 GHz (profiles/r1_exp_gather_frontend.txt)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--scale', type=int, default=64, help='ml1m(s) scale factor of the synthetic graph (64; 256 = the larger scaling workload)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the ml1m(s=1), hybrid-head and CPU legs')
    ap.add_argument('--no-s256', action='store_true', help='skip the ml1m(s=256) leg (the larger scaling workload, run by default next to --scale 64)')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks as fresh children.  This process has not touched the GPU
    (importing torch does not initialise HIP) and never will: it only waits for the launcher and passes its exit code on."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.run(cmd, env=env).returncode


def supervise(rank):
    """More than one rank: the rank's work runs in a CHILD of this process, which itself never touches the GPU.  The multi-rank step
    replayed from a hipGraph — RCCL collectives captured with the kernels — has only ever run with one rank or rehearsed on one GPU in
    the builder's environment; a capture or replay that aborts or stalls on a real node (the process group's watchdog ends a stalled
    collective after 300 s) would otherwise leave the scaling run without a line.  If the child fails and the graph was on, the
    rank runs once more with eager steps (AMAR_STEP_GRAPH=0), and the line says so (`config.retry`).  Peers of a failed rank fail
    too — their collectives break or time out — and meet it again in the second rendezvous, which waits up to 15 minutes."""
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, AMAR_BENCH_CHILD='1')
    first = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    if first.returncode == 0 or os.environ.get('AMAR_STEP_GRAPH', '1') == '0':
        sys.stdout.buffer.write(first.stdout)
        sys.stdout.flush()
        return first.returncode
    sys.stderr.write("bench.py: rank {}: the run with the hipGraph-replayed step ended with code {}; once more with eager steps\n".format(rank, first.returncode))
    sys.stderr.flush()
    env.update(AMAR_STEP_GRAPH='0', AMAR_BENCH_RETRY='the run with the hipGraph-replayed step ended with code {} on rank {}'.format(first.returncode, rank))
    second = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    sys.stdout.buffer.write(second.stdout)
    sys.stdout.flush()
    return second.returncode


def csrc_sha():
    """Identity of the kernel sources a PMC profile belongs to: sha256 over csrc/*.hip, *.h (sorted), first 16 hex digits."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'deep_cbrs_amar_renaissance_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h')):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), 'rb').read())
    return h.hexdigest()[:16]


def pmc_profile(scale, kind):
    """HBM/fabric bytes per launch from a separate rocprofv3 --pmc run of this same command (tools/profile_bench.sh ->
    profiles/spmm_pmc_latest.json), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950).  Only valid for
    the profiled scale, kernel form AND kernel sources: anything else returns None for the numbers and says why."""
    path = os.path.join(ROOT, 'profiles', 'spmm_pmc_latest.json')
    if not os.path.exists(path):
        return None, 'no profile (profiles/spmm_pmc_latest.json missing)'
    pmc = json.load(open(path))
    src = 'profiles/spmm_pmc_latest.json @ csrc {}'.format(pmc.get('csrc_sha', 'unrecorded'))
    if pmc.get('scale') != scale or pmc.get('kind', 'xs') != kind:
        return None, src + ' (profiled scale/kernel form differs from this run: not applied)'
    if pmc.get('csrc_sha') != csrc_sha():
        return None, src + ' (csrc/ changed since: stale, not applied)'
    return pmc, src


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(s1, headline=None):
    """Oracle (numpy/scipy port of the reference arithmetic) on the SAME ml1m(s=1) graph, weights and pair list as the
    `ml1m_s1` GPU leg: hoisted (one propagation, then all pairs) and faithful (propagation re-run per 2 048-pair batch as
    basic.py:61-63 does, batch size config.yaml:47) on ONE thread, hoisted again on all usable cores (the reference runs
    with n_workers: 12, config.yaml:2; what scales here is the BLAS pool of the Dense layers — scipy's CSR product is
    single-threaded whatever the pool).  gcn_filter is the model-construction preprocess (gnn.py:283) and is outside every
    timed region, as it is outside the GPU's.  `headline`: the inputs of the headline workload itself — one hoisted repetition
    on all usable cores, scores compared with the GPU leg's."""
    from threadpoolctl import threadpool_limits
    from oracle import graph as ograph, layers as olayers, models as om
    adj, gnn, head, u, i = s1['adj'], s1['gnn'], s1['head'], s1['u'], s1['i']
    a_hat = ograph.gcn_filter(adj)

    def propagate(a=a_hat, g=gnn):
        x = g['embeddings'].astype(np.float32)
        hs = [x]
        for lw in g['layers']:
            x = olayers.gcn_conv(x, a, lw['kernel'], lw['bias'])
            hs.append(x)
        return olayers.reduce_layers(hs, 'concatenation')

    def hoisted():
        e = propagate()
        return om.basic_rs(e[u], e[i], head)

    def faithful():
        outs = []
        for lo in range(0, len(u), 2048):
            e = propagate()
            outs.append(om.basic_rs(e[u[lo:lo + 2048]], e[i[lo:lo + 2048]], head))
        return np.concatenate(outs)

    def timed(fn, budget, min_reps):
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget or reps < min_reps:
            fn()
            reps += 1
        return (time.perf_counter() - t0) / reps, reps

    with threadpool_limits(limits=1):
        scores = hoisted()                                          # warm; also the parity check of the GPU leg
        dt, reps = timed(hoisted, 6.0, 3)
        fdt, freps = timed(faithful, 6.0, 1)
    cores = usable_cores()
    # the BLAS pool that serves this workload best: the Dense layers are [pairs, 24..48] x [24..48, 24..48] products, far too small
    # for one thread per core of a 256-thread host (64 threads: 3-4x SLOWER than one) — a short sweep picks the pool, all are reported
    sweep = {}
    for pool in sorted({p for p in (2, 4, 8, 16, 32, 64) if p <= cores}):
        with threadpool_limits(limits=pool):
            hoisted()
            sweep[pool] = timed(hoisted, 1.0, 2)[0]
    threads = min(sweep, key=sweep.get) if sweep else 1
    with threadpool_limits(limits=threads):
        hoisted()
        adt, areps = timed(hoisted, 3.0, 3)
    err = float(np.abs(scores.reshape(-1) - s1['gpu_scores'].reshape(-1)).max())
    s1_legs = {'workload': 'ml1m(s=1): the graph / weights / {} test pairs of the ml1m_s1 GPU leg'.format(len(u)),
               'one_core': {'value': len(u) / dt, 'unit': 'pairs/s', 'cores': 1,
                            'sample': '2-layer GCN propagation + all pairs, hoisted, {} reps of {:.3f} s on one thread'.format(reps, dt)},
               'faithful': {'value': len(u) / fdt, 'unit': 'pairs/s', 'cores': 1, 'batch': 2048,
                            'sample': 'propagation re-run per 2048-pair batch (basic.py:61-63), {} passes of {:.2f} s'.format(freps, fdt)},
               'best_pool': {'value': len(u) / adt, 'unit': 'pairs/s', 'cores': threads, 'usable_cores': cores, 'host_cpus': os.cpu_count(),
                             'pool_sweep_pairs_per_s': {str(p): len(u) / t for p, t in sweep.items()},
                             'sample': 'the same hoisted pass on the BLAS pool that serves it best — a sweep, NOT all cores: {} threads of {} usable '
                                       '({} reps of {:.3f} s); scipy\'s CSR product stays single-threaded'.format(threads, cores, areps, adt)},
               'max_abs_score_diff_vs_gpu': err}
    if headline is None:
        out = dict(s1_legs['one_core'], kind='port', workload=s1_legs['workload'], max_abs_score_diff_vs_gpu=err, ml1m_s1=s1_legs)
        return out
    from scipy import sparse
    hg, hh, hu, hi_ = headline['gnn'], headline['head'], headline['u'], headline['i']
    rowptr, colidx, vals = headline['a_hat']
    ah = sparse.csr_matrix((vals, colidx, rowptr), shape=(len(rowptr) - 1, len(rowptr) - 1))
    e1 = propagate(ah, hg)                                          # (untimed: warms the pages; also the sweep's input)
    hsweep = {}
    for pool in sorted({p for p in (8, 32, 64, 128) if p <= cores}):          # a 1 Mi-pair batch has work for a larger pool than ml1m(s=1)
        with threadpool_limits(limits=pool):
            t0 = time.perf_counter()
            om.basic_rs(e1[hu[:1 << 20]], e1[hi_[:1 << 20]], hh)
            hsweep[pool] = time.perf_counter() - t0
    threads = min(hsweep, key=hsweep.get) if hsweep else 1
    del e1
    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        e = propagate(ah, hg)
        t_prop = time.perf_counter() - t0
        worst = 0.0
        for lo in range(0, len(hu), 1 << 20):                       # pair batches bound the host memory, not the arithmetic
            sc = om.basic_rs(e[hu[lo:lo + (1 << 20)]], e[hi_[lo:lo + (1 << 20)]], hh)
            worst = max(worst, float(np.abs(sc.reshape(-1) - headline['gpu_scores'][lo:lo + (1 << 20)].reshape(-1)).max()))
        hdt = time.perf_counter() - t0
    # the headline figure is LIKE FOR LIKE: the workload `value` is quoted on; the ml1m(s=1) legs (one core, faithful, pool sweep) nest below
    out = {'value': len(hu) / hdt, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port', 'workload': headline['workload'],
           'sample': 'ONE hoisted repetition on the headline workload itself ({}): propagation {:.2f} s (scipy CSR product, one thread), '
                     'all {} pairs in 1 Mi-pair batches on a BLAS pool of {} threads (the fastest of a sweep over {}), {:.2f} s in all; A_hat is the '
                     'device-built matrix copied to the host (bit-identical to the oracle\'s gcn_filter: '
                     'tests/test_models_gpu.py::test_device_gcn_filter_matches_host)'.format(
                         headline['workload'], t_prop, len(hu), threads, sorted(hsweep), hdt),
           'max_abs_score_diff_vs_gpu': worst, 'ml1m_s1': s1_legs}
    return out


def ml1m_true_size(dev):
    """configs[1] at its real size, ml1m(s=1) (launch-latency-bound): hoisted and faithful pairs/s.
    Returns the JSON object and what the CPU baseline needs to time the same inputs."""
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
    for name, fn, reps in (('hoisted_eager', hoisted, 20), ('faithful_eager', faithful, 3)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name + '_pairs_per_s'] = p / dt
        out[name + '_ms'] = 1e3 * dt

    # the default route: Model.predict() on the test Sequence (batches of 2 048 id pairs, config.yaml:47) replays the whole
    # pass from a hipGraph it captured itself — nothing for the caller to manage
    u_np, i_np = u.cpu().numpy().astype(np.int64), i.cpu().numpy().astype(np.int64)

    class TestSequence:
        order_version = 0                                       # fixed order: predict() reads the batches once

        def __len__(self):
            return (p + 2047) // 2048

        def __getitem__(self, b):
            return (u_np[b * 2048:(b + 1) * 2048], i_np[b * 2048:(b + 1) * 2048]), np.zeros(min(2048, p - b * 2048))

    seq = TestSequence()
    for name, hoist, reps in (('hoisted', True, 50), ('faithful', False, 5)):
        scores = model._predict_graphed(seq, hoist)                 # device part of predict(): capture on first use
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            model._predict_graphed(seq, hoist)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name + '_pairs_per_s'], out[name + '_ms'] = p / dt, 1e3 * dt
        assert scores.shape[0] == p
    model.predict(seq)                                              # (mode switch: captures the hoisted pass again)
    t0 = time.perf_counter()
    host_scores = model.predict(seq)                                # the replayed pass plus the copy of the scores to a host ndarray
    out['predict_call_ms'] = 1e3 * (time.perf_counter() - t0)
    assert host_scores.shape == (p, 1)
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
    # the same inputs for the CPU leg: symmetric unit adjacency (preprocess.py:68-86 + math.py:13-20), weights, pairs
    from scipy import sparse
    from tests import helpers
    tp = data['train_pos'].cpu().numpy()
    adj = sparse.coo_matrix((np.ones(2 * len(tp), np.float32), (np.concatenate([tp[:, 0], tp[:, 1]]), np.concatenate([tp[:, 1], tp[:, 0]]))),
                            shape=(n, n))
    s1 = {'adj': adj, 'gnn': helpers.gnn_to_oracle(model.gnn), 'head': helpers.basic_head_to_oracle(model.rs),
          'u': u.cpu().numpy().astype(np.int64), 'i': i.cpu().numpy().astype(np.int64), 'gpu_scores': hoisted().cpu().numpy()}
    return out, s1


def hybrid_head(dev, scale):
    """econfigs/hybrid-gnn.yaml grid1 (HybridBertGCN, 768-d BERT rows, feature_based) at the same ml1m(s): the MLP head on
    MFMA — per-entity first-stage networks (the 768->256->64 towers on amar_dense_f32 / dense_mfma_kernel) and the pair stage
    (amar_dual_chain_f32) — timed with HIP events, as TFLOP/s against the fp32 MFMA peak."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import hybrid
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    data = synthetic.ml1m_device(scale, device=dev)
    nu, ni = data['n_users'], data['n_items']
    n = nu + ni
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    engine.set_seed(42)
    model = hybrid.HybridBertGCN(a, **HYBRID_GRID1)
    model.n_users, model.n_items = nu, ni
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    bert = torch.randn((n, 768), device=dev, generator=gen) * 0.5
    model.set_bert_table(bert)
    model.rs.build_head(model.gnn.output_dim(), 768)
    perm = torch.randperm(data['test'].shape[0], device=dev, generator=gen)
    u = data['test'][perm, 0].to(torch.int32).contiguous()
    i = data['test'][perm, 1].to(torch.int32).contiguous()
    p = int(u.numel())
    del data, perm
    emb = model.gnn(None)
    rs = model.rs

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    ms_bert_u = timed(lambda: rs.dense2a.apply2(bert[:nu]))
    flop_bert_u = nu * (768 * 256 + 256 * 64) * 2.0
    tw = rs.towers(emb[:nu], emb[nu:], bert[:nu], bert[nu:])
    ms_towers = timed(lambda: rs.towers(emb[:nu], emb[nu:], bert[:nu], bert[nu:]))
    from deep_cbrs_amar_renaissance_amd.models.basic import PairPlan
    plan = PairPlan(u, i)                                              # the pair list does not change between steps: prepared once
    ms_pairs = timed(lambda: rs.score_towers(tw, u, i, 0, nu, pair_plan=plan))
    flop_pair = 2 * (2 * 64 * 64 + 128 * 64 + 64 * 64 + 64)            # dense3a/3b second layers + clf 128-64-64-1 per pair
    f32_form = os.environ.get('AMAR_PAIR_MFMA') == 'f32'
    f32_dense = os.environ.get('AMAR_DENSE_SPLIT') == '0'
    split_note = ('products on v_mfma_f32_*_bf16 with both operands split three ways (x = hi + mid + lo exactly, six part products in '
                  'f32: as accurate as the f32 instruction, DESIGN 4b): mfma_frac = the time the matrix pipe needs for the EXECUTED '
                  'instructions at their own dense peaks (bf16 2 500 TFLOP/s, f32 157.3) over the elapsed time — a utilisation, never above 1; '
                  'f32_equiv_tflops is the f32 work the launch stands for (secondary)')
    # executed instructions: a split product is six bf16 part products; the 256 -> 64 layer of the tower stays on the f32 instruction
    l1, l2 = nu * 768 * 256 * 2.0, nu * 256 * 64 * 2.0
    pipe_ms_bert = 1e3 * ((l1 / (MFMA_F32_PEAK_TFLOPS * 1e12) if f32_dense else 6 * l1 / (MFMA_BF16_PEAK_TFLOPS * 1e12)) + l2 / (MFMA_F32_PEAK_TFLOPS * 1e12))
    pipe_ms_pairs = 1e3 * p * (flop_pair / (MFMA_F32_PEAK_TFLOPS * 1e12) if f32_form else 6 * flop_pair / (MFMA_BF16_PEAK_TFLOPS * 1e12))
    del plan
    del model, bert, tw, emb
    torch.cuda.empty_cache()
    return {'config': 'econfigs/hybrid-gnn.yaml grid1: HybridBertGCN d=8 L=2, dense [[24,24],[256,64],[64,64]], clf [64,64], 768-d BERT, ml1m(s={})'.format(scale),
            'bert_tower': {'kernel': ('dense_mfma128_kernel' if f32_dense else 'dense_split128_kernel (amar_dense_split_f32)') +
                                     ' / dense_mfma_kernel (amar_dense_f32): 768->256->64 over the {} user rows'.format(nu),
                           'ms': ms_bert_u, 'mfma_frac': pipe_ms_bert / ms_bert_u,
                           'executed': {'first_layer': 'v_mfma_f32_32x32x2_f32' if f32_dense else 'v_mfma_f32_32x32x16_bf16 x 6 part products',
                                        'first_layer_tflops': (1 if f32_dense else 6) * l1 / ms_bert_u / 1e9,
                                        'first_layer_peak_tflops': MFMA_F32_PEAK_TFLOPS if f32_dense else MFMA_BF16_PEAK_TFLOPS,
                                        'second_layer': 'v_mfma_f32_32x32x2_f32', 'note': 'tflops over the time of BOTH layers'},
                           'f32_equiv_tflops': flop_bert_u / ms_bert_u / 1e9, 'f32_peak_tflops': MFMA_F32_PEAK_TFLOPS,
                           'note': None if f32_dense else 'first layer: ' + split_note},
            'entity_towers_ms': ms_towers,
            'pair_stage': {'kernel': ('dual_chain_full_kernel' if f32_form else 'dual_chain_split_kernel') +
                                     ' (amar_dual_chain_indexed_f32 on the prepared pair list) + scatter_windows_kernel', 'pairs': p, 'ms': ms_pairs,
                           'pairs_per_s': p / ms_pairs * 1e3, 'flop_per_pair': flop_pair,
                           'mfma_frac': pipe_ms_pairs / ms_pairs,
                           'executed_tflops': p * (1 if f32_form else 6) * flop_pair / ms_pairs / 1e9,
                           'executed_peak_tflops': MFMA_F32_PEAK_TFLOPS if f32_form else MFMA_BF16_PEAK_TFLOPS,
                           'executed_instruction': 'v_mfma_f32_16x16x4_f32' if f32_form else 'v_mfma_f32_16x16x32_bf16 x 6 part products',
                           'f32_equiv_tflops': p * flop_pair / ms_pairs / 1e9, 'f32_peak_tflops': MFMA_F32_PEAK_TFLOPS,
                           'pmc_busy_frac_note': 'matrix pipe busy 0.45-0.50 by SQ_VALU_MFMA_BUSY_CYCLES (profiles/r3_exp_pair_split.txt)',
                           'note': None if f32_form else split_note}}


def train_true_size():
    """SURVEY 8(f) N1 at the reference's real size: `model.fit` on ml1m(s=1) for the config the reference publishes a training time
    for (doc.pdf p.22 Table 5: BasicGCN 16 channels x 2 layers, dense [48, 48], clf [64, 64], batch 1024, Adam 1e-3 — 211 s per 25
    epochs on an RTX 3060, other hardware, for orientation): full-graph propagation, BCE + L2, reverse pass and Adam per batch, each
    batch replayed from a hipGraph (training.py).  One warm-up epoch, one timed epoch."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraph
    from deep_cbrs_amar_renaissance_amd.experiment import Adam
    from deep_cbrs_amar_renaissance_amd.models import basic
    from tests import helpers
    g = helpers.ml1m_indexed(1)
    engine.set_seed(42)
    model = basic.BasicGCN(g['adj_ui'], embedding_dim=16, n_hiddens=[16, 16], dense_units=[48, 48], clf_units=[64, 64], l2_regularizer=1e-4)
    model.compile(loss='binary_crossentropy', optimizer=Adam(learning_rate=1e-3), metrics=['accuracy'])
    train = UserItemGraph(g['train'], g['users'], g['items'], g['adj_ui'], batch_size=1024, shuffle=True)
    first = model.fit(train, epochs=1, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hist = model.fit(train, epochs=1, verbose=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'config': 'BasicGCN d=16 [16,16], dense [48,48], clf [64,64], batch 1024, Adam(1e-3), l2 1e-4 on ml1m(s=1): {} train pairs, {} batches per epoch'.format(len(g['train']), len(train)),
            's_per_epoch': dt, 'ms_per_batch': 1e3 * dt / len(train), 'train_pairs_per_s': len(g['train']) / dt,
            's_per_25_epochs': 25 * dt, 'loss_epoch1': float(first['loss'][-1]), 'loss_epoch2': float(hist['loss'][-1]),
            'reference_published': '211 s per 25 epochs on an RTX 3060 (doc.pdf p.22 Table 5; other hardware)'}


def leaf_spmm_timers(capi, events, kinds_seen=None):
    """HIP-event wrappers around the LEAF propagation entry points (capi.spmm_xs forwards an LDS-tiled image to capi.spmm_lt: only
    the leaf records, so a launch is timed once).  Returns the function that restores the originals."""
    names = ('gcn_layer', 'spmm_sj', 'spmm_xs', 'spmm_lt')
    raw = {name: getattr(capi, name) for name in names}

    def timed(name, fn):
        def wrapper(*a, **k):
            if name == 'spmm_xs' and hasattr(a[0], 'words'):      # forwarded to spmm_lt, which records
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*a, **k)
            e1.record()
            events.append((e0, e1))
            if kinds_seen is not None:
                kinds_seen.append(name)
        return wrapper
    for name in names:
        setattr(capi, name, timed(name, raw[name]))

    def restore():
        for name in names:
            setattr(capi, name, raw[name])
    return restore


def timed_model(model, u, i, steps, n, nnz, f):
    """Hoisted step of a Basic* model replayed from a hipGraph + the HIP-event time of its fused propagation layers over a few
    eager steps, against the 8(d) bytes of an [n, n] graph with nnz non-zeros at width f."""
    from deep_cbrs_amar_renaissance_amd import capi, parallel
    runner = parallel.SingleRunner(model, u, i)
    events = []
    for _ in range(10):
        runner.step()
    restore = leaf_spmm_timers(capi, events)
    for _ in range(5):
        runner.step()
    torch.cuda.synchronize()
    restore()
    layer_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in events])) if events else float('nan')
    prop_ms = runner.last_propagation_ms()
    for _ in range(40):                                           # untimed replays until the clocks have settled, as in main()
        runner.step_graphed()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step_graphed()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    alg = nnz * 8 + (n + 1) * 4 + 2 * n * f * 4
    p = int(u.numel())
    return {'ms_per_step': 1e3 * dt, 'pairs_per_s': p / dt, 'propagation_ms': prop_ms,
            'gcn_layer': {'avg_launch_ms': layer_ms, 'launches_timed': len(events), 'algorithmic_bytes_per_launch': alg,
                          'achieved_gbps': alg / (layer_ms * 1e-3) / 1e9, 'frac_of_hbm_peak': alg / (layer_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}}


def shuffled_test_pairs(data, dev, seed=42):
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    perm = torch.randperm(data['test'].shape[0], device=dev, generator=gen)
    return data['test'][perm, 0].to(torch.int32).contiguous(), data['test'][perm, 1].to(torch.int32).contiguous()


def uip_graph(dev, scale, steps):
    """configs[2] / the graph of configs[4]: econfigs/basic-gnn-uip-2relconf.yaml grid1 — the same BasicGCN over the
    user-item-PROPERTY graph (three node types, duplicate item-property links kept, preprocess.py:149-168) at the same
    ml1m(s): hoisted step and the fused GCN layer's HIP-event time against the 8(d) bytes of that graph."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    data = synthetic.ml1m_device(scale, device=dev, with_props=True)
    nu, ni, npr = data['n_users'], data['n_items'], data['n_props']
    n = nu + ni + npr
    rows = torch.cat([data['train_pos'][:, 0], data['item_prop'][:, 0]])
    cols = torch.cat([data['train_pos'][:, 1], data['item_prop'][:, 1]])
    a = gcn_filter_device(rows, cols, n)
    engine.set_seed(42)
    model = basic.BasicGCN(a, **GRID1)
    model.n_users, model.n_items = nu, ni
    u, i = shuffled_test_pairs(data, dev)
    del data, rows, cols
    out = {'config': 'econfigs/basic-gnn-uip-2relconf.yaml grid1: BasicGCN d=8 L=2 on the user-item-property graph, ml1m(s={}): '
                     'N={} nodes ({} users, {} items, {} properties), nnz(A_hat)={}, {} test pairs'.format(scale, n, nu, ni, npr, a.nnz, int(u.numel()))}
    out.update(timed_model(model, u, i, steps, n, a.nnz, GRID1['n_hiddens'][0]))
    del model, a
    torch.cuda.empty_cache()
    return out


def wider_layers(dev, scale, steps):
    """econfigs/basic-gnn.yaml grid2 (d = 16) and grid3 (d = 32) — /root/reference/econfigs/basic-gnn.yaml:13-33 — at the same
    ml1m(s): the same BasicGCN, two fused layers of width 16 / 32, against the 8(d) bytes at that width."""
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    u, i = shuffled_test_pairs(data, dev)
    nu, ni = data['n_users'], data['n_items']
    del data
    out = {}
    for name, cfg in (('grid2_F16', GRID2), ('grid3_F32', GRID3)):
        engine.set_seed(42)
        model = basic.BasicGCN(a, **cfg)
        model.n_users, model.n_items = nu, ni
        f = cfg['n_hiddens'][0]
        res = {'config': 'econfigs/basic-gnn.yaml {}: BasicGCN d={} L=2 concat, dense {}, clf {}, ml1m(s={})'.format(
            name.split('_')[0], f, cfg['dense_units'], cfg['clf_units'], scale)}
        res.update(timed_model(model, u, i, steps, n, a.nnz, f))
        out[name] = res
        del model
        a.__dict__.pop('_lt_cache', None)                          # one width's image at a time
        torch.cuda.empty_cache()
    return out


def model_families(dev, scale, steps):
    """The other layer kinds of econfigs/basic-gnn.yaml (model.name: basic.BasicLightGCN / BasicGraphSage / BasicGAT, grid1 dims) at
    the same ml1m(s): the hoisted step replayed from a hipGraph, and the propagation alone (HIP events over eager steps).  GraphSAGE
    and GAT take the raw symmetric edge list (gnn.py:316-319, 349-352), LightGCN the gcn-filtered matrix."""
    from deep_cbrs_amar_renaissance_amd import engine, parallel
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device, DeviceCSR
    data = synthetic.ml1m_device(scale, device=dev)
    nu, ni = data['n_users'], data['n_items']
    n = nu + ni
    a_hat = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    rp = a_hat.rowptr.long()
    rows = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
    keep = rows != a_hat.colidx.long()
    rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(rows[keep], minlength=n), 0)
    edges = DeviceCSR(rowptr.to(torch.int32), a_hat.colidx[keep].contiguous(), None, (n, n))
    edges.row_breaks = (nu,)
    u, i = shuffled_test_pairs(data, dev)
    p = int(u.numel())
    del data, rows, keep, rp
    out = {'config': 'econfigs/basic-gnn.yaml grid1 dims (d=8, L=2, dense [24,24], clf [48,48]) at ml1m(s={}), {} test pairs'.format(scale, p)}
    cfg = dict(GRID1, aggregate='mean', dropout_rate=0.0)
    for name, adj in (('BasicLightGCN', a_hat), ('BasicGraphSage', edges), ('BasicGAT', edges)):
        engine.set_seed(42)
        model = getattr(basic, name)(adj, **cfg)
        model.n_users, model.n_items = nu, ni
        runner = parallel.SingleRunner(model, u, i)
        for _ in range(5):
            runner.step()
        prop_ms = runner.last_propagation_ms()
        for _ in range(40):
            runner.step_graphed()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            runner.step_graphed()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[name] = {'ms_per_step': 1e3 * dt, 'pairs_per_s': p / dt, 'propagation_ms': prop_ms}
        del runner, model
        for key in ('_lt_cache', '_lt_mean_cache', '_lt_gat_cache'):
            adj.__dict__.pop(key, None)
        torch.cuda.empty_cache()
    return out


def larger_scale_leg(scale, steps, rank, world, local_rank, multi, rehearse):
    """The same step on ml1m(scale) (default 256: 2.36 M nodes, 224 M non-zeros, 48 M pairs — the node tables leave the 32 MB of L2),
    timed by the same protocol as the headline (graph replay, barrier + synchronize on both sides, MAX over ranks): at N > 1 the
    exchange is 4x the bytes on 4x the compute, which is where >= 5x at 8 GPUs is reachable; at N = 1 it is the
    single-GPU time the scaling of this leg is measured against."""
    from deep_cbrs_amar_renaissance_amd import engine, parallel
    from deep_cbrs_amar_renaissance_amd.data import synthetic
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter_device
    dev = torch.device('cuda', local_rank)
    cdev = 'cpu' if rehearse else dev
    engine.set_seed(42)
    data = synthetic.ml1m_device(scale, device=dev)
    n = data['n_users'] + data['n_items']
    a_hat = gcn_filter_device(data['train_pos'][:, 0], data['train_pos'][:, 1], n)
    model = basic.BasicGCN(a_hat, **GRID1)
    model.n_users, model.n_items = data['n_users'], data['n_items']
    u, i = shuffled_test_pairs(data, dev)
    n_pairs = int(u.numel())
    del data
    torch.cuda.empty_cache()
    if world == 1 and multi:                                       # AMAR_FORCE_DIST: the partitioned runner + RCCL with a single rank
        runner = parallel.PartitionedGCNRunner(model, u, i, rank, world)
    else:
        runner = parallel.make_runner(model, u, i, rank, world, dist=parallel.SharedDeviceCollectives(rank, world) if rehearse else None)

    def barrier():
        if multi:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(5):
        runner.step()
    barrier()
    phases = runner.phase_times() if hasattr(runner, 'phase_times') else None
    step, graphed = runner.step, False
    if os.environ.get('AMAR_STEP_GRAPH', '1') != '0' and not rehearse:
        ok = 1
        try:
            runner.capture_step()
        except Exception as exc:
            sys.stderr.write("bench.py: rank {}: hipGraph capture of the ml1m(s={}) step failed ({}); timing eager steps\n".format(rank, scale, exc))
            ok = 0
        if multi:
            flag = torch.tensor([ok], device=cdev, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            ok = int(flag.item())
        if ok:
            step, graphed = runner.step_graphed, True
    for _ in range(20):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if multi:
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t[0].item())
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, dict(phases or {}, rank=rank, rows=int(runner.local_rows), nnz=int(runner.local_nnz),
                                                             pairs=int(runner.u_ids.numel())))
    out = {'workload': 'ml1m(s={}) user-item graph: N={} nodes, nnz(A_hat)={}, {} test pairs; same model and step as the headline'.format(
               scale, n, a_hat.nnz, n_pairs),
           'scale': scale, 'n_gpus': world, 'steps': steps, 'ms_per_step': 1e3 * dt / steps, 'value': n_pairs * steps / dt, 'unit': 'pairs/s',
           'replayed_from_hipgraph': graphed, 'parallelism': runner.describe()}
    if per_rank is not None:
        out['per_rank'] = per_rank
    del runner, model, a_hat, u, i
    torch.cuda.empty_cache()
    return out


def value_spread(scale):
    """min / max ms per step of the default run over the boxes of the build session (profiles/r<round>_bench_repeats.json, written by
    tools/collect_spread.py from the bench.py lines of different gpurun boxes), with the csrc/ hash it was measured at."""
    paths = [os.path.join(ROOT, 'profiles', 'r{}_bench_repeats.json'.format(r)) for r in (4, 3)]
    path = next((q for q in paths if os.path.exists(q)), None)
    if path is None:
        return None
    rep = json.load(open(path))
    if rep.get('scale') != scale:
        return None
    rep['stale'] = rep.get('csrc_sha') != csrc_sha()
    return rep


def onchip_floor(capi, a_hat, f, nnz, fused_ms, pmc):
    """The dispatched LDS-tiled kernel against ITS OWN on-chip floors (the HBM roofline stays in `roofline`): (1) the L2 line
    requests it issues per launch — this build's TCC_REQ counter, profiles/spmm_pmc_latest.json — at the rate a CU front end
    sustains (profiles/r1_exp_gather_frontend.txt); (2) the same launch with the gathers compiled out (index walk + one LDS
    read-add-write per non-zero: AMAR_LT_VARIANT=24, wrong sums, timing only), measured here on the same image."""
    lt = a_hat.tiled_image(f)
    n = a_hat.shape[0]
    x = torch.randn((n, f), device=a_hat.rowptr.device)
    y = torch.empty_like(x)

    def ms(reps=10):
        for _ in range(3):
            capi.spmm_lt(lt, x, y, prescaled=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            capi.spmm_lt(lt, x, y, prescaled=True)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps
    plain_ms = ms()
    ablation = {}
    for name, variant in (('paced', '24'), ('unpaced', '35')):     # 24: gathers compiled out, the shipped barrier pacing; 35: no pacing either
        os.environ['AMAR_LT_VARIANT'] = variant
        try:
            ablation[name] = ms()
        finally:
            os.environ.pop('AMAR_LT_VARIANT', None)
    no_gather_ms = min(ablation.values())
    req = (pmc or {}).get('TCC', {}).get('REQ') if pmc else None
    floor_ms = 1e3 * req / L2_REQUESTS_PER_S if req else None
    return {'kernel': 'spmm_lt_kernel<8>', 'plain_launch_ms': plain_ms, 'fused_launch_ms': fused_ms,
            'l2_requests_per_launch': req, 'l2_requests_per_nonzero': req / nnz if req else None,
            'l2_requests_per_s_sustained': L2_REQUESTS_PER_S, 'request_floor_ms': floor_ms,
            'frac_of_request_floor': floor_ms / plain_ms if floor_ms else None,
            'no_gather_ms': no_gather_ms, 'no_gather_paced_ms': ablation['paced'], 'no_gather_unpaced_ms': ablation['unpaced'],
            'frac_of_no_gather_floor': no_gather_ms / plain_ms,
            'structural_ceiling_frac_of_hbm_peak': (nnz * 8 + (n + 1) * 4 + 2 * n * f * 4) / (no_gather_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            'ceiling_note': 'with every gather compiled out (AMAR_LT_VARIANT 24 / 35: wrong sums, timing only) the walk — index stream, one '
                            'LDS read-add-write per non-zero, with and without the window barriers — still takes no_gather_ms (the faster of '
                            'the two): that caps this kernel STRUCTURE at structural_ceiling_frac_of_hbm_peak of the HBM roofline on the 8(d) '
                            'bytes at F = 8, fp32, whatever the gathers cost (DESIGN.md 4a); the request floor counts this build\'s own L2 '
                            'requests (about 0.5 per non-zero: neighbouring entries of the column-ordered walk share L1 lines), not one per '
                            'non-zero as round 1\'s XS model did'}


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world > 1 and os.environ.get('AMAR_BENCH_CHILD') != '1' and os.environ.get('AMAR_BENCH_SUPERVISE', '1') != '0':
        sys.exit(supervise(rank))
    retry = os.environ.get('AMAR_BENCH_RETRY')
    if os.environ.get('AMAR_BENCH_FAIL_FIRST') == '1' and not retry and world > 1:     # (tests: the supervisor's second attempt)
        os._exit(134)
    if world != args.gpus:
        sys.exit("bench.py: --gpus {} but WORLD_SIZE={}: start it as `python bench.py --gpus N` (it launches its own ranks) or "
                 "under torchrun with --nproc-per-node equal to --gpus".format(args.gpus, world))
    # stdout carries exactly ONE line, the JSON: whatever libraries print meanwhile (RCCL's version banner goes to stdout)
    # is sent to stderr until then
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP path has no CPU fallback"
    # AMAR_REHEARSE_ONE_GPU=1: all ranks on device 0 over gloo (parallel.SharedDeviceCollectives) — a rehearsal of the multi-rank
    # code path on a one-GPU box, not a measurement (the ranks share the GPU; RCCL and the captured step are not exercised)
    rehearse = os.environ.get('AMAR_REHEARSE_ONE_GPU') == '1' and world > 1
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dist = bool(os.environ.get('AMAR_FORCE_DIST'))      # rehearse the RCCL path with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        if force_dist and world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29531')
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        if rehearse:
            dist.init_process_group('gloo')
        else:
            import datetime
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=datetime.timedelta(seconds=900 if retry else 300))   # (a stuck collective ends the run instead of hanging it; a second attempt waits for the peers whose first one is still timing out)

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

    runner = parallel.make_runner(model, u_all, i_all, rank, world, dist=parallel.SharedDeviceCollectives(rank, world) if rehearse else None) \
        if not force_dist else parallel.PartitionedGCNRunner(model, u_all, i_all, rank, world)

    spmm_events, kinds_seen = [], []
    # The timed steps replay the whole step — on several ranks its collectives too — from a hipGraph the runner captures itself
    # (AMAR_STEP_GRAPH=0: the eager steps are the timed ones).  No per-launch event can be recorded inside a capture, so the
    # kernel-level objects below (roofline, pair_stage) are timed over K EAGER steps run just before, whose own rate is reported
    # as `eager`; the kernels and their durations are the same in both.
    graph_step = hasattr(runner, 'step_graphed') and os.environ.get('AMAR_STEP_GRAPH', '1') != '0' and not rehearse
    restore_spmm = leaf_spmm_timers(capi, spmm_events, kinds_seen)       # whichever form the layer dispatches to, timed once
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
    scatter_events, raw_scatter = [], capi.scatter

    def timed_scatter(*a, **k):                                # the second launch of the pair stage: scores into the caller's order
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        raw_scatter(*a, **k)
        e1.record()
        scatter_events.append((e0, e1))
    capi.scatter = timed_scatter
    multi = world > 1 or force_dist
    cdev = 'cpu' if rehearse else dev                          # gloo reduces host tensors

    def barrier():
        if multi:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 20)):                   # the kernel-level timings below want settled clocks too
        runner.step()
    barrier()
    spmm_events.clear()
    pair_events.clear()
    scatter_events.clear()
    del kinds_seen[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step()
    eager_host_dt = time.perf_counter() - t0
    barrier()
    eager_dt = time.perf_counter() - t0
    restore_spmm()
    capi.chain = raw_chain
    capi.scatter = raw_scatter
    phases = runner.phase_times() if hasattr(runner, 'phase_times') else None      # the last eager step, by phase (typed partition)
    step = runner.step
    if graph_step:
        # Capture first, replay only once EVERY rank has a graph: a replay enqueues the step's collectives, which a rank whose
        # capture failed would never match (the others would hang in their next synchronisation).  All ranks switch to eager steps
        # together if any capture failed.
        ok = 1
        try:
            runner.capture_step()
        except Exception as exc:                              # a capture the runtime refuses: the eager steps stand
            sys.stderr.write("bench.py: rank {}: hipGraph capture of the step failed ({}); timing eager steps\n".format(rank, exc))
            ok = 0
        if multi:
            flag = torch.tensor([ok], device=cdev, dtype=torch.int32)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            ok = int(flag.item())
        graph_step = bool(ok)
        if graph_step:
            # untimed replays until the clocks have settled: the eager steps before leave gaps between kernels, and the first few
            # dozen replays of the graph run below the sustained rate (10 timed steps right after the capture: 1.26 ms per step;
            # 100: 1.19 ms — profiles/r2_bench_repeats.txt)
            for _ in range(max(args.warmup, 40)):
                runner.step_graphed()
            step = runner.step_graphed
        barrier()
    t0 = time.perf_counter()
    if graph_step:
        for _ in range(args.steps):
            step()
        host_dt = time.perf_counter() - t0                    # time to ENQUEUE the steps (host-side launch cost)
        barrier()
        dt = time.perf_counter() - t0
    else:
        host_dt, dt = eager_host_dt, eager_dt
    per_rank = None
    if multi:
        t = torch.tensor([dt, eager_dt], device=cdev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt, eager_dt = float(t[0].item()), float(t[1].item())
        # Self-check of the partitioned step (after the timed region): every rank runs the SINGLE-GPU step on the same inputs once (it
        # holds the whole graph) and compares the scores of its own pair shard — the multi-rank exchange has nothing else to vouch
        # for it on a node the builder never had (the oracle-checked tests cover 2-3 ranks over gloo and rank threads on one GPU).
        diff = float('nan')
        if hasattr(runner, 'pair_index') and os.environ.get('AMAR_BENCH_SELF_CHECK', '1') != '0':
            try:
                ref = parallel.SingleRunner(model, u_all, i_all).step().view(-1)
                got = runner.step().view(-1)
                diff = float((got - ref[runner.pair_index]).abs().max()) if got.numel() else 0.0
                del ref, got
            except Exception as exc:
                sys.stderr.write("bench.py: rank {}: self-check against the single-GPU step failed to run ({})\n".format(rank, exc))
            a_hat.__dict__.pop('_lt_cache', None)
            torch.cuda.empty_cache()
        mine = dict(phases or {}, rank=rank, rows=int(runner.local_rows), nnz=int(runner.local_nnz), pairs=int(runner.u_ids.numel()),
                    max_abs_score_diff_vs_single_gpu=diff)
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)

    spmm_ms = [e0.elapsed_time(e1) for e0, e1 in spmm_events]
    rows_local = runner.local_rows
    nnz_local = runner.local_nnz
    f = GRID1['n_hiddens'][0]
    kind = {'gcn_layer': 'csr', 'spmm_sj': 'sj', 'spmm_xs': 'xs', 'spmm_lt': 'lt'}[kinds_seen[0]] if kinds_seen else 'none'
    alg_bytes = nnz_local * 8 + (rows_local + 1) * 4 + (n_nodes + rows_local) * f * 4
    avg_ms = float(np.mean(spmm_ms)) if spmm_ms else float('nan')
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    pmc, traffic_source = (pmc_profile(args.scale, kind) if not multi else (None, 'not profiled for multi-rank runs'))
    kernel_names = {'sj': 'spmm_sj_kernel<8>', 'xs': 'spmm_xs_partial_kernel<8> + spmm_xs_combine_kernel<8>', 'csr': 'spmm_stream_kernel<8>',
                    'lt': 'spmm_lt_kernel<8> (LDS-tiled, one launch)', 'none': 'none'}

    s256 = None
    if args.scale == 64 and not args.no_s256 and not rehearse:     # every rank takes part (its collectives); rank 0 reports
        s256 = larger_scale_leg(256, args.steps, rank, world, local_rank, multi, rehearse)

    if rank == 0:
        # what the dispatched image actually streams per launch (the 8(d) formula counts a canonical 8-byte CSR entry)
        image_entry_bytes = {'csr': 8, 'sj': 8, 'xs': 4, 'lt': 4, 'none': 0}[kind]
        out = {
            'metric': '(user,item) pairs scored/sec, ML-1M basic-gnn 2-layer, 1/2/4/8 MI355X',
            'value': n_pairs * args.steps / dt, 'unit': 'pairs/s', 'csrc_sha': csrc_sha(),
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
            'dtype_note': 'f32 values and f32 sums everywhere; the pair stage takes its products on the bf16 matrix instruction with both operands '
                          'split into three bf16 parts (x = hi + mid + lo exactly, six part products accumulated in f32): as close to a float64 '
                          'evaluation as the f32 instruction (DESIGN 4b); AMAR_PAIR_MFMA=f32 / AMAR_DENSE_SPLIT=0 select v_mfma_f32_*_f32',
            'data': 'synthetic',
            'config': {'workload': 'ml1m(s={}) user-item graph: N={} nodes, nnz(A_hat)={}, {} test pairs; '
                                   'econfigs/basic-gnn.yaml grid1 BasicGCN d=8 L=2 concat, dense [24,24], clf [48,48]; '
                                   'one propagation + per-entity towers + all pairs (shuffled order) per step (hoisted)'.format(args.scale, n_nodes, nnz, n_pairs),
                       'scale': args.scale, 'parallelism': runner.describe() + (' (step replayed from a hipGraph)' if graph_step else '') +
                       (' — one-GPU rehearsal over gloo: NOT a scaling measurement' if rehearse else ''),
                       **({'retry': retry + ': eager steps timed'} if retry else {})},
            'roofline': {'bound': 'hbm', 'kernel': kernel_names[kind] + ' (fused GCN layer: SpMM + bias + ReLU + next X.W)', 'achieved': achieved,
                         'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS,
                         'traffic': pmc['traffic_bytes_per_launch'] if pmc else None, 'traffic_source': traffic_source,
                         'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_ms': avg_ms,
                         'launches_timed': len(spmm_ms), 'timed_in': 'the K eager steps just before the replayed ones' if graph_step else 'timed region',
                         'note': 'achieved = SURVEY 8(d) bytes (canonical CSR: 8 B per non-zero) / time, i.e. a CSR-equivalent effective '
                                 'bandwidth; the dispatched image streams {} B per non-zero'.format(image_entry_bytes)},
            'propagation_ms': runner.last_propagation_ms(),
            'host_enqueue_ms_per_step': 1e3 * host_dt / args.steps,
            'eager': {'ms_per_step': 1e3 * eager_dt / args.steps, 'value': n_pairs * args.steps / eager_dt, 'unit': 'pairs/s',
                      'host_enqueue_ms_per_step': 1e3 * eager_host_dt / args.steps,
                      'note': 'the same K steps launched one kernel at a time, each SpMM / pair-stage launch bracketed by HIP events'},
        }
        if s256 is not None:
            out['s256'] = s256
        spread = value_spread(args.scale)
        if spread is not None and not multi:
            out['value_spread_boxes'] = spread
        if per_rank is not None:
            diffs = [r.get('max_abs_score_diff_vs_single_gpu', float('nan')) for r in per_rank]
            out['multi_rank_parity'] = {'max_abs_score_diff_vs_single_gpu': max(diffs) if all(d == d for d in diffs) else None,
                                        'ok': bool(all(d == d and d < 1e-5 for d in diffs)),
                                        'note': 'every rank re-ran the single-GPU step on the same inputs after the timed region and compared the scores of its pair shard '
                                                '(two fp32 summation orders of the same arithmetic: expected <= 1e-6)'}
            out['per_rank'] = {'ranks': per_rank,
                               'note': 'one EAGER step by phase on every rank, HIP events on the compute stream: local_spmm_ms and pair_stage_ms '
                                       'shrink with the rank count, replicated_ms (X_0 . W_1 over all rows + the item tower) does not, exposed_exchange_ms '
                                       '(= exchange_ms) is the time the compute stream spent issuing all-gathers and waiting for the sections it needs next: '
                                       'a layer runs one launch per node type and gathers each type\'s section of the next table behind the other types\' launches '
                                       '(parallel.py), so what shows here is the part of the exchange compute did not cover'}
        if kind == 'lt' and not multi:
            out['roofline_onchip'] = onchip_floor(capi, a_hat, f, nnz_local, avg_ms, pmc)
        pair_ms = [e0.elapsed_time(e1) for e0, e1 in pair_events]
        if pair_ms:
            # the other large kernel of a step: the sum-input chain kernel gathers two 48-float per-entity rows per pair
            c1, c2 = GRID1['clf_units'][0], GRID1['clf_units'][1]
            pairs_local = int(runner.u_ids.numel())
            pair_alg = pairs_local * (2 * c1 * 4 + 2 * 4 + 4)                 # what the kernel gathers: two folded 192-B rows
            pair_alg_8d = pairs_local * 204                                    # two 24-float rows + ids + score
            pms = float(np.mean(pair_ms))
            sms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in scatter_events])) if scatter_events else 0.0
            flop_pair = 2 * (c1 * c2 + c2)
            f32_form = os.environ.get('AMAR_PAIR_MFMA') == 'f32'
            # executed on the bf16 pipe: 6 part products x (tiles of 16 outputs) x (k-steps of 32) x 16x16x32 MACs per 16 pairs
            bf16_flop_pair = 6 * (-(-c2 // 16)) * (-(-c1 // 32)) * (16 * 16 * 32 * 2) / 16.0
            out['pair_stage'] = {'kernel': 'chain_pipe_kernel<3,2> (relu(T_u[u] + T_i[i]) -> Dense 48 -> Dense 1, sigmoid)' +
                                           (' + scatter_windows_kernel (scores into the caller\'s order)' if scatter_events else ''),
                                 'avg_launch_ms': pms, 'scatter_launch_ms': sms, 'stage_ms': pms + sms,
                                 'pairs_per_launch': pairs_local, 'bound': 'mfma',
                                 'bound_note': 'with the prepared pair list the gathered rows come from the L2s (traffic = what the memory side moved, '
                                               'about half the gathered bytes); the scores go back to the caller\'s order in two steps (window streams '
                                               'from the kernel, amar_scatter_f32 inside the windows: scatter_launch_ms) because 12 M single-word stores '
                                               'at random cost 0.19 ms of memory-side work inside the launch; the kernel itself is bound by vector issue: '
                                               + ('fp32 MFMA and VALU time add up on gfx950' if f32_form else
                                                  'its products run on the bf16 matrix instruction with both operands split three ways (f32-accurate, '
                                                  'DESIGN 4b) and the splitting shares the SIMD\'s issue port with the MFMAs') +
                                               '; the byte fractions below are kept for comparison with SURVEY 8(d) and may exceed 1',
                                 'algorithmic_bytes_per_launch': pair_alg, 'achieved': pair_alg / (pms * 1e-3) / 1e9,
                                 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': pair_alg / (pms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                 'frac_label': '396 B/pair: the two 48-float per-entity rows (classifier layer 1 folded into the towers) + ids + score',
                                 'frac_8d': pair_alg_8d / (pms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                 'frac_8d_label': '204 B/pair as SURVEY 8(d) counts it (two 24-float rows): the fold doubles the gathered bytes to halve the MFMA work',
                                 'traffic': pmc.get('pair_stage_traffic_bytes_per_launch') if pmc else None, 'traffic_source': traffic_source,
                                 # the same launch against the matrix-pipe peaks: the f32 work it stands for against the f32 MFMA peak
                                 # (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense), and what it executes against the dense bf16 peak
                                 'mfma': {'frac': (pairs_local * flop_pair / (pms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS) if f32_form else
                                                  (pairs_local * bf16_flop_pair / (pms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS),
                                          'frac_note': 'EXECUTED matrix instructions over the dense peak of that instruction ({}): a utilisation'.format(
                                              'v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s' if f32_form else 'v_mfma_f32_16x16x32_bf16, 2 500 TFLOP/s; six part products per f32 product, k padded 48 -> 64'),
                                          'executed_flop_per_pair': flop_pair if f32_form else bf16_flop_pair,
                                          'executed_tflops': pairs_local * (flop_pair if f32_form else bf16_flop_pair) / (pms * 1e-3) / 1e12,
                                          'executed_peak_tflops': MFMA_F32_PEAK_TFLOPS if f32_form else MFMA_BF16_PEAK_TFLOPS,
                                          'pmc_busy_frac': pmc.get('pair_stage_mfma_busy_frac') if pmc else None,
                                          'f32_equiv': {'flop_per_pair': flop_pair,
                                                        'flop_note': 'per-pair f32 work after hoisting the towers and the first classifier layer per entity '
                                                                     '(SURVEY 8(d) counts 13 920 flop/pair for the un-hoisted head; test_faithful_equals_hoisted_and_predict keeps the two equal)',
                                                        'tflops': pairs_local * flop_pair / (pms * 1e-3) / 1e12, 'f32_peak_tflops': MFMA_F32_PEAK_TFLOPS,
                                                        'note': 'the f32 work the launch stands for (secondary; an exact-f32 formulation tops out at the f32 peak)'}}}
        if world == 1 and not args.no_cpu_baseline:
            # the headline workload's own inputs for the CPU leg (host copies), before the GPU objects go
            from tests import helpers
            headline = {'a_hat': (a_hat.rowptr.cpu().numpy(), a_hat.colidx.cpu().numpy(), a_hat.vals.cpu().numpy()),
                        'gnn': helpers.gnn_to_oracle(model.gnn), 'head': helpers.basic_head_to_oracle(model.rs),
                        'u': u_all.cpu().numpy().astype(np.int64), 'i': i_all.cpu().numpy().astype(np.int64),
                        'gpu_scores': runner.step().cpu().numpy(), 'workload': 'ml1m(s={})'.format(args.scale)}
            del runner, model, a_hat, u_all, i_all
            torch.cuda.empty_cache()
            out['wider_layers'] = wider_layers(dev, args.scale, args.steps)
            out['hybrid_head'] = hybrid_head(dev, args.scale)
            out['uip_graph'] = uip_graph(dev, args.scale, args.steps)
            out['model_families'] = model_families(dev, args.scale, args.steps)
            out['ml1m_s1'], s1 = ml1m_true_size(dev)
            out['train_s1'] = train_true_size()
            out['cpu_baseline'] = cpu_baseline(s1, headline)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
