"""CPU, world_size 2 and 3 over gloo: the typed node-range partition + per-layer all-gather logic.

The HIP kernels cannot run here, so the runner's kernel provider is replaced by a numpy stand-in
(test infrastructure) while partitioning, the type-major table layout, the per-type row blocks, CSR column remapping and the
collectives are the shipped code.  Each rank's own blocks and gathered tables must equal the
single-process oracle propagation.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from scipy import sparse

from tests import helpers

GRID1 = dict(embedding_dim=8, n_hiddens=[8, 8], dense_units=[24, 24], clf_units=[48, 48])


class NumpyOps:
    """CPU stand-ins with the signatures of the capi functions the partitioned runner calls."""

    @staticmethod
    def copy_columns(src, dst, ids=None, base=0):
        dst.copy_(src if ids is None else src[ids.long() - base])

    @staticmethod
    def rowwise_xw(X, W, H, copy_to=None, row_ids=None, row_scale=None, **kw):
        if row_ids is not None:                                     # amar_rowwise_xw_gather_f32: a negative id leaves a zero row
            src = X[row_ids.long().clamp(min=0)] * (row_ids >= 0).to(X.dtype)[:, None]
            h = src @ W.detach()
            H.copy_(h if row_scale is None else h * row_scale[:, None])
            return
        H.copy_(X @ W.detach())
        if copy_to is not None:
            copy_to.copy_(X)

    @staticmethod
    def gcn_layer(rowptr, colidx, vals, H, bias, Y, Wnext=None, Hnext=None):
        n = rowptr.numel() - 1
        a = sparse.csr_matrix((vals.numpy(), colidx.numpy(), rowptr.numpy()), shape=(n, H.shape[0]))
        y = np.maximum(a @ H.numpy() + bias.detach().numpy(), 0)
        Y.copy_(torch.from_numpy(y))
        if Wnext is not None:
            Hnext.copy_(torch.from_numpy(y @ Wnext.detach().numpy()))

    @staticmethod
    def spmm_csr(rowptr, colidx, vals, X, Y=None, acc_in=None, acc_out=None, acc_div=None, **kw):
        n = rowptr.numel() - 1
        a = sparse.csr_matrix((vals.numpy(), colidx.numpy(), rowptr.numpy()), shape=(n, X.shape[0]))
        y = torch.from_numpy(a @ X.detach().numpy())
        if Y is not None:
            Y.copy_(y)
        if acc_out is not None:                                     # LightGCN's running layer sum / mean in the epilogue
            acc_out.copy_((acc_in + y) / (acc_div if acc_div is not None else 1))

    @staticmethod
    def add_inplace(dst, src, scale=1.0):
        dst.add_(src, alpha=scale)

    @staticmethod
    def row_affine(a, scale, out, b=None):
        out.copy_((a if b is None else a + b) * scale[:, None])


def _typed_worker(rank, world, port, out_dir, phases):
    """The typed partition (GCN stack of a model that knows its user / item split): equal-height blocks per node type, per layer one
    all-gather of the next gathered table and one of the item rows."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), AMAR_PART_PHASES=phases)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from deep_cbrs_amar_renaissance_amd import engine, parallel
        from deep_cbrs_amar_renaissance_amd.models import basic
        engine.set_seed(42)
        g = helpers.tiny_graph(n_users=70, n_items=45, n_ratings=1500, seed=8, n_props=25, n_links=90)
        model = basic.BasicGCN(g['adj'], **GRID1)
        model.n_users, model.n_items = g['n_users'], g['n_items']
        helpers.randomize_biases(model, seed=3)
        u, i = torch.from_numpy(g['u_ids']), torch.from_numpy(g['i_ids'])
        runner = parallel.PartitionedGCNRunner(model, u, i, rank, world, ops=NumpyOps, dist=dist, timing=False)
        assert runner.typed and runner.tpart.G == (2 if phases == '1' else 1)
        x_local, x_items = runner.propagate_typed()
        runner.wait_exchange()
        # LightGCN ('mean' over the layers, accumulated on the rank's own rows) on the same partition; and a model that does not know
        # its user / item split: one node type, pairs follow whichever node they name first
        light = basic.BasicLightGCN(g['adj'], **dict(GRID1, n_layers=3))
        light.n_users, light.n_items = g['n_users'], g['n_items']
        lrun = parallel.PartitionedGCNRunner(light, u, i, rank, world, ops=NumpyOps, dist=dist, timing=False)
        l_local, l_items = lrun.propagate_typed()
        lrun.wait_exchange()
        blind = basic.BasicGCN(g['adj'], **GRID1)
        brun = parallel.PartitionedGCNRunner(blind, u, i, rank, world, ops=NumpyOps, dist=dist, timing=False)
        assert brun.typed and brun.tpart.T == 1 and brun.n_items == g['adj'].shape[0]
        b_local, b_items = brun.propagate_typed()
        brun.wait_exchange()
        tp = runner.tpart
        np.savez(os.path.join(out_dir, 'typed{}.npz'.format(rank)),
                 owned=np.array([tp.owned(rank, t) for t in range(tp.T)]), off=np.array(tp.off), h=np.array(tp.h),
                 pair_index=runner.pair_index.numpy(), u_ids=runner.u_ids.numpy(), i_ids=runner.i_ids.numpy(),
                 nnz=np.array(runner.local_nnz), **{'xl%d' % k: x.numpy() for k, x in enumerate(x_local)},
                 **{'xi%d' % k: x.numpy() for k, x in enumerate(x_items)},
                 light_local=l_local[0].detach().numpy(), light_items=l_items[0].detach().numpy(),
                 blind_rows=np.array(brun.tpart.owned(rank, 0)), blind_pairs=brun.pair_index.numpy(),
                 **{'bl%d' % k: x.numpy() for k, x in enumerate(b_local)}, **{'bi%d' % k: x.numpy() for k, x in enumerate(b_items)})
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize('phases', ['1', '0'])
@pytest.mark.parametrize('world', [2, 3])
def test_typed_partition_matches_oracle(tmp_path, world, phases):
    from oracle import models as om
    from deep_cbrs_amar_renaissance_amd import engine
    from deep_cbrs_amar_renaissance_amd.models import basic
    from deep_cbrs_amar_renaissance_amd.utilities.math import gcn_filter
    mp.spawn(_typed_worker, args=(world, _free_port(), str(tmp_path), phases), nprocs=world, join=True)
    engine.set_seed(42)
    g = helpers.tiny_graph(n_users=70, n_items=45, n_ratings=1500, seed=8, n_props=25, n_links=90)
    model = basic.BasicGCN(g['adj'], **GRID1)
    helpers.randomize_biases(model, seed=3)
    want = om.propagate(g['adj'], helpers.gnn_to_oracle(model.gnn), np.float64)      # [N, 24] = [X_0 || X_1 || X_2]
    light = basic.BasicLightGCN(g['adj'], **dict(GRID1, n_layers=3))                  # same seed order as in the workers
    want_light = om.propagate(g['adj'], helpers.gnn_to_oracle(light.gnn), np.float64)
    blind = basic.BasicGCN(g['adj'], **GRID1)
    want_blind = om.propagate(g['adj'], helpers.gnn_to_oracle(blind.gnn), np.float64)
    nu, ni = g['n_users'], g['n_items']
    n = g['adj'].shape[0]
    shards, blind_shards, total_nnz = [], [], 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), 'typed{}.npz'.format(r)))
        for k in range(2):
            cols = slice(8 * (k + 1), 8 * (k + 2))
            # the gathered item rows are the item table in the reference's item order, on every rank
            assert helpers.rel_err(z['xi%d' % k][:ni], want[nu:nu + ni, cols]) < 1e-5
            # the rank's own block, type after type at the block offsets
            for t, (lo, hi) in enumerate(z['owned']):
                if hi > lo:
                    assert helpers.rel_err(z['xl%d' % k][z['off'][t]:z['off'][t] + hi - lo], want[lo:hi, cols]) < 1e-5
        ulo, uhi = z['owned'][0]
        assert ((z['u_ids'] >= ulo) & (z['u_ids'] < uhi)).all()                     # pairs follow their user's owner
        assert np.array_equal(z['u_ids'], g['u_ids'][z['pair_index']]) and np.array_equal(z['i_ids'], g['i_ids'][z['pair_index']])
        shards.append(z['pair_index'])
        total_nnz += int(z['nnz'])
        # LightGCN: the mean over the layers for the rank's own rows and, gathered, for every item
        assert helpers.rel_err(z['light_items'][:ni], want_light[nu:nu + ni]) < 1e-5
        for t, (lo, hi) in enumerate(z['owned']):
            if hi > lo:
                assert helpers.rel_err(z['light_local'][z['off'][t]:z['off'][t] + hi - lo], want_light[lo:hi]) < 1e-5
        # no user / item split known: one type, the "item" gather is the whole table
        lo, hi = z['blind_rows']
        for k in range(2):
            cols = slice(8 * (k + 1), 8 * (k + 2))
            assert helpers.rel_err(z['bi%d' % k][:n], want_blind[:, cols]) < 1e-5
            assert helpers.rel_err(z['bl%d' % k][:hi - lo], want_blind[lo:hi, cols]) < 1e-5
        blind_shards.append(z['blind_pairs'])
    assert np.array_equal(np.sort(np.concatenate(shards)), np.arange(len(g['u_ids'])))   # every pair scored exactly once
    assert np.array_equal(np.sort(np.concatenate(blind_shards)), np.arange(len(g['u_ids'])))
    assert total_nnz == gcn_filter(g['adj']).nnz


def test_typed_partition_layout():
    from deep_cbrs_amar_renaissance_amd.parallel import TypedPartition
    for bounds, world in (([0, 70, 115, 140], 4), ([0, 6036, 9228], 8), ([0, 3, 5], 8), ([0, 10, 10, 17], 2)):
        T = len(bounds) - 1
        for groups in (None, [list(range(T))]) + (([[0, 2], [1]],) if T == 3 else ()):
            tp = TypedPartition(bounds, world, groups=groups)
            n = bounds[-1]
            ids = torch.arange(n)
            p = tp.padded_index(ids)
            assert len(torch.unique(p)) == n and int(p.max()) < world * tp.R
            assert torch.equal(tp.node_of_row('cpu')[p].long(), ids) and int((tp.node_of_row('cpu') >= 0).sum()) == n
            assert tp.R == sum(tp.h) and tp.goff[-1] == world * tp.R and sum(tp.gh) == tp.R
            for t in range(tp.T):
                g = tp.group_of[t]
                spans = [tp.owned(r, t) for r in range(world)]
                assert spans[0][0] == bounds[t] and spans[-1][1] == bounds[t + 1]
                assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:])) and all(hi - lo <= tp.h[t] for lo, hi in spans)
                # a type's rows, taken out of the blocks in rank order, are the type in id order: block r starts at r * h_t
                for r, (lo, hi) in enumerate(spans):
                    if hi > lo:
                        assert lo - bounds[t] == r * tp.h[t]
                        # group-major tables: rank r's block of group g starts at block_row0(r, g), the type at its offset inside the block
                        assert torch.equal(p[lo:hi], tp.block_row0(r, g) + tp.in_group[t] + torch.arange(hi - lo))
                        assert tp.section(g)[0] <= int(p[lo]) and int(p[hi - 1]) < tp.section(g)[1]
                        assert torch.equal(tp.group_of_row(p[lo:hi]), torch.full((hi - lo,), g))
            if groups is None:                                      # one group per type: type-major, a type's section IS the type in id order
                for t in range(tp.T):
                    assert torch.equal(p[bounds[t]:bounds[t + 1]], tp.goff[t] + torch.arange(bounds[t + 1] - bounds[t]))
            elif len(groups) == 1:                                  # one group: the rank-major layout of round 3
                for t in range(tp.T):
                    for r in range(world):
                        lo, hi = tp.owned(r, t)
                        assert torch.equal(p[lo:hi], r * tp.R + tp.off[t] + torch.arange(hi - lo))
            assert world * tp.R - n <= sum(world for _ in range(tp.T)) + sum(tp.h)   # padding: O(world) rows per type (+ a short last block)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_supervisor_flow_without_a_gpu():
    """bench.py under a multi-rank launcher runs the rank's work in a child and repeats a failed graph-replayed run once with eager steps
    (bench.supervise).  Without a GPU both children stop at `bench.py needs a GPU`: the parent must report the first failure, start the
    second attempt with AMAR_STEP_GRAPH=0 and pass the second exit code on — and print nothing on stdout."""
    import subprocess
    import sys
    import torch
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("covers the no-GPU flow (tests/test_parallel_nccl_gpu.py has the GPU one)")
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29577')
    for k in ('AMAR_BENCH_CHILD', 'AMAR_BENCH_RETRY', 'AMAR_STEP_GRAPH', 'AMAR_BENCH_SUPERVISE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ''
    assert 'once more with eager steps' in r.stderr and r.stderr.count('AssertionError: bench.py needs a GPU') == 2, r.stderr[-1500:]
    # with the graph already off there is nothing to fall back to: one attempt only
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=dict(env, AMAR_STEP_GRAPH='0'), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'once more with eager steps' not in r.stderr and r.stderr.count('AssertionError: bench.py needs a GPU') == 1
