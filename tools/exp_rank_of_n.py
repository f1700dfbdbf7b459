#!/usr/bin/env python
"""Device time of ONE rank's step in a world-W node-range partition of bench.py's workload, on one GPU (development aid for the
scaling estimate of DESIGN.md §6): the rank's own kernels run for real — its row block on LT / XS, the replicated X.W, towers,
its pair shard — while the all-gathers are replaced by a local copy of the rank's own block (wrong neighbours' data, right
sizes), so the number is the step WITHOUT the exchange; EXP_WIRE=1 adds the same step under an EMULATED wire (class EmulatedWire:
a device-side delay per collective on one side stream) and prints the exposed part of the exchange.
`python tools/exp_rank_of_n.py [scale] [world ...]`."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

GRID1 = dict(embedding_dim=8, n_hiddens=[8, 8], n_layers=2, dense_units=[24, 24], clf_units=[48, 48], l2_regularizer=1e-4,
             final_node='concatenation', activation='relu')


class _Done:
    def wait(self):
        pass


class LocalCopy:
    """all_gather_into_tensor stand-in: only this rank's block lands (at its place); the other blocks keep last step's bytes.
    EXP_NO_COPY=1: not even that (the first call of every buffer still copies, so the tables hold finite values) — the step
    without ANY exchange work."""
    def __init__(self, rank):
        self.rank, self.seen = rank, set()

    def all_gather_into_tensor(self, out, inp, async_op=False):
        r = inp.shape[0]
        if not (os.environ.get('EXP_NO_COPY') == '1' and out.data_ptr() in self.seen):
            self.seen.add(out.data_ptr())
            out[self.rank * r:(self.rank + 1) * r].copy_(inp)
        return _Done() if async_op else None


class _WireHandle:
    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class EmulatedWire:
    """all_gather_into_tensor stand-in with an EMULATED WIRE: collectives run one after the other on ONE side
    stream (as RCCL's do), each waits for the compute stream at the point of issue, then takes
        EXP_WIRE_LATENCY_US (20) + block bytes / EXP_WIRE_GBPS (75 GB/s per link; every peer has its own link)
    of device time (tools/micro/spin.hip: a one-wave kernel polling s_memrealtime) and lands this rank's own block.  async_op=True
    returns a handle whose wait() makes the current stream wait for that — what torch.distributed's Work does on the nccl backend.
    The number it gives is the step WITH the exposed part of the exchange under that wire model; nothing crosses a real link."""
    supports_async = True

    def __init__(self, rank):
        import ctypes
        self.rank = rank
        self.lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'libexp_spin.so'))
        self.lib.exp_spin_us.argtypes = [ctypes.c_double, ctypes.c_void_p]
        self.lib.exp_spin_us.restype = ctypes.c_int
        self.side = torch.cuda.Stream()
        self.lat = float(os.environ.get('EXP_WIRE_LATENCY_US', 20))
        self.gbps = float(os.environ.get('EXP_WIRE_GBPS', 75))
        self.calls, self.wire_us = 0, 0.0

    def all_gather_into_tensor(self, out, inp, async_op=False):
        r = inp.shape[0]
        us = self.lat + inp.numel() * inp.element_size() / (self.gbps * 1e3)
        self.calls += 1
        self.wire_us += us
        cur = torch.cuda.current_stream()
        self.side.wait_stream(cur)
        with torch.cuda.stream(self.side):
            assert self.lib.exp_spin_us(us, self.side.cuda_stream) == 0
            out[self.rank * r:(self.rank + 1) * r].copy_(inp)
            ev = torch.cuda.Event()
            ev.record(self.side)
        if async_op:
            return _WireHandle(ev)
        cur.wait_event(ev)
        return None


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
    wire = os.environ.get('EXP_WIRE') == '1'
    for world in worlds:
        ranks = [int(r) for r in os.environ['EXP_RANKS'].split(',') if int(r) < world] if os.environ.get('EXP_RANKS') else sorted({0, world // 2, world - 1})
        for rank in ranks:
            def timed(dist_):
                runner = parallel.PartitionedGCNRunner(model, u, i, rank, world, dist=dist_, timing=False)
                for _ in range(40):
                    runner.step_graphed()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    runner.step_graphed()
                torch.cuda.synchronize()
                return runner, (time.perf_counter() - t0) / 20
            os.environ['EXP_NO_COPY'] = '1'
            runner, dt = timed(LocalCopy(rank))
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
            if wire:
                w = EmulatedWire(rank)
                runner, dtw = timed(w)
                per_step = w.calls and (w.wire_us / w.calls, w.calls)
                w.calls, w.wire_us = 0, 0.0
                runner.step()
                print('      emulated wire (%.0f us + bytes / %.0f GB/s per collective, one collective stream): %.4f ms per step -> exposed exchange %.1f us; '
                      '%d collectives, %.1f us of wire time per step; speed-up %.2fx' % (w.lat, w.gbps, 1e3 * dtw, 1e6 * (dtw - dt), w.calls, w.wire_us, t1 / dtw), flush=True)
                del runner, w


if __name__ == '__main__':
    main()
