"""Multi-GPU execution of the hot path: node-range partition + per-layer all-gather.

The reference is single-device; this layer is new design.  One process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  The graph propagation Y = A_hat.X
is row-separable, the scoring head is pair-separable:

* rows of A_hat are split into contiguous ranges of (nearly) equal non-zero count, one per rank;
* node tables live in a *padded* index space: node j owned by rank r at local offset o sits at
  row r*R + o (R = largest range), so that ``all_gather_into_tensor`` of equal [R, C] shards
  lands directly in the layout the next SpMM gathers from — the local CSR's column indices are
  remapped once, at partition time, and no compaction pass is needed;
* per GCN layer: ONE local SpMM kernel (bias + ReLU fused), then ONE all-gather of the layer's own
  [R, C] output block (the bipartite id grouping means every rank needs nearly all rows of the
  other node type, so a plain all-gather beats a sparse halo exchange); the gathered block is both
  the layer's slice of the final node table and the input of the next layer's tiny X.W, which every
  rank recomputes for all rows (replicated weights) rather than exchanging a second block;
* the weights (node table included) are replicated, so the X_0.W_1 prologue needs no exchange;
* every rank then holds the whole [N, F_cat] table and scores its contiguous 1/G slice of the pairs.

``ops`` is the kernel provider (the ctypes binding by default); tests inject a CPU stand-in to
exercise the partition / exchange logic under ``gloo`` without a GPU.
"""
import os

import numpy as np
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.layers.dgcf_conv import DGCFConv
from deep_cbrs_amar_renaissance_amd.layers.gat_conv import GATConv
from deep_cbrs_amar_renaissance_amd.layers.gcn_conv import GCNConv
from deep_cbrs_amar_renaissance_amd.layers.graphsage_conv import GraphSageConv
from deep_cbrs_amar_renaissance_amd.layers.lightgcn_conv import LightGCNConv
from deep_cbrs_amar_renaissance_amd.utilities.math import DeviceCSR


def partition_rows_by_nnz(rowptr, world):
    """Boundaries b[0..world] of contiguous row ranges with (nearly) equal non-zero counts."""
    rowptr = rowptr.to(torch.int64)
    n = rowptr.numel() - 1
    nnz = int(rowptr[-1])
    targets = torch.arange(1, world, dtype=torch.int64, device=rowptr.device) * nnz // world
    cuts = torch.searchsorted(rowptr, targets, right=False).clamp_(0, n)
    bounds = [0] + [int(c) for c in cuts.cpu()] + [n]
    for k in range(1, len(bounds)):                      # monotone even for degenerate inputs
        bounds[k] = max(bounds[k], bounds[k - 1])
    return bounds


class RowPartition:
    def __init__(self, bounds):
        self.bounds = list(bounds)
        self.world = len(bounds) - 1
        self.n = bounds[-1]
        self.R = max(1, max(bounds[k + 1] - bounds[k] for k in range(self.world)))
        # round the shard height up so that every shard base stays 16-byte aligned for any width
        self.R = (self.R + 3) // 4 * 4

    def rows(self, rank):
        return self.bounds[rank + 1] - self.bounds[rank]

    def padded_index(self, ids):
        """Global node ids (int tensor) -> rows of the padded [world*R, *] tables."""
        b = torch.tensor(self.bounds, dtype=torch.int64, device=ids.device)
        ids = ids.to(torch.int64)
        owner = torch.searchsorted(b, ids, right=True) - 1
        owner.clamp_(0, self.world - 1)
        return owner * self.R + (ids - b[owner])

    def pad_table(self, table):
        """[n, C] table in global order -> [world*R, C] padded layout (padding rows zero)."""
        out = torch.zeros((self.world * self.R, table.shape[1]), dtype=table.dtype, device=table.device)
        idx = self.padded_index(torch.arange(self.n, device=table.device))
        out[idx] = table
        return out

    def local_csr(self, a, rank):
        """Rows [b_r, b_{r+1}) of `a` with column indices remapped to the padded space."""
        lo, hi = self.bounds[rank], self.bounds[rank + 1]
        rp = a.rowptr[lo:hi + 1].to(torch.int64)
        p0, p1 = int(rp[0]), int(rp[-1])
        colidx = self.padded_index(a.colidx[p0:p1]).to(torch.int32).contiguous()
        vals = a.vals[p0:p1].contiguous() if a.vals is not None else None
        local = DeviceCSR((rp - p0).to(torch.int32).contiguous(), colidx, vals, (hi - lo, self.world * self.R),
                          gcn_filtered=a.gcn_filtered)
        local.diag_offset = rank * self.R                            # padded column of local row 0's own entry
        if getattr(a, 'dinv', None) is not None and getattr(a, 'mult', None) is not None:
            local.dinv = self.pad_table(a.dinv.view(-1, 1)).view(-1).contiguous()      # over the padded columns
            local.mult = a.mult[p0:p1].contiguous()
        return local


class SharedDeviceCollectives:
    """`torch.distributed` stand-in for REHEARSING several ranks on ONE GPU (AMAR_REHEARSE_ONE_GPU=1: a `gloo` process group, every
    rank a process of its own on device 0).  RCCL refuses two ranks on one device and gloo moves GPU tensors for all_reduce /
    broadcast only, so the per-layer exchange is an all_reduce of the zero-padded table — the same bytes in the same layout as
    `all_gather_into_tensor` (x + 0 is exact).  It exercises everything of a multi-rank run except RCCL itself: the launcher, the
    row partition and the padded layout per process, pair sharding, the per-rank images and pair plans, the rank-0 report."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def all_gather_into_tensor(self, out, inp):
        rows = inp.shape[0]
        out.zero_()
        out[self.rank * rows:(self.rank + 1) * rows] = inp
        torch.distributed.all_reduce(out)


class SingleRunner:
    """world == 1: the model's own path, no padding, no exchange."""

    def __init__(self, model, u_ids, i_ids):
        self.model, self.u_ids, self.i_ids = model, u_ids, i_ids
        a = model.gnn.gnn_layers.adj_matrix
        self.local_rows, self.local_nnz = a.shape[0], a.nnz
        self._prop_ms = None
        # the pair list does not change between steps: prepared once in XCD-affine item ranges (models/basic.py:PairPlan)
        self.pair_plan = None
        if hasattr(model.rs, 'unet') and os.environ.get('AMAR_PAIR_PLAN', '1') != '0' and u_ids.numel() >= (1 << 16):
            from deep_cbrs_amar_renaissance_amd.models.basic import PairPlan
            self.pair_plan = PairPlan(u_ids, i_ids)

    def step(self):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        emb = self.model.gnn(None)
        e1.record()
        self._events = (e0, e1)
        return self._score(emb)

    def step_graphed(self):
        """The same step replayed from a hipGraph (captured on first use, after one eager step): the launch gaps of the eight
        kernels go away.  The graph is valid while the weights keep their storage and the Dense weights their values (packed
        blobs are made on the host): call it between weight updates only, as bench.py does."""
        state = self.__dict__.setdefault('_graph_state', {})
        key = self.model.weights_version
        if state.get('key') != key:
            self.step()
            from deep_cbrs_amar_renaissance_amd.engine import capture_graph

            def body():
                emb = self.model.gnn(None)
                return self._score(emb)
            state['graph'], state['out'] = capture_graph(body)
            state['key'] = key
        state['graph'].replay()
        return state['out']

    def _score(self, emb):
        # per-entity towers, then gather + classifier per pair (nothing is cached across steps)
        nu, ni = self.model.n_users, self.model.n_items
        kw = {'pair_plan': self.pair_plan} if self.pair_plan is not None else {}
        if nu is None or ni is None:
            return self.model.rs.score_towers(self.model.rs.towers(emb, emb), self.u_ids, self.i_ids, **kw)
        return self.model.rs.score_towers(self.model.rs.towers(emb[:nu], emb[nu:nu + ni]), self.u_ids, self.i_ids, 0, nu, **kw)

    def last_propagation_ms(self):
        e0, e1 = self._events
        e1.synchronize()
        return e0.elapsed_time(e1)

    def describe(self):
        return 'single GPU'


class PartitionedGCNRunner:
    """Basic* / HybridBert* models with a GCN ('concatenation') or LightGCN ('mean') stack over `world` ranks:
    row-range SpMM + per-layer all-gather, per-entity towers (item tower replicated, user tower over the rank's own user
    range), scoring of the pairs of that user range."""

    def __init__(self, model, u_ids, i_ids, rank, world, ops=capi, dist=None, timing=True):
        self.ops, self.rank, self.world, self.timing = ops, rank, world, timing
        self.dist = dist if dist is not None else torch.distributed
        if not hasattr(model.gnn, 'gnn_layers'):
            raise NotImplementedError("the partitioned runner covers single-graph models; TwoStep / TwoWay stacks run on one GPU")
        seq = model.gnn.gnn_layers
        layers = list(seq.seq_layers)
        if layers and all(isinstance(l, GCNConv) for l in layers) and seq.final_node == 'concatenation':
            self.kind = 'gcn'
        elif layers and all(isinstance(l, LightGCNConv) for l in layers) and seq.final_node == 'mean':
            self.kind = 'lightgcn'
        elif layers and all(isinstance(l, DGCFConv) for l in layers) and seq.final_node == 'mean':
            self.kind = 'dgcf'
        elif layers and all(isinstance(l, GraphSageConv) for l in layers) and seq.final_node == 'concatenation':
            self.kind = 'sage'
        elif layers and all(isinstance(l, GATConv) for l in layers) and seq.final_node == 'concatenation' and \
                all(l.channels in (8, 16, 32) for l in layers):
            self.kind = 'gat'
        else:
            raise NotImplementedError("the partitioned runner covers GCN / GraphSAGE / GAT (8, 16 or 32 channels) stacks with "
                                      "'concatenation' and LightGCN / DGCF stacks ('mean')")
        if self.kind in ('sage', 'gat') and ops is not capi:
            raise NotImplementedError("partitioned GraphSAGE / GAT run on the XCD-sliced HIP kernels only")
        self.hybrid = hasattr(model.rs, 'dense1a')
        self.model, self.seq = model, seq
        a = seq.adj_matrix
        self.part = RowPartition(partition_rows_by_nnz(a.rowptr, world))
        self.csr = self.part.local_csr(a, rank)
        self.local_rows, self.local_nnz = self.csr.shape[0], self.csr.nnz
        self.widths = seq.layer_widths()
        # replicated node table in the padded layout (rebuilt when the weights change)
        self._x0_version, self._x0p = None, None
        self._events = None
        p = int(u_ids.numel())
        known = getattr(model, 'n_users', None) is not None and getattr(model, 'n_items', None) is not None
        if known and world > 1 and os.environ.get('AMAR_PAIR_SHARDING', 'user') == 'user':
            # Pairs sharded BY USER RANGE (equal pair counts): the user tower of a rank then only covers its own 1/world of the
            # users (the item tower stays replicated), instead of every rank running both towers over all entities — at 8 ranks
            # the replicated towers are ~15 % of the step.  `pair_index` = positions of the rank's pairs in the caller's list;
            # inside the shard the order is re-shuffled (seeded), so no gather locality is bought by the sort.
            order = torch.argsort(u_ids.to(torch.int64), stable=True)
            lo, hi = p * rank // world, p * (rank + 1) // world
            mine = order[lo:hi]
            gen = torch.Generator(device=mine.device)
            gen.manual_seed(1234 + rank)
            mine = mine[torch.randperm(mine.numel(), device=mine.device, generator=gen)]
            self.pair_index, self.pair_range = mine, None
        else:
            # this rank's contiguous slice of the pair list
            lo, hi = p * rank // world, p * (rank + 1) // world
            self.pair_index = torch.arange(lo, hi, device=u_ids.device)
            self.pair_range = (lo, hi)
        my_u, my_i = u_ids[self.pair_index], i_ids[self.pair_index]
        self.u_ids = self.part.padded_index(my_u).to(torch.int32).contiguous()      # ids moved to the padded space
        self.i_ids = self.part.padded_index(my_i).to(torch.int32).contiguous()
        # users are the first n_users global ids, items the next n_items (loaders.py:43-56): in the padded layout they
        # occupy two row ranges that overlap by at most one rank's block, so each tower runs on its own range only
        if not known:
            self.u_rows = self.i_rows = (0, self.world * self.part.R)        # unknown split: both towers over every row
        else:
            nu, ni = int(model.n_users), int(model.n_items)
            last = torch.tensor([nu - 1, nu, nu + ni - 1], device=u_ids.device)
            pu_end, pi_beg, pi_end = [int(v) for v in self.part.padded_index(last).cpu()]
            self.u_rows, self.i_rows = (0, pu_end + 1), (pi_beg, pi_end + 1)
            if self.pair_range is None and self.u_ids.numel():
                self.u_rows = (int(self.u_ids.min()), int(self.u_ids.max()) + 1)   # the user rows this shard touches
        self._bert_version, self._bert_pad = None, None
        self.pair_plan = None
        if ops is capi and not self.hybrid and os.environ.get('AMAR_PAIR_PLAN', '1') != '0' and self.u_ids.numel() >= (1 << 16):
            from deep_cbrs_amar_renaissance_amd.models.basic import PairPlan
            self.pair_plan = PairPlan(self.u_ids, self.i_ids)          # the rank's pairs, fixed for the runner's lifetime

    def _x0_padded(self):
        emb = self.seq.embeddings
        if self._x0_version != emb._version:
            self._x0p, self._x0_version = self.part.pad_table(emb.detach()), emb._version
        return self._x0p

    def propagate(self):
        """Returns the [world*R, F_cat] table of final node representations (padded layout).

        Exchange per layer: ONE all-gather of the layer's own output block [R, C_l] — it is needed for the final
        table anyway — after which every rank recomputes the next layer's tiny dense product X_l . W_{l+1} for all
        rows (replicated weights, ~N*C*C flops) instead of gathering a second [N, C] block.  Bytes on xGMI per
        propagation: N * sum(C_l) * 4 (37.8 MB at s=64), half of what fusing X.W into the SpMM epilogue would move.
        """
        ops, R, dev = self.ops, self.part.R, self.seq.embeddings.device
        layers, widths = list(self.seq.seq_layers), self.widths
        rows = self.local_rows
        x0p = self._x0_padded()
        if self.kind in ('lightgcn', 'dgcf'):
            # X_{l+1} = A X_l on the local rows (DGCF: A_dgcf (X_l * sigmoid(w_l)), the gate applied to the whole replicated
            # table), gathered; the mean over layers accumulates on the full table
            acc = x0p.clone()
            x = x0p
            for k, layer in enumerate(layers):
                y_local = torch.zeros((R, widths[0]), dtype=torch.float32, device=dev)
                if self.kind == 'dgcf':
                    gated = torch.empty_like(x)
                    ops.locality_scale(x, self._gate_padded(k, layer), gated)
                    x = gated
                ops.spmm_csr(self.csr.rowptr, self.csr.colidx, self.csr.vals, x, y_local[:rows])
                x = torch.empty((self.world * R, widths[0]), dtype=torch.float32, device=dev)
                self.dist.all_gather_into_tensor(x, y_local)
                ops.add_inplace(acc, x)
            out = torch.empty_like(acc)
            ops.row_affine(acc, self._mean_scale(acc.shape[0], len(layers) + 1, dev), out)
            return out
        f_cat = sum(widths)
        offs = np.cumsum([0] + widths)
        e_all = self._buffer(('e',), (self.world * R, f_cat)) if self.kind == 'gcn' else \
            torch.empty((self.world * R, f_cat), dtype=torch.float32, device=dev)
        if self.kind in ('sage', 'gat'):
            ops.copy_columns(x0p, e_all[:, :widths[0]])                     # X_0 is a replicated weight
            # the rank's row block on the XCD-sliced forms (amar_spmm_xs_f32 mean aggregate / amar_gat_xs_f32): both take a
            # block whose own rows sit at column offset rank * R of the replicated table
            lo = self.rank * R
            x_full = x0p
            for k, layer in enumerate(layers):
                f, c = widths[k], widths[k + 1]
                y_local = torch.zeros((R, c), dtype=torch.float32, device=dev)
                if self.kind == 'sage':
                    agg = torch.empty((rows, f), dtype=torch.float32, device=dev)
                    ops.spmm_xs(self.csr.tiled_mean_image(f, layer.self_loops), x_full, agg, prescaled=True)
                    if ops.sage_tail_supported(f, c):
                        ops.sage_tail(x_full[lo:lo + rows], agg, layer.kernel, layer.bias, y_local[:rows])
                    else:
                        xa = torch.empty((rows, 2 * f), dtype=torch.float32, device=dev)
                        ops.copy_columns(x_full[lo:lo + rows], xa[:, :f])
                        ops.copy_columns(agg, xa[:, f:])
                        z = torch.empty((rows, c), dtype=torch.float32, device=dev)
                        ops.dense(xa, layer.kernel, layer.bias, z, act=None)
                        nrm, inv = torch.empty_like(z), torch.empty(rows, dtype=torch.float32, device=dev)
                        ops.l2norm_fwd(z, nrm, inv, y_local[:rows], act='relu')
                else:
                    h = torch.empty((self.world * R, c), dtype=torch.float32, device=dev)
                    s_self = torch.empty(self.world * R, dtype=torch.float32, device=dev)
                    s_neigh = torch.empty(self.world * R, dtype=torch.float32, device=dev)
                    ops.rowwise_xw(x_full, layer.kernel.view(-1, c), h, a_self=layer.attn_kernel_self.view(c),
                                   a_neigh=layer.attn_kernel_neighs.view(c), s_self=s_self, s_neigh=s_neigh)
                    ops.gat_xs(self.csr.xcd_sliced(), h, s_self, s_neigh, layer.bias, y_local[:rows], self_loop=layer.add_self_loops)
                x_full = torch.empty((self.world * R, c), dtype=torch.float32, device=dev)
                self.dist.all_gather_into_tensor(x_full, y_local)
                ops.copy_columns(x_full, e_all[:, offs[k + 1]:offs[k + 2]])
            return e_all
        def pre_scale(width):
            # value-free XCD-sliced image: the gathered table is pre-scaled by d^-1/2 inside the X.W launch
            if not self._use_xs(width):
                return None
            xs = self.csr.tiled_image(width)
            return xs.col_scale if xs.row_scale is not None else None

        # persistent buffers: at 8 ranks the local kernels take tens of microseconds, so per-step allocations and the
        # memsets of the padded blocks would show up next to them (the pad rows are written once, here, and never again)
        h = self._buffer(('h', 0), (self.world * R, widths[1]))
        scale = pre_scale(widths[1])
        ops.rowwise_xw(x0p, layers[0].kernel, h, copy_to=e_all[:, :widths[0]], row_scale=scale)   # X_0 slice rides along
        for k, layer in enumerate(layers):
            y_local = self._buffer(('y', k), (R, widths[k + 1]), zero=True)
            if self._use_xs(widths[k + 1]):
                # the rank's row block on the XCD-sliced image (value-free when A_hat's factors are known): same kernels
                # as the single-GPU path, the block's own rows sit at column offset rank * R of the padded table
                ops.spmm_xs(self.csr.tiled_image(widths[k + 1]), h, y_local[:rows], bias=layer.bias, relu=True, prescaled=scale is not None)
            else:
                ops.gcn_layer(self.csr.rowptr, self.csr.colidx, self.csr.vals, h, layer.bias, y_local[:rows])
            x_full = self._buffer(('x', k), (self.world * R, widths[k + 1]))
            self.dist.all_gather_into_tensor(x_full, y_local)
            if k + 1 < len(layers):
                # the gathered block's copy into its slice of the final table rides on the next layer's X.W launch
                h = self._buffer(('h', k + 1), (self.world * R, widths[k + 2]))
                scale = pre_scale(widths[k + 2])
                ops.rowwise_xw(x_full, layers[k + 1].kernel, h, copy_to=e_all[:, offs[k + 1]:offs[k + 2]], row_scale=scale)
            else:
                ops.copy_columns(x_full, e_all[:, offs[k + 1]:offs[k + 2]])
        return e_all

    def _buffer(self, key, shape, zero=False):
        """A per-runner float32 device buffer, allocated (and zeroed, if asked) on first use and then reused every step."""
        cache = self.__dict__.setdefault('_buffers', {})
        buf = cache.get(key)
        if buf is None or tuple(buf.shape) != tuple(shape):
            dev = self.seq.embeddings.device
            buf = cache[key] = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=torch.float32, device=dev)
        return buf

    def step_graphed(self):
        """The same step replayed from a hipGraph, collectives included (at 8 ranks a step is ~0.4 ms of device work behind
        ~0.3 ms of host-side enqueueing: replayed, the host side is one call).  Captured on first use, after one eager step.
        Rehearsed by the builder with one rank only (RCCL all-gather inside the capture: profiles/r2_partitioned_1rank.txt)."""
        state = self.__dict__.setdefault('_graph_state', {})
        if 'graph' not in state:
            self.step()                                              # eager once: lazy image builds, persistent buffers
            timing, self.timing = self.timing, False                 # no event records inside a capture
            from deep_cbrs_amar_renaissance_amd.engine import capture_graph
            g, out = capture_graph(self.step)
            self.timing = timing
            state['graph'], state['out'] = g, out
        state['graph'].replay()
        return state['out']

    def step(self):
        if self.timing:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        emb = self.propagate()
        if self.timing:
            e1.record()
            self._events = (e0, e1)
        # replicated per-entity towers, each over its own row range of the padded table
        (u0, u1), (i0, i1) = self.u_rows, self.i_rows
        if self.hybrid:
            bert = self._bert_padded()
            towers = self.model.rs.towers(emb[u0:u1], emb[i0:i1], bert[u0:u1], bert[i0:i1])
        else:
            towers = self.model.rs.towers(emb[u0:u1], emb[i0:i1])
        if self.pair_plan is not None:
            return self.model.rs.score_towers(towers, self.u_ids, self.i_ids, u0, i0, pair_plan=self.pair_plan)
        return self.model.rs.score_towers(towers, self.u_ids, self.i_ids, u0, i0)

    def _use_xs(self, width):
        """XCD-sliced local SpMM when the gathered table exceeds the per-XCD L2s (as utilities.math.spmm_kind decides
        for the single-GPU path); AMAR_SPMM_KIND=csr|xs overrides."""
        if self.ops is not capi or not hasattr(self.csr, 'diag_offset') or width > 16 or width % 4:
            return False
        forced = os.environ.get('AMAR_SPMM_KIND')
        if forced in ('csr', 'xs'):
            return forced == 'xs'
        return self.world * self.part.R * width * 4 >= ((8 << 20) if width <= 8 else (16 << 20))

    def _gate_padded(self, k, layer):
        """DGCF's per-node gate weights of layer k in the padded layout (rebuilt when they change)."""
        cache = self.__dict__.setdefault('_gates', {})
        version = layer.w._version
        if cache.get(k, (None, None))[0] != version:
            cache[k] = (version, self.part.pad_table(layer.w.detach().view(-1, 1)).view(-1).contiguous())
        return cache[k][1]

    def _mean_scale(self, n_rows, n_terms, dev):
        if getattr(self, '_mean', None) is None or self._mean.numel() != n_rows:
            self._mean = torch.full((n_rows,), 1.0 / n_terms, dtype=torch.float32, device=dev)
        return self._mean

    def _bert_padded(self):
        """The resident BERT table (rows = users then items) in the padded layout; property nodes get zero rows."""
        table = self.model.bert_table
        if table is None:
            raise ValueError("the hybrid model needs its BERT table registered (set_bert_table) for the partitioned run")
        key = (table.data_ptr(), table._version)
        if self._bert_version != key:
            full = torch.zeros((self.part.n, table.shape[1]), dtype=torch.float32, device=table.device)
            full[:min(self.part.n, table.shape[0])] = table[:self.part.n]
            self._bert_pad, self._bert_version = self.part.pad_table(full), key
            if not self.model.rs.built:
                self.model.rs.build_head(self.model.gnn.output_dim(), table.shape[1])
        return self._bert_pad

    def last_propagation_ms(self):
        if not self._events:
            return None
        e0, e1 = self._events
        e1.synchronize()
        return e0.elapsed_time(e1)

    def describe(self):
        return 'node-range partition over {} GPUs (equal nnz), per-layer RCCL all-gather, pairs sharded {}'.format(
            self.world, 'by user range (equal counts)' if self.pair_range is None else 'in contiguous slices')


def make_runner(model, u_ids, i_ids, rank=0, world=1, dist=None):
    if world == 1:
        return SingleRunner(model, u_ids, i_ids)
    return PartitionedGCNRunner(model, u_ids, i_ids, rank, world, dist=dist)
