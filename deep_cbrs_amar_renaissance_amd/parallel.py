"""Multi-GPU execution of the hot path: node-range partition + per-layer all-gather.

The reference is single-device; this layer is new design.  One process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  The graph propagation Y = A_hat.X
is row-separable, the scoring head is pair-separable.

GCN stacks of models that know their user / item split run on the TYPED partition (round 3, `TypedPartition` and
`PartitionedGCNRunner._step_typed`): every node type is cut into `world` blocks of equal height, a rank owns one block of
each type, and nothing in a rank's step passes over the whole node table except the X_0 . W_1 prologue:

* the fused SpMM layer leaves BOTH its own rows of X_l and its own rows of the next layer's pre-scaled gathered table
  H_{l+1} = S (X_l W_{l+1}) (epilogue of amar_spmm_lt_f32 / amar_spmm_xs_f32): the exchange per layer is one all-gather of the
  H_{l+1} block (every rank's SpMM gathers from all of it) plus one all-gather of the ITEM rows of X_l (all the item tower
  reads), issued asynchronously so that the item-row gathers hide behind the next layer's SpMM and the user tower;
* the towers read [X_0 || X_1 || ... ] in place from the per-layer tables (capi.ConcatTable -> amar_chain_segments_f32): the
  user tower over the rank's own users only (pairs are sharded by the SAME user ranges), the item tower over all items;
* ids stay the reference's ids: the only remapped index space is the column index of the local CSR block.

The other layer kinds (and models without a known user / item split) keep the round-1/2 scheme below:

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

class TypedPartition:
    """Node-range partition of a graph whose ids are grouped by node TYPE (users | items [| properties], loaders.py:43-68).

    Type t with n_t nodes is cut into `world` contiguous blocks of EQUAL HEIGHT h_t = ceil(n_t / world) (only a type's last
    blocks can be short or empty); rank r owns block r of every type.  A rank's rows, in local order, are
    [its users | its items | its properties], each padded to h_t rows: R = sum(h_t) rows per rank, and the gathered tables the
    SpMM reads from are RANK-MAJOR, [world * R, C] — `all_gather_into_tensor` of the ranks' [R, C] blocks lands in place.
    Node j of type t sits at row  (j' // h_t) * R + off_t + j' % h_t,  j' = j - first id of type t.
    Because a type's blocks are equally tall, its rows taken out of the ranks' blocks in rank order ARE the type in id order
    (plus padding at the very end): the all-gather of the item parts of the blocks is the item table in the reference's own
    item order, and a rank's user rows are a contiguous range of user ids — towers and pair ids need no remapping.
    Against equal-nnz row ranges (RowPartition) the padded index space is N + O(world) rows instead of 1.3 N at ml1m(s=64);
    the price is that non-zeros are balanced only as far as degrees are unrelated to id order (`nnz_imbalance`)."""

    def __init__(self, type_bounds, world):
        tb = [int(b) for b in type_bounds]
        if len(tb) < 2 or tb[0] != 0 or any(tb[k] > tb[k + 1] for k in range(len(tb) - 1)):
            raise ValueError("type_bounds must be an ascending list starting at 0")
        self.tb, self.world, self.n, self.T = tb, int(world), tb[-1], len(tb) - 1
        self.h = [max(1, -(-(tb[t + 1] - tb[t]) // self.world)) for t in range(self.T)]
        self.off = [0]
        for h in self.h:
            self.off.append(self.off[-1] + h)
        self.R = self.off[-1]

    def owned(self, rank, t):
        """[lo, hi) of the ids of type t that rank owns (empty when the type ran out before this rank's block)."""
        lo = min(self.tb[t] + rank * self.h[t], self.tb[t + 1])
        return lo, min(lo + self.h[t], self.tb[t + 1])

    def padded_index(self, ids):
        """Global node ids (int tensor) -> rows of the rank-major [world * R, *] tables."""
        ids = ids.to(torch.int64)
        dev = ids.device
        tb = torch.tensor(self.tb, dtype=torch.int64, device=dev)
        t = (torch.searchsorted(tb, ids, right=True) - 1).clamp_(0, self.T - 1)
        h = torch.tensor(self.h, dtype=torch.int64, device=dev)[t]
        j = ids - tb[t]
        return (j // h) * self.R + torch.tensor(self.off[:-1], dtype=torch.int64, device=dev)[t] + j % h

    def node_of_row(self, device):
        """int32 [world * R]: the node id held by every row of the rank-major tables, -1 for padding rows."""
        out = torch.full((self.world * self.R,), -1, dtype=torch.int32, device=device)
        ids = torch.arange(self.n, device=device)
        out[self.padded_index(ids)] = ids.to(torch.int32)
        return out

    def pad_vector(self, v):
        """[n] per-node vector -> [world * R] in the rank-major layout (padding rows zero)."""
        out = torch.zeros(self.world * self.R, dtype=v.dtype, device=v.device)
        out[self.padded_index(torch.arange(self.n, device=v.device))] = v
        return out

    def local_block(self, a, rank):
        """The rank's rows of `a` (a square DeviceCSR over the n nodes) as an [R, world * R] DeviceCSR in local row order, column
        indices in the rank-major layout; carries what the tiled images need of a row block: `diag_offset` (column of local row 0's
        own entry), `row_breaks` (local rows where the node type changes), and the value-free factors when `a` has them."""
        dev = a.rowptr.device
        rp = a.rowptr.to(torch.int64)
        deg = torch.zeros(self.R, dtype=torch.int64, device=dev)
        cols, vals, mult = [], [], []
        for t in range(self.T):
            lo, hi = self.owned(rank, t)
            if hi <= lo:
                continue
            p0, p1 = int(rp[lo]), int(rp[hi])
            deg[self.off[t]:self.off[t] + hi - lo] = rp[lo + 1:hi + 1] - rp[lo:hi]
            cols.append(self.padded_index(a.colidx[p0:p1]).to(torch.int32))
            if a.vals is not None:
                vals.append(a.vals[p0:p1])
            if getattr(a, 'mult', None) is not None:
                mult.append(a.mult[p0:p1])
        rowptr = torch.zeros(self.R + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(deg, 0)
        empty = torch.zeros(0, dtype=torch.int32, device=dev)
        local = DeviceCSR(rowptr.to(torch.int32).contiguous(), torch.cat(cols).contiguous() if cols else empty,
                          (torch.cat(vals).contiguous() if vals else torch.zeros(0, dtype=torch.float32, device=dev)) if a.vals is not None else None,
                          (self.R, self.world * self.R), gcn_filtered=a.gcn_filtered)
        local.diag_offset = rank * self.R
        local.row_breaks = tuple(self.off[1:-1])
        if getattr(a, 'dinv', None) is not None and getattr(a, 'mult', None) is not None:
            local.dinv = self.pad_vector(a.dinv).contiguous()
            local.mult = torch.cat(mult).contiguous() if mult else empty
        return local

    def nnz_imbalance(self, rowptr):
        """max over ranks of the rank's non-zero count / the mean: 1.0 = perfectly balanced."""
        rp = rowptr.to(torch.int64).cpu()
        per = [sum(int(rp[hi] - rp[lo]) for lo, hi in (self.owned(r, t) for t in range(self.T))) for r in range(self.world)]
        return max(per) * self.world / max(1, sum(per))


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
        state = self.capture_step()
        state['graph'].replay()
        return state['out']

    def capture_step(self):
        state = self.__dict__.setdefault('_graph_state', {})
        key = self.model.weights_version
        if state.get('key') != key:
            state.clear()
            self.step()
            from deep_cbrs_amar_renaissance_amd.engine import capture_graph

            def body():
                emb = self.model.gnn(None)
                return self._score(emb)
            state['graph'], state['out'] = capture_graph(body)
            state['key'] = key
        return state

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
        self._events, self.phase_ms = None, None
        known_types = getattr(model, 'n_users', None) is not None and getattr(model, 'n_items', None) is not None
        # GCN stacks with a known user / item split: equal-height blocks per node type, no pass over the whole table but the
        # X_0 . W_1 prologue (AMAR_PARTITION=rows keeps the equal-nnz row ranges of rounds 1-2 for A/B runs)
        self.typed = self.kind == 'gcn' and known_types and os.environ.get('AMAR_PARTITION', 'typed') != 'rows'
        if self.typed:
            self._init_typed(model, u_ids, i_ids, a)
            return
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

    # ---- typed partition (GCN stacks) -------------------------------------------------------------------------------------
    def _init_typed(self, model, u_ids, i_ids, a):
        nu, ni, n = int(model.n_users), int(model.n_items), int(a.shape[0])
        if nu + ni > n:
            raise ValueError("n_users + n_items exceeds the graph's node count")
        self.part = self.tpart = TypedPartition([0, nu, nu + ni] + ([n] if n > nu + ni else []), self.world)
        self.csr = self.tpart.local_block(a, self.rank)
        self.local_rows = sum(hi - lo for lo, hi in (self.tpart.owned(self.rank, t) for t in range(self.tpart.T)))
        self.local_nnz = self.csr.nnz
        self.nnz_imbalance = self.tpart.nnz_imbalance(a.rowptr)
        self.widths = self.seq.layer_widths()
        dev = u_ids.device
        self.row_ids = self.tpart.node_of_row(dev)
        # pairs follow their user: the rank scores the pairs of the users it owns, so its user tower reads its own rows only
        self.u_lo, self.u_hi = self.tpart.owned(self.rank, 0)
        self.i_lo, self.n_items = nu, ni
        mine = torch.nonzero((u_ids >= self.u_lo) & (u_ids < self.u_hi)).view(-1)
        self.pair_index, self.pair_range = mine, None
        self.u_ids = u_ids[mine].to(torch.int32).contiguous()            # the reference's ids, unchanged
        self.i_ids = i_ids[mine].to(torch.int32).contiguous()
        self.pair_plan = None
        if self.ops is capi and not self.hybrid and os.environ.get('AMAR_PAIR_PLAN', '1') != '0' and self.u_ids.numel() >= (1 << 16):
            from deep_cbrs_amar_renaissance_amd.models.basic import PairPlan
            self.pair_plan = PairPlan(self.u_ids, self.i_ids)
        # item-row gathers behind the next kernels: only with a real process group (stand-ins copy synchronously)
        self.async_exchange = self.dist is torch.distributed and os.environ.get('AMAR_EXCHANGE_ASYNC', '1') != '0'

    def _gather(self, out, inp, defer=False):
        """all_gather_into_tensor of equal blocks; defer=True: issued on the collective's own stream, returns the handle to wait on."""
        if defer and self.async_exchange:
            return self.dist.all_gather_into_tensor(out, inp, async_op=True)
        self.dist.all_gather_into_tensor(out, inp)
        return None

    def _mark(self, name):
        if self.timing:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._marks.append((name, e))

    def propagate_typed(self):
        """One propagation on the typed partition.  Returns (x_local, x_items): per layer l = 1..L the rank's own [R, C_l] block of
        X_l (local row order: users | items | properties, each padded to its block height) and the all-gathered item rows
        [world * h_items, C_l] (the reference's item order; rows past n_items are padding).  X_0 is the node table itself."""
        ops, tp, dev = self.ops, self.tpart, self.seq.embeddings.device
        layers, widths = list(self.seq.seq_layers), self.widths
        n_tab, R = self.world * tp.R, tp.R
        emb = self.seq.embeddings.detach()
        tiled = [self._use_xs(w) for w in widths[1:]]
        images = [self.csr.tiled_image(w) if t else None for w, t in zip(widths[1:], tiled)]
        pre = all(tiled) and all(im.row_scale is not None for im in images)       # the chain of gathered tables stays pre-scaled by d^-1/2
        h = self._buffer(('h', 0), (n_tab, widths[1]))
        ops.rowwise_xw(emb, layers[0].kernel, h, row_ids=self.row_ids, row_scale=images[0].col_scale if pre else None)
        self._mark('prologue')
        i0, hi_ = tp.off[1], tp.h[1]
        x_local, x_items, pending = [], [], []
        for k, layer in enumerate(layers):
            nxt = layers[k + 1] if k + 1 < len(layers) else None
            y = self._buffer(('y', k), (R, widths[k + 1]), zero=True)
            hn = self._buffer(('hl', k + 1), (R, widths[k + 2]), zero=True) if nxt is not None else None
            if tiled[k]:
                ops.spmm_xs(images[k], h, y, bias=layer.bias, relu=True, Wnext=nxt.kernel if nxt is not None else None, Hnext=hn,
                            prescaled=pre, scale_next=pre and nxt is not None)
            else:
                ops.gcn_layer(self.csr.rowptr, self.csr.colidx, self.csr.vals, h, layer.bias, y,
                              Wnext=nxt.kernel if nxt is not None else None, Hnext=hn)
            self._mark('spmm')
            wait_h = None
            if nxt is not None:
                h = self._buffer(('h', k + 1), (n_tab, widths[k + 2]))
                wait_h = self._gather(h, hn, defer=True)
            xi = self._buffer(('xi', k), (self.world * hi_, widths[k + 1]))
            pending.append(self._gather(xi, y[i0:i0 + hi_], defer=True))
            if wait_h is not None:
                wait_h.wait()
            self._mark('exchange')
            x_local.append(y)
            x_items.append(xi)
        self._pending = [w for w in pending if w is not None]
        return x_local, x_items

    def wait_exchange(self):
        """Make the compute stream wait for the item-row gathers still in flight (a no-op with synchronous collectives)."""
        for w in getattr(self, '_pending', []):
            w.wait()
        self._pending = []

    def _item_bert_sharded(self):
        """Hybrid heads: the item-side BERT tower (768 -> 256 -> 64 at econfigs/hybrid-gnn*.yaml: 0.8 ms over all items of ml1m(s=64))
        on the rank's OWN items only, its [h_items, D] block gathered — issued first thing in the step on the collective's stream: it
        depends on no layer and hides behind the whole propagation.  Returns (handle or None, full table view [n_items, D])."""
        rs, tp = self.model.rs, self.tpart
        bert = self.model.bert_table
        lo, hi = tp.owned(self.rank, 1)
        hi_ = tp.h[1]
        part = rs.item_bert_part(bert[lo:hi]) if hi > lo else None
        width = int(part.shape[1]) if part is not None else int(rs.item_bert_part(bert[self.i_lo:self.i_lo + 1]).shape[1])
        blk = self._buffer(('ibl', width), (hi_, width), zero=True)
        if part is not None:
            self.ops.copy_columns(part, blk[:hi - lo])
        full = self._buffer(('ibf', width), (self.world * hi_, width))
        return self._gather(full, blk, defer=True), full[:self.n_items]

    def _step_typed(self):
        self._marks = []
        self._mark('start')
        ib_wait = ib_full = None
        if self.hybrid and os.environ.get('AMAR_HYBRID_ITEM_TOWER', 'sharded') == 'sharded':      # (every rank takes part: it is a collective)
            if self.model.bert_table is None:
                raise ValueError("the hybrid model needs its BERT table registered (set_bert_table) for the partitioned run")
            if not self.model.rs.built:
                self.model.rs.build_head(self.model.gnn.output_dim(), self.model.bert_table.shape[1])
            ib_wait, ib_full = self._item_bert_sharded()
            self._mark('item_bert')
        x_local, x_items = self.propagate_typed()
        emb = self.seq.embeddings.detach()
        nu_loc = self.u_hi - self.u_lo
        u_table = capi.ConcatTable([emb[self.u_lo:self.u_hi]] + [x[:nu_loc] for x in x_local])
        i_table = capi.ConcatTable([emb[self.i_lo:self.i_lo + self.n_items]] + [x[:self.n_items] for x in x_items])
        rs = self.model.rs
        if self.u_ids.numel() == 0:                                    # a rank without users (more ranks than user blocks): nothing to score
            self.wait_exchange()
            if ib_wait is not None:
                ib_wait.wait()
            return torch.empty((0, 1), dtype=torch.float32, device=emb.device)
        # the user tower runs first: it reads the rank's own rows only and hides the last item-row gather
        if self.hybrid:
            bert = self.model.bert_table
            if bert is None:
                raise ValueError("the hybrid model needs its BERT table registered (set_bert_table) for the partitioned run")
            if not rs.built:
                rs.build_head(self.model.gnn.output_dim(), bert.shape[1])
            self.wait_exchange()
            if ib_wait is not None:
                ib_wait.wait()
            towers = rs.towers(u_table, i_table, bert[self.u_lo:self.u_hi], bert[self.i_lo:self.i_lo + self.n_items], ib_done=ib_full)
            self._mark('towers')
            out = rs.score_towers(towers, self.u_ids, self.i_ids, self.u_lo, self.i_lo)
        else:
            tu = rs.tower('u', u_table)
            self._mark('user_tower')
            self.wait_exchange()
            self._mark('exchange')
            ti = rs.tower('i', i_table)
            self._mark('item_tower')
            kw = {'pair_plan': self.pair_plan} if self.pair_plan is not None else {}
            out = rs.score_towers((tu, ti, rs.split_ready()), self.u_ids, self.i_ids, self.u_lo, self.i_lo, **kw)
        self._mark('pairs')
        return out

    def phase_times(self):
        """Milliseconds of the last eager step by phase (HIP events on the compute stream): `replicated` = work every rank repeats
        whatever the world size (X_0 . W_1 over all rows, the item tower), `local` = work that shrinks with it (SpMM blocks, user
        tower, pair stage), `exchange` = time the compute stream spent issuing and waiting for all-gathers."""
        if not getattr(self, '_marks', None):
            return None
        self._marks[-1][1].synchronize()
        out = {}
        for (_, e0), (name, e1) in zip(self._marks[:-1], self._marks[1:]):
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        spmm, pairs = out.get('spmm', 0.0), out.get('pairs', 0.0)
        return {'local_spmm_ms': spmm, 'exchange_ms': out.get('exchange', 0.0),
                'replicated_ms': out.get('prologue', 0.0) + out.get('item_tower', 0.0),
                'user_tower_ms': out.get('user_tower', 0.0), 'towers_ms': out.get('towers', 0.0) + out.get('item_bert', 0.0), 'pair_stage_ms': pairs,
                'prologue_ms': out.get('prologue', 0.0), 'item_tower_ms': out.get('item_tower', 0.0)}

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
                    # the rank's row block on the LDS-tiled walk where its density allows (amar_gat_lt_f32 takes a block whose own
                    # rows sit at a column offset, like the plain sum), else on the XCD-sliced online-softmax form
                    lt = self.csr.tiled_gat_image(c)
                    if lt is not None:
                        ops.gat_lt(lt, self.csr, h, s_self, s_neigh, layer.bias, y_local[:rows], self_loop=layer.add_self_loops)
                    else:
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
        state = self.capture_step()
        state['graph'].replay()
        return state['out']

    def capture_step(self):
        """Capture the step (no replay).  Keyed on the model's weights version: the Dense weights are packed on the host at capture
        time, so a weight update invalidates the graph.  Several ranks should agree that EVERY rank captured before any of them
        replays (a replay enqueues collectives the others must match): bench.py all-reduces a flag between the two."""
        state = self.__dict__.setdefault('_graph_state', {})
        key = self.model.weights_version
        if state.get('key') != key:
            state.clear()
            self.step()                                              # eager once: lazy image builds, persistent buffers
            timing, self.timing = self.timing, False                 # no event records inside a capture
            from deep_cbrs_amar_renaissance_amd.engine import capture_graph
            try:
                g, out = capture_graph(self.step)
            finally:
                self.timing = timing
            state['graph'], state['out'], state['key'] = g, out, key
        return state

    def step(self):
        if self.typed:
            return self._step_typed()
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
        if self.typed:
            ph = self.phase_times()
            return None if ph is None else ph['prologue_ms'] + ph['local_spmm_ms'] + ph['exchange_ms']
        if not self._events:
            return None
        e0, e1 = self._events
        e1.synchronize()
        return e0.elapsed_time(e1)

    def describe(self):
        if self.typed:
            return ('node-range partition over {} GPUs (equal-height blocks per node type, nnz imbalance {:.3f}), per layer one RCCL all-gather of '
                    'the next gathered table + one of the item rows, pairs sharded by the same user ranges').format(self.world, self.nnz_imbalance)
        return 'node-range partition over {} GPUs (equal nnz), per-layer RCCL all-gather, pairs sharded {}'.format(
            self.world, 'by user range (equal counts)' if self.pair_range is None else 'in contiguous slices')


def make_runner(model, u_ids, i_ids, rank=0, world=1, dist=None):
    if world == 1:
        return SingleRunner(model, u_ids, i_ids)
    return PartitionedGCNRunner(model, u_ids, i_ids, rank, world, dist=dist)
