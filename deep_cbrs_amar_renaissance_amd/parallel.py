"""Multi-GPU execution of the hot path: node-range partition + per-layer all-gather.

The reference is single-device; this layer is new design.  One process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  The graph propagation Y = A_hat.X
is row-separable, the scoring head is pair-separable.

`TypedPartition`: every node TYPE (users | items [| properties]; one type when the model does not know its split) is cut into
`world` blocks of equal height, a rank owns one block of each type.  `PartitionedGCNRunner` runs a step on it so that nothing
passes over the whole node table except the layer-1 prologue, and (round 4) so that the per-layer exchange hides behind compute:

* the table a layer gathers from is TYPE-MAJOR — [all users | all items | all properties], each type padded to world x its block
  height — and is made by the PRODUCER of the previous layer from its own rows, one all-gather PER NODE TYPE landing in place
  (GCN: H_{l+1} = S (X_l W_{l+1}) straight out of the fused SpMM's epilogue; LightGCN: S X_l; DGCF: X_l sigmoid(w_{l+1});
  GraphSAGE: X_l; GAT: X_l W_{l+1} with its neighbour scalars);
* a layer runs as one launch per node type (tiles never straddle a type anyway), in an order that alternates from layer to layer:
  [users, properties | items] then [items | users, properties] ...  The adjacency of a rating graph is bipartite / tripartite —
  user rows read item columns only, item rows read user and property columns, property rows item columns — so the section a
  phase produces first is gathered while the layer's other phase runs, and the next layer starts with the phase that needs
  exactly that section; the rows' own entries (the diagonal of A_hat) are read from the rank's own block, not from the table;
* the ITEM rows of every X_l (all the item tower reads) are gathered on the side, asynchronously; 'mean' stacks accumulate on
  the rank's own rows and gather the items' mean once; the last layer skips node types no tower reads (properties);
* the towers read [X_0 || X_1 || ... ] in place from the per-layer tables (capi.ConcatTable -> amar_chain_segments_f32): the
  user tower over the rank's own users only (pairs follow their user's owner), the item tower over all items; hybrid heads run
  the item-side BERT tower on the rank's own items and gather its output;
* ids stay the reference's ids: the only remapped index space is the column index of the local CSR blocks.

(Rounds 1-2 split rows into equal-nnz ranges in a padded index space, gathered every layer's own block and re-ran X.W over the
whole table on every rank; round 3 had rank-major tables, one launch per layer and an exposed all-gather between the layers:
DESIGN.md 6 keeps the numbers.)

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


class TypedPartition:
    """Node-range partition of a graph whose ids are grouped by node TYPE (users | items [| properties], loaders.py:43-68).

    Type t with n_t nodes is cut into `world` contiguous blocks of EQUAL HEIGHT h_t = ceil(n_t / world) (only a type's last
    blocks can be short or empty); rank r owns block r of every type.  Node types are bundled into GROUPS (`groups`, a list of
    lists of types; default: every type its own group).  A rank's rows, in local order, are its blocks group after group, type
    after type inside a group, each padded to h_t rows: R = sum(h_t) rows per rank.  The gathered tables the SpMM reads from,
    [world * R, C], are GROUP-MAJOR: section g = rows [goff_g, goff_g + world * R_g) holds the ranks' [R_g, C] blocks of group g
    in rank order (R_g = the group's rows per rank), so `all_gather_into_tensor` of the ranks' blocks of ONE group lands in place:
      * one group of all types = the rank-major tables of round 3 — one launch and one collective per layer;
      * one group per type (or [users, properties | items]) = type-major tables — a layer runs one launch per group and each
        group's collective can start as soon as its launch is enqueued (PartitionedGCNRunner: the exchange behind compute).
    Node j of type t (group g) sits at row  goff_g + (j' // h_t) * R_g + o_t + j' % h_t,  j' = j - first id of type t, o_t = the
    type's offset inside its group's block.
    Because a type's blocks are equally tall, its rows taken out of the ranks' blocks in rank order ARE the type in id order
    (plus padding at the very end): the all-gather of the item parts of the blocks is the item table in the reference's own
    item order, and a rank's user rows are a contiguous range of user ids — towers and pair ids need no remapping.
    Against the equal-nnz row ranges of rounds 1-2 the padded index space is N + O(world) rows instead of 1.3 N at ml1m(s=64);
    the price is that non-zeros are balanced only as far as degrees are unrelated to id order (`nnz_imbalance`)."""

    def __init__(self, type_bounds, world, groups=None):
        tb = [int(b) for b in type_bounds]
        if len(tb) < 2 or tb[0] != 0 or any(tb[k] > tb[k + 1] for k in range(len(tb) - 1)):
            raise ValueError("type_bounds must be an ascending list starting at 0")
        self.tb, self.world, self.n, self.T = tb, int(world), tb[-1], len(tb) - 1
        self.h = [max(1, -(-(tb[t + 1] - tb[t]) // self.world)) for t in range(self.T)]
        self.groups = [[t] for t in range(self.T)] if groups is None else [[int(t) for t in g] for g in groups]
        if sorted(t for g in self.groups for t in g) != list(range(self.T)):
            raise ValueError("groups must hold every node type exactly once")
        self.G = len(self.groups)
        self.group_of, self.in_group = [0] * self.T, [0] * self.T    # a type's group / its offset inside the group's block
        self.off = [0] * self.T                                       # ... its offset in the rank's local row order
        self.gh, self.goff, self.loff = [], [], []                    # per group: rows per rank, first table row, first local row
        local = 0
        for g, types in enumerate(self.groups):
            self.loff.append(local)
            inside = 0
            for t in types:
                self.group_of[t], self.in_group[t], self.off[t] = g, inside, local + inside
                inside += self.h[t]
            self.gh.append(inside)
            self.goff.append(self.world * local)
            local += inside
        self.R = local
        self.goff.append(self.world * self.R)

    def owned(self, rank, t):
        """[lo, hi) of the ids of type t that rank owns (empty when the type ran out before this rank's block)."""
        lo = min(self.tb[t] + rank * self.h[t], self.tb[t + 1])
        return lo, min(lo + self.h[t], self.tb[t + 1])

    def section(self, g):
        """[lo, hi) of the rows of group g's section in the gathered tables."""
        return self.goff[g], self.goff[g + 1]

    def block_row0(self, rank, g):
        """First row of rank's block of group g in the gathered tables."""
        return self.goff[g] + rank * self.gh[g]

    def local_rows(self, g):
        """[lo, hi) of group g's rows in the rank's local row order."""
        return self.loff[g], self.loff[g] + self.gh[g]

    def padded_index(self, ids):
        """Global node ids (int tensor) -> rows of the group-major [world * R, *] tables."""
        ids = ids.to(torch.int64)
        dev = ids.device
        as_t = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)
        t = (torch.searchsorted(as_t(self.tb), ids, right=True) - 1).clamp_(0, self.T - 1)
        j = ids - as_t(self.tb)[t]
        h = as_t(self.h)[t]
        g = as_t(self.group_of)[t]
        return as_t(self.goff[:-1])[g] + (j // h) * as_t(self.gh)[g] + as_t(self.in_group)[t] + j % h

    def node_of_row(self, device):
        """int32 [world * R]: the node id held by every row of the gathered tables, -1 for padding rows."""
        out = torch.full((self.world * self.R,), -1, dtype=torch.int32, device=device)
        ids = torch.arange(self.n, device=device)
        out[self.padded_index(ids)] = ids.to(torch.int32)
        return out

    def pad_vector(self, v):
        """[n] per-node vector -> [world * R] in the layout of the gathered tables (padding rows zero)."""
        out = torch.zeros(self.world * self.R, dtype=v.dtype, device=v.device)
        out[self.padded_index(torch.arange(self.n, device=v.device))] = v
        return out

    def group_of_row(self, rows):
        """Group (section) of rows of the gathered tables (int64 tensor -> int64 tensor)."""
        goff = torch.tensor(self.goff, dtype=torch.int64, device=rows.device)
        return (torch.searchsorted(goff, rows.to(torch.int64), right=True) - 1).clamp_(0, self.G - 1)

    def local_block(self, a, rank, g):
        """The rank's rows of group g of `a` (a square DeviceCSR over the n nodes) as an [R_g, world * R] DeviceCSR in local row
        order, column indices in the layout of the gathered tables; carries what the tiled images need of a row block:
        `diag_offset` (column of row 0's own entry: the block's rows are contiguous in the tables), `row_breaks` (rows where the
        node type changes), `active_cols` (columns of the sections it touches: what its entry density is measured against),
        `reads` (the groups its off-diagonal entries fall in), and the value-free factors when `a` has them."""
        dev = a.rowptr.device
        rp = a.rowptr.to(torch.int64)
        hg = self.gh[g]
        deg = torch.zeros(hg, dtype=torch.int64, device=dev)
        empty = torch.zeros(0, dtype=torch.int32, device=dev)
        cols, vals, mult = [], [], []
        for t in self.groups[g]:
            lo, hi = self.owned(rank, t)
            if hi <= lo:
                continue
            p0, p1 = int(rp[lo]), int(rp[hi])
            deg[self.in_group[t]:self.in_group[t] + hi - lo] = rp[lo + 1:hi + 1] - rp[lo:hi]
            cols.append(self.padded_index(a.colidx[p0:p1]).to(torch.int32))
            if a.vals is not None:
                vals.append(a.vals[p0:p1])
            if getattr(a, 'mult', None) is not None:
                mult.append(a.mult[p0:p1])
        rowptr = torch.zeros(hg + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(deg, 0)
        colidx = torch.cat(cols).contiguous() if cols else empty
        local = DeviceCSR(rowptr.to(torch.int32).contiguous(), colidx,
                          (torch.cat(vals).contiguous() if vals else torch.zeros(0, dtype=torch.float32, device=dev)) if a.vals is not None else None,
                          (hg, self.world * self.R), gcn_filtered=a.gcn_filtered)
        local.diag_offset = self.block_row0(rank, g)
        local.row_breaks = tuple(self.in_group[t] for t in self.groups[g][1:])
        # the groups the block's OFF-diagonal entries fall in (a rating graph: users -> items, items -> users + properties, ...)
        rows_of = torch.repeat_interleave(torch.arange(hg, device=dev), deg)
        off_diag = colidx.long() != rows_of + local.diag_offset
        local.reads = tuple(int(x) for x in torch.unique(self.group_of_row(colidx.long()[off_diag])).tolist()) if colidx.numel() else ()
        local.has_diagonal = bool((~off_diag).any()) if colidx.numel() else False
        local.active_cols = max(1, sum(self.goff[x + 1] - self.goff[x] for x in set(local.reads) | {g}))
        if getattr(a, 'dinv', None) is not None and getattr(a, 'mult', None) is not None:
            local.dinv = self._padded_dinv(a)
            local.mult = torch.cat(mult).contiguous() if mult else empty
        return local

    def _padded_dinv(self, a):
        cache = self.__dict__.setdefault('_dinv_cache', {})
        key = id(a)
        if key not in cache:
            cache.clear()
            cache[key] = self.pad_vector(a.dinv).contiguous()
        return cache[key]

    def nnz_imbalance(self, rowptr):
        """max over ranks of the rank's non-zero count / the mean: 1.0 = perfectly balanced."""
        rp = rowptr.to(torch.int64).cpu()
        per = [sum(int(rp[hi] - rp[lo]) for lo, hi in (self.owned(r, t) for t in range(self.T))) for r in range(self.world)]
        return max(per) * self.world / max(1, sum(per))

    def pair_imbalance(self, u_ids):
        """max over ranks of the pairs a rank scores (pairs follow their user's owner) / the mean."""
        if u_ids.numel() == 0:
            return 1.0
        owner = ((u_ids.to(torch.int64) - self.tb[0]) // self.h[0]).clamp_(0, self.world - 1)
        per = torch.bincount(owner, minlength=self.world)
        return float(per.max()) * self.world / float(per.sum())


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

    def _propagate(self):
        """model.gnn(None) for a scoring step: the towers read user and item rows only, so the LAST layer's rows past them (the
        property rows of a user-item-property graph) need not be computed — `rows_needed` on the stack for the duration of the call
        (the table a caller of model.gnn(None) gets is always complete)."""
        seq = getattr(self.model.gnn, 'gnn_layers', None)
        nu, ni = self.model.n_users, self.model.n_items
        hint = seq is not None and nu is not None and ni is not None and os.environ.get('AMAR_ROWS_NEEDED', '1') != '0'
        if hint:
            seq.rows_needed = int(nu) + int(ni)
        try:
            return self.model.gnn(None)
        finally:
            if hint:
                seq.rows_needed = None

    def step(self):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        emb = self._propagate()
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
                emb = self._propagate()
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
    """Basic* / HybridBert* models with a GCN / GraphSAGE / GAT ('concatenation') or LightGCN / DGCF ('mean') stack over `world`
    ranks on the typed partition: the rank's row block of every layer, per layer one all-gather of the next gathered table and
    one of the item rows, user tower over the rank's own users, item tower over all items, scoring of the pairs of its users."""

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
        self.typed = True                      # (one partition scheme since round 3: equal-height blocks per node type)
        n = int(a.shape[0])
        known = getattr(model, 'n_users', None) is not None and getattr(model, 'n_items', None) is not None
        if known:
            nu, ni = int(model.n_users), int(model.n_items)
            if nu + ni > n:
                raise ValueError("n_users + n_items exceeds the graph's node count")
            bounds, self.item_type = [0, nu, nu + ni] + ([n] if n > nu + ni else []), 1
        else:                                    # unknown split: one node type; "users" are whoever the pairs name first, "items" every node
            nu, ni, bounds, self.item_type = 0, n, [0, n], 0
        self.split_known = known
        self.widths = self.seq.layer_widths()
        self.part = self.tpart = TypedPartition(bounds, self.world, groups=self._choose_groups(bounds, a))
        # one row block per GROUP of node types: a layer runs as one launch per group (tiles never straddle a type anyway)
        self.blocks = [self.tpart.local_block(a, self.rank, g) for g in range(self.tpart.G)]
        self.csr = self.blocks[0]                               # (the first block: what tools and tests look at)
        self.local_rows = sum(hi - lo for lo, hi in (self.tpart.owned(self.rank, t) for t in range(self.tpart.T)))
        self.local_nnz = sum(b.nnz for b in self.blocks)
        self.nnz_imbalance = self.tpart.nnz_imbalance(a.rowptr)
        self.pair_imbalance = self.tpart.pair_imbalance(u_ids) if known else 1.0
        if self.pair_imbalance > 1.15 and self.rank == 0:
            import warnings
            warnings.warn("partitioned run: pairs per rank are unbalanced (max / mean = {:.2f}): user activity follows the user ids; a "
                          "degree-interleaved relabelling of the users at load time would balance the equal-height blocks".format(self.pair_imbalance))
        dev = u_ids.device
        self.row_ids = self.tpart.node_of_row(dev)
        self.row_ids0 = self.row_ids.clamp(min=0).contiguous()           # (padding rows copy node 0: finite values nobody gathers)
        # the node held by every LOCAL row (users | items | properties of this rank, each padded to its block height)
        self.local_ids0 = torch.cat([self.row_ids0[self.tpart.block_row0(self.rank, g):self.tpart.block_row0(self.rank, g) + self.tpart.gh[g]]
                                     for g in range(self.tpart.G)]).contiguous()
        # pairs follow their user: the rank scores the pairs of the users it owns, so its user tower reads its own rows only
        self.u_lo, self.u_hi = self.tpart.owned(self.rank, 0)
        self.i_lo, self.n_items = nu, ni
        mine = torch.nonzero((u_ids >= self.u_lo) & (u_ids < self.u_hi)).view(-1)
        self.pair_index = mine
        self.u_ids = u_ids[mine].to(torch.int32).contiguous()            # the reference's ids, unchanged
        self.i_ids = i_ids[mine].to(torch.int32).contiguous()
        self.pair_plan = None
        if self.ops is capi and os.environ.get('AMAR_PAIR_PLAN', '1') != '0' and self.u_ids.numel() >= (1 << 16):
            from deep_cbrs_amar_renaissance_amd.models.basic import PairPlan
            self.pair_plan = PairPlan(self.u_ids, self.i_ids)
        # item-row gathers behind the next kernels: only with a real process group (stand-ins copy synchronously)
        self.async_exchange = getattr(self.dist, 'supports_async', self.dist is torch.distributed) and os.environ.get('AMAR_EXCHANGE_ASYNC', '1') != '0'

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

    # -- the schedule of a layer ------------------------------------------------------------------------------------
    PHASE_MIN_BYTES = 4 << 20

    def _choose_groups(self, bounds, a):
        """How the node types are bundled into launches (TypedPartition `groups`).  PHASED: [users, properties... | items] — a layer is
        two launches and two all-gathers, each section in flight behind the other group's launch; it hides the exchange but pays
        for two short launches per layer (fixed costs per launch, tiles half as tall).  UNPHASED: all types in one group — one
        launch per layer on tiles twice as tall, the layer's all-gather exposed between the layers (round 3's scheme).
        Measured with an emulated wire (tools/exp_rank_of_n.py): at ml1m(s=64) on 8 ranks (2.4 MB per rank and layer)
        the phases cost more than they hide; from about 4 MB per rank and layer on they win.  AMAR_PART_PHASES=0|1 overrides."""
        T = len(bounds) - 1
        if T == 1:
            return [[0]]
        forced = os.environ.get('AMAR_PART_PHASES')
        per_rank = -(-int(a.shape[0]) // self.world) * max(self.widths[1:] or [self.widths[0]]) * 4
        phased = forced == '1' or (forced != '0' and per_rank >= self.PHASE_MIN_BYTES)
        if not phased:
            return [list(range(T))]
        return [[t for t in range(T) if t != self.item_type], [self.item_type]]

    def phase_order(self, k):
        """Groups in the order layer k (0-based) runs them.  Phased, the order alternates: even layers [users (+ properties), items],
        odd layers [items, users (+ properties)] — the section of the next gathered table a layer finishes FIRST is in flight while
        its other launch runs, and the next layer starts with the launch that reads that section (user and property rows read item
        columns only, item rows read user and property columns)."""
        G = self.tpart.G
        order = list(range(G))
        return order if k % 2 == 0 else order[::-1]

    def _wait_sections(self, handles, k, types):
        """Make the compute stream wait for the all-gathers of table k's sections `types` (each is waited for once)."""
        for t in types:
            w = handles.pop((k, t), None)
            if w is not None:
                w.wait()

    def propagate_typed(self):
        """One propagation on the typed partition.  Returns (x_local, x_items).  'concatenation' stacks (GCN, GraphSAGE, GAT): per
        layer l = 1..L the rank's own [R, C_l] block of X_l (local row order: the partition's groups, users first, each type padded
        to its block height) and the all-gathered item rows [world * h_items, C_l] (the reference's item order; rows past n_items
        are padding) — X_0 is the node table itself.  'mean' stacks (LightGCN, DGCF): one entry each, the mean over the layers.
        Every kind follows the same scheme: a group-major table T_l [world * R, C] the next layer gathers from, made by the PRODUCER
        from its own rows of X_l (GCN: S (X_l W_{l+1}) in the SpMM epilogue; LightGCN: S X_l; DGCF: X_l sigmoid(w_{l+1}); GraphSAGE:
        X_l; GAT: X_l W_{l+1} and its neighbour scalars), one all-gather per group issued as soon as the group's launch is enqueued
        and waited for by the first launch of the next layer that reads that section."""
        ops, tp, dev = self.ops, self.tpart, self.seq.embeddings.device
        layers, widths = list(self.seq.seq_layers), self.widths
        n_tab, R = self.world * tp.R, tp.R
        emb = self.seq.embeddings.detach()
        item = self.item_type
        i0, hi_ = tp.off[item], tp.h[item]
        x_local, x_items, pending, handles = [], [], [], {}
        n_l, G = len(layers), tp.G
        rows_of = [slice(*tp.local_rows(g)) for g in range(G)]                             # a group's rows in the local [R, *] buffers
        sect_of = [slice(*tp.section(g)) for g in range(G)]                                # ... its section of a gathered table
        own_of = [slice(tp.block_row0(self.rank, g), tp.block_row0(self.rank, g) + tp.gh[g]) for g in range(G)]
        item_group = tp.group_of[item]

        def gather_items(key, block):                                   # the item rows of a local [R, C] block, behind the next kernels
            xi = self._buffer(('xi', key), (self.world * hi_, block.shape[1]))
            pending.append(self._gather(xi, block[i0:i0 + hi_], defer=True))
            x_items.append(xi)

        def gather_section(k, g, table, block):                        # group g's rows of a local [R, ...] block -> its section of table k
            handles[(k, g)] = self._gather(table[sect_of[g]], block[rows_of[g]], defer=True)

        def reads_of(g, own_landed):
            """Sections of the gathered table a launch over group g's rows reads: where its off-diagonal entries fall, plus its own
            section unless the launch takes the rows' own entries from the rank's own block (`xself`)."""
            r = set(self.blocks[g].reads)
            if own_landed:
                r.add(g)
            return r

        if self.kind == 'gcn':
            tiled = [self._use_xs(w) for w in widths[1:]]
            images = [[self.blocks[g].tiled_image(w) if use else None for g in range(G)] for w, use in zip(widths[1:], tiled)]
            pre = all(tiled) and all(im.row_scale is not None for ims in images for im in ims)   # the chain of gathered tables stays pre-scaled by d^-1/2
            h = self._buffer(('t', 0), (n_tab, widths[1]))
            ops.rowwise_xw(emb, layers[0].kernel, h, row_ids=self.row_ids, row_scale=images[0][0].col_scale if pre else None)
            self._mark('prologue')
            hl_prev = None
            for k, layer in enumerate(layers):
                last = k == n_l - 1
                nxt = layers[k + 1] if not last else None
                y = self._buffer(('y', k), (R, widths[k + 1]), zero=True)
                hn = self._buffer(('hl', k + 1), (R, widths[k + 2]), zero=True) if nxt is not None else None
                h_next = self._buffer(('t', k + 1), (n_tab, widths[k + 2])) if nxt is not None else None
                for g in self.phase_order(k):
                    blk, rs = self.blocks[g], rows_of[g]
                    own_local = tiled[k] and pre and hl_prev is not None          # the diagonal term from the rank's own block of H_k
                    if k > 0:
                        self._wait_sections(handles, k, reads_of(g, not own_local and blk.has_diagonal))
                        self._mark('exchange')
                    if tiled[k]:
                        ops.spmm_xs(images[k][g], h, y[rs], bias=layer.bias, relu=True, Wnext=nxt.kernel if nxt is not None else None,
                                    Hnext=hn[rs] if hn is not None else None, prescaled=pre, scale_next=pre and nxt is not None,
                                    xself=hl_prev[rs] if own_local else None)
                    else:
                        ops.gcn_layer(blk.rowptr, blk.colidx, blk.vals, h, layer.bias, y[rs],
                                      Wnext=nxt.kernel if nxt is not None else None, Hnext=hn[rs] if hn is not None else None)
                    self._mark('spmm')
                    if nxt is not None:
                        gather_section(k + 1, g, h_next, hn)
                    if g == item_group:
                        gather_items(k, y)
                    self._mark('exchange')
                if k > 0:
                    self._wait_sections(handles, k, range(G))                     # (sections nobody read: drained before the table is reused)
                h, hl_prev = h_next, hn
                x_local.append(y)

        elif self.kind in ('lightgcn', 'dgcf'):
            d = widths[0]
            tiled = self._use_xs(d)
            images = [self.blocks[g].tiled_image(d) if tiled else None for g in range(G)]
            value_free = tiled and all(im.row_scale is not None for im in images)
            tab = self._buffer(('t', 0), (n_tab, d))
            if self.kind == 'dgcf':                                      # layer 1 gathers X_0 . sigmoid(w_1)
                gated = self._buffer(('g0',), (emb.shape[0], d))
                ops.locality_scale(emb, layers[0].w.detach().view(-1), gated)
                ops.copy_columns(gated, tab, ids=self.row_ids0)
            elif value_free:                                             # ... S X_0 (the identity kernel keeps the bits of X_0)
                ops.rowwise_xw(emb, self._identity(d, dev), tab, row_ids=self.row_ids, row_scale=images[0].col_scale)
            else:
                ops.copy_columns(emb, tab, ids=self.row_ids0)
            acc = self._buffer(('acc', 0), (R, d))
            ops.copy_columns(emb, acc, ids=self.local_ids0)              # the running sum starts at the rank's own rows of X_0
            self._mark('prologue')
            for k, layer in enumerate(layers):
                last = k == n_l - 1
                y = None if last else self._buffer(('y', k), (R, d), zero=True)
                acc_out = self._buffer(('acc', k + 1), (R, d))
                nxt_local = None if last else (self._buffer(('tl', k + 1), (R, d)) if (self.kind == 'dgcf' or value_free) else y)
                tab_next = None if last else self._buffer(('t', k + 1), (n_tab, d))
                gate = self._gate_local(k + 1, layers[k + 1]) if (self.kind == 'dgcf' and not last) else None
                for g in self.phase_order(k):
                    blk, rs = self.blocks[g], rows_of[g]
                    if k > 0:
                        self._wait_sections(handles, k, reads_of(g, True))
                        self._mark('exchange')
                    kw = dict(acc_in=acc[rs], acc_out=acc_out[rs], acc_div=n_l + 1 if last else None)
                    if tiled:
                        ops.spmm_xs(images[g], tab, y[rs] if y is not None else None, prescaled=value_free, **kw)
                    else:
                        ops.spmm_csr(blk.rowptr, blk.colidx, blk.vals, tab, y[rs] if y is not None else None, **kw)
                    self._mark('spmm')
                    if not last:
                        if self.kind == 'dgcf':
                            ops.locality_scale(y[rs], gate[rs], nxt_local[rs])
                        elif value_free:
                            ops.row_affine(y[rs], images[g].row_scale, nxt_local[rs])
                        gather_section(k + 1, g, tab_next, nxt_local)
                        self._mark('exchange')
                if k > 0:
                    self._wait_sections(handles, k, range(G))
                acc, tab = acc_out, tab_next
            gather_items('mean', acc)
            self._mark('exchange')
            x_local.append(acc)

        else:                                                            # GraphSAGE / GAT: edge-list graphs, layers gather X_l (GAT: X_l W)
            x0 = self._buffer(('t', 0), (n_tab, widths[0]))
            ops.copy_columns(emb, x0, ids=self.row_ids0)
            tab = x0
            if self.kind == 'gat':
                l0, c = layers[0], widths[1]
                tab = self._buffer(('h', 0), (n_tab, c))
                s_self, s_neigh = self._buffer(('ss', 0), (n_tab,)), self._buffer(('sn', 0), (n_tab,))
                ops.rowwise_xw(x0, l0.kernel.view(-1, c), tab, a_self=l0.attn_kernel_self.view(c), a_neigh=l0.attn_kernel_neighs.view(c),
                               s_self=s_self, s_neigh=s_neigh)
            self._mark('prologue')
            for k, layer in enumerate(layers):
                f, c = widths[k], widths[k + 1]
                last = k == n_l - 1
                y = self._buffer(('y', k), (R, c), zero=True)
                if not last:
                    if self.kind == 'sage':
                        tab_next = self._buffer(('t', k + 1), (n_tab, c))
                    else:                                                # the producer's X_l . W_{l+1} and attention scalars, own rows only
                        nxt, c2 = layers[k + 1], widths[k + 2]
                        tab_next = self._buffer(('h', k + 1), (n_tab, c2))
                        h_local = self._buffer(('hl', k + 1), (R, c2))
                        ss_next = self._buffer(('ss', k + 1), (n_tab,), zero=True)
                        sn_next = self._buffer(('sn', k + 1), (n_tab,))
                        sn_local = self._buffer(('snl', k + 1), (R,))
                if k > 0:                                                # (these kinds read their own rows from the table, and GAT reduces s_neigh over
                    self._wait_sections(handles, k, range(G))            #  all of it: every section of table k lands before the layer starts)
                    self._wait_sections(handles, ('sn', k), range(G))
                    self._mark('exchange')
                for g in self.phase_order(k):
                    blk, rs, own = self.blocks[g], rows_of[g], own_of[g]
                    if self.kind == 'sage':
                        agg = self._buffer(('agg', k), (R, f))
                        ops.spmm_xs(blk.tiled_mean_image(f, layer.self_loops), tab, agg[rs], prescaled=True)
                        if ops.sage_tail_supported(f, c):
                            ops.sage_tail(tab[own], agg[rs], layer.kernel, layer.bias, y[rs])
                        else:
                            xa = self._buffer(('xa', k), (R, 2 * f))
                            ops.copy_columns(tab[own], xa[rs][:, :f])
                            ops.copy_columns(agg[rs], xa[rs][:, f:])
                            z = self._buffer(('z', k), (R, c))
                            ops.dense(xa[rs], layer.kernel, layer.bias, z[rs], act=None)
                            nrm, inv = self._buffer(('nrm', k), (R, c)), self._buffer(('inv', k), (R,))
                            ops.l2norm_fwd(z[rs], nrm[rs], inv[rs], y[rs], act='relu')
                    else:
                        lt = blk.tiled_gat_image(c)
                        if lt is not None:
                            ops.gat_lt(lt, blk, tab, s_self, s_neigh, layer.bias, y[rs], self_loop=layer.add_self_loops)
                        else:
                            ops.gat_xs(blk.xcd_sliced(), tab, s_self, s_neigh, layer.bias, y[rs], self_loop=layer.add_self_loops)
                    self._mark('spmm')
                    if g == item_group:
                        gather_items(k, y)
                    if not last:
                        if self.kind == 'sage':
                            gather_section(k + 1, g, tab_next, y)
                        else:
                            ops.rowwise_xw(y[rs], nxt.kernel.view(-1, c2), h_local[rs], a_self=nxt.attn_kernel_self.view(c2),
                                           a_neigh=nxt.attn_kernel_neighs.view(c2), s_self=ss_next[own], s_neigh=sn_local[rs])
                            gather_section(k + 1, g, tab_next, h_local)
                            handles[(('sn', k + 1), g)] = self._gather(sn_next[sect_of[g]], sn_local[rs], defer=True)
                    self._mark('exchange')
                if not last:
                    tab = tab_next
                    if self.kind == 'gat':
                        s_self, s_neigh = ss_next, sn_next
                x_local.append(y)
        self._pending = [w for w in pending if w is not None] + [w for w in handles.values() if w is not None]
        return x_local, x_items

    def _identity(self, d, dev):
        cache = self.__dict__.setdefault('_eye', {})
        if d not in cache:
            cache[d] = torch.eye(d, dtype=torch.float32, device=dev).contiguous()
        return cache[d]

    def _gate_local(self, k, layer):
        """DGCF's per-node gate weights of layer k for the rank's own rows, in local row order (rebuilt when they change)."""
        cache = self.__dict__.setdefault('_gates_local', {})
        version = layer.w._version
        if cache.get(k, (None, None))[0] != version:
            cache[k] = (version, layer.w.detach().view(-1)[self.local_ids0.long()].contiguous())
        return cache[k][1]

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
        # (a model that does not know its user / item split has ONE node type: no item blocks to shard the tower over — replicated)
        if self.hybrid and self.split_known and os.environ.get('AMAR_HYBRID_ITEM_TOWER', 'sharded') == 'sharded':      # (every rank takes part: it is a collective)
            if self.model.bert_table is None:
                raise ValueError("the hybrid model needs its BERT table registered (set_bert_table) for the partitioned run")
            if not self.model.rs.built:
                self.model.rs.build_head(self.model.gnn.output_dim(), self.model.bert_table.shape[1])
            ib_wait, ib_full = self._item_bert_sharded()
            self._mark('item_bert')
        x_local, x_items = self.propagate_typed()
        emb = self.seq.embeddings.detach()
        nu_loc = self.u_hi - self.u_lo
        if self.kind in ('lightgcn', 'dgcf'):                           # 'mean' reduction: one table
            u_table, i_table = x_local[0][:nu_loc], x_items[0][:self.n_items]
        else:                                                           # 'concatenation': [X_0 || X_1 || ...] read in place
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
            kw = {'pair_plan': self.pair_plan} if self.pair_plan is not None else {}
            out = rs.score_towers(towers, self.u_ids, self.i_ids, self.u_lo, self.i_lo, **kw)
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
        tower, pair stage), `exchange` = `exposed_exchange` = time the compute stream spent issuing all-gathers and WAITING for the
        sections / item rows it needs next (the collectives themselves run on RCCL's stream behind the other types' launches: what
        shows here is the part of the exchange that compute did not cover, plus the host-side issue gaps of an eager step)."""
        if not getattr(self, '_marks', None):
            return None
        self._marks[-1][1].synchronize()
        out = {}
        for (_, e0), (name, e1) in zip(self._marks[:-1], self._marks[1:]):
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        spmm, pairs = out.get('spmm', 0.0), out.get('pairs', 0.0)
        return {'local_spmm_ms': spmm, 'exchange_ms': out.get('exchange', 0.0), 'exposed_exchange_ms': out.get('exchange', 0.0),
                'replicated_ms': out.get('prologue', 0.0) + out.get('item_tower', 0.0),
                'user_tower_ms': out.get('user_tower', 0.0), 'towers_ms': out.get('towers', 0.0) + out.get('item_bert', 0.0), 'pair_stage_ms': pairs,
                'prologue_ms': out.get('prologue', 0.0), 'item_tower_ms': out.get('item_tower', 0.0)}

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
        return self._step_typed()

    def _use_xs(self, width):
        """XCD-sliced local SpMM when the gathered table exceeds the per-XCD L2s (as utilities.math.spmm_kind decides
        for the single-GPU path); AMAR_SPMM_KIND=csr|xs overrides."""
        if self.ops is not capi or not hasattr(self.csr, 'diag_offset') or width > 16 or width % 4:
            return False
        forced = os.environ.get('AMAR_SPMM_KIND')
        if forced in ('csr', 'xs'):
            return forced == 'xs'
        return self.world * self.part.R * width * 4 >= ((8 << 20) if width <= 8 else (16 << 20))

    def last_propagation_ms(self):
        ph = self.phase_times()
        return None if ph is None else ph['prologue_ms'] + ph['local_spmm_ms'] + ph['exchange_ms']

    def describe(self):
        how = ('per layer one launch and one RCCL all-gather per group of node types {} (the next gathered table, group-major), each in flight '
               'behind the other group\'s launch'.format(self.tpart.groups) if self.tpart.G > 1 else
               'per layer one launch and one RCCL all-gather of the next gathered table (rank-major)')
        return ('node-range partition over {} GPUs (equal-height blocks per node type, nnz imbalance {:.3f}, pair imbalance {:.3f}), {} + one '
                'all-gather of the item rows behind the next kernels; pairs sharded by the same user ranges').format(
                    self.world, self.nnz_imbalance, self.pair_imbalance, how)


def make_runner(model, u_ids, i_ids, rank=0, world=1, dist=None):
    if world == 1:
        return SingleRunner(model, u_ids, i_ids)
    return PartitionedGCNRunner(model, u_ids, i_ids, rank, world, dist=dist)
