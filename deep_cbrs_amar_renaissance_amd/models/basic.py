"""Basic recommender heads — mirrors `/root/reference/src/models/basic.py:11-120`.

``BasicRS``: two dense towers (user, item), Concatenate, dense classifier ending in
Dense(1, sigmoid).  ``BasicGNN``: full-graph propagation -> embedding lookup of the batch's
(user, item) ids -> BasicRS.  The class factory at the bottom generates ``BasicGCN``,
``BasicGAT``, ``BasicGraphSage``, ``BasicLightGCN``, ``BasicDGCF``, ``BasicTS*`` and ``BasicTW*`` exactly like
`basic.py:90-120`.

Device mapping: the lookup is fused into the first tower layer's load (`amar_dense_f32` with
``ids``), each tower's last layer writes its half of the concatenation buffer directly.
"""
import abc

import os

import numpy as np

import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Model, ids_to_device, to_device_tensor
from deep_cbrs_amar_renaissance_amd.models.dense import build_dense_network, build_dense_classifier
from deep_cbrs_amar_renaissance_amd.models.gnn import GCN, GAT, GraphSage, LightGCN, DGCF
from deep_cbrs_amar_renaissance_amd.models.tsgnn import TwoStepGCN, TwoStepGraphSage, TwoStepGAT, TwoStepLightGCN, TwoStepDGCF
from deep_cbrs_amar_renaissance_amd.models.twgnn import TwoWayGCN, TwoWayGraphSage, TwoWayGAT, TwoWayLightGCN, TwoWayDGCF


class BasicRS(Model):
    def __init__(
            self,
            dense_units=(512, 256, 128),
            clf_units=(64, 64),
            activation='relu',
            **kwargs
    ):
        """
        :param dense_units: units of the user / item towers.
        :param clf_units: units of the classifier (a final Dense(1, sigmoid) is appended).
        :param activation: hidden activation.
        :param kwargs: unused.
        """
        super().__init__()
        self.dense_units = list(dense_units)
        self.unet = build_dense_network(dense_units, activation=activation)
        self.inet = build_dense_network(dense_units, activation=activation)
        self.clf = build_dense_classifier(clf_units, n_classes=1, activation=activation)

    def build(self, input_shape):
        self.build_head(input_shape[0][-1], input_shape[1][-1])

    def build_head(self, u_dim, i_dim):
        """Create all weights for known input widths (Keras would do it at the first call)."""
        d = self.unet.build_chain(u_dim)
        self.inet.build_chain(i_dim)
        self.clf.build_chain(2 * d)
        self.built = True

    def call(self, inputs, u_ids=None, i_ids=None, **kwargs):
        """inputs = (u, i): [B, F] feature blocks — or, with ``u_ids``/``i_ids``, two tables to gather from."""
        u, i = inputs
        u, i = to_device_tensor(u), to_device_tensor(i)
        if len(self.dense_units) == 0:
            raise NotImplementedError("BasicRS needs at least one tower layer")
        if not self.built:
            self.build_head(u.shape[1], i.shape[1])
        towers = self.towers(u, i, u_ids=u_ids, i_ids=i_ids)
        return self.score_towers(towers, None, None)

    def fit(self, sequence, epochs=1, **kwargs):
        """Keras ``fit`` on pre-computed embedding rows (basic-kge / hybrid-kge configs): BCE + Adam on the Dense stacks."""
        from deep_cbrs_amar_renaissance_amd import training
        return training.fit(self, sequence, epochs=epochs, **kwargs)

    # The classifier's first Dense layer is linear in its concatenated input, [u || i] . W1 = u . W1[:d] + i . W1[d:],
    # so its two halves ride at the end of the towers (per entity when hoisted) and the pair stage starts at
    # act(T'u[u] + T'i[i] (+ b1, folded into T'i)): half the per-pair MFMA work of the classifier's first two layers.
    def _split_plan(self, u_dim, i_dim):
        """Packed stacks for the split formulation, or None when the fused chain cannot run these shapes."""
        version = self.weights_version
        cache = getattr(self, '_split_cache', None)
        if cache is not None and cache[0] == (version, u_dim, i_dim):
            return cache[1]
        plan = None
        clf, d = list(self.clf.layers), self.dense_units[-1]
        if len(clf) >= 3:
            c1 = clf[0].units
            udims = [u_dim] + [l.units for l in self.unet.layers] + [c1]
            idims = [i_dim] + [l.units for l in self.inet.layers] + [c1]
            rdims = [c1] + [l.units for l in clf[1:]]
            if (capi.chain_supported(udims, u_dim) and capi.chain_supported(idims, i_dim)
                    and capi.chain_supported(rdims, c1, c1, sum_inputs=True)):
                w1 = clf[0].kernel.detach().cpu().numpy()
                b1 = clf[0].bias.detach().cpu().numpy()
                dev = clf[0].kernel.device
                np_ = lambda p: p.detach().cpu().numpy()
                ub, _ = capi.chain_pack([np_(l.kernel) for l in self.unet.layers] + [w1[:d]],
                                        [np_(l.bias) for l in self.unet.layers] + [0 * b1])
                ib, _ = capi.chain_pack([np_(l.kernel) for l in self.inet.layers] + [w1[d:]],
                                        [np_(l.bias) for l in self.inet.layers] + [b1])
                rb, _ = capi.chain_pack([np_(l.kernel) for l in clf[1:]], [np_(l.bias) for l in clf[1:]])
                to = lambda x: torch.from_numpy(x).to(dev)
                plan = {'u': (to(ub), udims, [l.activation for l in self.unet.layers] + [None]),
                        'i': (to(ib), idims, [l.activation for l in self.inet.layers] + [None]),
                        'rest': (to(rb), rdims, [l.activation for l in clf[1:]]), 'in_act': clf[0].activation}
        self._split_cache = ((version, u_dim, i_dim), plan)
        return plan

    def towers(self, u_table, i_table, u_ids=None, i_ids=None):
        """Tower outputs for the given rows (all rows of the tables, or the gathered ids): Dense stacks act row by
        row, so in hoisted mode unet/inet run once per user/item instead of once per pair."""
        plan = self._split_plan(u_table.shape[1], i_table.shape[1])
        if plan is None:
            return (self.unet.apply2(u_table, ids_a=u_ids), self.inet.apply2(i_table, ids_a=i_ids), False)
        out = []
        for key, table, ids in (('u', u_table, u_ids), ('i', i_table, i_ids)):
            blob, dims, acts = plan[key]
            m = ids.numel() if ids is not None else table.shape[0]
            t = torch.empty((m, dims[-1]), dtype=torch.float32, device=table.device)
            capi.chain(table, blob, dims, acts, t, ids_a=ids)
            out.append(t)
        return (out[0], out[1], True)

    def tower(self, key, table, ids=None):
        """One tower ('u' or 'i') of `towers` on its own — the partitioned runner evaluates the user tower (its own rows) while the
        last item rows are still arriving.  `table` may be a capi.ConcatTable (per-layer tables read in place)."""
        plan = self._split_plan(table.shape[1], table.shape[1])
        if plan is None:
            return (self.unet if key == 'u' else self.inet).apply2(table, ids_a=ids)
        blob, dims, acts = plan[key]
        m = ids.numel() if ids is not None else table.shape[0]
        t = torch.empty((m, dims[-1]), dtype=torch.float32, device=table.device)
        capi.chain(table, blob, dims, acts, t, ids_a=ids)
        return t

    def split_ready(self):
        """Whether the last tower() / towers() call produced the split form (classifier layer 1 folded into the towers)."""
        return self._split_cache[1] is not None

    def score_towers(self, towers, u_ids, i_ids, u_base=0, i_base=0, pair_plan=None):
        """Scores of the pairs (u_ids[p], i_ids[p]) from per-entity tower tables.  `pair_plan` (PairPlan of the same id lists):
        the launch walks the list in the plan's XCD-affine order and writes every score to its place in the caller's order."""
        tu, ti, split = towers
        if not split:
            return self.clf.apply2(tu, ti, ids_a=u_ids, base_a=u_base, ids_b=i_ids, base_b=i_base)
        plan = self._split_cache[1]
        blob, dims, acts = plan['rest']
        m = u_ids.numel() if u_ids is not None else tu.shape[0]
        out = torch.empty((m, 1), dtype=torch.float32, device=tu.device)
        if u_ids is None or i_ids is None:
            # rows themselves (the per-batch call, towers evaluated per pair): numbered 0..m-1, so that this call too runs the pair-stage
            # kernel — the same products as the hoisted path, hence the same bits (the generic kernel multiplies on the f32 instruction)
            rows = getattr(self, '_row_ids', None)
            if rows is None or rows.numel() < m or rows.device != tu.device:
                rows = self._row_ids = torch.arange(max(m, 4096), dtype=torch.int32, device=tu.device)
            u_ids, u_base = (rows[:m], 0) if u_ids is None else (u_ids, u_base)
            i_ids, i_base = (rows[:m], 0) if i_ids is None else (i_ids, i_base)
        if pair_plan is not None:
            pair_plan.check(u_ids, i_ids)
            if pair_plan.mid_index is not None:
                capi.chain(tu, blob, dims, acts, pair_plan.mid.view(-1, 1), ids_a=pair_plan.u_ids, base_a=u_base, B=ti, ids_b=pair_plan.i_ids,
                           base_b=i_base, sum_inputs=True, in_act=plan['in_act'], out_index=pair_plan.mid_index)
                capi.scatter(pair_plan.mid, pair_plan.final_index, out, pair_plan.window_off, pair_plan.n_windows)
                return out
            capi.chain(tu, blob, dims, acts, out, ids_a=pair_plan.u_ids, base_a=u_base, B=ti, ids_b=pair_plan.i_ids, base_b=i_base,
                       sum_inputs=True, in_act=plan['in_act'], out_index=pair_plan.out_index)
            return out
        capi.chain(tu, blob, dims, acts, out, ids_a=u_ids, base_a=u_base, B=ti, ids_b=i_ids, base_b=i_base,
                   sum_inputs=True, in_act=plan['in_act'])
        return out


class PairPlan:
    """A pair list prepared ONCE per dataset for the pair stage (the test Sequence's pairs are constant across steps and
    epochs: datasets.py:199-203 only reshuffles the TRAIN order).  The scoring kernel deals its 128-pair chunks to workgroups
    round-robin, and workgroups to the eight XCDs round-robin, so the pairs at positions p with (p >> 7) % 8 == x run on XCD x.
    The plan sorts the pairs by item id, cuts that order into eight contiguous item ranges sized to the positions each XCD
    owns, and places range x on XCD x's positions — inside a range ordered by user id.  Every XCD then gathers item-tower rows of one eighth of the items (~5 MB at
    ml1m(s=64): L2-resident) instead of all of them; `out_index` sends each score back to the caller's position."""

    N_XCD, CHUNK = 8, 128

    def __init__(self, u_ids, i_ids, phases=None):
        p = int(u_ids.numel())
        dev = u_ids.device
        if phases is None:
            phases = int(os.environ.get('AMAR_PAIR_PHASES', '1'))
        pos = torch.arange(p, device=dev)
        # position -> class: (phase, XCD).  A workgroup walks its chunks in ascending position, so the first 1/phases of the
        # positions are visited first by every workgroup: with `phases` > 1 an XCD works through `phases` item ranges one after
        # the other, each a 1/(8 phases) of the items
        per_phase = -(-p // phases)
        cls = (pos // per_phase) * self.N_XCD + (pos // self.CHUNK) % self.N_XCD
        n_cls = phases * self.N_XCD
        slots = torch.bincount(cls, minlength=n_cls)                           # positions owned by each class
        by_item = torch.argsort(i_ids.to(torch.int64), stable=True)
        # item ranges in class order: range k (k-th slice of the item-sorted pairs) goes to the class with the k-th ... the ranges are
        # sized to the classes taken in (XCD, phase) order, so that XCD x gets `phases` neighbouring ranges
        order_cls = torch.arange(n_cls, device=dev).view(phases, self.N_XCD).t().reshape(-1)     # (xcd, phase) -> class id
        bucket_sizes = slots[order_cls]
        bucket_of_rank = torch.repeat_interleave(order_cls, bucket_sizes)       # class of the k-th pair in item order
        if os.environ.get('AMAR_PAIR_INNER', 'user') == 'user' and p:
            # inside a range by user id, so that consecutive pairs also share user-tower rows (0.674 against 0.689 ms at
            # ml1m(s=64) for the original order inside a range, AMAR_PAIR_INNER=pos; the scattered score writes cost nothing extra)
            order = by_item[torch.argsort((bucket_of_rank * (int(u_ids.max()) + 1) + u_ids[by_item].to(torch.int64)) * p + by_item)]
        else:
            order = by_item[torch.argsort(bucket_of_rank * p + by_item)]        # (class, original position)
        place = torch.argsort(cls * p + pos)                                    # positions grouped by class, ascending inside
        src = torch.empty(p, dtype=torch.int64, device=dev)
        src[place] = order                                                      # position -> original pair
        self.u_ids = u_ids[src].contiguous()
        self.i_ids = i_ids[src].contiguous()
        self.out_index = src.to(torch.int32).contiguous()
        self._key = (u_ids.data_ptr(), i_ids.data_ptr(), p)
        # The way back to the caller's order in two steps (round 3).  Stored straight to out[out_index[.]], 12 M single words at random
        # over 48 MB are each a line fetched and written back: 0.19 ms of the 0.60 ms launch at ml1m(s=64), memory-side work next to
        # the gathers'.  Instead the kernel stores to mid[mid_index[.]], a scratch vector ordered by (WINDOW of the final position, XCD,
        # list position) — each XCD appends to one open line per window, which its L2 completes —, and amar_scatter_f32 finishes inside
        # the windows, one window per XCD at a time (window_off).  Only for long lists: up to a few million scores the destination
        # mostly stays in the L2s and the direct store is the faster one (a rank of 4: 3.0 M pairs 0.115 against 0.125 ms; a rank of
        # 2: 6.0 M pairs 0.289 against 0.242 ms) — AMAR_PAIR_WINDOW_MIN pairs (default 4 Mi); AMAR_PAIR_WINDOW=0: always direct.
        # window: about 185 of them (64 Ki scores at ml1m(s=64)'s 12 M pairs, 256 Ki at ml1m(s=256)'s 48 M: every XCD keeps one open line
        # per window, and the second launch's window stays a piece of the destination an L2 holds — ml1m(s=256): 5.05 ms per step
        # with 256 Ki against 5.14 with 64 Ki and 5.18 with the direct store)
        auto = 1 << max(16, min(19, int(round(np.log2(max(p, 1) / 185.0)))))
        self.window = int(os.environ.get('AMAR_PAIR_WINDOW', str(auto)))
        self.mid_index = self.final_index = self.window_off = self.mid = None
        self.n_windows = 0
        if self.window > 0 and p > 2 * self.window and p >= int(os.environ.get('AMAR_PAIR_WINDOW_MIN', str(1 << 22))):
            n_win = -(-p // self.window)
            key = (src // self.window) * self.N_XCD + (pos // self.CHUNK) % self.N_XCD
            by_key = torch.argsort(key * p + pos)
            mid_index = torch.empty(p, dtype=torch.int32, device=dev)
            mid_index[by_key] = torch.arange(p, dtype=torch.int32, device=dev)
            self.mid_index = mid_index
            self.final_index = self.out_index[by_key].contiguous()
            counts = torch.bincount(src // self.window, minlength=n_win)
            self.window_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(counts, 0)]).to(torch.int32)
            self.n_windows = n_win
            self.mid = torch.empty(p, dtype=torch.float32, device=dev)

    def check(self, u_ids, i_ids):
        if (u_ids.data_ptr(), i_ids.data_ptr(), int(u_ids.numel())) != self._key:
            raise ValueError("this PairPlan was prepared for another pair list")


class BasicGNN(Model, abc.ABC):
    def __init__(
            self,
            dense_units=(32, 16),
            clf_units=(16, 16),
            activation='relu',
            **kwargs
    ):
        super().__init__()
        self.rs = BasicRS(dense_units, clf_units, activation=activation)
        self.n_users = self.n_items = None      # optional: lets the hoisted mode run each tower on its own rows only
        self._towers = None
        self.built = True

    def call(self, inputs, **kwargs):
        updated_embeddings = self.gnn(None)
        return self.embed_recommend(updated_embeddings, inputs)

    def embed_recommend(self, embeddings, inputs):
        """Look up the user / item rows of `embeddings` and score them: inputs = (user ids, item ids).

        Faithful mode gathers per pair.  Hoisted mode (predict) computes the two towers once per entity —
        same kernel, same per-row arithmetic, so both modes return identical bits."""
        n_nodes = embeddings.shape[0]
        u, i = ids_to_device(inputs[0], n_nodes), ids_to_device(inputs[1], n_nodes)
        if not self.gnn.hoist:
            return self.rs([embeddings, embeddings], u_ids=u, i_ids=i)
        key = (self.weights_version, embeddings.data_ptr())
        if self._towers is None or self._towers[0] != key:
            n = embeddings.shape[0]
            nu = self.n_users if self.n_users is not None else n
            lo = nu if self.n_users is not None else 0
            hi = nu + self.n_items if (self.n_users is not None and self.n_items is not None) else n
            self._towers = (key, self.rs.towers(embeddings[:nu], embeddings[lo:hi]), lo)
        _, towers, lo = self._towers
        return self.rs.score_towers(towers, u, i, 0, lo)

    def fit(self, sequence, epochs=1, **kwargs):
        """Keras ``fit``: BCE + L2 + Adam over the batches of `sequence` (training.py)."""
        from deep_cbrs_amar_renaissance_amd import training
        return training.fit(self, sequence, epochs=epochs, **kwargs)

    def _hoist_begin(self, hoist):
        self.gnn.hoist = bool(hoist)

    def _hoist_end(self):
        self.gnn.hoist = False
        self.gnn._hoisted = None
        self._towers = None

    def _graph_predict_supported(self, sequence):
        """Batches of id pairs only (datasets.UserItemGraph): nothing but device work between the first and the last launch."""
        if len(sequence) == 0 or not hasattr(self.gnn, 'gnn_layers'):     # single-graph stacks (TwoStep / TwoWay stay eager)
            return False
        inputs, _ = sequence[0]
        return isinstance(inputs, tuple) and len(inputs) == 2


class BasicTSGNN(BasicGNN):
    pass


class BasicTWGNN(BasicGNN):
    pass


class BasicKnowledgeGCN(BasicGNN):
    pass


def BasicGNNFactory(name, Parent, GNN):
    def __init__(self, *args, **kwargs):
        Parent.__init__(self, **kwargs)
        self.gnn = self.gnn_class(*args, **kwargs)
        self.gnn.build_layers()
        self.rs.build_head(self.gnn.output_dim(), self.gnn.output_dim())

    return type(name, (Parent,), {"gnn_class": GNN, "__init__": __init__})


BASIC_GNNS = [
    (BasicGNN, [GCN, GAT, GraphSage, LightGCN, DGCF], None),
    (BasicTSGNN, [TwoStepGCN, TwoStepGraphSage, TwoStepGAT, TwoStepLightGCN, TwoStepDGCF], lambda name: 'BasicTS' + name[7:]),
    (BasicTWGNN, [TwoWayGCN, TwoWayGraphSage, TwoWayGAT, TwoWayLightGCN, TwoWayDGCF], lambda name: 'BasicTW' + name[6:]),
]


def generate_basics():
    for parent, gnns, name_getter in BASIC_GNNS:
        for gnn in gnns:
            name = name_getter(gnn.__name__) if name_getter is not None else 'Basic' + gnn.__name__
            globals()[name] = BasicGNNFactory(name, parent, gnn)


generate_basics()
