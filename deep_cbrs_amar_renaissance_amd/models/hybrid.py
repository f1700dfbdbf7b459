"""Hybrid (graph + BERT) recommender heads — mirrors `/root/reference/src/models/hybrid.py:13-181`.

``HybridCBRS`` takes four inputs (user-graph, item-graph, user-BERT, item-BERT):

    ug -> dense1a, ig -> dense1b, ub -> dense2a, ib -> dense2b
    feature_based:  x1 = dense3a([ug || ig]),  x2 = dense3b([ub || ib])
    entity based:   x1 = dense3a([ug || ub]),  x2 = dense3b([ig || ib])
    x = fuse2([x1, x2]);  out = clf(x)   or, with residual=True,   clf(act(residual(x) + x1 + x2))

FusionLayer('concatenate') everywhere is the setting of the BASELINE configs and the one with the fused pair-stage
kernels (folded first layers, `amar_dual_chain_f32`); 'attention' fusion (fuse2 when feature based, fuse1a/1b
otherwise) and the residual classifier (`hybrid-gnn-tweaks*.yaml`) run on per-branch chains + the mix / add
kernels.  ``HybridBertGNN`` = propagation -> lookup -> HybridCBRS; the factory generates
``HybridBert{GCN,GAT,GraphSage,LightGCN,DGCF}``.

The BERT blocks may be given per batch ([B, 768] host arrays, as the reference's Sequence
does, `datasets.py:65-66`) or as a resident table + ids (``bert_table``), which removes the
12.6 MB host->device copy per 2 048-pair batch.
"""
import abc
import os

import numpy as np
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import Model, ids_to_device, to_device_tensor
from deep_cbrs_amar_renaissance_amd.layers.fusion import FusionLayer
from deep_cbrs_amar_renaissance_amd.models.dense import build_dense_classifier, build_dense_network, build_residual_dense_network
from deep_cbrs_amar_renaissance_amd.models.gnn import GCN, GAT, GraphSage, LightGCN, DGCF
from deep_cbrs_amar_renaissance_amd.models.tsgnn import TwoStepGCN, TwoStepGraphSage, TwoStepGAT, TwoStepLightGCN, TwoStepDGCF
from deep_cbrs_amar_renaissance_amd.models.twgnn import TwoWayGCN, TwoWayGraphSage, TwoWayGAT, TwoWayLightGCN, TwoWayDGCF


class HybridCBRS(Model):
    """Hybrid recommender system that receives inputs from two sources."""

    def __init__(
            self,
            feature_based=True,
            dense_units=((512, 256, 128), (512, 256, 128), (64, 64)),
            clf_units=(64, 64),
            activation='relu',
            fusion_method='concatenate',
            residual=False,
            **kwargs
    ):
        super().__init__()
        self.feature_based = feature_based
        if feature_based:
            self.fuse1a, self.fuse1b, self.fuse2 = FusionLayer('concatenate'), FusionLayer('concatenate'), FusionLayer(fusion_method)
        else:
            self.fuse1a, self.fuse1b, self.fuse2 = FusionLayer(fusion_method), FusionLayer(fusion_method), FusionLayer('concatenate')
        if residual and dense_units[2][-1] != clf_units[-1]:
            raise ValueError("The last dense units before the last fusion layer "
                             "must be equal to the last classifier units for residual connections")
        self.dense_units = [list(d) for d in dense_units]
        if any(len(d) == 0 for d in self.dense_units):
            raise NotImplementedError("every hybrid branch needs at least one layer")
        self.dense1a = build_dense_network(dense_units[0], activation=activation)
        self.dense1b = build_dense_network(dense_units[0], activation=activation)
        self.dense2a = build_dense_network(dense_units[1], activation=activation)
        self.dense2b = build_dense_network(dense_units[1], activation=activation)
        self.dense3a = build_dense_network(dense_units[2], activation=activation)
        self.dense3b = build_dense_network(dense_units[2], activation=activation)
        if residual:                                          # hybrid.py:61-65
            self.residual = build_residual_dense_network(clf_units, activation=activation)
            self.activation = activation
            self.clf = build_dense_classifier([], n_classes=1)
        else:
            self.residual = self.activation = None
            self.clf = build_dense_classifier(clf_units, n_classes=1, activation=activation)
        self.fusion_method = fusion_method

    def build(self, input_shape):
        self.build_head(input_shape[0][-1], input_shape[2][-1])

    def build_head(self, g_dim, b_dim):
        """Create all weights for known graph / BERT input widths."""
        d1 = self.dense1a.build_chain(g_dim)
        self.dense1b.build_chain(g_dim)
        d2 = self.dense2a.build_chain(b_dim)
        self.dense2b.build_chain(b_dim)
        ins = ((d1, d1), (d2, d2)) if self.feature_based else ((d1, d2), (d1, d2))
        for fuse, (da, db) in zip((self.fuse1a, self.fuse1b), ins):
            if not fuse.built:
                fuse.build([(None, da), (None, db)])
                fuse.built = True
        d3 = self.dense3a.build_chain(self.fuse1a.output_dim(*ins[0]))
        self.dense3b.build_chain(self.fuse1b.output_dim(*ins[1]))
        if not self.fuse2.built:
            self.fuse2.build([(None, d3), (None, d3)])
            self.fuse2.built = True
        d_fused = self.fuse2.output_dim(d3, d3)
        if self.residual is not None:
            self.clf.build_chain(self.residual.build_chain(d_fused))
        else:
            self.clf.build_chain(d_fused)
        self.built = True

    def call(self, inputs, g_ids=None, b_ids=None, **kwargs):
        """inputs = (ug, ig, ub, ib) blocks; ``g_ids`` / ``b_ids`` = (user ids, item ids) turn the graph /
        BERT inputs into tables to gather from."""
        ug, ig, ub, ib = [to_device_tensor(t) for t in inputs]
        gu, gi = g_ids if g_ids is not None else (None, None)
        bu, bi = b_ids if b_ids is not None else (None, None)
        if not self.built:
            self.build_head(ug.shape[1], ub.shape[1])
        towers = (self.dense1a.apply2(ug, ids_a=gu), self.dense1b.apply2(ig, ids_a=gi),
                  self.dense2a.apply2(ub, ids_a=bu), self.dense2b.apply2(ib, ids_a=bi), False)
        return self.score_towers(towers, None, None)

    def fit(self, sequence, epochs=1, **kwargs):
        """Keras ``fit`` on pre-computed embedding rows (basic-kge / hybrid-kge configs): BCE + Adam on the Dense stacks."""
        from deep_cbrs_amar_renaissance_amd import training
        return training.fit(self, sequence, epochs=epochs, **kwargs)

    # As in BasicRS: dense3a / dense3b start with a Dense layer over a concatenation, which is linear in its two
    # halves — [a || b] . W = a . W[:da] + b . W[da:] — so each half is applied once per ENTITY at the end of the
    # corresponding first-stage network and the pair stage starts at act(A'[x] + B'[y]) (sum-input chain).
    def _fold(self, first_net, second_net, fuse_net, d_first):
        """Device tables-producing closures are avoided: returns (W_first_half, W_second_half, bias, rest layers)."""
        l0 = fuse_net.layers[0]
        w = l0.kernel.detach()
        return w[:d_first].contiguous(), w[d_first:].contiguous(), l0.bias.detach(), list(fuse_net.layers[1:]), l0.activation

    def _can_fold(self):
        d3 = [l.units for l in self.dense3a.layers]
        if self.fuse1a.method != 'concatenate':              # the first Dense of dense3a/3b then sees a mixed block, not a concatenation
            return False
        return (len(self.dense3a.layers) >= 2 and len(self.dense3b.layers) >= 2 and max(d3) <= capi.CHAIN_MAX_WIDTH
                and all(u % 4 == 0 for u in d3))

    def towers(self, ug_table, ig_table, ub_table, ib_table, ib_done=None):
        """Per-ENTITY outputs of the four first-stage networks (row-wise independent, hence hoistable); when the
        fused chain can run the second stage, each table is further multiplied by its half of dense3a/3b's first layer.
        ib_done: the item-side BERT table of the result, already computed (`item_bert_part` over row blocks, gathered by the
        partitioned runner): `ib_table` is then not read."""
        t = [self.dense1a.apply2(ug_table), self.dense1b.apply2(ig_table),
             self.dense2a.apply2(ub_table), self.dense2b.apply2(ib_table) if ib_done is None else None]
        if not self._can_fold():
            return (t[0], t[1], t[2], t[3] if ib_done is None else ib_done, False)
        if self.feature_based:
            pairs = ((0, 1, self.dense3a), (2, 3, self.dense3b))          # x1 = dense3a([ug || ig]), x2 = dense3b([ub || ib])
        else:
            pairs = ((0, 2, self.dense3a), (1, 3, self.dense3b))          # x1 = dense3a([ug || ub]), x2 = dense3b([ig || ib])
        folded = [None] * 4
        for ia, ib, net in pairs:
            wa, wb, bias, _, _ = self._fold(None, None, net, t[ia].shape[1])
            ya = torch.empty((t[ia].shape[0], wa.shape[1]), dtype=torch.float32, device=wa.device)
            capi.dense(t[ia], wa, None, ya, act=None)
            folded[ia] = ya
            if ib == 3 and ib_done is not None:
                folded[ib] = ib_done
                continue
            yb = torch.empty((t[ib].shape[0], wb.shape[1]), dtype=torch.float32, device=wb.device)
            capi.dense(t[ib], wb, bias, yb, act=None)
            folded[ib] = yb
        return (folded[0], folded[1], folded[2], folded[3], True)

    def item_bert_part(self, ib_rows):
        """The item-side BERT table `towers` would hold for these rows — dense2b, then (when the second stage is folded) its half of
        dense3b's first layer with the bias — on ANY block of item rows: row-wise independent, so a node-partitioned run computes
        it on every rank's own items and gathers the blocks instead of every rank running the 768-wide tower over all items."""
        t3 = self.dense2b.apply2(ib_rows)
        if not self._can_fold():
            return t3
        d_first = self.dense2a.output_units if self.feature_based else self.dense1b.output_units
        _, wb, bias, _, _ = self._fold(None, None, self.dense3b, d_first)
        yb = torch.empty((t3.shape[0], wb.shape[1]), dtype=torch.float32, device=wb.device)
        capi.dense(t3, wb, bias, yb, act=None)
        return yb

    def _rest(self, net):
        """Packed remaining layers (after the folded first one) of dense3a / dense3b, cached per weight version."""
        key = ('rest', id(net), self.weights_version)
        cache = self.__dict__.setdefault('_rest_cache', {})
        if cache.get('key_' + str(id(net))) != key:
            layers = list(net.layers[1:])
            blob, dims = capi.chain_pack([l.kernel.detach().cpu().numpy() for l in layers],
                                         [l.bias.detach().cpu().numpy() for l in layers])
            cache['key_' + str(id(net))] = key
            cache[id(net)] = (torch.from_numpy(blob).to(layers[0].kernel.device), dims, [l.activation for l in layers],
                              net.layers[0].activation)
        return cache[id(net)]

    def _dual_plan(self):
        """Blob + shapes for the fused two-branch kernel (rest of dense3a, rest of dense3b, classifier), or None."""
        key = self.weights_version
        cache = self.__dict__.get('_dual_cache')
        if cache is not None and cache[0] == key:
            return cache[1]
        plan = None
        ra, rb, clf = list(self.dense3a.layers[1:]), list(self.dense3b.layers[1:]), list(self.clf.layers)
        D = self.dense3a.layers[0].units
        same = all(l.units == D for l in ra + rb) and self.dense3b.layers[0].units == D and len(ra) == len(rb) and \
            self.dense3a.layers[0].activation == self.dense3b.layers[0].activation and \
            all(x.activation == y.activation for x, y in zip(ra, rb))
        trunk_dims = [2 * D] + [l.units for l in clf]
        if capi.dual_chain_supported(D, same, trunk_dims):
            np_ = lambda prm: prm.detach().cpu().numpy()
            blobs = []
            for layers in (ra, rb, clf):
                if layers:
                    blob, _ = capi.chain_pack([np_(l.kernel) for l in layers], [np_(l.bias) for l in layers])
                    blobs.append(torch.from_numpy(blob))
            plan = {'D': D, 'in_act': self.dense3a.layers[0].activation, 'branch_acts': [l.activation for l in ra],
                    'trunk_dims': trunk_dims, 'trunk_acts': [l.activation for l in clf],
                    'wpack': torch.cat(blobs).to(clf[0].kernel.device)}
        self.__dict__['_dual_cache'] = (key, plan)
        return plan

    def score_towers(self, towers, u_ids, i_ids, u_base=0, i_base=0, pair_plan=None):
        """dense3a / dense3b over the fused (concatenated) tower rows of each pair, then the classifier.  `pair_plan`
        (models/basic.py:PairPlan of the same id lists; used by the fused two-branch head only): the launch walks the list in the
        plan's XCD-affine order — four 256-byte rows per pair out of tables that together exceed the L2s — and the scores return to
        the caller's order in the plan's two steps."""
        tug, tig, tub, tib, folded = towers
        if self.feature_based:
            args1 = dict(ids_a=u_ids, base_a=u_base, ids_b=i_ids, base_b=i_base)
            args2 = args1
            in1, in2 = (tug, tig), (tub, tib)
        else:
            args1 = dict(ids_a=u_ids, base_a=u_base, ids_b=u_ids, base_b=u_base)
            args2 = dict(ids_a=i_ids, base_a=i_base, ids_b=i_ids, base_b=i_base)
            in1, in2 = (tug, tub), (tig, tib)
        plain = self.fuse2.method == 'concatenate' and self.residual is None
        if not folded:
            if self.fuse1a.method == 'concatenate':
                x1 = self.dense3a.apply2(in1[0], in1[1], **args1)
                x2 = self.dense3b.apply2(in2[0], in2[1], **args2)
            else:                                             # attention over the gathered rows of each pair
                xs = []
                for fuse, net, (ta, tb), args in ((self.fuse1a, self.dense3a, in1, args1), (self.fuse1b, self.dense3b, in2, args2)):
                    ga, gb = self._rows(ta, args['ids_a'], args['base_a']), self._rows(tb, args['ids_b'], args['base_b'])
                    xs.append(net.apply2(fuse([ga, gb])))
                x1, x2 = xs
            return self.clf.apply2(x1, x2) if plain else self._tail(x1, x2)
        plan = self._dual_plan() if plain else None
        if plan is not None:
            m = args1['ids_a'].numel() if args1['ids_a'] is not None else in1[0].shape[0]
            out = torch.empty((m, 1), dtype=torch.float32, device=in1[0].device)
            if pair_plan is not None and u_ids is not None and i_ids is not None:
                pair_plan.check(u_ids, i_ids)
                pu, pi = pair_plan.u_ids, pair_plan.i_ids
                ia, ib = ((pu, pu), (pi, pi)) if self.feature_based else ((pu, pi), (pu, pi))
                two_step = pair_plan.mid_index is not None
                capi.dual_chain((in1[0], in2[0]), (in1[1], in2[1]), ia, ib,
                                (args1['base_a'], args2['base_a']), (args1['base_b'], args2['base_b']),
                                plan['D'], plan['in_act'], plan['branch_acts'], plan['trunk_dims'], plan['trunk_acts'], plan['wpack'],
                                pair_plan.mid.view(-1, 1) if two_step else out,
                                out_index=pair_plan.mid_index if two_step else pair_plan.out_index)
                if two_step:
                    capi.scatter(pair_plan.mid, pair_plan.final_index, out, pair_plan.window_off, pair_plan.n_windows)
                return out
            capi.dual_chain((in1[0], in2[0]), (in1[1], in2[1]), (args1['ids_a'], args2['ids_a']), (args1['ids_b'], args2['ids_b']),
                            (args1['base_a'], args2['base_a']), (args1['base_b'], args2['base_b']),
                            plan['D'], plan['in_act'], plan['branch_acts'], plan['trunk_dims'], plan['trunk_acts'], plan['wpack'], out)
            return out
        outs = []
        for net, (ta, tb), args in ((self.dense3a, in1, args1), (self.dense3b, in2, args2)):
            blob, dims, acts, in_act = self._rest(net)
            m = args['ids_a'].numel() if args['ids_a'] is not None else ta.shape[0]
            x = torch.empty((m, dims[-1]), dtype=torch.float32, device=ta.device)
            capi.chain(ta, blob, dims, acts, x, B=tb, sum_inputs=True, in_act=in_act, **args)
            outs.append(x)
        return self.clf.apply2(outs[0], outs[1]) if plain else self._tail(outs[0], outs[1])

    @staticmethod
    def _rows(table, ids, base):
        if ids is None:
            return table
        rows = torch.empty((ids.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
        capi.copy_columns(table, rows, ids=ids, base=base)
        return rows

    def _tail(self, x1, x2):
        """hybrid.py:85-89 for the tweak configs: x = fuse2([x1, x2]) (attention or concatenation), then either the
        classifier, or clf(activation(residual(x) + x1 + x2))."""
        x = self.fuse2([x1, x2])
        if self.residual is None:
            return self.clf.apply2(x)
        r = self.residual.apply2(x)
        s = torch.empty_like(r)
        capi.add3_act(r, x1, x2, s, act=self.activation)
        return self.clf.apply2(s)


class HybridBertGNN(Model, abc.ABC):
    def __init__(
            self,
            dense_units=(32, 16),
            clf_units=(16, 16),
            feature_based=False,
            activation='relu',
            fusion_method='concatenate',
            residual=False,
            **kwargs
    ):
        super().__init__()
        self.rs = HybridCBRS(
            feature_based=feature_based,
            dense_units=dense_units,
            clf_units=clf_units,
            activation=activation,
            fusion_method=fusion_method,
            residual=residual
        )
        self.bert_table = None
        self.n_users = self.n_items = None
        self._towers = None
        self.built = True

    def set_bert_table(self, table):
        """Keep the [|U|+|I|, 768] BERT rows resident in HBM; batches may then pass ids only."""
        self.bert_table = to_device_tensor(table)

    def call(self, inputs, **kwargs):
        updated_embeddings = self.gnn(None)
        return self.embed_recommend(updated_embeddings, inputs)

    def embed_recommend(self, embeddings, inputs):
        """inputs = (user ids, item ids, user BERT block, item BERT block); the BERT blocks may be None
        when a resident table was registered with :meth:`set_bert_table`.  With a resident table, hoisted
        mode (predict) evaluates the four first-stage networks once per entity."""
        ug, ig, ub, ib = inputs
        u, i = ids_to_device(ug), ids_to_device(ig)
        if ub is None and ib is None:
            if self.bert_table is None:
                raise ValueError("no BERT blocks in the batch and no resident table registered")
            bert = self.bert_table
            if not self.rs.built:
                self.rs.build_head(embeddings.shape[1], bert.shape[1])
            if not self.gnn.hoist:
                return self.rs([embeddings, embeddings, bert, bert], g_ids=(u, i), b_ids=(u, i))
            key = (self.weights_version, embeddings.data_ptr(), bert.data_ptr())
            if self._towers is None or self._towers[0] != key:
                n = min(embeddings.shape[0], bert.shape[0])
                nu = self.n_users if self.n_users is not None else n
                lo = nu if self.n_users is not None else 0
                hi = nu + self.n_items if (self.n_users is not None and self.n_items is not None) else n
                self._towers = (key, self.rs.towers(embeddings[:nu], embeddings[lo:hi], bert[:nu], bert[lo:hi]), lo)
            return self.rs.score_towers(self._towers[1], u, i, 0, self._towers[2])
        return self.rs([embeddings, embeddings, ub, ib], g_ids=(u, i))

    def fit(self, sequence, epochs=1, **kwargs):
        """Keras ``fit``: BCE + L2 + Adam over the batches of `sequence` (training.py)."""
        from deep_cbrs_amar_renaissance_amd import training
        return training.fit(self, sequence, epochs=epochs, **kwargs)

    def resident_ids(self, sequence):
        """The reference's hybrid batch Sequence (data.datasets.UserItemGraphEmbeddings: ids + the BERT rows of the batch, gathered on the
        host from ONE table indexed by node id and uploaded every batch — 6 MB at batch 1 024) read as ids only: its table is registered
        on the device once (`set_bert_table`) and the ids Sequence inside it is returned; None for any other Sequence or with
        AMAR_RESIDENT_BERT=0 (the batches as they come).  The same rows either way, gathered on the device."""
        from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraphEmbeddings
        if os.environ.get('AMAR_RESIDENT_BERT', '1') == '0' or not isinstance(sequence, UserItemGraphEmbeddings):
            return None
        table = getattr(sequence.embeddings, 'embeddings', None)
        if not (isinstance(table, np.ndarray) and table.ndim == 2 and table.shape[0] >= len(sequence.users) + len(sequence.items)):
            return None
        if getattr(self, '_bert_table_source', None) is not table:
            self.set_bert_table(np.ascontiguousarray(table, dtype=np.float32))
            self._bert_table_source = table
        return sequence.graph_ids

    def predict(self, sequence, hoist=True, **kwargs):
        """Model.predict; the reference's hybrid Sequence is read as ids against the resident table (`resident_ids`), so that the hoisted
        pass evaluates the first-stage networks once per entity instead of once per pair of every batch."""
        ids = self.resident_ids(sequence)
        if ids is not None:
            sequence = _IdsWithoutBlocks(ids)
        return super().predict(sequence, hoist=hoist, **kwargs)

    def _hoist_begin(self, hoist):
        self.gnn.hoist = bool(hoist)

    def _hoist_end(self):
        self.gnn.hoist = False
        self.gnn._hoisted = None
        self._towers = None


class _IdsWithoutBlocks:
    """A Sequence of (user ids, item ids) batches seen as hybrid batches whose BERT blocks are None (taken from the resident table)."""

    def __init__(self, ids_sequence):
        self.ids = ids_sequence

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, b):
        (u, i), y = self.ids[b]
        return (u, i, None, None), y


def BasicGNNFactory(name, Parent, GNN):
    def __init__(self, *args, **kwargs):
        Parent.__init__(self, **kwargs)
        self.gnn = self.gnn_class(*args, **kwargs)
        self.gnn.build_layers()

    return type(name, (Parent,), {"gnn_class": GNN, "__init__": __init__})


class HybridBertTSGNN(HybridBertGNN):
    pass


class HybridBertTWGNN(HybridBertGNN):
    pass


HYBRID_GNNS = [
    (HybridBertGNN, [GCN, GAT, GraphSage, LightGCN, DGCF], None),
    (HybridBertTSGNN, [TwoStepGCN, TwoStepGraphSage, TwoStepGAT, TwoStepLightGCN, TwoStepDGCF], lambda name: 'HybridBertTS' + name[7:]),
    (HybridBertTWGNN, [TwoWayGCN, TwoWayGraphSage, TwoWayGAT, TwoWayLightGCN, TwoWayDGCF], lambda name: 'HybridBertTW' + name[6:]),
]


def generate_hybrids():
    for parent, gnns, name_getter in HYBRID_GNNS:
        for gnn in gnns:
            name = name_getter(gnn.__name__) if name_getter is not None else 'HybridBert' + gnn.__name__
            globals()[name] = BasicGNNFactory(name, parent, gnn)


generate_hybrids()
