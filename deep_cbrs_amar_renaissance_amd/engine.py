"""Minimal Keras-protocol layer/model base on top of torch tensors.

The reference's driver talks to its models through the Keras protocol (`experiment.py:155-198`:
``compile``, ``model(batch)``, ``summary``, ``trainable_weights``, ``fit``, ``evaluate``,
``predict``).  This module provides that protocol for models whose arithmetic is the HIP
library behind :mod:`deep_cbrs_amar_renaissance_amd.capi`; torch only owns the weights
(device memory) and the stream.  Weights are created lazily on the first call, like Keras
``build``, from a seeded numpy generator so that a given seed gives the same weights on every
machine (`experiment.py:56` seeds TensorFlow the same way; the streams differ, the
distributions — glorot_uniform kernels, zero biases — do not).
"""
import os

import numpy as np
import torch

_SEED = 42
_RNG = np.random.default_rng(_SEED)


def set_seed(seed):
    """Counterpart of ``tf.random.set_seed(config.seed)`` (experiment.py:56)."""
    global _SEED, _RNG
    _SEED = int(seed)
    _RNG = np.random.default_rng(_SEED)


def default_device():
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


def glorot_uniform(shape, rng=None):
    """Keras GlorotUniform with `_compute_fans`: leading dims of rank>2 kernels are a receptive field."""
    shape = tuple(int(s) for s in shape)
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return (rng or _RNG).uniform(-limit, limit, size=shape).astype(np.float32)


class L2:
    """keras.regularizers.l2 stand-in: carries the factor; training.py adds l2 * sum(w^2) to the loss / 2 l2 w to the gradient."""
    def __init__(self, l2=0.01):
        self.l2 = float(l2)


class Layer(torch.nn.Module):
    """A Keras-like layer: weights appear at the first call (``build``), then ``call`` runs."""

    def __init__(self):
        super().__init__()
        self.built = False

    def add_weight(self, name, shape, initializer='glorot_uniform', regularizer=None, trainable=True):
        if initializer == 'glorot_uniform':
            value = glorot_uniform(shape)
        elif initializer == 'zeros':
            value = np.zeros(shape, dtype=np.float32)
        elif initializer == 'ones':
            value = np.ones(shape, dtype=np.float32)
        else:
            raise ValueError("Unknown initializer {}".format(initializer))
        p = torch.nn.Parameter(torch.from_numpy(value).to(default_device()), requires_grad=trainable)
        p.regularizer = regularizer
        self.__dict__.pop(name, None)                 # a `self.kernel = None` placeholder set in __init__
        self.register_parameter(name, p)
        return p

    def build(self, input_shape):
        pass

    def call(self, inputs, **kwargs):
        raise NotImplementedError

    @staticmethod
    def _shape_of(inputs):
        if isinstance(inputs, (list, tuple)):
            return [Layer._shape_of(i) for i in inputs]
        return tuple(inputs.shape) if hasattr(inputs, 'shape') else None

    def forward(self, inputs=None, **kwargs):
        if not self.built:
            self.build(self._shape_of(inputs))
            self.built = True
        with torch.no_grad():
            return self.call(inputs, **kwargs)

    @property
    def trainable_weights(self):
        return [p for p in self.parameters() if p.requires_grad]

    @property
    def non_trainable_weights(self):
        return [p for p in self.parameters() if not p.requires_grad]

    @property
    def weights_version(self):
        """Changes whenever any weight is modified in place or replaced (hoisting cache key)."""
        return tuple((id(p), p._version) for p in self.parameters())


def to_device_tensor(x, dtype=torch.float32):
    """numpy / torch input -> contiguous device tensor of `dtype` (no copy if already there)."""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x))
    return t.to(device=default_device(), dtype=dtype).contiguous()


def ids_to_device(ids, n_rows=None):
    """Host int64 ids (numpy default, datasets.py:203) -> device int32.  Host arrays are range-checked here — against the
    table height `n_rows` where the caller knows it: the kernels gather / scatter rows by id without a bounds check, so an
    id past the table would be an out-of-bounds device access where TensorFlow's embedding_lookup raises.  Device tensors are
    taken as they are (checking them would force a synchronisation); producers of device ids own their range."""
    if isinstance(ids, torch.Tensor):
        t = ids
    else:
        t = torch.from_numpy(np.ascontiguousarray(ids))
    if not t.is_cuda and t.numel():
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= 2 ** 31:
            raise ValueError("ids must be in [0, 2^31)")
        if n_rows is not None and hi >= n_rows:
            raise IndexError("id {} is out of range for a table of {} rows".format(hi, n_rows))
    return t.to(device=default_device(), dtype=torch.int32).contiguous()


def stage_ids(dst_host, ids, n_rows=None):
    """Host ids -> the int32 HOST tensor `dst_host` (pinned staging memory of a replayed training batch), range-checked like
    ids_to_device: the kernels gather / scatter rows by id without a bounds check."""
    a = ids.numpy() if isinstance(ids, torch.Tensor) else np.asarray(ids)
    if a.size:
        lo, hi = int(a.min()), int(a.max())
        if lo < 0 or hi >= 2 ** 31:
            raise ValueError("ids must be in [0, 2^31)")
        if n_rows is not None and hi >= n_rows:
            raise IndexError("id {} is out of range for a table of {} rows".format(hi, n_rows))
    dst_host.numpy()[...] = a


def _drain_collective_watchdog():
    """With an RCCL process group alive, wait until its watchdog thread has retired every collective issued so far.

    torch's ProcessGroupNCCL keeps each eager collective's Work in a list that a watchdog thread walks every 100 ms, asking the
    Work's end event whether it has completed (hipEventQuery).  While a stream capture in torch's default (global) error mode is in
    progress, HIP refuses that query from ANY thread ("operation not permitted when stream is capturing"), the watchdog rethrows
    and the process aborts — a race between the watchdog's period and the start of the capture that synchronising the device does
    not close (the events are complete, but the list still holds them until the next walk).  Collectives issued INSIDE a capture are
    not put on that list.  Three watchdog periods of sleep after the device is idle empty it."""
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or 'nccl' not in str(dist.get_backend()).lower():
            return
    except Exception:                                   # (no process group of this kind: nothing to wait for)
        return
    import time
    time.sleep(0.35)


def capture_graph(fn):
    """Capture `fn()`'s launches on the current device into a hipGraph; returns (graph, fn's result).

    The cyclic garbage collector is held off for the duration of the capture: a collection that happens to free an OLDER
    graph (a model that went out of use still holds its predict / training graphs) would call hipGraphExecDestroy in the
    middle of the capture, which HIP rejects — the process aborts."""
    import gc
    torch.cuda.synchronize()
    _drain_collective_watchdog()
    gc.collect()
    graph = torch.cuda.CUDAGraph()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph):
            out = fn()
    finally:
        if was_enabled:
            gc.enable()
    return graph, out



class _RenamedArchive:
    """An .npz archive seen under other key names (Model.load_weights(name_map=...))."""

    def __init__(self, archive, new_to_old):
        self._archive, self._map = archive, dict(new_to_old)
        self.files = list(self._map)

    def __getitem__(self, key):
        return self._archive[self._map[key]]


class Model(Layer):
    """Keras ``Model`` protocol used by the reference driver (experiment.py:155-198)."""

    def compile(self, loss=None, optimizer=None, metrics=None):
        self.loss, self.optimizer, self.metrics = loss, optimizer, list(metrics or [])

    def summary(self, print_fn=print, expand_nested=False):
        print_fn('Model: "{}"'.format(type(self).__name__))
        rows = self.named_parameters() if expand_nested else \
            ((n, p) for n, p in self.named_parameters())
        for name, p in rows:
            print_fn('  {:<48s} {:>18s} {:>10d}'.format(name, str(tuple(p.shape)), p.numel()))
        trainable = sum(p.numel() for p in self.trainable_weights)
        print_fn('Trainable params: {}'.format(trainable))
        print_fn('Non-trainable params: {}'.format(sum(p.numel() for p in self.non_trainable_weights)))

    def fit(self, *args, **kwargs):
        raise NotImplementedError(
            "fit() is implemented by the Basic* / HybridBert* GNN models (training.py: BCE + L2 + Adam reverse pass); "
            "{} has no training path".format(type(self).__name__))

    # -- inference ---------------------------------------------------------------------------
    def _predict_batches(self, sequence):
        outs = []
        for b in range(len(sequence)):
            inputs, _ = sequence[b]
            outs.append(self(inputs))
        return torch.cat(outs, dim=0) if outs else torch.empty((0, 1), device=default_device())

    def predict(self, sequence, hoist=True, graph=None, **kwargs):
        """Scores for every pair of `sequence` as an ``ndarray[P, 1]`` (experiment.py:197-198).

        hoist=True runs the (input-independent, gnn.py:263-264) graph propagation once for the
        whole call; hoist=False re-runs it for every batch exactly like the reference does.

        graph (default on; AMAR_PREDICT_GRAPH=0 or graph=False turns it off): models whose batches are id pairs only replay
        the whole call from a captured hipGraph — at ML-1M size the launches of a predict pass are mostly gaps (hoisted:
        8 launches, 0.13 ms eager vs 0.065 ms replayed; per-batch: ~650 launches).  The ids are uploaded once per Sequence,
        the capture is redone when the batch sizes, the mode or any weight changes (a reshuffled Sequence of the same batch
        sizes only refreshes the id buffers), and replayed scores equal the eager ones bit for bit (tests/test_models_gpu.py).
        """
        if graph is None:
            graph = os.environ.get('AMAR_PREDICT_GRAPH', '1') != '0'
        if graph and self._graph_predict_supported(sequence):
            return self._predict_graphed(sequence, bool(hoist)).cpu().numpy()
        self._hoist_begin(hoist)
        try:
            return self._predict_batches(sequence).cpu().numpy()
        finally:
            self._hoist_end()

    # -- predict() replayed from a hipGraph ----------------------------------------------------
    def _graph_predict_supported(self, sequence):
        return False

    def _sequence_ids(self, sequence):
        """All (user, item) id pairs of a Sequence on the device.  The host ids are read on every call (a shuffling Sequence
        changes its order between epochs, datasets.py:205-213) and compared with what the device buffers hold; a changed list of
        the same batch sizes is copied INTO those buffers, so that a captured graph reading them stays valid.  A Sequence that
        carries an `order_version` counter (data/datasets.py bumps it on every reshuffle) is not re-read while the counter, the
        object and its length stay the same: at ML-1M size reading 81 batches costs more than the replayed pass itself."""
        cache = self.__dict__.get('_seq_ids')
        stamp = (id(sequence), len(sequence), getattr(sequence, 'order_version', None), id(getattr(sequence, 'ratings', None)))
        if cache is not None and stamp[2] is not None and cache.get('stamp') == stamp:
            return cache['u'], cache['i'], cache['sizes']
        us, its, sizes = [], [], []
        for b in range(len(sequence)):
            (u, i), _ = sequence[b]
            us.append(np.asarray(u))
            its.append(np.asarray(i))
            sizes.append(len(us[-1]))
        u_host = np.concatenate(us).astype(np.int64) if us else np.zeros(0, np.int64)
        i_host = np.concatenate(its).astype(np.int64) if its else np.zeros(0, np.int64)
        if cache is not None and cache['sizes'] == sizes:
            if not (np.array_equal(cache['u_host'], u_host) and np.array_equal(cache['i_host'], i_host)):
                cache['u'].copy_(ids_to_device(u_host))
                cache['i'].copy_(ids_to_device(i_host))
                cache['u_host'], cache['i_host'] = u_host, i_host
        else:
            cache = {'sizes': sizes, 'u_host': u_host, 'i_host': i_host, 'u': ids_to_device(u_host), 'i': ids_to_device(i_host)}
            self.__dict__['_seq_ids'] = cache
        cache['stamp'], cache['keepalive'] = stamp, sequence          # (the object is kept alive: its id() is part of the stamp)
        return cache['u'], cache['i'], sizes

    def _predict_graphed(self, sequence, hoist):
        u_all, i_all, sizes = self._sequence_ids(sequence)
        if u_all.numel() == 0:
            return torch.empty((0, 1), device=default_device())

        def run():
            self._hoist_begin(hoist)
            try:
                if hoist:
                    return self((u_all, i_all))                          # one propagation, towers once, every pair in one launch
                outs, lo = [], 0
                for n in sizes:                                          # basic.py:61-63: propagation + scoring per batch
                    outs.append(self((u_all[lo:lo + n], i_all[lo:lo + n])))
                    lo += n
                return torch.cat(outs, dim=0)
            finally:
                self._hoist_end()

        # weight VALUES are part of the key: the Dense stacks' weights reach the kernels as blobs packed on the host (amar_chain_pack_f32),
        # which a replay cannot redo — any weight update (fit() bumps the version counters) re-captures on the next predict()
        key = (hoist, u_all.data_ptr(), i_all.data_ptr(), tuple(sizes), self.weights_version,
               tuple(p.data_ptr() for p in self.parameters()))
        cached = self.__dict__.get('_predict_graph')
        if cached is None or cached[0] != key:
            run()                                                        # eager once: lazy builds (graph images, packed weights, kernel attributes)
            g, out = capture_graph(run)
            cached = (key, g, out)
            self.__dict__['_predict_graph'] = cached
        cached[1].replay()
        return cached[2]

    def _hoist_begin(self, hoist):
        pass

    def _hoist_end(self):
        pass

    # -- weights export / import -----------------------------------------------------------------
    def keras_variable_names(self):
        """[(name, parameter)] for every weight under a REPO-NATIVE naming scheme in Keras' style: '<layer>/<variable>:0', the layer
        name being the class name in snake_case with a per-model, per-class counter in creation order ('dense', 'dense_1', ...,
        'gcn_conv', 'gcn_conv_1', 'sequential_gnn').  These are NOT the keys of a tf.keras checkpoint of the reference: Keras puts the
        variables of nested subclassed models under nested name scopes ('basic_gcn/basic_rs/sequential/dense/kernel:0') and numbers
        layers with counters that are global to the session.  The archive matches a model of the same architecture by these names,
        shapes and order; importing real Keras weights needs a name map (`load_weights(path, name_map=...)`).  The reference keeps its
        trained models through mlflow.tensorflow.autolog() (utilities/utils.py:108); none of its checkpoints is available here."""
        import re
        counters, out = {}, []
        for module in self.modules():
            own = list(module.named_parameters(recurse=False))
            if not own:
                continue
            base = re.sub(r'(?<=[a-z0-9])(?=[A-Z])|(?<=[A-Z])(?=[A-Z][a-z])', '_', type(module).__name__).lower()
            k = counters.get(base, 0)
            counters[base] = k + 1
            layer = base if k == 0 else '{}_{}'.format(base, k)
            out.extend(('{}/{}:0'.format(layer, name), prm) for name, prm in own)
        return out

    def save_weights(self, path):
        """Write every weight to an .npz archive keyed by `keras_variable_names()` (fp32, host byte order)."""
        arrays = {name: prm.detach().cpu().numpy() for name, prm in self.keras_variable_names()}
        np.savez(path if str(path).endswith('.npz') else str(path) + '.npz', **arrays)

    def load_weights(self, path, name_map=None):
        """Read the archive `save_weights` wrote into this model's weights (same architecture: names and shapes must match).
        `name_map`: {archive key: this model's name} (or a callable key -> name) for archives keyed differently — e.g. arrays exported
        from a real tf.keras model of the reference, whose variable names carry nested scopes and session-global counters.
        Every weight's version counter moves, so hoisted tables, packed Dense blobs and captured graphs are rebuilt on next use."""
        z = np.load(path if str(path).endswith('.npz') else str(path) + '.npz')
        if name_map is not None:
            rename = name_map if callable(name_map) else (lambda k: name_map.get(k, k))
            z = _RenamedArchive(z, {rename(k): k for k in z.files})
        names = self.keras_variable_names()
        missing = [n for n, _ in names if n not in z.files]
        extra = sorted(set(z.files) - {n for n, _ in names})
        if missing or extra:
            raise ValueError("load_weights: the archive does not match this model (missing {}, unexpected {})".format(missing[:4], extra[:4]))
        with torch.no_grad():
            for name, prm in names:
                value = z[name]
                if tuple(value.shape) != tuple(prm.shape):
                    raise ValueError("load_weights: {} is {} in the archive, {} in the model".format(name, tuple(value.shape), tuple(prm.shape)))
                prm.copy_(torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32)).to(prm.device))

    def evaluate(self, sequence, **kwargs):
        """Loss and accuracy on `sequence` (experiment.py:194).  As Keras' evaluate(), 'loss' is the binary cross-entropy plus
        the regularisation losses of the model (l2 * sum(w^2) for every weight carrying a regulariser: gnn.py:45,293-294),
        the same sum fit() reports per epoch, so train and test losses of one run are comparable."""
        pred = self.predict(sequence).reshape(-1).astype(np.float64)
        y = np.concatenate([np.asarray(sequence[b][1]).reshape(-1) for b in range(len(sequence))]).astype(np.float64)
        eps = 1e-7                                                   # keras backend epsilon
        p = np.clip(pred, eps, 1 - eps)
        loss = float(-np.mean(y * np.log(p) + (1 - y) * np.log(1 - p))) if len(y) else 0.0
        for w in self.parameters():
            reg = getattr(w, 'regularizer', None)
            if reg is not None and getattr(reg, 'l2', 0.0):
                loss += float(reg.l2) * float((w.detach().double() ** 2).sum())
        acc = float(np.mean((pred > 0.5) == (y > 0.5))) if len(y) else 0.0
        return [loss, acc]
