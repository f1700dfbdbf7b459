"""Oracle for the training step.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what Keras does for one batch of `Experimenter.train` (experiment.py:155-188, config.yaml:50-58):

    forward   BasicGNN.call (basic.py:61-75): full-graph propagation, lookup, BasicRS
    loss      binary_crossentropy (Keras backend form, epsilon 1e-7, mean over the batch)
              + L2 regularisers l2 * sum(w^2) on the node table and the conv kernels / biases
                (gnn.py:45, 293-294; Dense layers of the head carry none)
    backward  manual reverse pass below; `torch_grads` is an independent torch-autograd check
    update    Adam(learning_rate, beta_1, beta_2=0.999, epsilon=1e-7) as keras.optimizers.Adam applies it:
              lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  w -= lr_t * m / (sqrt(v) + eps)

Covers the GCN and LightGCN stacks with the 'concatenation' / 'mean' reductions of the BASELINE configs.
"""
import numpy as np

from oracle import graph as ograph

EPS = 1e-7


def _dense_fwd(x, net, acts):
    cache = []
    for (w, b), act in zip(net, acts):
        z = x @ w + b
        y = np.maximum(z, 0) if act == 'relu' else (1.0 / (1.0 + np.exp(-z)) if act == 'sigmoid' else z)
        cache.append((x, y))
        x = y
    return x, cache


def _dense_bwd(dy, net, acts, cache):
    grads = []
    for (w, b), act, (x, y) in zip(reversed(net), reversed(acts), reversed(cache)):
        dz = dy * (y > 0) if act == 'relu' else (dy * y * (1 - y) if act == 'sigmoid' else dy)
        grads.append((x.T @ dz, dz.sum(0)))
        dy = dz @ w.T
    return dy, list(reversed(grads))


def loss_and_grads(adj, gnn, head, u_ids, i_ids, y, l2=0.0, dtype=np.float64):
    """Returns (loss, grads) with grads shaped like the weight containers of oracle/models.py."""
    kind = gnn['kind']
    a_hat = ograph.gcn_filter(adj).astype(dtype)
    x0 = gnn['embeddings'].astype(dtype)
    cast = lambda net: [(w.astype(dtype), b.astype(dtype)) for w, b in net]
    unet, inet, clf = cast(head['unet']), cast(head['inet']), cast(head['clf'])
    # ---- forward
    hs, pre = [x0], []
    x = x0
    for lw in gnn['layers']:
        if kind == 'gcn':
            h = x @ lw['kernel'].astype(dtype)
            x_new = np.maximum(a_hat @ h + lw['bias'].astype(dtype), 0)
            pre.append(x)
        else:
            x_new = a_hat @ x
        hs.append(x_new)
        x = x_new
    if kind == 'gcn':
        e = np.concatenate(hs, axis=1)
    else:
        e = sum(hs) / len(hs)
    tower_acts = ['relu'] * len(unet)
    clf_acts = ['relu'] * (len(clf) - 1) + ['sigmoid']
    tu, cu = _dense_fwd(e[u_ids], unet, tower_acts)
    ti, ci = _dense_fwd(e[i_ids], inet, tower_acts)
    p, cc = _dense_fwd(np.concatenate([tu, ti], axis=1), clf, clf_acts)
    p = p[:, 0]
    yv = np.asarray(y, dtype=dtype)
    pc = np.clip(p, EPS, 1 - EPS)
    bce = -np.mean(yv * np.log(pc + EPS) + (1 - yv) * np.log(1 - pc + EPS))
    reg = l2 * np.sum(x0 * x0)
    if kind == 'gcn':
        for lw in gnn['layers']:
            reg += l2 * (np.sum(lw['kernel'].astype(dtype) ** 2) + np.sum(lw['bias'].astype(dtype) ** 2))
    loss = bce + reg
    # ---- backward
    inside = (p >= EPS) & (p <= 1 - EPS)
    dp = -(yv / (pc + EPS) - (1 - yv) / (1 - pc + EPS)) / len(p) * inside
    dcat, g_clf = _dense_bwd(dp[:, None], clf, clf_acts, cc)
    d = tu.shape[1]
    dgu, g_unet = _dense_bwd(dcat[:, :d], unet, tower_acts, cu)
    dgi, g_inet = _dense_bwd(dcat[:, d:], inet, tower_acts, ci)
    de = np.zeros_like(e)
    np.add.at(de, u_ids, dgu)
    np.add.at(de, i_ids, dgi)
    g_layers = []
    if kind == 'gcn':
        widths = [h.shape[1] for h in hs]
        offs = np.cumsum([0] + widths)
        dxs = [de[:, offs[k]:offs[k + 1]].copy() for k in range(len(hs))]
        for k in range(len(gnn['layers']) - 1, -1, -1):
            lw = gnn['layers'][k]
            dz = dxs[k + 1] * (hs[k + 1] > 0)
            dh = a_hat.T @ dz
            g_layers.append({'kernel': pre[k].T @ dh + 2 * l2 * lw['kernel'].astype(dtype),
                             'bias': dz.sum(0) + 2 * l2 * lw['bias'].astype(dtype)})
            dxs[k] = dxs[k] + dh @ lw['kernel'].astype(dtype).T
        g_layers.reverse()
        dx0 = dxs[0]
    else:
        g = de / len(hs)
        dx0 = g.copy()
        acc = g
        for _ in gnn['layers']:
            acc = a_hat.T @ acc
            dx0 = dx0 + acc
        # d/dX0 of (X0 + A X0 + A^2 X0 + ...)/n summed term by term: g + A^T g + (A^T)^2 g + ...
        g_layers = [{} for _ in gnn['layers']]
    grads = {'gnn': {'embeddings': dx0 + 2 * l2 * x0, 'layers': g_layers},
             'head': {'unet': g_unet, 'inet': g_inet, 'clf': g_clf}}
    return float(loss), grads, p


def adam_update(w, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """One keras.optimizers.Adam step (t = 1 for the first step); returns (w, m, v)."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return w - lr_t * m / (np.sqrt(v) + eps), m, v


def torch_grads(adj, gnn, head, u_ids, i_ids, y, l2=0.0):
    """Independent check: the same loss written with torch ops (float64, CPU) and differentiated by autograd."""
    import torch
    a = ograph.gcn_filter(adj).tocoo()
    a_t = torch.sparse_coo_tensor(np.stack([a.row, a.col]), a.data.astype(np.float64), a.shape).coalesce()
    T = lambda arr: torch.tensor(np.asarray(arr, dtype=np.float64), requires_grad=True)
    x0 = T(gnn['embeddings'])
    layers = [{k: T(v) for k, v in lw.items()} for lw in gnn['layers']]
    nets = {name: [(T(w), T(b)) for w, b in head[name]] for name in ('unet', 'inet', 'clf')}
    hs, x = [x0], x0
    for lw in layers:
        x = torch.relu(torch.sparse.mm(a_t, x @ lw['kernel']) + lw['bias']) if gnn['kind'] == 'gcn' else torch.sparse.mm(a_t, x)
        hs.append(x)
    e = torch.cat(hs, 1) if gnn['kind'] == 'gcn' else sum(hs) / len(hs)

    def run(net, v, last_sigmoid=False):
        for k, (w, b) in enumerate(net):
            v = v @ w + b
            v = torch.sigmoid(v) if (last_sigmoid and k == len(net) - 1) else torch.relu(v)
        return v
    u = torch.as_tensor(np.asarray(u_ids), dtype=torch.long)
    i = torch.as_tensor(np.asarray(i_ids), dtype=torch.long)
    p = run(nets['clf'], torch.cat([run(nets['unet'], e[u]), run(nets['inet'], e[i])], 1), True)[:, 0]
    yv = torch.tensor(np.asarray(y, dtype=np.float64))
    pc = torch.clamp(p, EPS, 1 - EPS)
    loss = -torch.mean(yv * torch.log(pc + EPS) + (1 - yv) * torch.log(1 - pc + EPS)) + l2 * (x0 ** 2).sum()
    if gnn['kind'] == 'gcn':
        for lw in layers:
            loss = loss + l2 * ((lw['kernel'] ** 2).sum() + (lw['bias'] ** 2).sum())
    loss.backward()
    g = lambda t: t.grad.numpy() if t.grad is not None else np.zeros(tuple(t.shape))
    return float(loss), {'gnn': {'embeddings': g(x0), 'layers': [{k: g(v) for k, v in lw.items()} for lw in layers]},
                         'head': {name: [(g(w), g(b)) for w, b in nets[name]] for name in nets}}


def _torch_stack(adj, x, st, self_loops=True, force_mean=True):
    """One convolution stack (the layer loop of gnn.py:74-84 / 141-150 / 197-207) in differentiable float64 torch ops,
    restating oracle/layers.py op by op.  st = {'kind', 'layers': leaf tensors, 'final_node'}."""
    import torch
    kind, layers = st['kind'], st['layers']
    n = x.shape[0]
    hs = [x]
    np_dt = np.float32 if x.dtype == torch.float32 else np.float64   # float64 is the oracle; float32 only sizes the rounding floor
    if kind in ('gcn', 'lightgcn'):
        a = ograph.gcn_filter(adj).tocoo()
        a_t = torch.sparse_coo_tensor(np.stack([a.row, a.col]), a.data.astype(np_dt), a.shape).coalesce()
        for lw in layers:
            x = torch.relu(torch.sparse.mm(a_t, x @ lw['kernel']) + lw['bias']) if kind == 'gcn' else torch.sparse.mm(a_t, x)
            hs.append(x)
    elif kind == 'dgcf':
        a = ograph.dgcf_adjacency(adj).tocoo()
        a_t = torch.sparse_coo_tensor(np.stack([a.row, a.col]), a.data.astype(np_dt), a.shape).coalesce()
        for lw in layers:
            x = torch.sparse.mm(a_t, x * torch.sigmoid(lw['w']))
            hs.append(x)
    else:
        row, col, _ = ograph.reordered_coo(adj)
        if self_loops:
            row, col = ograph.add_self_loops_edges(row, col, n)
        src = torch.as_tensor(np.asarray(row), dtype=torch.long)
        tgt = torch.as_tensor(np.asarray(col), dtype=torch.long)
        count = torch.bincount(tgt, minlength=n).to(x.dtype).clamp(min=1.0)
        for lw in layers:
            if kind == 'sage':
                agg = torch.zeros_like(x).index_add(0, tgt, x[src]) / count[:, None]
                z = torch.cat([x, agg], 1) @ lw['kernel'] + lw['bias']
                z = z * torch.rsqrt(torch.clamp((z * z).sum(1, keepdim=True), min=1e-12))
                x = torch.relu(z)
            else:
                h = x @ lw['kernel']
                e = (h @ lw['attn_self'])[tgt] + (h @ lw['attn_neigh'])[src]
                e = torch.where(e > 0, e, 0.2 * e)
                seg_max = torch.full((n,), -float('inf'), dtype=x.dtype).scatter_reduce(0, tgt, e.detach(), 'amax')
                ex = torch.exp(e - seg_max[tgt])
                denom = torch.zeros(n, dtype=x.dtype).index_add(0, tgt, ex) + 1e-9
                alpha = ex / denom[tgt]
                x = torch.relu(torch.zeros_like(h).index_add(0, tgt, alpha[:, None] * h[src]) + lw['bias'])
            hs.append(x)
    final_node = st.get('final_node', 'concatenation')
    if force_mean and kind in ('lightgcn', 'dgcf'):
        final_node = 'mean'
    if final_node == 'concatenation':
        return torch.cat(hs, 1)
    if final_node == 'last':
        return hs[-1]
    if final_node == 'w-sum':                                        # reduction.py:54-55
        w = st['reduction_w'].reshape(-1)
        return sum((w[k] * w[k]) * h for k, h in enumerate(hs))
    return sum(hs) / (len(hs) if final_node == 'mean' else 1)


def torch_model_grads(adj, gnn, head, u_ids, i_ids, y, l2=0.0, self_loops=True, bert=None, feature_based=True, n_users=None,
                      n_items=None, dtype=np.float64):
    """Loss and gradients of ANY model on the path by torch autograd (float64, CPU): the four GNN kinds of
    oracle/models.py:propagate under the Basic head (keys unet/inet/clf) or the Hybrid head (dense1a..dense3b, clf;
    `bert` = (user rows [B, D], item rows [B, D]) as the batch Sequence delivers them, hybrid.py:119-140).
    Compound layouts: gnn = {'step_one', 'step_two'} with adj = (user-item, item-property) is a TwoStep model, gnn =
    {'way_one', 'way_two', 'step_two'} with adj = (user-item, item-property, user-property) a TwoWay one (n_users, n_items
    required); their gradients come back as grads['gnn'][stack name].

    The forward below restates oracle/layers.py op by op with differentiable torch ops (tests check that its scores
    equal the numpy oracle's); what Keras adds is autodiff of exactly this graph plus the L2 terms of the node table
    and of the conv kernels / biases (gnn.py:45, 293-294, 324-327, 357-360; attention vectors carry none).
    """
    import torch
    T = lambda arr: torch.tensor(np.asarray(arr, dtype=dtype), requires_grad=True)
    fusers = {name: {k: T(v) for k, v in head[name].items()} for name in head if name.startswith('fuse')}
    nets = {name: [(T(w), T(b)) for w, b in head[name]] for name in head if not name.startswith('fuse')}
    # every stack as a dict of leaf tensors — compound layouts chain them (tsgnn.py:99-101, twgnn.py:98-105)
    stacks = {}

    def leaf(name, w, table_l2):
        t = {'kind': w['kind'], 'final_node': w.get('final_node', 'concatenation'),
             'layers': [{k: T(v) for k, v in lw.items()} for lw in w['layers']], 'table_l2': table_l2}
        if 'embeddings' in w:
            t['embeddings'] = T(w['embeddings'])
        if t['final_node'] == 'w-sum':                               # gnn.py:62 builds ReductionLayer(final_node) without a regulariser
            t['reduction_w'] = T(w['reduction_w'] if w.get('reduction_w') is not None else np.ones(len(w['layers']) + 1))
        stacks[name] = t
        return t
    if 'step_one' in gnn:
        adj_ui, adj_kg = adj
        one, two = leaf('step_one', gnn['step_one'], True), leaf('step_two', gnn['step_two'], False)   # tsgnn.py:77-81: no regulariser on the user table
        x = _torch_stack(adj_kg, one['embeddings'], one, self_loops, False)
        e_all = _torch_stack(adj_ui, torch.cat([two['embeddings'], x[:n_items]], 0), two, self_loops, False)
    elif 'way_one' in gnn:
        adj_ui, adj_ip, adj_up = adj
        one, two, three = leaf('way_one', gnn['way_one'], True), leaf('way_two', gnn['way_two'], True), leaf('step_two', gnn['step_two'], False)
        users = _torch_stack(adj_up, one['embeddings'], one, self_loops, False)
        items = _torch_stack(adj_ip, two['embeddings'], two, self_loops, False)
        e_all = _torch_stack(adj_ui, torch.cat([users[:n_users], items[:n_items]], 0), three, self_loops, False)
    else:
        only = leaf('gnn', gnn, True)
        e_all = _torch_stack(adj, only['embeddings'], only, self_loops, True)

    def run(net, v, last_sigmoid=False):
        for k, (w, b) in enumerate(net):
            v = v @ w + b
            v = torch.sigmoid(v) if (last_sigmoid and k == len(net) - 1) else torch.relu(v)
        return v
    u = torch.as_tensor(np.asarray(u_ids), dtype=torch.long)
    i = torch.as_tensor(np.asarray(i_ids), dtype=torch.long)
    if 'unet' in nets:
        p = run(nets['clf'], torch.cat([run(nets['unet'], e_all[u]), run(nets['inet'], e_all[i])], 1), True)[:, 0]
    else:
        ub = torch.tensor(np.asarray(bert[0], dtype=dtype))
        ib = torch.tensor(np.asarray(bert[1], dtype=dtype))
        g1, g2, b1, b2 = run(nets['dense1a'], e_all[u]), run(nets['dense1b'], e_all[i]), run(nets['dense2a'], ub), run(nets['dense2b'], ib)

        def fuse(name, a, b):                                        # fusion.py:51-68
            if name not in fusers:
                return torch.cat([a, b], 1)
            fw = fusers[name]
            if 'proj_weight' in fw:
                a, b = (a @ fw['proj_weight'], b) if a.shape[1] < b.shape[1] else (a, b @ fw['proj_weight'])
            x = torch.stack([a, b], 1)
            return (torch.softmax(torch.tanh(x @ fw['att_weight']), 1) * x).sum(1)
        # hybrid.py:72-84: feature based = (graph user, graph item) | (bert user, bert item); otherwise one branch per entity
        x1 = run(nets['dense3a'], fuse('fuse1a', g1, g2) if feature_based else fuse('fuse1a', g1, b1))
        x2 = run(nets['dense3b'], fuse('fuse1b', b1, b2) if feature_based else fuse('fuse1b', g2, b2))
        x = fuse('fuse2', x1, x2)
        if 'residual' in nets:                                       # hybrid.py:86-89
            res = nets['residual']
            r = run(res[:-1], x) if len(res) > 1 else x
            x = torch.relu(r @ res[-1][0] + res[-1][1] + x1 + x2)
        p = run(nets['clf'], x, True)[:, 0]
    yv = torch.tensor(np.asarray(y, dtype=dtype))
    pc = torch.clamp(p, EPS, 1 - EPS)
    loss = -torch.mean(yv * torch.log(pc + EPS) + (1 - yv) * torch.log(1 - pc + EPS))
    for st in stacks.values():
        if st['table_l2'] and 'embeddings' in st:
            loss = loss + l2 * (st['embeddings'] ** 2).sum()
        for lw in st['layers']:
            for name in ('kernel', 'bias', 'w'):                     # LocalityAdaptive's w carries the regulariser too (dgcf_conv.py:97)
                if name in lw:
                    loss = loss + l2 * (lw[name] ** 2).sum()
    loss.backward()
    g = lambda t: t.grad.numpy() if t.grad is not None else np.zeros(tuple(t.shape))

    def export(st):
        out = {'layers': [{k: g(v) for k, v in lw.items()} for lw in st['layers']]}
        if 'embeddings' in st:
            out['embeddings'] = g(st['embeddings'])
        if 'reduction_w' in st:
            out['reduction_w'] = g(st['reduction_w'])
        return out
    grads = {'gnn': export(stacks['gnn']) if 'gnn' in stacks else {name: export(st) for name, st in stacks.items()},
             'head': {name: [(g(w), g(b)) for w, b in nets[name]] for name in nets}}
    grads['head'].update({name: {k: g(v) for k, v in fw.items()} for name, fw in fusers.items()})
    return float(loss.detach()), grads, p.detach().numpy()
