"""Synthetic MovieLens-1M-shape inputs, ``ml1m(s, seed)``.

The reference ships no data (``datasets/movielens.dvc`` etc. are DVC pointers), so every
test and benchmark runs on graphs of ML-1M *shape*, scaled by an integer factor ``s``:

    |U| = 6 036 s, |I| = 3 192 s, ratings = 946 772 s (57.22 % positive), 80/20 split,
    RS2 properties: |P| = 17 554 s with 70 341 s links   (doc.pdf p.19 Table 3)

Everything here is host-side numpy, seeded, and independent of both the HIP path and the
oracle: it only produces the arrays / files both of them consume.

The on-disk layout written by :func:`write_dataset` is the one the reference's loaders read
(`/root/reference/src/data/loaders.py:43-68`, `datasets/README.md:17-29`): ``train2id.tsv`` and
``test2id.tsv`` rows ``user\\titem\\t{0,1}`` without header, and a props file ``item\\tprop\\trel``.
"""
import json
import os
from dataclasses import dataclass

import numpy as np

ML1M_USERS = 6036
ML1M_ITEMS = 3192
ML1M_RATINGS = 946772
ML1M_POSITIVE = 541788
ML1M_PROPS_RS2 = 17554
ML1M_PROP_LINKS_RS2 = 70341
ML1M_ITEMS_WITH_PROPS = 2825
GRAPH_SEED = 20240101


@dataclass
class SyntheticML1M:
    """Raw-id ratings as the TSV files would hold them (not yet remapped)."""
    train: np.ndarray        # [R_train, 3] int64 (raw user, raw item, rating)
    test: np.ndarray         # [R_test, 3] int64
    props: np.ndarray        # [L, 3] int64 (raw item, raw prop, relation) or None
    scale: int
    seed: int


def _user_degrees(rng, n_users, n_items, total):
    # log-normal activity, minimum 20 ratings per user as in MovieLens, rescaled to `total`
    deg = rng.lognormal(mean=0.0, sigma=1.0, size=n_users)
    deg = 20.0 + deg * (total / n_users - 20.0) / deg.mean()
    # MovieLens-1M's heaviest user has 2 314 ratings: cap so that row lengths do not grow with s
    deg = np.minimum(np.rint(deg).astype(np.int64), min(int(0.6 * n_items), 2400))
    # fix the rounding drift on the heaviest users so the total matches exactly
    drift = int(total - deg.sum())
    order = np.argsort(-deg)
    k = min(len(order), max(1, abs(drift)))
    step = np.zeros(n_users, dtype=np.int64)
    step[order[:k]] = drift // k
    step[order[:abs(drift) - (abs(drift) // k) * k]] += int(np.sign(drift))
    deg = np.maximum(deg + step, 1)
    return deg


_POP_ALPHA = 1.1
_POP_OFFSET = 0.03


def _popularity_rank(uniform, n_items):
    """Inverse CDF of a Zipf-Mandelbrot popularity law w(r) ~ (r + c)^-alpha, c = 0.03 |I|.

    Heavy tail with a flattened head: the most popular item is rated by roughly 40 % of the
    users at every scale, as in MovieLens-1M. Closed form (continuous approximation), so that
    60 M draws at s=64 cost a vectorised pow instead of a binary search.
    """
    c = _POP_OFFSET * n_items
    e = 1.0 - _POP_ALPHA
    lo, hi = c ** e, (n_items + c) ** e
    r = (lo + uniform * (hi - lo)) ** (1.0 / e) - c
    return np.clip(r.astype(np.int64), 0, n_items - 1)


def _sample_distinct(rng, deg, n_items, item_of_rank):
    """For each user u draw (up to) deg[u] distinct items by popularity; returns sorted keys u*|I|+i.

    Draw-with-replacement, dedupe, then top up each user's deficit for a few rounds. Users so
    heavy that they keep colliding may end a few ratings short: the total is fixed by the seed.
    """
    n_users = len(deg)
    need = deg.copy()
    parts = []
    for _ in range(8):
        if not need.any():
            break
        users = np.repeat(np.arange(n_users, dtype=np.int64), need)
        ranks = _popularity_rank(rng.random(len(users)), n_items)
        keys = np.unique(users * n_items + item_of_rank[ranks])
        for old in parts:
            pos = np.minimum(np.searchsorted(old, keys), len(old) - 1)
            keys = keys[old[pos] != keys]
        need = need - np.bincount(keys // n_items, minlength=n_users)
        parts.append(keys)
    keys = np.concatenate(parts)
    keys.sort()
    return keys


def ml1m(scale=1, seed=GRAPH_SEED, with_props=True):
    """Generate the synthetic ML-1M-shape dataset at integer scale ``scale``."""
    s = int(scale)
    rng = np.random.default_rng(seed)
    n_users, n_items = ML1M_USERS * s, ML1M_ITEMS * s
    total = ML1M_RATINGS * s

    # sparse, non-contiguous raw ids: exercises the np.unique remap of loaders.py:47-56
    raw_users = np.sort(rng.choice(3 * n_users, size=n_users, replace=False)).astype(np.int64)
    raw_items = np.sort(rng.choice(3 * n_items, size=n_items, replace=False)).astype(np.int64)

    deg = _user_degrees(rng, n_users, n_items, total)
    item_of_rank = rng.permutation(n_items).astype(np.int64)   # popularity is unrelated to id order
    keys = _sample_distinct(rng, deg, n_items, item_of_rank)
    u, i = keys // n_items, keys % n_items
    del keys
    rating = (rng.random(len(u)) < ML1M_POSITIVE / ML1M_RATINGS).astype(np.int64)

    # 80/20 split per user; every test user and item must also occur in train
    is_test = rng.random(len(u)) < 0.2
    first = np.searchsorted(u, np.arange(n_users))
    is_test[first[first < len(u)]] = False                      # >= 1 train rating per user
    train_cnt = np.bincount(i[~is_test], minlength=n_items)
    is_test &= train_cnt[i] > 0                                 # items unseen in train go back to train
    # shuffle the file order (files are not sorted by user in general)
    perm_tr = rng.permutation(int((~is_test).sum()))
    perm_te = rng.permutation(int(is_test.sum()))
    cols = np.stack([raw_users[u], raw_items[i], rating], axis=1)
    train = cols[~is_test][perm_tr]
    test = cols[is_test][perm_te]

    props = None
    if with_props:
        n_props = ML1M_PROPS_RS2 * s
        n_links = ML1M_PROP_LINKS_RS2 * s
        train_items = np.unique(i[~is_test])
        carriers = rng.choice(train_items, size=min(len(train_items), ML1M_ITEMS_WITH_PROPS * s), replace=False)
        raw_props = np.sort(rng.choice(3 * n_props, size=n_props, replace=False)).astype(np.int64)
        # every property gets >= 1 link; the remainder follows a heavy-tailed property degree
        p_idx = np.concatenate([
            np.arange(n_props, dtype=np.int64),
            np.minimum((n_props * rng.random(n_links - n_props) ** 2.5).astype(np.int64), n_props - 1)])
        it_idx = carriers[rng.integers(0, len(carriers), size=n_links)]
        rel = rng.integers(0, 11, size=n_links)
        # ~3 % of the links are a second relation between an already linked (item, prop) pair
        n_dup = int(0.03 * n_links)
        src = rng.integers(0, n_links, size=n_dup)
        dst = rng.choice(np.arange(n_props, n_links), size=n_dup, replace=False)
        it_idx[dst], p_idx[dst] = it_idx[src], p_idx[src]
        rel[dst] = (rel[src] + 1 + rng.integers(0, 10, size=n_dup)) % 11
        props = np.stack([raw_items[it_idx], raw_props[p_idx], rel], axis=1)
        props = props[np.lexsort((props[:, 1], props[:, 0]))]
    return SyntheticML1M(train=train, test=test, props=props, scale=s, seed=seed)


def entity_embeddings(n_rows, dim=768, kind='bert', seed=GRAPH_SEED + 1):
    """BERT-like N(0,1)*0.5 or KGE-like U(-0.1,0.1) rows, fp32."""
    rng = np.random.default_rng(seed + (0 if kind == 'bert' else 7))
    if kind == 'bert':
        return (0.5 * rng.standard_normal((n_rows, dim), dtype=np.float32)).astype(np.float32)
    return rng.uniform(-0.1, 0.1, size=(n_rows, dim)).astype(np.float32)


def write_dataset(ds, dirpath, bert_dim=None, kge_dim=None):
    """Write the TSV (and optionally JSON embedding) files in the reference's on-disk formats."""
    os.makedirs(dirpath, exist_ok=True)
    paths = {
        'train_ratings_filepath': os.path.join(dirpath, 'train2id.tsv'),
        'test_ratings_filepath': os.path.join(dirpath, 'test2id.tsv'),
    }
    np.savetxt(paths['train_ratings_filepath'], ds.train, fmt='%d', delimiter='\t')
    np.savetxt(paths['test_ratings_filepath'], ds.test, fmt='%d', delimiter='\t')
    if ds.props is not None:
        paths['props_triples_filepath'] = os.path.join(dirpath, 'props2id-2relconf.tsv')
        np.savetxt(paths['props_triples_filepath'], ds.props, fmt='%d', delimiter='\t')
    users = np.unique(ds.train[:, 0])
    items = np.unique(ds.train[:, 1])
    if bert_dim:
        # embeddings/README.md:7-21 — list of {"ID_OpenKE", "profile_embedding" | "embedding"}
        ub = entity_embeddings(len(users), bert_dim, 'bert')
        ib = entity_embeddings(len(items), bert_dim, 'bert', seed=GRAPH_SEED + 2)
        paths['bert_user_filepath'] = os.path.join(dirpath, 'user-lastlayer.json')
        paths['bert_item_filepath'] = os.path.join(dirpath, 'item-lastlayer.json')
        with open(paths['bert_user_filepath'], 'w') as fp:
            json.dump([{'ID_OpenKE': int(u), 'profile_embedding': ub[k].tolist()} for k, u in enumerate(users)], fp)
        with open(paths['bert_item_filepath'], 'w') as fp:
            json.dump([{'ID_OpenKE': int(i), 'embedding': ib[k].tolist()} for k, i in enumerate(items)], fp)
    if kge_dim:
        # loaders.py:85-94,109-121 — {"ent_embeddings": [[...]]} indexed by raw entity id
        n_ent = int(max(users.max(), items.max())) + 1
        paths['graph_filepath'] = os.path.join(dirpath, '%dTransD.json' % kge_dim)
        with open(paths['graph_filepath'], 'w') as fp:
            json.dump({'ent_embeddings': entity_embeddings(n_ent, kge_dim, 'kge').tolist()}, fp)
    return paths


def ml1m_device(scale=64, seed=GRAPH_SEED, device=None, with_props=False):
    """ml1m(s) drawn directly in index space on the GPU (torch as plumbing), for large scales.

    Same shape laws as :func:`ml1m` (user activity, item popularity, 57.22 % positives, 80/20
    split) but a different random stream: torch's device generator instead of numpy, and a
    single draw-and-dedupe round.  Returns a dict of device tensors:
    ``train_pos`` [E, 2] (user index, item index + |U|) of positive train ratings,
    ``test`` [P, 2] pairs, plus ``n_users``, ``n_items``.  with_props: also ``item_prop`` [L, 2] (item index + |U|,
    property index + |U| + |I|) and ``n_props`` — the RS2 item-property links of :func:`ml1m` (every property linked at
    least once, heavy-tailed property degree, ~3 % duplicate (item, property) pairs, carriers among the train items).
    Used by bench.py only.
    """
    import torch
    s = int(scale)
    device = device or torch.device('cuda')
    n_users, n_items = ML1M_USERS * s, ML1M_ITEMS * s
    rng = np.random.default_rng(seed)
    deg = torch.from_numpy(_user_degrees(rng, n_users, n_items, ML1M_RATINGS * s)).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    item_of_rank = torch.randperm(n_items, device=device, generator=gen)
    users = torch.repeat_interleave(torch.arange(n_users, device=device), deg)
    uni = torch.rand(users.numel(), device=device, generator=gen, dtype=torch.float64)
    c = _POP_OFFSET * n_items
    e = 1.0 - _POP_ALPHA
    lo, hi = c ** e, (n_items + c) ** e
    ranks = ((lo + uni * (hi - lo)) ** (1.0 / e) - c).long().clamp_(0, n_items - 1)
    del uni
    keys = torch.unique(users * n_items + item_of_rank[ranks])
    del users, ranks
    u, i = keys // n_items, keys % n_items
    del keys
    liked = torch.rand(u.numel(), device=device, generator=gen) < ML1M_POSITIVE / ML1M_RATINGS
    is_test = torch.rand(u.numel(), device=device, generator=gen) < 0.2
    train_cnt = torch.bincount(i[~is_test], minlength=n_items)
    is_test &= train_cnt[i] > 0
    tr = liked & ~is_test
    train_pos = torch.stack([u[tr], i[tr] + n_users], dim=1)
    test = torch.stack([u[is_test], i[is_test] + n_users], dim=1)
    out = {'train_pos': train_pos, 'test': test, 'n_users': n_users, 'n_items': n_items, 'scale': s}
    if with_props:
        n_props, n_links = ML1M_PROPS_RS2 * s, ML1M_PROP_LINKS_RS2 * s
        train_items = torch.nonzero(train_cnt > 0).view(-1)
        carriers = train_items[torch.randperm(train_items.numel(), device=device, generator=gen)[:ML1M_ITEMS_WITH_PROPS * s]]
        tail = (n_props * torch.rand(n_links - n_props, device=device, generator=gen, dtype=torch.float64) ** 2.5).long().clamp_(0, n_props - 1)
        p_idx = torch.cat([torch.arange(n_props, device=device), tail])
        it_idx = carriers[torch.randint(0, carriers.numel(), (n_links,), device=device, generator=gen)]
        n_dup = int(0.03 * n_links)
        src = torch.randint(0, n_links, (n_dup,), device=device, generator=gen)
        dst = n_props + torch.randperm(n_links - n_props, device=device, generator=gen)[:n_dup]
        src_it, src_p = it_idx[src].clone(), p_idx[src].clone()
        it_idx[dst], p_idx[dst] = src_it, src_p
        out['item_prop'] = torch.stack([it_idx + n_users, p_idx + n_users + n_items], dim=1)
        out['n_props'] = n_props
    return out
