"""Adjacency construction from contiguous-id rating triples.

Mirrors `/root/reference/src/data/preprocess.py:9-170`: ``'unary'`` (user-item graph of positive ratings),
``'unary-uip'`` (user-item-properties graph), ``'unary-kg'`` (the user-item graph AND the item-property graph, for the
TwoStep / TwoWay stacks), ``get_user_properties`` (the two-hop user-property graph) and the offline
``process_item_properties_graph`` filter that produces the property files.  Output is a scipy COO with the
same triplets, in the same order, as the reference builds — including duplicate (item, property) links and the
un-deduplicated symmetric copy.  ``'binary'`` only feeds the BPR sampler, which is out of scope, and raises.
"""
import numpy as np
from scipy import sparse

from deep_cbrs_amar_renaissance_amd.utilities.math import symmetrize_matrix


def get_user_properties(ui_adj, ip_adj, n_users, n_items):
    """User-property adjacency [|U|+|P|, |U|+|P|] (users first): 1 where a property is two hops away from a user.

    The reference squares the user-item-property matrix and copies two blocks out of its DENSE image
    (`preprocess.py:33-41`, 5.7 GB at ML-1M size); the same entries come out of two sparse block products here:
    (A.A)[u, p] = sum_i A[u, i] A[i, p] only runs over items, because users link to items and properties to items only.
    Float64 values and row-major entry order, as `sparse.coo_matrix(dense)` yields them.
    """
    n_props = ip_adj.shape[0] - n_items
    ui, ip = sparse.csr_matrix(ui_adj), sparse.csr_matrix(ip_adj)
    user_item, item_user = ui[:n_users, n_users:], ui[n_users:, :n_users]
    item_prop, prop_item = ip[:n_items, n_items:], ip[n_items:, :n_items]
    upper = sparse.csr_matrix(user_item @ item_prop)           # users -> properties
    lower = sparse.csr_matrix(prop_item @ item_user)           # properties -> users
    for block in (upper, lower):
        block.eliminate_zeros()
        block.data = np.ones(len(block.data))
    size = n_users + n_props
    up = sparse.bmat([[None, upper], [lower, None]], format='csr', dtype=np.float64)
    up.resize((size, size))
    up.sort_indices()
    return up.tocoo()


def build_adjacency_matrix(
        bi_ratings,
        users,
        items,
        props_triples=None,
        props=None,
        type_adjacency='unary',
        sparse_adjacency=True,
        symmetric_adjacency=True
):
    """
    :param bi_ratings: [R, 3] array (user index, item index + |U|, rating in {0, 1}).
    :param users: original user identifiers (only the count is used).
    :param items: original item identifiers (only the count is used).
    :param props_triples: [L, 3] array (item index, property index + |I|, 1) or None.
    :param props: original property identifiers or None.
    :param type_adjacency: 'unary', 'unary-uip' or 'unary-kg'.
    :param sparse_adjacency: must be True (the HIP path consumes CSR).
    :param symmetric_adjacency: append the transposed triplets.
    :return: scipy COO float32 adjacency; for 'unary-kg' the pair (user-item [|U|+|I|]^2, item-property [|I|+|P|]^2).
    """
    if type_adjacency == 'binary':
        raise NotImplementedError("type_adjacency 'binary' only feeds the BPR sampler, which is out of scope")
    if type_adjacency not in ('unary', 'unary-uip', 'unary-kg'):
        raise ValueError("Unknown adjacency matrix type named {}".format(type_adjacency))
    if not sparse_adjacency:
        raise NotImplementedError("dense adjacency matrices are not supported by the HIP path")

    n_ui = len(users) + len(items)
    liked = bi_ratings[:, 2] == 1
    rows, cols, data = bi_ratings[liked, 0], bi_ratings[liked, 1], bi_ratings[liked, 2]
    size = n_ui
    if type_adjacency == 'unary-kg':
        # preprocess.py:134-152: two separate graphs, the item-property one with items at rows 0..|I|-1
        if props is None or props_triples is None:
            raise ValueError("KG adjacency matrix requires properties info")
        n_kg = len(items) + len(props)
        adj_bi = sparse.coo_matrix((data, (rows, cols)), shape=[n_ui, n_ui], dtype=np.float32)
        adj_kg = sparse.coo_matrix((props_triples[:, 2], (props_triples[:, 0], props_triples[:, 1])),
                                   shape=[n_kg, n_kg], dtype=np.float32)
        if symmetric_adjacency:
            adj_bi, adj_kg = symmetrize_matrix(adj_bi), symmetrize_matrix(adj_kg)
        return adj_bi, adj_kg
    if type_adjacency == 'unary-uip':
        if props is None or props_triples is None:
            raise ValueError("KG adjacency matrix requires properties info")
        rows = np.concatenate([rows, props_triples[:, 0] + len(users)])
        cols = np.concatenate([cols, props_triples[:, 1] + len(users)])
        data = np.concatenate([data, props_triples[:, 2]])
        size = n_ui + len(props)
    adj = sparse.coo_matrix((data, (rows, cols)), shape=[size, size], dtype=np.float32)
    return symmetrize_matrix(adj) if symmetric_adjacency else adj


def process_item_properties_graph(ratings_filepath, graph_filepath, kg_filepath, sep='\t'):
    """Offline filter that produces the `props2id-*.tsv` files the loaders expect (`preprocess.py:173-198` of the reference).

    `graph_filepath` holds, after one header line, the training ratings followed by the knowledge-graph triples
    (item, property, relation).  The triples whose item occurs in the training ratings are written to `kg_filepath`,
    sorted by (item, property) — stable for equal keys —, without header.  (`loaders.index_props` fails on a property file
    that names an item absent from the training ratings, exactly like `np.stack` does at `loaders.py:68`.)

    :param ratings_filepath: training ratings `user<sep>item<sep>rating`.
    :param graph_filepath: ratings + KG interactions, first line skipped.
    :param kg_filepath: output path.
    :param sep: column separator.
    """
    import pandas as pd
    ratings = pd.read_csv(ratings_filepath, sep=sep, header=None).to_numpy()
    items = np.unique(ratings[:, 1])
    graph = pd.read_csv(graph_filepath, sep=sep, header=None, skiprows=1).to_numpy()
    kg = graph[len(ratings):]
    kg = kg[np.isin(kg[:, 0], items)]
    order = np.lexsort((kg[:, 1], kg[:, 0]))
    pd.DataFrame(data=kg[order]).to_csv(kg_filepath, sep=sep, header=False, index=False)
