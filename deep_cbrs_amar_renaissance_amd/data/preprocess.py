"""Adjacency construction from contiguous-id rating triples.

Mirrors `/root/reference/src/data/preprocess.py:44-170` for the two adjacency types of the hot
path: ``'unary'`` (user-item graph of positive ratings) and ``'unary-uip'`` (user-item-properties
graph).  Output is a float32 scipy COO with the same triplets, in the same order, as the
reference builds — including duplicate (item, property) links and the un-deduplicated symmetric
copy.  ``'binary'`` and ``'unary-kg'`` feed models that are out of scope and raise.
"""
import numpy as np
from scipy import sparse

from deep_cbrs_amar_renaissance_amd.utilities.math import symmetrize_matrix


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
    :param type_adjacency: 'unary' or 'unary-uip'.
    :param sparse_adjacency: must be True (the HIP path consumes CSR).
    :param symmetric_adjacency: append the transposed triplets.
    :return: scipy COO float32 adjacency.
    """
    if type_adjacency in ('binary', 'unary-kg'):
        raise NotImplementedError(
            "type_adjacency '{}' only feeds TwoStep/TwoWay/BPR models, which are out of scope".format(type_adjacency))
    if type_adjacency not in ('unary', 'unary-uip'):
        raise ValueError("Unknown adjacency matrix type named {}".format(type_adjacency))
    if not sparse_adjacency:
        raise NotImplementedError("dense adjacency matrices are not supported by the HIP path")

    n_ui = len(users) + len(items)
    liked = bi_ratings[:, 2] == 1
    rows, cols, data = bi_ratings[liked, 0], bi_ratings[liked, 1], bi_ratings[liked, 2]
    size = n_ui
    if type_adjacency == 'unary-uip':
        if props is None or props_triples is None:
            raise ValueError("KG adjacency matrix requires properties info")
        rows = np.concatenate([rows, props_triples[:, 0] + len(users)])
        cols = np.concatenate([cols, props_triples[:, 1] + len(users)])
        data = np.concatenate([data, props_triples[:, 2]])
        size = n_ui + len(props)
    adj = sparse.coo_matrix((data, (rows, cols)), shape=[size, size], dtype=np.float32)
    return symmetrize_matrix(adj) if symmetric_adjacency else adj
