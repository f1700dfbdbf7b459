"""Dataset entry points named in the configs — mirrors `/root/reference/src/data/loaders.py:11-442`.

``dataset.load_function_name`` in ``config.yaml`` / ``econfigs/*.yaml`` resolves to one of the
``load_*`` functions here (`experiment.py:118`).  File formats are the reference's:
TSV ratings ``user\\titem\\t{0,1}`` without header, TSV property triples ``item\\tprop\\trel``,
KGE JSON ``{"ent_embeddings": [[...]]}`` indexed by raw id, BERT JSON
``[{"ID_OpenKE": id, "profile_embedding" | "embedding": [...]}]``.

Raw ids become contiguous indices in ascending raw-id order of the TRAIN file
(``np.unique``), items offset by |U|, properties by |U|+|I| (`loaders.py:43-68`).
"""
import numpy as np
import pandas as pd

from deep_cbrs_amar_renaissance_amd.data import jsonstream
from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemEmbeddings, HybridUserItemEmbeddings
from deep_cbrs_amar_renaissance_amd.data.datasets import UserItemGraph, UserItemGraphEmbeddings
from deep_cbrs_amar_renaissance_amd.data.preprocess import build_adjacency_matrix, get_user_properties


def _read_tsv(filepath, sep):
    return pd.read_csv(filepath, sep=sep, header=None).to_numpy()


def _lookup(sorted_ids, values, what):
    pos = np.searchsorted(sorted_ids, values)
    pos = np.minimum(pos, len(sorted_ids) - 1)
    if not np.array_equal(sorted_ids[pos], values):
        raise ValueError("{} contain identifiers that do not occur in the training ratings".format(what))
    return pos


def index_ratings(train_ratings, test_ratings):
    """Raw (user, item, rating) triples -> contiguous indices; returns ((train, test), (users, items))."""
    users, u_idx = np.unique(train_ratings[:, 0], return_inverse=True)
    items, i_idx = np.unique(train_ratings[:, 1], return_inverse=True)
    train = np.stack([u_idx, i_idx + len(users), train_ratings[:, 2]], axis=1)
    test = np.stack([_lookup(users, test_ratings[:, 0], "test users"),
                     _lookup(items, test_ratings[:, 1], "test items") + len(users),
                     test_ratings[:, 2]], axis=1)
    return (train, test), (users, items)


def index_props(props_triples, items):
    """Raw (item, prop, relation) triples -> (item index, prop index + |I|, 1); the relation is dropped."""
    it = _lookup(items, props_triples[:, 0], "property triples")
    props, p_idx = np.unique(props_triples[:, 1], return_inverse=True)
    ones = np.ones(len(p_idx), dtype=props_triples.dtype)
    return np.stack([it, p_idx + len(items), ones], axis=1), props


def load_train_test_ratings(
        train_filepath,
        test_filepath,
        props_filepath=None,
        sep='\t',
        return_adjacency=False,
        type_adjacency='unary',
        sparse_adjacency=True,
        symmetric_adjacency=True
):
    """Load train and test ratings (ids made sequential) and optionally the training adjacency matrix."""
    (train_ratings, test_ratings), (users, items) = index_ratings(_read_tsv(train_filepath, sep),
                                                                  _read_tsv(test_filepath, sep))
    if not return_adjacency:
        return (train_ratings, test_ratings), (users, items)

    props = props_triples = None
    if type_adjacency in ('unary-kg', 'unary-uip') and props_filepath is not None:
        props_triples, props = index_props(_read_tsv(props_filepath, sep), items)

    adj_matrix = build_adjacency_matrix(
        train_ratings, users, items,
        props_triples=props_triples, props=props,
        type_adjacency=type_adjacency,
        sparse_adjacency=sparse_adjacency,
        symmetric_adjacency=symmetric_adjacency
    )
    return (train_ratings, test_ratings), (users, items), adj_matrix


def json_load_graph_embeddings(filepath):
    """The 'ent_embeddings' rows (loaders.py:85-95) as a float32 array, streamed (the files reach 5.5 GB of JSON text)."""
    return jsonstream.stream_ent_embeddings(filepath)


def json_load_bert_embeddings(filepath):
    """DataFrame sorted by ID_OpenKE like the reference's (loaders.py:98-105); the loaders below stream instead."""
    return pd.read_json(filepath).sort_values(by=['ID_OpenKE'])


def load_graph_user_item_embeddings(filepath, users, items):
    """[|U|+|I|, D] fp32: KGE rows of the users followed by those of the items (rows indexed by raw id)."""
    table = json_load_graph_embeddings(filepath)
    return np.concatenate([table[users], table[items]], axis=0)


def load_bert_user_item_embeddings(user_filepath, item_filepath, users, items):
    """[|U|+|I|, D] fp32: BERT rows of the users followed by those of the items (matched on ID_OpenKE)."""
    def rows(filepath, column, ids):
        known, table = jsonstream.stream_bert_records(filepath, column)
        order = np.argsort(known, kind='stable')
        known, table = known[order], table[order]
        return table[_lookup(known, ids, "BERT file " + filepath)]
    return np.concatenate([rows(user_filepath, 'profile_embedding', users),
                           rows(item_filepath, 'embedding', items)], axis=0)


def _embedding_sequences(seq_class, train_ratings, test_ratings, users, items, tables, shuffle,
                         train_batch_size, test_batch_size):
    data_train = seq_class(train_ratings, users, items, *tables, batch_size=train_batch_size, shuffle=shuffle)
    data_test = seq_class(test_ratings, users, items, *tables, batch_size=test_batch_size, shuffle=False)
    return data_train, data_test


def load_graph_embeddings(
        train_ratings_filepath,
        test_ratings_filepath,
        graph_filepath,
        sep='\t',
        shuffle=True,
        train_batch_size=1024,
        test_batch_size=2048
):
    """Train / test sequences of pre-computed graph (KGE) embeddings for BasicRS."""
    (train_ratings, test_ratings), (users, items) = load_train_test_ratings(
        train_ratings_filepath, test_ratings_filepath, sep=sep, return_adjacency=False)
    graph_embeddings = load_graph_user_item_embeddings(graph_filepath, users, items)
    return _embedding_sequences(UserItemEmbeddings, train_ratings, test_ratings, users, items,
                                (graph_embeddings,), shuffle, train_batch_size, test_batch_size)


def load_bert_embeddings(
        train_ratings_filepath,
        test_ratings_filepath,
        bert_user_filepath,
        bert_item_filepath,
        sep='\t',
        shuffle=True,
        train_batch_size=1024,
        test_batch_size=2048
):
    """Train / test sequences of pre-computed BERT embeddings for BasicRS."""
    (train_ratings, test_ratings), (users, items) = load_train_test_ratings(
        train_ratings_filepath, test_ratings_filepath, sep=sep, return_adjacency=False)
    bert_embeddings = load_bert_user_item_embeddings(bert_user_filepath, bert_item_filepath, users, items)
    return _embedding_sequences(UserItemEmbeddings, train_ratings, test_ratings, users, items,
                                (bert_embeddings,), shuffle, train_batch_size, test_batch_size)


def load_hybrid_embeddings(
        train_ratings_filepath,
        test_ratings_filepath,
        graph_filepath,
        bert_user_filepath,
        bert_item_filepath,
        sep='\t',
        shuffle=True,
        train_batch_size=1024,
        test_batch_size=2048
):
    """Train / test sequences of (graph, BERT) embeddings for HybridCBRS."""
    (train_ratings, test_ratings), (users, items) = load_train_test_ratings(
        train_ratings_filepath, test_ratings_filepath, sep=sep, return_adjacency=False)
    graph_embeddings = load_graph_user_item_embeddings(graph_filepath, users, items)
    bert_embeddings = load_bert_user_item_embeddings(bert_user_filepath, bert_item_filepath, users, items)
    return _embedding_sequences(HybridUserItemEmbeddings, train_ratings, test_ratings, users, items,
                                (graph_embeddings, bert_embeddings), shuffle, train_batch_size, test_batch_size)


def _graph_ratings(train_ratings_filepath, test_ratings_filepath, props_triples_filepath, sep, type_adjacency,
                   sparse_adjacency, symmetric_adjacency, user_properties):
    ratings, (users, items), adj_matrix = load_train_test_ratings(
        train_ratings_filepath, test_ratings_filepath, props_triples_filepath,
        sep=sep, return_adjacency=True, type_adjacency=type_adjacency,
        sparse_adjacency=sparse_adjacency, symmetric_adjacency=symmetric_adjacency)
    if user_properties and type_adjacency != 'unary-uip':
        # loaders.py:318-321: TwoWay models take (user-item, item-property, user-property); like the reference this
        # needs the 'unary-kg' pair
        ui_adj, ip_adj = adj_matrix
        adj_matrix = (ui_adj, ip_adj, get_user_properties(ui_adj, ip_adj, len(users), len(items)))
    return ratings, (users, items), adj_matrix


def load_user_item_graph(
        train_ratings_filepath,
        test_ratings_filepath,
        props_triples_filepath=None,
        sep='\t',
        type_adjacency='unary',
        sparse_adjacency=True,
        symmetric_adjacency=True,
        user_properties=False,
        shuffle=True,
        train_batch_size=1024,
        test_batch_size=2048
):
    """Train / test sequences of graph node ids for GNN-based models (plus the training adjacency)."""
    (train_ratings, test_ratings), (users, items), adj_matrix = _graph_ratings(
        train_ratings_filepath, test_ratings_filepath, props_triples_filepath, sep, type_adjacency,
        sparse_adjacency, symmetric_adjacency, user_properties)
    data_train = UserItemGraph(train_ratings, users, items, adj_matrix, batch_size=train_batch_size, shuffle=shuffle)
    data_test = UserItemGraph(test_ratings, users, items, adj_matrix, batch_size=test_batch_size, shuffle=False)
    return data_train, data_test


def load_user_item_graph_sample(*args, **kwargs):
    raise NotImplementedError("BPR positive/negative sampling is out of scope (SURVEY.md §2 row 9)")


def load_user_item_graph_bert_embeddings(
        train_ratings_filepath,
        test_ratings_filepath,
        bert_user_filepath,
        bert_item_filepath,
        props_triples_filepath=None,
        sep='\t',
        type_adjacency='unary',
        sparse_adjacency=True,
        symmetric_adjacency=True,
        shuffle=True,
        train_batch_size=1024,
        test_batch_size=2048,
        user_properties=None):
    """Train / test sequences of graph node ids + BERT rows for hybrid GNN models."""
    (train_ratings, test_ratings), (users, items), adj_matrix = _graph_ratings(
        train_ratings_filepath, test_ratings_filepath, props_triples_filepath, sep, type_adjacency,
        sparse_adjacency, symmetric_adjacency, user_properties)
    bert_embeddings = load_bert_user_item_embeddings(bert_user_filepath, bert_item_filepath, users, items)
    data_train = UserItemGraphEmbeddings(train_ratings, users, items, adj_matrix, bert_embeddings,
                                         batch_size=train_batch_size, shuffle=shuffle)
    data_test = UserItemGraphEmbeddings(test_ratings, users, items, adj_matrix, bert_embeddings,
                                        batch_size=test_batch_size, shuffle=False)
    return data_train, data_test
