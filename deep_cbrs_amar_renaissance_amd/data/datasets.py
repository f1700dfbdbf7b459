"""Batch sequences fed to the models — mirrors `/root/reference/src/data/datasets.py:8-373`.

Same constructor arguments, attributes (``ratings``, ``users``, ``items``, ``adj_matrix``,
``embeddings``) and batch tuple layouts as the reference's ``keras.utils.Sequence`` classes:

    UserItemEmbeddings        ((u_emb[B,D], i_emb[B,D]), y[B])                      datasets.py:65-67
    HybridUserItemEmbeddings  ((u_graph, i_graph, u_bert, i_bert), y)               datasets.py:127-133
    UserItemGraph             ((u_ids[B], i_ids[B]), y[B])                          datasets.py:203
    UserItemGraphEmbeddings   ((u_ids, i_ids, u_emb[B,D], i_emb[B,D]), y)           datasets.py:366

ids are int64 with item ids already offset by |U|; the last batch is short.  Shuffling uses
``np.random.RandomState(seed)`` re-drawn at every epoch end, like the reference.
``UserItemGraphPosNegSample`` (BPR sampling) is out of scope.
"""
import numpy as np


class _RatingsSequence:
    def __init__(self, ratings, users, items, batch_size=512, shuffle=False, seed=42):
        self.ratings = ratings
        self.users = users
        self.items = items
        self.batch_size = batch_size
        self.shuffle = shuffle
        self.seed = seed
        self.indexes = None
        self.random_state = None
        self.order_version = 0          # bumped whenever the batch order changes: lets predict() keep its uploaded id list
        self.on_epoch_end()

    def __len__(self):
        return int(np.ceil(len(self.ratings) / self.batch_size))

    def __iter__(self):
        return (self[b] for b in range(len(self)))

    def _batch_ratings(self, idx):
        if idx < 0 or idx >= len(self):
            raise IndexError(idx)
        lo = idx * self.batch_size
        hi = min(lo + self.batch_size, len(self.ratings))
        if self.shuffle:
            return self.ratings[self.indexes[lo:hi]]
        return self.ratings[lo:hi]

    def on_epoch_end(self):
        if self.shuffle:
            if self.random_state is None:
                self.random_state = np.random.RandomState(self.seed)
            self.indexes = np.arange(len(self.ratings))
            self.random_state.shuffle(self.indexes)
            self.order_version += 1


class UserItemEmbeddings(_RatingsSequence):
    """Pairs as pre-computed embedding rows (KGE or BERT), gathered on the host per batch."""

    def __init__(self, ratings, users, items, embeddings, batch_size=512, shuffle=False, seed=42):
        self.embeddings = embeddings
        super().__init__(ratings, users, items, batch_size=batch_size, shuffle=shuffle, seed=seed)

    def __getitem__(self, idx):
        r = self._batch_ratings(idx)
        return (self.embeddings[r[:, 0]], self.embeddings[r[:, 1]]), r[:, 2]


class HybridUserItemEmbeddings(_RatingsSequence):
    """Pairs as (graph, BERT) embedding rows for HybridCBRS."""

    def __init__(self, ratings, users, items, graph_embeddings, bert_embeddings, batch_size=512, shuffle=False,
                 seed=42):
        self.graph_embeddings = graph_embeddings
        self.bert_embeddings = bert_embeddings
        super().__init__(ratings, users, items, batch_size=batch_size, shuffle=shuffle, seed=seed)

    def __getitem__(self, idx):
        r = self._batch_ratings(idx)
        u, i = r[:, 0], r[:, 1]
        return (self.graph_embeddings[u], self.graph_embeddings[i],
                self.bert_embeddings[u], self.bert_embeddings[i]), r[:, 2]


class UserItemGraph(_RatingsSequence):
    """Pairs as node ids of the user-item(-properties) graph."""

    def __init__(self, ratings, users, items, adj_matrix, batch_size=512, shuffle=False, seed=42):
        self.adj_matrix = adj_matrix
        super().__init__(ratings, users, items, batch_size=batch_size, shuffle=shuffle, seed=seed)

    def __getitem__(self, idx):
        r = self._batch_ratings(idx)
        return (r[:, 0], r[:, 1]), r[:, 2]


class UserItemGraphEmbeddings:
    """Node ids plus the matching BERT rows (HybridBertGNN batches)."""

    def __init__(self, ratings, users, items, adj_matrix, embeddings, batch_size=512, shuffle=False, seed=42):
        self.ratings = ratings
        self.users = users
        self.items = items
        self.adj_matrix = adj_matrix
        self.graph_ids = UserItemGraph(ratings, users, items, adj_matrix,
                                       batch_size=batch_size, shuffle=shuffle, seed=seed)
        self.embeddings = UserItemEmbeddings(ratings, users, items, embeddings,
                                             batch_size=batch_size, shuffle=shuffle, seed=seed)

    def __len__(self):
        return len(self.graph_ids)

    def __iter__(self):
        return (self[b] for b in range(len(self)))

    def __getitem__(self, idx):
        (user_ids, item_ids), ratings = self.graph_ids[idx]
        (user_embeddings, item_embeddings), _ = self.embeddings[idx]
        return (user_ids, item_ids, user_embeddings, item_embeddings), ratings

    def on_epoch_end(self):
        self.graph_ids.on_epoch_end()
        self.embeddings.on_epoch_end()


class UserItemGraphPosNegSample:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("BPR positive/negative sampling is out of scope (SURVEY.md §2 row 10)")
