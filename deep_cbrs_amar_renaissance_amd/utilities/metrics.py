"""Ranking step after the scoring head — mirrors `/root/reference/src/utilities/metrics.py:11-80`.

``top_k_predictions`` keeps, for every user, the k best-scored items among THAT USER'S test
pairs (`metrics.py:11-34`), on the GPU (`amar_topk_segmented_f32`: one wavefront per user).
Ordering is (user ascending, score descending); equal scores — left to an unstable quicksort
in the reference (`metrics.py:27`) — break on item id ascending here.

``top_k_metrics`` in the reference shells out to ``java -jar binaries/mimir.jar`` (RiVal
Precision/Recall, `metrics.py:60-65`); no JVM exists where this runs, so a plain host-side
Precision/Recall/F1@k over the same files is provided instead.  It is a stand-in for the jar,
not a bit-exact restatement of it.
"""
import logging
import os

import numpy as np
import pandas as pd
import torch

from deep_cbrs_amar_renaissance_amd import capi
from deep_cbrs_amar_renaissance_amd.engine import default_device

logging.basicConfig(format="%(message)s", level=logging.INFO)
logger = logging.getLogger(__name__)


def top_k_arrays(u_idx, i_idx, scores, k):
    """Per-user top-k on the device. Returns (user index [n], item index [n, k] (-1 padded), score [n, k])."""
    u_idx = np.asarray(u_idx, dtype=np.int64)
    i_idx = np.asarray(i_idx, dtype=np.int64)
    scores = np.asarray(scores, dtype=np.float32).reshape(-1)
    order = np.argsort(u_idx, kind='stable')
    seg_users, counts = np.unique(u_idx[order], return_counts=True)
    seg_ptr = np.zeros(len(seg_users) + 1, dtype=np.int64)
    np.cumsum(counts, out=seg_ptr[1:])
    dev = default_device()
    items_dev = torch.from_numpy(i_idx[order].astype(np.int32)).to(dev)
    scores_dev = torch.from_numpy(scores[order]).to(dev)
    seg_dev = torch.from_numpy(seg_ptr.astype(np.int32)).to(dev)
    out_items, out_scores = capi.topk_segmented(seg_dev, items_dev, scores_dev, int(k))
    return seg_users, out_items.cpu().numpy(), out_scores.cpu().numpy()


def top_k_predictions(predictions, users, items, k=5):
    """
    Top-K suggested items for each user.

    :param predictions: [P, 3] array (user index, item index (offset by |U|), score).
    :param users: original user identifiers.
    :param items: original item identifiers.
    :param k: the K parameter.
    :return: DataFrame (users, items, scores) with the original identifiers, k rows per user at most.
    """
    seg_users, top_items, top_scores = top_k_arrays(predictions[:, 0], predictions[:, 1], predictions[:, 2], k)
    valid = top_items >= 0
    df = pd.DataFrame()
    df['users'] = np.asarray(users)[np.repeat(seg_users, k).reshape(-1, k)[valid]]
    df['items'] = np.asarray(items)[top_items[valid] - len(users)]
    df['scores'] = top_scores[valid].astype(np.float64)
    return df


def precision_recall_f1_at_k(test_filepath, predictions_filepath, k, sep='\t'):
    """Macro-averaged Precision/Recall/F1@k over users: relevant = test rating 1, predicted = listed items."""
    test = pd.read_csv(test_filepath, sep=sep, header=None).to_numpy()
    pred = pd.read_csv(predictions_filepath, sep=sep, header=None).to_numpy()
    liked = test[test[:, 2] == 1]
    liked_keys = set(zip(liked[:, 0].astype(np.int64).tolist(), liked[:, 1].astype(np.int64).tolist()))
    n_liked = pd.Series(liked[:, 0].astype(np.int64)).value_counts().to_dict()
    hits = {}
    for u, i in zip(pred[:, 0].astype(np.int64).tolist(), pred[:, 1].astype(np.int64).tolist()):
        hits[u] = hits.get(u, 0) + ((u, i) in liked_keys)
    users = sorted(hits)
    precision = float(np.mean([hits[u] / k for u in users])) if users else 0.0
    recall = float(np.mean([hits[u] / n_liked[u] for u in users if n_liked.get(u, 0) > 0])) if users else 0.0
    f1 = 2 * precision * recall / (precision + recall) if precision + recall > 0 else 0.0
    return precision, recall, f1


def top_k_metrics(test_filepath, predictions_path):
    """Write ``results.tsv`` (label, precision, recall, F1) next to every ``predictions*`` file found."""
    if not os.path.isdir(predictions_path):
        logger.error("Invalid predictions path specified. Unable to run evaluator.")
        return
    for root, _, files in os.walk(predictions_path):
        found = sorted(f for f in files if f.startswith("predictions"))
        if not found:
            continue
        cutoff = int(str(root)[root.rfind(os.sep):].split("_")[1])
        rows = [precision_recall_f1_at_k(test_filepath, os.path.join(root, f), cutoff) for f in found]
        p, r, f1 = np.mean(np.asarray(rows), axis=0)
        pd.DataFrame([["top_{}".format(cutoff), p, r, f1]]).to_csv(
            os.path.join(root, "results.tsv"), sep='\t', header=False, index=False)
