"""Ranking step after the scoring head — mirrors `/root/reference/src/utilities/metrics.py:11-80`.

``top_k_predictions`` keeps, for every user, the k best-scored items among THAT USER'S test
pairs (`metrics.py:11-34`), on the GPU (`amar_topk_segmented_f32`: one wavefront per user).
Ordering is (user ascending, score descending); equal scores — left to pandas' sort in the reference
(`metrics.py:27`: whatever order `sort_values` leaves a tie in) — break on item id ascending here.  The reference's own
function, run on committed inputs, gives the same rows in the same order (tests/golden/topk_reference.npz).

``top_k_metrics`` in the reference shells out to ``java -jar binaries/mimir.jar`` (RiVal
Precision/Recall, `metrics.py:60-65`); no JVM exists where this runs, so the holdout
Precision/Recall/F1@k is restated on the host from RiVal's ranking-metric semantics
(``precision_recall_f1_at_k``: hand-computed vectors in tests/test_metrics_cpu.py; the choices the
jar's behaviour cannot settle here are explicit switches) and writes the same ``results.tsv``
(label, P, R, F1 — `experiment.py:211-213` reads columns 1..3).
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


def precision_recall_f1_at_k(test_filepath, predictions_filepath, k, sep='\t', short_lists='skip', no_relevant='skip',
                             relevance_threshold=1.0, counts=None):
    """Precision / Recall / F1 @k of a top-k predictions file against the test ratings, as `mimir.jar -holdout -cutoff k`
    computes them (metrics.py:60-65) — restated from RiVal's ranking metrics, the library inside the jar
    (net.recommenders.rival.evaluation.metric.ranking.{AbstractRankingMetric,Precision,Recall}; there is no JVM here, so
    the jar itself cannot be run: what its behaviour cannot settle is an explicit switch below):

    * per TEST user with predictions, items are ranked by predicted score (the file's order inside a user is kept: the
      reference writes each user's rows best first, metrics.py:27-33) and each carries its test relevance
      (rating >= relevance_threshold -> relevant; an item absent from the user's test rows is not relevant);
    * P@k(u) = (relevant among the first k) / k, R@k(u) = (relevant among the first k) / (relevant test items of u);
    * RiVal records a user's value at cutoff k only when the ranked list REACHES rank k (`if (rank == at)`), so a user with
      fewer than k predicted items contributes to neither mean: short_lists='skip' (default).  'count' keeps such users
      with P = hits / k, R = hits / relevant — what an evaluator that pads short lists would report;
    * a user without any relevant test item has R = 0 / 0 = NaN, which RiVal's getValueAt drops from the recall mean
      (no_relevant='skip', default; 'zero' counts it as recall 0); its precision is 0 and is counted;
    * both means are plain averages over the remaining users; F1 = 2 P R / (P + R) of the two MEANS (the jar's f1Measure
      takes the aggregated precision and recall), 0 when both are 0.
    `relevance_threshold`: the jar's constant is not recoverable from its call site; with the {0, 1} ratings of every
    dataset in the reference any threshold in (0, 1] gives the same result.
    `counts` (a dict, optional) receives how many users each mean covers and how many the switches dropped — the defaults are
    RiVal's behaviour as restated from its source, NOT checked against mimir.jar, so what they exclude is
    reported with every result.
    """
    if short_lists not in ('skip', 'count') or no_relevant not in ('skip', 'zero'):
        raise ValueError("short_lists must be 'skip' or 'count', no_relevant 'skip' or 'zero'")
    test = pd.read_csv(test_filepath, sep=sep, header=None).to_numpy()
    pred = pd.read_csv(predictions_filepath, sep=sep, header=None).to_numpy()
    test_users = set(test[:, 0].astype(np.int64).tolist())
    liked = test[test[:, 2] >= relevance_threshold]
    liked_keys = set(zip(liked[:, 0].astype(np.int64).tolist(), liked[:, 1].astype(np.int64).tolist()))
    n_liked = pd.Series(liked[:, 0].astype(np.int64)).value_counts().to_dict()
    hits, listed = {}, {}
    for u, i in zip(pred[:, 0].astype(np.int64).tolist(), pred[:, 1].astype(np.int64).tolist()):
        if u not in test_users:
            continue                                              # RiVal walks the test model's users
        rank = listed.get(u, 0)
        if rank < k:                                              # only the first k rows of a user count
            hits[u] = hits.get(u, 0) + ((u, i) in liked_keys)
        listed[u] = rank + 1
    users = sorted(u for u in listed if short_lists == 'count' or listed[u] >= k)
    prec = [hits[u] / k for u in users]
    rec = [hits[u] / n_liked[u] if n_liked.get(u, 0) > 0 else (0.0 if no_relevant == 'zero' else None) for u in users]
    rec = [r for r in rec if r is not None]
    precision = float(np.mean(prec)) if prec else 0.0
    recall = float(np.mean(rec)) if rec else 0.0
    f1 = 2 * precision * recall / (precision + recall) if precision + recall > 0 else 0.0
    info = {'users_with_predictions': len(listed), 'users_in_precision_mean': len(prec), 'users_in_recall_mean': len(rec),
            'skipped_short_list': len(listed) - len(users), 'skipped_no_relevant_item': len(users) - len(rec),
            'short_lists': short_lists, 'no_relevant': no_relevant}
    if info['skipped_short_list'] or info['skipped_no_relevant_item']:
        logger.info("P/R/F1@%d of %s: %d users with fewer than %d predictions left out of both means (short_lists='%s'), %d users "
                    "without a relevant test item left out of the recall mean (no_relevant='%s')", k, predictions_filepath,
                    info['skipped_short_list'], k, short_lists, info['skipped_no_relevant_item'], no_relevant)
    if counts is not None:
        counts.update(info)
    return precision, recall, f1


def top_k_metrics(test_filepath, predictions_path, short_lists=None, no_relevant=None):
    """Write ``results.tsv`` (label, precision, recall, F1) next to every ``predictions*`` file found, and ``results_users.tsv``
    (users in the precision / recall means, users skipped by each switch) beside it.  The switches come from the arguments, else
    from AMAR_METRICS_SHORT_LISTS / AMAR_METRICS_NO_RELEVANT, else the RiVal defaults ('skip', 'skip')."""
    short_lists = short_lists or os.environ.get('AMAR_METRICS_SHORT_LISTS', 'skip')
    no_relevant = no_relevant or os.environ.get('AMAR_METRICS_NO_RELEVANT', 'skip')
    if not os.path.isdir(predictions_path):
        logger.error("Invalid predictions path specified. Unable to run evaluator.")
        return
    for root, _, files in os.walk(predictions_path):
        found = sorted(f for f in files if f.startswith("predictions"))
        if not found:
            continue
        cutoff = int(str(root)[root.rfind(os.sep):].split("_")[1])
        infos = [{} for _ in found]
        rows = [precision_recall_f1_at_k(test_filepath, os.path.join(root, f), cutoff, short_lists=short_lists, no_relevant=no_relevant,
                                         counts=info) for f, info in zip(found, infos)]
        p, r, f1 = np.mean(np.asarray(rows), axis=0)
        pd.DataFrame([["top_{}".format(cutoff), p, r, f1]]).to_csv(
            os.path.join(root, "results.tsv"), sep='\t', header=False, index=False)
        keys = ['users_with_predictions', 'users_in_precision_mean', 'users_in_recall_mean', 'skipped_short_list', 'skipped_no_relevant_item',
                'short_lists', 'no_relevant']
        pd.DataFrame([[f] + [info[key] for key in keys] for f, info in zip(found, infos)], columns=['file'] + keys).to_csv(
            os.path.join(root, "results_users.tsv"), sep='\t', index=False)
