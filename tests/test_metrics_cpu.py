"""Known-answer tests of the holdout Precision / Recall / F1 @k evaluator (utilities/metrics.py) that stands in for
`java -jar binaries/mimir.jar -holdout -cutoff K` (reference src/utilities/metrics.py:60-65), and of the results.tsv
layout src/experiment.py:211-213 relies on.  Every expected number below is computed by hand in the comments."""
import os

import numpy as np
import pandas as pd
import pytest

from deep_cbrs_amar_renaissance_amd.utilities import metrics


def _write(path, rows):
    pd.DataFrame(rows).to_csv(path, sep='\t', header=False, index=False)


@pytest.fixture
def toy(tmp_path):
    """k = 2.  Raw ids; test file rows (user, item, rating):
       user 10: items 1 (1), 2 (0), 3 (1)          -> 2 relevant; predicted [1, 2]     -> 1 hit
       user 20: item  4 (1)                         -> 1 relevant; predicted [4]        -> list shorter than k
       user 30: items 5 (0), 6 (0)                  -> 0 relevant; predicted [5, 6]     -> 0 hits, recall undefined
       user 40: items 7 (1), 8 (0), 9 (1); scores of 8 and 9 tie for rank 2: the deterministic rule (item id ascending)
                lists [7, 8]                        -> 2 relevant, 1 hit
       user 50 appears in the predictions only (not a test user)                       -> ignored"""
    test = [(10, 1, 1), (10, 2, 0), (10, 3, 1), (20, 4, 1), (30, 5, 0), (30, 6, 0), (40, 7, 1), (40, 8, 0), (40, 9, 1)]
    pred = [(10, 1, 0.9), (10, 2, 0.8), (20, 4, 0.7), (30, 5, 0.6), (30, 6, 0.5), (40, 7, 0.9), (40, 8, 0.4), (50, 1, 0.99), (50, 2, 0.98)]
    t, p = str(tmp_path / 'test2id.tsv'), str(tmp_path / 'predictions_1.tsv')
    _write(t, test)
    _write(p, pred)
    return t, p


def test_rival_holdout_semantics_default(toy):
    """RiVal: a user's value at cutoff k exists only if its ranked list reaches rank k -> user 20 is left out of both means;
    user 30's recall is 0/0 = NaN and is dropped from the recall mean, its precision 0 is counted.
       P = mean(10: 1/2, 30: 0/2, 40: 1/2) = 1/3        R = mean(10: 1/2, 40: 1/2) = 1/2
       F1 = 2 (1/3)(1/2) / (1/3 + 1/2) = (1/3) / (5/6) = 0.4"""
    p, r, f1 = metrics.precision_recall_f1_at_k(*toy, 2)
    assert p == pytest.approx(1 / 3) and r == pytest.approx(0.5) and f1 == pytest.approx(0.4)


def test_switch_short_lists_count(toy):
    """short_lists='count': user 20 stays with P = 1/2 (one hit over k = 2), R = 1/1.
       P = mean(1/2, 1/2, 0, 1/2) = 3/8     R = mean(1/2, 1, 1/2) = 2/3     F1 = 2 (3/8)(2/3) / (3/8 + 2/3) = (1/2) / (25/24) = 0.48"""
    p, r, f1 = metrics.precision_recall_f1_at_k(*toy, 2, short_lists='count')
    assert p == pytest.approx(3 / 8) and r == pytest.approx(2 / 3) and f1 == pytest.approx(0.48)


def test_switch_no_relevant_zero(toy):
    """no_relevant='zero': user 30 enters the recall mean with 0.  R = mean(1/2, 0, 1/2) = 1/3; P as default = 1/3; F1 = 1/3."""
    p, r, f1 = metrics.precision_recall_f1_at_k(*toy, 2, no_relevant='zero')
    assert p == pytest.approx(1 / 3) and r == pytest.approx(1 / 3) and f1 == pytest.approx(1 / 3)


def test_only_the_first_k_rows_of_a_user_count(toy, tmp_path):
    """A predictions file holding more than k rows per user (top-10 file evaluated at k = 2): rows beyond rank k are ignored.
    user 10 gets a third row (item 3, relevant): still 1 hit at k = 2."""
    t, p = toy
    rows = pd.read_csv(p, sep='\t', header=None).values.tolist()
    rows.insert(2, [10, 3, 0.1])
    p2 = str(tmp_path / 'predictions_long.tsv')
    _write(p2, rows)
    assert metrics.precision_recall_f1_at_k(t, p2, 2) == pytest.approx((1 / 3, 0.5, 0.4))
    # at k = 3 only user 10 reaches rank 3: P = 2/3, R = 2/2
    assert metrics.precision_recall_f1_at_k(t, p2, 3) == pytest.approx((2 / 3, 1.0, 0.8))


def test_degenerate_inputs(tmp_path):
    t, p = str(tmp_path / 't.tsv'), str(tmp_path / 'p.tsv')
    _write(t, [(1, 1, 0), (1, 2, 0)])
    _write(p, [(1, 1, 0.5), (1, 2, 0.4)])
    assert metrics.precision_recall_f1_at_k(t, p, 2) == (0.0, 0.0, 0.0)          # no relevant item anywhere
    with pytest.raises(ValueError):
        metrics.precision_recall_f1_at_k(t, p, 2, short_lists='pad')


def test_results_tsv_layout(toy, tmp_path):
    """top_k_metrics writes <top_K>/results.tsv = one row (label, P, R, F1), no header: experiment.py:211-213 drops column 0
    and reads P, R, F1 from the rest.  The cutoff comes from the directory name `top_<K>` (metrics.py:55)."""
    t, p = toy
    d = tmp_path / 'predictions' / 'top_2'
    os.makedirs(d)
    os.replace(p, str(d / 'predictions_1.tsv'))
    metrics.top_k_metrics(t, str(d))
    results = pd.read_csv(str(d / 'results.tsv'), sep='\t', header=None)
    assert results.shape == (1, 4) and results.iloc[0, 0] == 'top_2'
    vals = results.drop(0, axis=1).to_numpy().squeeze()                         # exactly what experiment.py does
    assert vals == pytest.approx([1 / 3, 0.5, 0.4])
