#!/usr/bin/env python
"""Golden vectors for row A9 (per-user top-k) produced by the REFERENCE'S OWN function, run in the build container.

`/root/reference/src/utilities/metrics.py:11-34` (`top_k_predictions`) is the one function of the hot path whose module imports
without TensorFlow / Spektral / mlflow (os, subprocess, logging, numpy, pandas only), so it can be executed here; every other
module of the reference fails with ModuleNotFoundError (an ordinary import error).  The reference targets pandas < 2
(`DataFrame.append`, removed in pandas 2.0, at metrics.py:33); the pandas of this image is 2.x, so for the duration of the call
`DataFrame.append(other)` is restored as its documented equivalent `pd.concat([self, other])` — nothing of the reference is
copied or altered.  The vectors (inputs and the function's outputs) go to tests/golden/topk_reference.npz; the reference tree
does not travel, the fixture does.

    python tests/golden/make_topk_reference_golden.py      (needs /root/reference; CPU only)
"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, '/root/reference/src')
    from utilities.metrics import top_k_predictions                     # the reference's function, unmodified
    if not hasattr(pd.DataFrame, 'append'):
        pd.DataFrame.append = lambda self, other: pd.concat([self, other])      # pandas < 2 API the reference was written against
    rng = np.random.default_rng(20240101)
    n_users, n_items = 60, 45
    users = np.sort(rng.choice(1000, n_users, replace=False)).astype(np.int64)          # raw ids, ascending like np.unique
    items = np.sort(rng.choice(5000, n_items, replace=False)).astype(np.int64)
    out = {'users': users, 'items': items}
    for name, ties in (('distinct', False), ('ties', True)):
        rows = []
        for u in range(n_users):
            cnt = 2 if u == 7 else int(rng.integers(5, 41))                            # user 7 has fewer pairs than any k
            its = rng.choice(n_items, cnt, replace=False)
            sc = rng.random(cnt)
            if ties and cnt >= 6:
                sc[1] = sc[0]                                                          # two equal scores, one of them maybe at the cut
                sc[5] = sc[4]
            rows += [(u, i + n_users, s) for i, s in zip(its, sc)]
        pred = np.array(rows, dtype=np.float64)
        pred = pred[rng.permutation(len(pred))]                                        # test-file order is arbitrary
        out['pred_' + name] = pred
        for k in (5, 10):
            df = top_k_predictions(pred, users, items, k=k)
            ru, ri, rs = df['users'].to_numpy(), df['items'].to_numpy(), df['scores'].to_numpy()
            order = np.argsort(ru, kind='stable')                                      # the function walks set(users): any user order
            out['{}_k{}_users'.format(name, k)] = ru[order].astype(np.int64)
            out['{}_k{}_items'.format(name, k)] = ri[order].astype(np.int64)
            out['{}_k{}_scores'.format(name, k)] = rs[order].astype(np.float64)
    np.savez_compressed(os.path.join(HERE, 'topk_reference.npz'), **out)
    print('wrote', os.path.join(HERE, 'topk_reference.npz'), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
