#!/usr/bin/env python
"""Per-batch launch counts and durations from the kernel statistics of tools/profile_train.sh (rocprofv3 --kernel-trace --stats of
tools/exp_train.py 1 table5: 2 epochs x 741 batches).  usage: python tools/train_launches.py <dir with *kernel_stats.csv> [batches]"""
import csv
import glob
import os
import sys

out = sys.argv[1]
batches = float(sys.argv[2]) if len(sys.argv) > 2 else 1482.0
for f in glob.glob(os.path.join(out, '**', '*kernel_stats.csv'), recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot_calls = sum(float(r['Calls']) for r in rows) / batches
    tot_us = sum(float(r['TotalDurationNs']) for r in rows) / batches / 1e3
    print('%.1f launches and %.1f us of kernel time per batch' % (tot_calls, tot_us))
    for r in rows:
        c = float(r['Calls']) / batches
        if c >= 0.5:
            print('  %5.2f x %6.1f us = %6.1f us  %s' % (c, float(r['AverageNs']) / 1e3, c * float(r['AverageNs']) / 1e3, r['Name'][:110]))
