#!/bin/bash
# rocprofv3 kernel-trace statistics of fit() on ml1m(s=1), the config doc.pdf p.22 Table 5 publishes a time for (BasicGCN 16 x 2,
# dense [48,48], clf [64,64], batch 1 024): one warm-up epoch + one epoch of 741 hipGraph-replayed batches (tools/exp_train.py 1 table5).
# usage: tools/profile_train.sh <outdir>     (repo root, GPU box)
set -u
OUT=$1
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/train" -- python $ROOT/tools/exp_train.py 1 table5 > "$ROOT/$OUT/train.log" 2>&1
echo "train rc=$?"
cd "$ROOT"
find "$OUT/train" -name '*kernel_trace.csv' -delete          # hundreds of thousands of rows: only the statistics travel back
cat "$OUT/train.log" | grep -v amdgpu.ids
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, 'train', '**', '*kernel_stats.csv'), recursive=True):
    rows = list(csv.reader(open(f)))
    total = sum(float(r[2]) for r in rows[1:])
    print('## kernel stats: %d kernels, %.1f ms of kernel time in all' % (len(rows) - 1, total / 1e6))
    for i, row in enumerate(rows):
        if i < 45:
            print(','.join(c[:100] for c in row))
PY
