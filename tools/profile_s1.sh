#!/bin/bash
# rocprofv3 kernel statistics of the replayed predict() pass at ml1m(s=1), hoisted and faithful.  usage: tools/profile_s1.sh <outdir>
set -u
OUT=$1; mkdir -p "$OUT"; export TMPDIR=/tmp; ROOT=$(pwd); cd /tmp
for mode in hoisted faithful; do
  EXP_MODE=$mode timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/$mode" -- python $ROOT/tools/exp_s1_nodes.py > "$ROOT/$OUT/$mode.log" 2>&1
  find "$ROOT/$OUT/$mode" -name '*kernel_trace.csv' -delete
  grep "ms per pass" "$ROOT/$OUT/$mode.log"
done
cd "$ROOT"
python tools/train_launches.py "$OUT/hoisted" 501
python tools/train_launches.py "$OUT/faithful" 1701       # 21 passes x 81 batches
