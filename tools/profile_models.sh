#!/bin/bash
# rocprofv3 kernel-trace statistics of tools/exp_models_s64.py (the four Basic* model families at ml1m(s)) and of bench.py's
# uip_graph leg workload (tools/exp_uip.py).  usage: tools/profile_models.sh <outdir> [scale]     (repo root, GPU box)
set -u
OUT=$1; SCALE=${2:-64}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/models" -- python $ROOT/tools/exp_models_s64.py $SCALE > "$ROOT/$OUT/models.log" 2>&1
echo "models rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/uip" -- python $ROOT/tools/exp_uip.py $SCALE 8 > "$ROOT/$OUT/uip.log" 2>&1
echo "uip rc=$?"
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for leg in ('models', 'uip'):
    for f in glob.glob(os.path.join(out, leg, '**', '*kernel_stats.csv'), recursive=True):
        print('## kernel stats (%s): amar kernels only' % leg)
        for i, row in enumerate(csv.reader(open(f))):
            if i == 0 or 'anonymous namespace' in row[0]:
                print(','.join(c[:110] for c in row))
PY
