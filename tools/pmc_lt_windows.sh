#!/bin/bash
# L2 requests of the LT kernel by window size (development aid). usage: tools/pmc_lt_windows.sh <outdir> [scale] [F] [windows...]
set -u
OUT=$1; SCALE=${2:-64}; F=${3:-8}; shift 3
mkdir -p "$OUT"; export TMPDIR=/tmp
for W in "$@"; do
  export AMAR_LT_WINDOW=$W
  timeout -k 10 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-include-regex "spmm_lt" --output-format csv -d "$OUT/w$W" -- python tools/run_spmm_once.py $SCALE $F 5 lt > "$OUT/w$W.log" 2>&1
  echo "window $W rc=$?"
done
python - "$OUT" "$@" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for w in sys.argv[2:]:
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(out, 'w' + w, '**', '*counter_collection.csv'), recursive=True)):
        for row in csv.DictReader(open(f)):
            agg.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
    print('window', w, ' '.join('{}={:.4g}'.format(c, sum(v) / len(v)) for c, v in agg.items()))
PY
