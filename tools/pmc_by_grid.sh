#!/bin/bash
# FETCH_SIZE / TCC hit counters per dispatch, grouped by grid size. usage: tools/pmc_by_grid.sh <outdir> <regex> -- <python args>
set -u
OUT=$1; REGEX=$2; shift 3
mkdir -p "$OUT"; export TMPDIR=/tmp
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_128B_sum" "GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-include-regex "$REGEX" --output-format csv -d "$OUT/pmc_$name" -- python "$@" > "$OUT/pmc_$name.log" 2>&1
  echo "pass $name rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        agg.setdefault((row['Kernel_Name'][:40], int(row['Grid_Size']), row['Counter_Name']), []).append(float(row['Counter_Value']))
for (k, g, c), v in agg.items():
    print('{:<42s} grid={:<10d} {:<26s} n={} mean={:.6g}'.format(k, g, c, len(v), sum(v) / len(v)))
PY
