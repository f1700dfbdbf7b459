#!/bin/bash
# PMC passes for the XS SpMM kernels only (development aid). usage: tools/pmc_xs.sh <outdir> [scale] [F]
# NOTE: passes with TCP_* / TA_* / TD_* counters (and a 5-counter TCC pass) hung under rocprofv3 on this pool until the
# 200 s timeout, three times in a row (round 1): they are left out.  SQ_*, TCC_* (<= 4 per pass), FETCH/WRITE_SIZE work.
set -u
OUT=$1; SCALE=${2:-64}; F=${3:-8}
mkdir -p "$OUT"; export TMPDIR=/tmp
n=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
  "FETCH_SIZE" "WRITE_SIZE"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-include-regex "spmm_xs" --output-format csv -d "$OUT/p$n" -- python tools/run_spmm_once.py $SCALE $F 3 xs > "$OUT/p$n.log" 2>&1
  echo "pass $n rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, 'p*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        agg.setdefault((row['Kernel_Name'][:48], row['Counter_Name']), []).append(float(row['Counter_Value']))
for (k, c), v in agg.items():
    print('{:<50s} {:<40s} n={} mean={:.6g}'.format(k, c, len(v), sum(v) / len(v)))
PY
