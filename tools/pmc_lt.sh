#!/bin/bash
# PMC passes for the LT SpMM kernel (development aid). usage: tools/pmc_lt.sh <outdir> [scale] [F]
set -u
OUT=$1; SCALE=${2:-64}; F=${3:-8}
mkdir -p "$OUT"; export TMPDIR=/tmp
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-include-regex "spmm_lt|spmm_xs" --output-format csv -d "$OUT/pmc_$name" -- python tools/run_spmm_once.py $SCALE $F 5 lt > "$OUT/pmc_$name.log" 2>&1
  echo "pass $name rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        agg.setdefault((row['Kernel_Name'][:60], row['Counter_Name']), []).append(float(row['Counter_Value']))
for (k, c), v in agg.items():
    print('{:<62s} {:<26s} n={} mean={:.6g}'.format(k, c, len(v), sum(v) / len(v)))
PY
