#!/bin/bash
set -u
OUT=$1; mkdir -p "$OUT"; export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-include-regex "chain_kernel" --output-format csv -d "$OUT/pmc_$name" -- python tools/exp_chain.py 64 > "$OUT/pmc_$name.log" 2>&1
  echo "pass $name rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        agg.setdefault((row['Kernel_Name'][38:62], int(row['Grid_Size']), row['Counter_Name']), []).append(float(row['Counter_Value']))
for (k, g, c), v in agg.items():
    if g >= 1000000: print('{:<26s} grid={:<9d} {:<28s} n={} mean={:.6g}'.format(k, g, c, len(v), sum(v) / len(v)))
PY
