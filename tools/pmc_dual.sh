#!/bin/bash
# Counters of the hybrid head's pair-stage kernel (tools/exp_hybrid.py at ml1m(s=64)).  tools/pmc_dual.sh OUT
set -u
OUT=$1; mkdir -p "$OUT"; export TMPDIR=/tmp
i=0
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-include-regex "dual_chain" --output-format csv -d "$OUT/p$i" -- python tools/exp_hybrid.py 64 > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, '*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        agg.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
for c, v in agg.items():
    v = sorted(v)
    print('{:<34s} n={} median={:.6g}'.format(c, len(v), v[len(v) // 2]))
PY
