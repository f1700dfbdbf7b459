#!/bin/bash
# PMC passes for one kernel regex on a python command. usage: tools/pmc_kernel.sh <outdir> <regex> -- <python args...>
set -u
OUT=$1; REGEX=$2; shift 3
mkdir -p "$OUT"; export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_128B_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
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
        agg.setdefault((row['Kernel_Name'][:60], row['Counter_Name']), []).append(float(row['Counter_Value']))
for (k, c), v in agg.items():
    print('{:<62s} {:<26s} n={} mean={:.6g}'.format(k, c, len(v), sum(v) / len(v)))
PY
