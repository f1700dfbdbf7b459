#!/bin/bash
# PMC passes for the XS SpMM kernels only (development aid). usage: tools/pmc_xs.sh <outdir> [scale] [F]
set -u
OUT=$1; SCALE=${2:-64}; F=${3:-8}
mkdir -p "$OUT"; export TMPDIR=/tmp
n=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" \
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
