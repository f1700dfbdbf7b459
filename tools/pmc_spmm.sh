#!/bin/bash
# Collects rocprofv3 PMC passes for the SpMM kernel (development aid). usage: tools/pmc_spmm.sh <outdir> [scale] [F]
set -u
OUT=$1; SCALE=${2:-64}; F=${3:-8}
mkdir -p "$OUT"
export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python tools/run_spmm_once.py $SCALE $F 3 > "$OUT/$name.log" 2>&1
}
pass sqA SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY
pass sqB SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, '*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '')
        if 'spmm' not in k: continue
        key = (k[:60], row['Counter_Name'])
        agg.setdefault(key, []).append(float(row['Counter_Value']))
for (k, c), v in agg.items():
    print('{:<62s} {:<40s} n={} mean={:.4g}'.format(k, c, len(v), sum(v) / len(v)))
PY
