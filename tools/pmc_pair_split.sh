#!/bin/bash
# Counters of the pair-stage kernel on bench.py's pair list, f32 and split-bf16 forms.  tools/pmc_pair_split.sh OUT [width]
set -u
OUT=$1; W=${2:-48}; mkdir -p "$OUT"; export TMPDIR=/tmp
for tag in split f32; do
  if [ $tag = f32 ]; then export AMAR_PAIR_MFMA=f32; else unset AMAR_PAIR_MFMA; fi
  i=0
  for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --pmc $pass --kernel-include-regex "chain_pipe" --output-format csv -d "$OUT/${tag}_$i" -- python tools/exp_pair_split.py 64 $tag $W > "$OUT/${tag}_$i.log" 2>&1
    echo "$tag pass $i rc=$?"
  done
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, '*', '**', '*counter_collection.csv'), recursive=True)):
    tag = os.path.relpath(f, out).split(os.sep)[0].split('_')[0]
    for row in csv.DictReader(open(f)):
        agg.setdefault((tag, row['Counter_Name']), []).append(float(row['Counter_Value']))
for (t, c), v in agg.items():
    v = sorted(v)
    print('{:<6s} {:<34s} n={} median={:.6g}'.format(t, c, len(v), v[len(v) // 2]))
PY
