#!/bin/bash
# rocprofv3 evidence for bench.py: kernel-trace stats, then PMC passes restricted to the SpMM kernel.
# usage: tools/profile_bench.sh <outdir> [scale]
set -u
OUT=$1; SCALE=${2:-64}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --scale $SCALE --steps 5 --warmup 2 --no-cpu-baseline"
echo "== kernel trace" ; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python $ARGS > "$OUT/trace.log" 2>&1
echo "rc=$?"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  name=$(echo $pass | cut -d' ' -f1)
  echo "== pmc $pass"; date
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-include-regex "spmm_|chain_" --output-format csv -d "$OUT/pmc_$name" -- python $ARGS > "$OUT/pmc_$name.log" 2>&1
  echo "rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    print('## kernel stats', f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 16: print(','.join(c[:70] for c in row))
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        agg.setdefault((row['Kernel_Name'][:70], row['Counter_Name']), []).append(float(row['Counter_Value']))
print('## pmc (mean per dispatch)')
for (k, c), v in agg.items():
    print('{:<72s} {:<28s} n={} mean={:.6g}'.format(k, c, len(v), sum(v) / len(v)))
PY
