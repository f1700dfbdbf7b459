#!/bin/bash
# rocprofv3 evidence for bench.py: kernel-trace stats, then PMC passes restricted to the propagation and pair-stage kernels,
# then profiles/spmm_pmc_latest.json (what bench.py reports as roofline.traffic, tagged with the csrc/ hash it belongs to).
# usage: tools/profile_bench.sh <outdir> [scale]       (run from the repo root on the GPU box)
set -u
OUT=$1; SCALE=${2:-64}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$(pwd)
ARGS="$ROOT/bench.py --scale $SCALE --steps 5 --warmup 2 --no-cpu-baseline --no-s256"
cd /tmp
echo "== kernel trace" ; date
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/trace" -- python $ARGS > "$ROOT/$OUT/trace.log" 2>&1
echo "rc=$?"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo $pass | cut -d' ' -f1)
  echo "== pmc $pass"; date
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-include-regex "spmm_|chain_|scatter_" --output-format csv -d "$ROOT/$OUT/pmc_$name" -- python $ARGS > "$ROOT/$OUT/pmc_$name.log" 2>&1
  echo "rc=$?"
done
cd "$ROOT"
python - "$OUT" "$SCALE" <<'PY'
import csv, glob, os, sys, collections, json
sys.path.insert(0, os.getcwd())
out, scale = sys.argv[1], int(sys.argv[2])
for f in glob.glob(os.path.join(out, 'trace', '**', '*kernel_stats.csv'), recursive=True):
    print('## kernel stats', f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 18: print(','.join(c[:90] for c in row))
agg = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, 'pmc_*', '**', '*counter_collection.csv'), recursive=True)):
    for row in csv.DictReader(open(f)):
        agg.setdefault((row['Kernel_Name'][:70], row['Counter_Name']), []).append(float(row['Counter_Value']))
print('## pmc (mean per dispatch)')
mean = {}
for (k, c), v in agg.items():
    mean[(k, c)] = sum(v) / len(v)
    print('{:<72s} {:<28s} n={} mean={:.6g}'.format(k, c, len(v), mean[(k, c)]))
def get(kernel_part, counter):
    for (k, c), v in mean.items():
        if kernel_part in k and c == counter: return v
    return None
import bench
kind = 'lt' if get('spmm_lt_kernel', 'FETCH_SIZE') is not None else 'xs'
js = {'scale': scale, 'kind': kind, 'csrc_sha': bench.csrc_sha(),
      'workload': 'ml1m(s=%d) A_hat, F=8, fused GCN layer inside bench.py' % scale}
if kind == 'lt':
    f, w = get('spmm_lt_kernel', 'FETCH_SIZE'), get('spmm_lt_kernel', 'WRITE_SIZE')
    js['kernel'] = 'spmm_lt_kernel<8> (value-free LDS-tiled image, one launch per layer)'
    js['FETCH_SIZE_KB'], js['WRITE_SIZE_KB'] = f, w
    js['traffic_bytes_per_launch'] = (2 * f + w) * 1024
    js['note'] = 'FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE'
    js['TCC'] = {'REQ': get('spmm_lt_kernel', 'TCC_REQ_sum'), 'MISS': get('spmm_lt_kernel', 'TCC_MISS_sum'), 'HIT': get('spmm_lt_kernel', 'TCC_HIT_sum')}
cf, cw = get('chain_pipe_kernel', 'FETCH_SIZE'), get('chain_pipe_kernel', 'WRITE_SIZE')
if cf is not None:
    js['pair_stage_traffic_bytes_per_launch'] = (2 * cf + cw) * 1024
    busy, act = get('chain_pipe_kernel', 'SQ_VALU_MFMA_BUSY_CYCLES'), get('chain_pipe_kernel', 'GRBM_GUI_ACTIVE')
    if busy and act:
        js['pair_stage_mfma_busy_frac'] = busy / 1024.0 / (act / 8.0)     # busy cycles summed over 1024 SIMDs / active cycles summed over 8 XCDs
sf, sw = get('scatter_windows_kernel', 'FETCH_SIZE'), get('scatter_windows_kernel', 'WRITE_SIZE')
if sf is not None:
    js['pair_stage_scatter_traffic_bytes_per_launch'] = (2 * sf + sw) * 1024
json.dump(js, open(os.path.join('profiles', 'spmm_pmc_latest.json'), 'w'), indent=1)
print('## profiles/spmm_pmc_latest.json'); print(json.dumps(js, indent=1))
PY
