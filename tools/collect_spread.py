#!/usr/bin/env python
"""Collects the default bench.py lines of this session's gpurun boxes (gpurun_out/**/bench*.json, one fresh box per gpurun call) that
were measured on the CURRENT csrc/ sources into profiles/r4_bench_repeats.json (round 3: r3_bench_repeats.json) — what bench.py reports as `value_spread_boxes`.
Every bench line carries `csrc_sha` (bench.py), so lines of older builds are left out.  usage: python tools/collect_spread.py"""
import glob
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def main():
    sha = bench.csrc_sha()
    runs = []
    for path in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', '**', '*bench*.json'), recursive=True)):
        try:
            lines = [l for l in open(path).read().splitlines() if l.startswith('{')]
            d = json.loads(lines[-1])
        except (IndexError, ValueError):
            continue
        if d.get('csrc_sha') != sha or d.get('n_gpus') != 1 or d.get('config', {}).get('scale') != 64:
            continue
        if not d.get('config', {}).get('parallelism', '').startswith('single GPU'):      # (AMAR_FORCE_DIST runs the partitioned runner)
            continue
        runs.append({'file': os.path.relpath(path, ROOT), 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'steps': d['steps']})
    if not runs:
        sys.exit('no bench lines of csrc {} under gpurun_out/'.format(sha))
    ms = [r['ms_per_step'] for r in runs]
    out = {'scale': 64, 'csrc_sha': sha, 'boxes': len(runs), 'min_ms_per_step': min(ms), 'max_ms_per_step': max(ms),
           'min_value': min(r['value'] for r in runs), 'max_value': max(r['value'] for r in runs), 'runs': runs,
           'note': 'default `python bench.py` lines of this build on different gpurun boxes (one fresh MI355X box per call)'}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r4_bench_repeats.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
