#!/usr/bin/env python
"""Round 4: bench.py's `uip_graph` leg (BasicGCN on the user-item-property graph at ml1m(s)) with and without the scoring runner's
`rows_needed` hint (the last layer's property tiles not launched).  `python tools/exp_uip_rows_needed.py [scale]`"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 64
from deep_cbrs_amar_renaissance_amd import capi
capi.load()
dev = torch.device('cuda', 0)
for flag in ('0', '1', '0', '1'):
    os.environ['AMAR_ROWS_NEEDED'] = flag
    out = bench.uip_graph(dev, scale, 10)
    print('AMAR_ROWS_NEEDED=%s: %.4f ms per step, propagation %.4f ms, fused layer (mean of both) %.4f ms' % (
        flag, out['ms_per_step'], out['propagation_ms'], out['gcn_layer']['avg_launch_ms']), flush=True)
