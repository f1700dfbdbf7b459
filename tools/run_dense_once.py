#!/usr/bin/env python
"""A few calls of amar_dense_f32 at the BERT tower's first-layer shape (386 304 x 768 -> 256, ReLU) for rocprofv3 --pmc passes
(tools/pmc_kernel.sh <out> dense_mfma -- tools/run_dense_once.py).  Prints the event-timed rate too."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deep_cbrs_amar_renaissance_amd import capi
from tools.profile_step import timeit

capi.load()
dev = torch.device('cuda')
M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (386304, 768, 256)))
x = torch.randn((M, K), device=dev) * 0.5
w = torch.randn((K, N), device=dev) * 0.05
b = torch.zeros(N, device=dev)
y = torch.empty((M, N), device=dev)
for _ in range(5):
    capi.dense(x, w, b, y, act='relu')
t, tmin = timeit(lambda: capi.dense(x, w, b, y, act='relu'), reps=10)
print('dense %d x %d -> %d: %.4f ms (min %.4f) = %.1f TFLOP/s' % (M, K, N, t, tmin, 2.0 * M * K * N / t / 1e9), flush=True)
if capi.dense_split_supported(K, N):
    wq = torch.from_numpy(capi.dense_split_pack(w.cpu().numpy())).to(dev)
    y2 = torch.empty_like(y)
    t, tmin = timeit(lambda: capi.dense_split(x, wq, K, N, b, y2, act='relu'), reps=10)
    print('dense_split %d x %d -> %d: %.4f ms (min %.4f) = %.1f TFLOP/s (f32-equivalent); max |diff| to the f32 form %.3e' %
          (M, K, N, t, tmin, 2.0 * M * K * N / t / 1e9, float((y2 - y).abs().max())), flush=True)
