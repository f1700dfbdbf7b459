#!/usr/bin/env python
"""Cycle stamps of the Dense-stack forward kernel's phases (development aid, round 4).  Builds tools/libamar_hip_dsstamps.so with
-DAMAR_DS_STAMPS (the shipped library carries none), runs amar_dense_stack_f32 on batch-sized operands and prints, per stack, the median
over workgroups of: entry -> first barrier (input gather + first kernel), and per layer products / barrier wait / next kernel's staging.
    python tools/exp_ds_stamps.py build        (no GPU needed)
    python tools/exp_ds_stamps.py              (on the GPU box)"""
import ctypes
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, 'tools', 'libamar_hip_dsstamps.so')
CSRC = os.path.join(ROOT, 'deep_cbrs_amar_renaissance_amd', 'csrc')


def build():
    srcs = ['amar_capi.hip', 'amar_propagate.hip', 'amar_dense.hip', 'amar_layout.hip', 'amar_chain.hip', 'amar_train.hip']
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-mllvm', '-amdgpu-mfma-vgpr-form=1',
           '-DAMAR_DS_STAMPS', '-shared'] + [os.path.join(CSRC, s) for s in srcs] + ['-o', LIB]
    subprocess.run(cmd, check=True)
    print('built', LIB)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'build':
        return build()
    import numpy as np
    import torch
    from deep_cbrs_amar_renaissance_amd import capi
    capi.LIB_PATH = LIB
    lib = capi.load()
    lib.amar_ds_debug_copy.restype = ctypes.c_int
    lib.amar_ds_debug_copy.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device('cuda')
    for name, dims, acts, gather in (('tower 16-48-48 (gathered rows)', [16, 48, 48], ['relu', 'relu'], True),
                                     ('classifier 96-64-64-1', [96, 64, 64, 1], ['relu', 'relu', 'sigmoid'], False)):
        M = 1024
        x = torch.randn((9228 if gather else M, dims[0]), device=dev)
        ids = torch.randint(0, 9228, (M,), device=dev, dtype=torch.int32) if gather else None
        ws = [torch.randn((dims[l], dims[l + 1]), device=dev) * 0.3 for l in range(len(acts))]
        bs = [torch.randn(dims[l + 1], device=dev) * 0.1 for l in range(len(acts))]
        outs = [torch.empty((M, d), device=dev) for d in dims[1:]]
        xc = torch.empty((M, dims[0]), device=dev) if gather else None
        junk = torch.empty(64 << 20, device=dev)
        times = []
        rows = []
        for rep in range(12):
            junk.normal_()                                            # (push the operands out of the L2s: a training batch finds them cold)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            capi.dense_stack(x, ws, bs, acts, outs, ids=ids, xcopy=xc)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) * 1e3)
            buf = (ctypes.c_ulonglong * (64 * 16))()
            assert lib.amar_ds_debug_copy(buf, 64 * 16) == 0
            st = np.array(buf, dtype=np.int64).reshape(64, 16)[:min(64, (M + 15) // 16 if os.environ.get("AMAR_DENSE_STACK_ROWS") != "64" else M // 64)]
            rows.append(st)
        st = np.stack(rows[2:])                                       # [rep, block, stamp]
        L = len(acts)
        d = lambda i, j: float(np.median(st[:, :, j] - st[:, :, i]))
        total = d(0, 15)
        print('%s: launch %.1f us (event pair, median of %d); stamps in counter ticks, total %.0f' % (name, float(np.median(times[2:])), len(times) - 2, total), flush=True)
        print('   entry -> operands staged (first barrier): %.0f' % d(0, 1))
        for l in range(L):
            print('   layer %d: products + stores %.0f | barrier wait %.0f | staging of the next kernel %.0f' % (
                l, d(1 + 3 * l, 2 + 3 * l), d(2 + 3 * l, 3 + 3 * l), (d(3 + 3 * l, 4 + 3 * l) if l + 1 < L else d(3 + 3 * l, 15))))
        spread = np.median(st[:, :, 0].max(axis=1) - st[:, :, 0].min(axis=1))
        print('   first entry to last entry over the %d workgroups: %.0f; last exit - first entry: %.0f' % (
            st.shape[1], spread, float(np.median(st[:, :, 15].max(axis=1) - st[:, :, 0].min(axis=1)))))


if __name__ == '__main__':
    main()
