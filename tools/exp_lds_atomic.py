"""Runs tools/exp_lds_atomic.hip: entries (8 float adds each) per clock per CU into an LDS-resident Y tile."""
import ctypes, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'libexp_lds_atomic.so'))
lib.run_lds_atomic.restype = ctypes.c_float
out = torch.zeros(4, device='cuda')
iters, blocks = 2000, 256
for rows in (2048, 4096):
    for mode, name in ((0, 'random rows'), (2, 'random 32-row groups'), (1, 'consecutive rows')):
        ms = lib.run_lds_atomic(mode, rows, iters, blocks, ctypes.c_void_p(out.data_ptr()), 3)
        entries = blocks * 16 * iters * 32               # 16 waves x 32 entries per wave-iteration
        clk = ms * 1e-3 * 2.4e9
        print('Y tile %4d rows, %-22s: %.3f ms  %.3f entries/clk/CU  -> 55.9 M entries in %.3f ms chip-wide' %
              (rows, name, ms, entries / 256 / clk, 55.9e6 / (entries / ms)), flush=True)
