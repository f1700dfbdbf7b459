"""Runs tools/exp_gather.hip: rows gathered per clock per CU for several lane layouts and table sizes."""
import ctypes, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'libexp_gather.so'))
lib.run_gather.restype = ctypes.c_float
out = torch.zeros(4, device='cuda')
iters, blocks = 512, 256 * 8
rows_per_instr = {0: 32, 1: 32, 2: 16, 3: 8, 4: 64, 5: 32, 6: 32, 7: 32, 8: 32, 9: 32}
for log_rows, label in ((9, '16 KB (L1)'), (16, '2 MB (L2)'), (19, '16 MB (8 XCD L2s / MALL)')):
    x = torch.randn(((1 << log_rows) * 8,), device='cuda')
    for mode in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9):
        ms = lib.run_gather(mode, ctypes.c_void_p(x.data_ptr()), ctypes.c_uint((1 << log_rows) - 1), iters, blocks,
                            ctypes.c_void_p(out.data_ptr()), 5)
        instr = blocks * 4 * iters
        rows = instr * rows_per_instr[mode]
        clk = ms * 1e-3 * 2.4e9
        print('%-26s mode %d: %.3f ms  %.2f clk/instr/CU  %.3f rows/clk/CU  %.2f G rows/s' %
              (label, mode, ms, clk / (instr / 256), rows / 256 / clk, rows / ms / 1e6), flush=True)
