import ctypes, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
so = os.path.join(ROOT, 'tools', 'libexp_dpp.so')
lib = ctypes.CDLL(so)
x = torch.arange(64, dtype=torch.float32, device='cuda') + 1
o = [torch.zeros(64, device='cuda') for _ in range(4)]
lib.run_swap(ctypes.c_void_p(x.data_ptr()), *[ctypes.c_void_p(t.data_ptr()) for t in o]); torch.cuda.synchronize()
print('in   ', x.int().tolist())
print('16 r0', o[0].int().tolist()); print('16 r1', o[1].int().tolist())
print('32 r0', o[2].int().tolist()); print('32 r1', o[3].int().tolist())
x = torch.randn(64, device='cuda')
for s in (1, 2, 4, 8, 16):
    out = torch.zeros(64, device='cuda')
    lib.run(s, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr())); torch.cuda.synchronize()
    want = torch.stack([x[(torch.arange(64, device='cuda') % s) == (l % s)].sum() for l in range(64)])
    print('stride', s, 'max err', float((out - want).abs().max()))
out = torch.zeros(64, device='cuda')
lib.run(0, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr())); torch.cuda.synchronize()
print('max', float((out - x.max()).abs().max()))
