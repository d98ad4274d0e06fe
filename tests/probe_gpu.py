"""Diagnostics (not a pytest file): dumps what ds_read_b64_tr_b16 and the 16x16x32 bf16 MFMA
deliver per lane, to confirm the layouts the kernels assume.  The probes live in a library of their
own, not in the product's:  python vision_amd/build_ext.py --probe && python tests/probe_gpu.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "vision_amd", "lib", "libvamd_probe.so"))

dev = torch.device("cuda:0")
out = torch.zeros(256, dtype=torch.int16, device=dev)
lib.vamd_probe_tr16.argtypes = [C.c_void_p, C.c_void_p]
lib.vamd_probe_mfma.argtypes = [C.c_void_p, C.c_void_p]
assert lib.vamd_probe_tr16(out.data_ptr(), None) == 0
torch.cuda.synchronize()
o = out.cpu().view(64, 4).tolist()
ok = True
for l in range(64):
    # expectation: lane l gets column 16*(l>>4) + (l&15), rows 0..3
    exp = [r * 256 + 16 * (l >> 4) + (l & 15) for r in range(4)]
    if o[l] != exp:
        ok = False
    print(f"tr16 lane {l:2d}: got (row,col) {[(v >> 8, v & 255) for v in o[l]]}  expected {[(v >> 8, v & 255) for v in exp]}")
print("TR16 LAYOUT", "OK" if ok else "MISMATCH")
f = torch.zeros(256, dtype=torch.float32, device=dev)
assert lib.vamd_probe_mfma(f.data_ptr(), None) == 0
torch.cuda.synchronize()
o = f.cpu().view(64, 4).tolist()
ok = True
for l in range(64):
    exp = [float(((l >> 4) * 4 + r) * 16 + (l & 15)) for r in range(4)]   # D[row=4*(l>>4)+r][col=l&15] = B[row][col]
    if o[l] != exp:
        ok = False
        print(f"mfma lane {l:2d}: got {o[l]} expected {exp}")
print("MFMA LAYOUT", "OK" if ok else "MISMATCH")
