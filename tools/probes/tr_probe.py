#!/usr/bin/env python
"""What ds_read_b64_tr_b16 delivers (gfx950): LDS holds fp16 element indices 0..2047; every lane supplies a byte address; prints,
per lane, the four element indices it received.  Tools only (design input for a transposing fp16 weight-gradient loader)."""
import ctypes, os, subprocess, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(HERE, "tr_probe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(HERE, "tr_probe.hip"), "-o", so])
hip = ctypes.CDLL("libamdhip64.so")
lib = ctypes.CDLL(so)
mod = ctypes.c_void_p(); fn = ctypes.c_void_p()
# launch through hipModule API is unnecessary: use hipLaunchKernel on the symbol
hip.hipLaunchKernel.argtypes = [ctypes.c_void_p, ctypes.c_uint * 3, ctypes.c_uint * 3, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]


class dim3(ctypes.Structure):
    _fields_ = [("x", ctypes.c_uint), ("y", ctypes.c_uint), ("z", ctypes.c_uint)]


hip.hipLaunchKernel.argtypes = [ctypes.c_void_p, dim3, dim3, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_void_p]
sym = ctypes.cast(lib.probe, ctypes.c_void_p)
patterns = {
    "A: lane l -> byte 8*l (elements 4l..4l+3)": [8 * l for l in range(64)],
    "B: rows of 64 B; lane -> row (l&15)>>2 + 4*(l>>4), chunk l&3 (8 B)": [(((l & 15) >> 2) + 4 * (l >> 4)) * 64 + (l & 3) * 8 for l in range(64)],
    "C: rows of 256 B; lane -> row (l&15)>>2 + 4*(l>>4), chunk l&3": [(((l & 15) >> 2) + 4 * (l >> 4)) * 256 + (l & 3) * 8 for l in range(64)],
    "D: lane l -> byte 8*(l&15) + 512*(l>>4)": [8 * (l & 15) + 512 * (l >> 4) for l in range(64)],
}
for name, addrs in patterns.items():
    a = torch.tensor(addrs, dtype=torch.int32, device="cuda")
    out = torch.zeros(256, device="cuda")
    pa, po = ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr())
    args = (ctypes.c_void_p * 2)(ctypes.cast(ctypes.pointer(pa), ctypes.c_void_p), ctypes.cast(ctypes.pointer(po), ctypes.c_void_p))
    rc = hip.hipLaunchKernel(sym, dim3(1, 1, 1), dim3(64, 1, 1), args, 0, None)
    torch.cuda.synchronize()
    o = out.view(64, 4).cpu().int().tolist()
    print(name, "rc", rc)
    for l in range(64):
        print("  lane %2d addr %5d (elem %4d): %s" % (l, addrs[l], addrs[l] // 2, o[l]))
