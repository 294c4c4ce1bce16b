"""ctypes binding of libicgan_hip.so (the C-ABI in include/icgan_hip.h).

The signatures are parsed from the header itself, so the binding cannot drift from
the declared ABI.  There is NO fallback: if the shared library is missing or a call
fails, a RuntimeError is raised (the product path never routes around the HIP kernels).
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "icgan_hip.h")
LIB_PATH = os.path.join(_HERE, "lib", "libicgan_hip.so")

ICG_PRE_RELU, ICG_PRE_AFFINE, ICG_UPSAMPLE2X, ICG_RES_UPSAMPLE2X, ICG_RES_RELU_MASK, ICG_WINO_KEEP_V = 1, 2, 4, 8, 16, 32

_SCALARS = {
    "int": ctypes.c_int, "unsigned": ctypes.c_uint, "float": ctypes.c_float, "double": ctypes.c_double,
    "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t,
}


class AdamTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("numel", ctypes.c_int64)]


class SnLayer(ctypes.Structure):          # icg_sn_layer
    _fields_ = [(n, ctypes.c_void_p) for n in ("w", "u", "sv", "v_out", "u_out", "sigma_out", "w_ohwi", "w_dgrad",
                                               "w_up_fprop", "w_up_dgrad", "w_down_fprop", "w_down_dgrad", "scratch")] + \
               [("scratch_bytes", ctypes.c_size_t), ("rows", ctypes.c_int), ("Cin", ctypes.c_int), ("R", ctypes.c_int),
                ("reserved", ctypes.c_int)]


class SnBwdItem(ctypes.Structure):        # icg_sn_bwd_item
    _fields_ = [(n, ctypes.c_void_p) for n in ("dw_hwio", "dw_ohwi", "dw_up", "dw_down", "w_ohwi", "u", "v", "sigma", "dw", "scratch")] + \
               [("scratch_bytes", ctypes.c_size_t), ("rows", ctypes.c_int), ("Cin", ctypes.c_int), ("R", ctypes.c_int),
                ("accumulate", ctypes.c_int)]


class LinearItem(ctypes.Structure):       # icg_linear_item
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "w", "dy", "out")] + [("N", ctypes.c_int), ("reserved", ctypes.c_int)]


class WinoWeight(ctypes.Structure):        # icg_wino_weight
    _fields_ = [("w", ctypes.c_void_p), ("U", ctypes.c_void_p), ("N", ctypes.c_int), ("K", ctypes.c_int), ("planes", ctypes.c_int),
                ("reserved", ctypes.c_int)]


class EmaTensor(ctypes.Structure):
    _fields_ = [("target", ctypes.c_void_p), ("source", ctypes.c_void_p), ("numel", ctypes.c_int64)]


class Sg2Weight(ctypes.Structure):        # icg_sg2_weight
    _fields_ = [(n, ctypes.c_void_p) for n in ("w", "w_fwd", "w_adj", "wsq", "wscale", "warg")] + \
               [("O", ctypes.c_int), ("I", ctypes.c_int), ("R", ctypes.c_int), ("prenorm", ctypes.c_int), ("gain", ctypes.c_float),
                ("flip", ctypes.c_int), ("dtype", ctypes.c_int), ("reserved", ctypes.c_int)]


class F32Buffer(ctypes.Structure):        # icg_f32_buffer
    _fields_ = [("data", ctypes.c_void_p), ("numel", ctypes.c_int64)]


def parse_header(path: str = HEADER) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """-> {function name: (return type, [(arg type, arg name), ...])} for every prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    src = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(icg_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        alist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.+?)\s*(\w+)$", a)
                alist.append((mm.group(1).strip(), mm.group(2)))
        protos[name] = (ret, alist)
    return protos


def _ctype(t: str):
    t = t.replace("const ", "").strip()
    if t.endswith("*"):
        return ctypes.c_char_p if t == "char*" else ctypes.c_void_p
    return _SCALARS[t]


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"ic_gan_amd: HIP library not built: {LIB_PATH} missing. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (ic_gan_amd/csrc/build.sh). There is no CPU/PyTorch fallback for the hot path.")
        _lib = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        for name, (ret, args) in _protos.items():
            fn = getattr(_lib, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = ctypes.c_char_p if "char" in ret else _ctype(ret)
            fn.argtypes = [_ctype(t) for t, _ in args]
    return _lib


def protos():
    lib()
    return _protos


def _conv(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


_raw_stream = None if os.environ.get("ICG_CALL_STREAM_OBJECT") == "1" else getattr(torch._C, "_cuda_getCurrentRawStream", None)   # (switch: A/B)
_bound: Dict[str, object] = {}


def call(name: str, *args):
    """Call an `int icg_*` entry point on the current stream; raise on a non-zero status.
    Host cost matters: the StyleGAN2 iteration makes ~1 200 of these in 35 ms (profiles/r06_cfg4_host.txt), so the bound function is looked
    up once per name and the stream comes from torch's raw-stream getter (0.2 us) instead of a Stream object (2 - 3 us per call)."""
    fn = _bound.get(name)
    if fn is None:
        fn = _bound[name] = getattr(lib(), name)
    conv, dev = [], -1
    for a in args:
        if a is None or not isinstance(a, torch.Tensor):
            conv.append(a)
        else:
            conv.append(a.data_ptr())
            if dev < 0:
                dev = a.get_device()
    if _raw_stream is not None and dev >= 0:
        st = _raw_stream(dev)
    else:
        st = torch.cuda.current_stream().cuda_stream
    rc = fn(*conv, st)
    if rc != 0:
        l = lib()
        msg = l.icg_strerror(rc).decode()
        raise RuntimeError(f"{name} failed: {msg} (code {rc}, hip error {l.icg_last_hip_error()})")


def query(name: str, *args) -> int:
    """Call a `size_t icg_*_bytes` / `int icg_*_applies` query (host only; results are not cached: some read a switch per call)."""
    fn = _bound.get(name)
    if fn is None:
        fn = _bound[name] = getattr(lib(), name)
    return int(fn(*[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]))
