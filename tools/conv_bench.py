#!/usr/bin/env python
"""Kernel-level benchmark of the convolution triplet on the cfg3 layer shapes (tools only; not the product).
Prints TFLOP/s per layer for fprop / dgrad / wgrad, plus the sustained fp32-MFMA rate of this box."""
import ctypes, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_gan_amd._lib as L
if os.environ.get('ICG_LIB'):
    L.LIB_PATH = os.environ['ICG_LIB']

def ev_time(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def mfma_peak():
    so = os.path.join(ROOT, "tools", "mfma_peak.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "mfma_peak.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    res = {}
    for blocks in (256, 2048):
        out = torch.empty(blocks * 256, device="cuda")
        iters = 4000
        t = ev_time(lambda: lib.mfma_peak_launch(ctypes.c_void_p(out.data_ptr()), blocks, iters,
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 3, 1)
        flops = blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 2
        res[blocks] = round(flops / t / 1e12, 1)
    return res

# (name, B, H(out), W, Cin, Cout, R, flags)   cfg3: G with B=64, D with 2B=128
P_RELU, P_AFF, UP = 1, 2, 4
LAYERS = [
    ("G.b5.conv1 192->96 @256 up+bn", 64, 256, 256, 192, 96, 3, P_RELU | P_AFF | UP),
    ("G.b5.conv2 96->96 @256 bn", 64, 256, 256, 96, 96, 3, P_RELU | P_AFF),
    ("G.b4.conv1 384->192 @128 up+bn", 64, 128, 128, 384, 192, 3, P_RELU | P_AFF | UP),
    ("G.b3.conv1 768->384 @64 up+bn", 64, 64, 64, 768, 384, 3, P_RELU | P_AFF | UP),
    ("G.b1.conv1 1536->768 @16 up+bn", 64, 16, 16, 1536, 768, 3, P_RELU | P_AFF | UP),
    ("D.b1.conv1 96->192 @128 relu", 128, 128, 128, 96, 192, 3, P_RELU),
    ("D.b0.conv2 96->96 @256 relu", 128, 256, 256, 96, 96, 3, P_RELU),
    ("D.b3.conv2 768->768 @32 relu", 128, 32, 32, 768, 768, 3, P_RELU),
    ("G.b5.sc 1x1 192->96 @128", 64, 128, 128, 192, 96, 1, 0),
    # low-occupancy shapes: StyleGAN2 cfg4 (batch 16) and small-batch sampling
    ("SG2 512->512 @4  B16", 16, 4, 4, 512, 512, 3, 0),
    ("SG2 512->512 @8  B16", 16, 8, 8, 512, 512, 3, 0),
    ("SG2 512->512 @16 B16", 16, 16, 16, 512, 512, 3, 0),
    ("SG2 512->512 @32 B16", 16, 32, 32, 512, 512, 3, 0),
    ("SG2 256->256 @64 B16", 16, 64, 64, 256, 256, 3, 0),
    ("SG2 128->128 @128 B16", 16, 128, 128, 128, 128, 3, 0),
    ("SG2 64->64 @256 B16", 16, 256, 256, 64, 64, 3, 0),
    ("G.b0.conv2 1536->1536 @8 B8 (sampling)", 8, 8, 8, 1536, 1536, 3, P_RELU | P_AFF),
]

def main():
    only = sys.argv[1:]
    print("mfma_peak TF by blocks:", mfma_peak(), flush=True)
    for name, B, H, W, Cin, Cout, R, flags in LAYERS:
        if only and not any(o in name for o in only): continue
        up = 1 if flags & UP else 0
        dev = "cuda"
        x = torch.randn(B, Cin, H >> up, W >> up, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, R, R, Cin, device=dev) / (R * R * Cin) ** 0.5
        wd = torch.randn(Cin, R, R, Cout, device=dev) / (R * R * Cout) ** 0.5
        out = torch.empty(B, Cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(B, Cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        da = torch.empty(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
        sc = torch.rand(B, Cin, device=dev) + 0.5 if flags & P_AFF else None
        sh = torch.randn(B, Cin, device=dev) * 0.1 if flags & P_AFF else None
        ssb = Cin if flags & P_AFF else 0
        flops = 2.0 * B * H * W * Cin * Cout * R * R
        nf = L.query("icg_conv2d_fprop_workspace_bytes", B, H, W, Cin, Cout, R, flags)
        wf = torch.empty(max(nf, 16), dtype=torch.uint8, device=dev)
        t_f = ev_time(lambda: L.call("icg_conv2d_fprop_ws", x, w, None, None, out, sc, sh, ssb, B, H, W, Cin, Cout, R, flags, 1.0, wf, nf))
        nd = L.query("icg_conv2d_fprop_workspace_bytes", B, H, W, Cout, Cin, R, 0)
        wdd = torch.empty(max(nd, 16), dtype=torch.uint8, device=dev)
        t_d = ev_time(lambda: L.call("icg_conv2d_fprop_ws", dy, wd, None, None, da, None, None, 0, B, H, W, Cout, Cin, R, 0, 1.0, wdd, nd))
        nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, W, Cin, Cout, R)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
        dw = torch.empty(R * R * Cin * Cout, device=dev)
        t_w = ev_time(lambda: L.call("icg_conv2d_wgrad", x, dy, dw, sc, sh, ssb, B, H, W, Cin, Cout, R, flags, ws, nb))
        print(f"{name:34s} GF {flops/1e9:8.1f}  fprop {flops/t_f/1e12:6.1f} TF ({t_f*1e3:7.3f} ms)  dgrad {flops/t_d/1e12:6.1f} TF"
              f"  wgrad {flops/t_w/1e12:6.1f} TF ({t_w*1e3:7.3f} ms, ws {nb>>20} MiB)", flush=True)

if __name__ == "__main__":
    main()
