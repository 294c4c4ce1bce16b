#!/usr/bin/env python
"""Roofline check of the HBM-bound kernels (tools only): algorithmic bytes / HIP-event time at the sizes the step uses,
against 8 TB/s (spec) — a float4 copy reaches ≈6.3 TB/s on this part.  Prints one line per kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_gan_amd._lib as L
from ic_gan_amd import ops
from ic_gan_amd.stylegan_ops import bias_act as BA, upfirdn2d as UF

dev = "cuda:0"
PEAK = 8000.0


def ev(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, nbytes, t):
    gbs = nbytes / t / 1e9
    print(f"{name:58s} {nbytes / 1e9:7.2f} GB  {t * 1e3:8.3f} ms  {gbs:7.0f} GB/s  {gbs / PEAK:5.2f} of 8 TB/s", flush=True)


def cl(*shape):
    return torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last)


def main():
    # reference point: device copy
    a = torch.empty(256 << 20, device=dev); b = torch.empty_like(a)
    report("copy (torch) 1 GiB", 2 * a.numel() * 4, ev(lambda: b.copy_(a)))
    # ... and hand-written float4 copies (tools/hbm_copy.hip): this box's ceiling for a read + write stream and how the access shape
    # moves it -- U float4 in flight per thread, one block-iteration per workgroup (grid = n / (256 U)) or a grid-stride loop over a
    # fixed grid, non-temporal hints
    import ctypes, subprocess
    so = os.path.join(ROOT, "tools", "hbm_copy.so")
    if not os.path.exists(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "hbm_copy.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    n4 = a.numel() // 4
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    best = 0.0
    for name, var, u, byt in (("copy float4 x1", 0, 1, 8), ("copy float4 x2", 1, 2, 8), ("copy float4 x4", 2, 4, 8), ("copy float4 x8", 3, 8, 8),
                              ("copy float4 x4 non-temporal", 4, 4, 8), ("read-only float4 x4", 5, 4, 4), ("write-only float4 x4", 6, 4, 4)):
        for grid_name, blocks in (("full grid", (n4 + 256 * u - 1) // (256 * u)), ("2048 blocks", 2048), ("8192 blocks", 8192)):
            t = ev(lambda: lib.hbm_copy_launch(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_long(n4), var,
                                               int(blocks), st()))
            report(f"{name}, {grid_name}", byt * a.numel(), t)
            if byt == 8:
                best = max(best, 4.0 * a.numel() * 2 / t / 1e9)
    print(f"# copy ceiling of this box: {best:.0f} GB/s = {best / PEAK:.2f} of 8 TB/s", flush=True)
    del a, b

    # ---- BigGAN cfg3: largest BN layer [64, 96, 256, 256] ----
    B, C, H, W = 64, 96, 256, 256
    x = cl(B, C, H, W); n = x.numel(); rows = B * H * W
    rm = torch.zeros(C, device=dev)
    nb = L.query("icg_bn_workspace_bytes", rows, C); ws = ops._bytes(nb, dev)
    report("bn_partial_stats [64,96,256,256]", 4 * n, ev(lambda: L.call("icg_bn_partial_stats", x, rm, rows, C, ws, nb)))
    da = cl(B, C, H, W); dx = torch.empty_like(x)
    scale = torch.rand(B, C, device=dev) + 0.5; shift = torch.randn(B, C, device=dev); mean = torch.zeros(C, device=dev)
    ca = torch.randn(C, device=dev) * 0.01; cb = torch.randn(C, device=dev) * 0.01
    flags = 1 | 2
    nbb = L.query("icg_bn_bwd_workspace_bytes", B, H, W, C); wsb = ops._bytes(nbb, dev)
    sd = torch.empty(B, C, device=dev); sx = torch.empty(B, C, device=dev)
    report("bn_bwd_reduce  (reads x, da)", 8 * n,
           ev(lambda: L.call("icg_bn_bwd_reduce", x, da, scale, shift, C, mean, B, H, W, C, flags, wsb, nbb, sd, sx)))
    report("bn_bwd_apply   (reads x, da; writes dx)", 12 * n,
           ev(lambda: L.call("icg_bn_bwd_apply", x, da, scale, shift, C, mean, ca, cb, B, H, W, C, flags, dx)))
    y = torch.empty_like(x)
    report("bn_apply (stand-alone) r+w", 8 * n, ev(lambda: L.call("icg_bn_apply", x, scale, shift, C, B, H * W, C, flags, y)))
    report("relu_fwd r+w", 8 * n, ev(lambda: L.call("icg_relu_fwd", x, y, n)))
    nbc = L.query("icg_colsum_workspace_bytes", rows, C); wsc = ops._bytes(nbc, dev); cs = torch.empty(C, device=dev)
    report("colsum (bias gradient) [4.2M x 96]", 4 * n, ev(lambda: L.call("icg_colsum", x, rows, C, cs, wsc, nbc)))
    p = torch.empty(B, C, H // 2, W // 2, device=dev).contiguous(memory_format=torch.channels_last)
    report("avgpool2_fwd", 5 * n, ev(lambda: L.call("icg_avgpool2_fwd", x, None, p, B, H, W, C)))
    del da, dx, y, p
    img = cl(B, 3, H, W); out = torch.empty_like(img)
    report("tanh_fwd [64,3,256,256]", 8 * img.numel(), ev(lambda: L.call("icg_tanh_fwd", img, out, img.numel())))
    del x, img, out

    # attention softmax: beta [64*4096, 1024]
    r, c = 64 * 4096, 1024
    s_in = torch.randn(r, c, device=dev); s_out = torch.empty_like(s_in)
    report("softmax_fwd [262144 x 1024]", 8 * r * c, ev(lambda: L.call("icg_softmax_fwd", s_in, s_out, r, c)))
    g = torch.randn(r, c, device=dev)
    report("softmax_bwd", 12 * r * c, ev(lambda: L.call("icg_softmax_bwd", s_out, g, s_in, r, c)))
    del s_in, s_out, g

    # optimiser: 100 M parameters in 60 tensors
    ps = [torch.randn(1_700_000, device=dev) for _ in range(60)]
    gs = [torch.randn_like(t) for t in ps]; ms = [torch.zeros_like(t) for t in ps]; vs = [torch.zeros_like(t) for t in ps]
    tot = sum(t.numel() for t in ps)
    report("adam_multi 102 M params (28 B/param)", 28 * tot, ev(lambda: ops.adam_multi(ps, gs, ms, vs, 1e-4, 0.0, 0.999, 1e-6, 3)))
    report("ema_multi  102 M entries (12 B/entry)", 12 * tot, ev(lambda: ops.ema_multi(ms, ps, 0.999)))
    del ps, gs, ms, vs

    # spectral norm of the largest layer [1536, 1536*9]
    wgt = torch.randn(1536, 1536, 3, 3, device=dev) * 0.01
    u = torch.randn(1, 1536, device=dev); sv = torch.ones(1, device=dev)
    t = ev(lambda: ops.sn_prepare(wgt, u, sv, 1e-6, True, True))
    report("sn_forward [1536 x 13824] (3 reads + 2 writes of W)", 5 * wgt.numel() * 4, t)
    del wgt

    # ---- StyleGAN2 cfg4 sizes ----
    for fmt in ("nchw", "nhwc"):
        xs = torch.randn(16, 64, 256, 256, device=dev)
        if fmt == "nhwc":
            xs = xs.contiguous(memory_format=torch.channels_last)
        bias = torch.randn(64, device=dev)
        report(f"bias_act lrelu fwd [16,64,256,256] {fmt}", 8 * xs.numel(), ev(lambda: BA.bias_act(xs, bias, act="lrelu")))
        xg = xs.clone().requires_grad_(True)
        yy = BA.bias_act(xg, bias, act="lrelu"); gy = torch.randn_like(yy)
        report(f"bias_act lrelu bwd (dy, y -> dx) {fmt}", 12 * xs.numel(),
               ev(lambda: torch.autograd.grad(yy, xg, gy, retain_graph=True)))
        del xg, yy, gy
    f = UF.setup_filter([1, 3, 3, 1], device=dev)
    xs = torch.randn(16, 64, 257, 257, device=dev)
    report("upfirdn2d blur after up-conv [16,64,257,257]->256 (nchw)", 4 * (xs.numel() + 16 * 64 * 256 * 256),
           ev(lambda: UF.upfirdn2d(xs, f, padding=[1, 1, 1, 1], gain=4)))
    xs = torch.randn(16, 64, 256, 256, device=dev)
    report("upfirdn2d blur before down-conv [16,64,256,256]->257", 4 * (xs.numel() + 16 * 64 * 257 * 257),
           ev(lambda: UF.upfirdn2d(xs, f, padding=[2, 2, 2, 2])))
    xs = cl(16, 64, 257, 257)
    report("upfirdn2d_nhwc blur after up-conv [16,64,257,257]->256", 4 * (xs.numel() + 16 * 64 * 256 * 256),
           ev(lambda: UF.upfirdn2d(xs, f, padding=[1, 1, 1, 1], gain=4)))
    xs = cl(16, 64, 256, 256)
    report("upfirdn2d_nhwc blur before down-conv [16,64,256,256]->257", 4 * (xs.numel() + 16 * 64 * 257 * 257),
           ev(lambda: UF.upfirdn2d(xs, f, padding=[2, 2, 2, 2])))
    report("upfirdn2d_nhwc down=2 (1x1 skip) [16,64,256,256]->128", 4 * xs.numel() * 1.25,
           ev(lambda: UF.upfirdn2d(xs, f, down=2, padding=[1, 1, 1, 1])))
    xs = cl(16, 512, 16, 16)
    report("upfirdn2d_nhwc blur [16,512,17,17]-ish small", 4 * 2 * xs.numel(), ev(lambda: UF.upfirdn2d(xs, f, padding=[2, 1, 2, 1])))
    xs = torch.randn(16, 3, 128, 128, device=dev)
    report("upsample2d img [16,3,128,128]->256", 4 * xs.numel() * 5, ev(lambda: UF.upsample2d(xs, f)))
    xs = torch.randn(16, 64, 256, 256, device=dev)
    report("downsample2d-style 1x1 skip [16,64,256,256]->128", 4 * xs.numel() * 1.25,
           ev(lambda: UF.upfirdn2d(xs, f, down=2, padding=[1, 1, 1, 1])))
    # ---- fp16 storage (the reference's num_fp16_res blocks): half the bytes per element, fp32 arithmetic
    xs = cl(16, 64, 256, 256).half()
    bias = torch.randn(64, device=dev).half()
    report("bias_act lrelu fwd fp16 [16,64,256,256] nhwc", 4 * xs.numel(), ev(lambda: BA.bias_act(xs, bias, act="lrelu", clamp=256)))
    xs = cl(16, 64, 257, 257).half()
    report("upfirdn2d_nhwc fp16 blur after up-conv [16,64,257,257]->256", 2 * (xs.numel() + 16 * 64 * 256 * 256),
           ev(lambda: UF.upfirdn2d(xs, f, padding=[1, 1, 1, 1], gain=4)))
    # ---- fused StyleGAN2 layer kernels (csrc/sg2_fused.hip, round 5) at cfg4's largest fp16 layer [16, 64, 256, 256] and a wide one [16, 512, 32, 32]
    for (N, C, R) in ((16, 64, 256), (16, 512, 32)):
        HW = R * R
        x = cl(N, C, R, R).half(); y = torch.empty_like(x); y2 = torch.empty_like(x)
        n = x.numel()
        s = torch.rand(N, C, device=dev) + 0.5; d = torch.rand(N, C, device=dev) + 0.5
        noise = torch.randn(N, HW, device=dev); strength = torch.full((1,), 0.1, device=dev); bias = torch.randn(C, device=dev)
        tag = "[%d,%d,%d,%d] fp16" % (N, C, R, R)
        report("sg2_modulate " + tag, 4 * n, ev(lambda: L.call("icg_sg2_modulate", x, s, y, N, HW, C, 1)))
        report("sg2_act_fwd (demod + noise + bias + lrelu + clamp) " + tag, 4 * n,
               ev(lambda: L.call("icg_sg2_act_fwd", x, d, noise, HW, strength, bias, y, N, HW, C, 3, 0.2, 1.414, 256.0, 1)))
        sums, tot = torch.empty(N, 2 * C + 1, device=dev), torch.empty(2 * C + 1, device=dev)
        nb = L.query("icg_sg2_rows_workspace_bytes", N, HW, C, 2 * C + 1, 1); ws = ops._bytes(nb, dev)
        report("sg2_act_bwd (dy, y, c -> dc + bias / demod / noise sums) " + tag, 8 * n,
               ev(lambda: L.call("icg_sg2_act_bwd", x, y, y, d, noise, HW, y2, sums, tot, N, HW, C, 3, 0.2, 1.414, 256.0, 1, ws, nb)))
        ds = torch.empty(N, C, device=dev)
        nb2 = L.query("icg_sg2_rows_workspace_bytes", N, HW, C, C, 1); ws2 = ops._bytes(nb2, dev)
        report("sg2_modulate_bwd (dxs, x -> dx + style sums) " + tag, 6 * n,
               ev(lambda: L.call("icg_sg2_modulate_bwd", x, y, s, y2, ds, N, HW, C, 1, ws2, nb2)))
        w3 = torch.randn(3, C, device=dev) * 0.1; b3 = torch.randn(3, device=dev)
        img = torch.empty(N, 3, HW, device=dev); y3 = torch.empty(N, HW, 3, device=dev, dtype=torch.float16)
        report("sg2_torgb_fwd (x -> image) " + tag, 2 * n + 22 * N * HW,
               ev(lambda: L.call("icg_sg2_torgb_fwd", x, s, w3, b3, 256.0, None, img, y3, N, HW, C, 1)))
        sums4, tot4 = torch.empty(N, 4 * C + 3, device=dev), torch.empty(4 * C + 3, device=dev)
        nb3 = L.query("icg_sg2_torgb_bwd_workspace_bytes", N, HW, C, 1); ws3 = ops._bytes(nb3, dev)
        report("sg2_torgb_bwd (x -> dx + style / weight sums) " + tag, 4 * n + 18 * N * HW,
               ev(lambda: L.call("icg_sg2_torgb_bwd", img, y3, x, s, w3, 256.0, 1, y2, sums4, tot4, N, HW, C, 1, ws3, nb3)))
        if C <= 128:
            xi = torch.randn(N, 3, HW, device=dev).half(); wf = (torch.randn(C, 3, device=dev) * 0.5).half()
            report("sg2_fromrgb_fwd (image -> y) " + tag, 2 * n + 6 * N * HW,
                   ev(lambda: L.call("icg_sg2_fromrgb_fwd", xi, wf, bias, y, N, HW, C, 3, 0.2, 1.414, 256.0, 1)))
            tot5 = torch.empty(4 * C, device=dev); di = torch.empty_like(xi)
            nb5 = L.query("icg_sg2_rows_workspace_bytes", N, HW, C, 4 * C, 1); ws5 = ops._bytes(nb5, dev)
            report("sg2_fromrgb_bwd (dy, y -> weight sums + dimg) " + tag, 4 * n + 12 * N * HW,
                   ev(lambda: L.call("icg_sg2_fromrgb_bwd", x, y, xi, wf, di, tot5, N, HW, C, 3, 0.2, 1.414, 256.0, 1, ws5, nb5)))
        xin = cl(N, C, R + 1, R + 1).half()
        report("sg2_fir_act_fwd (blur 4x4 + epilogue; writes c and y) [%d,%d,%d,%d]->%d fp16" % (N, C, R + 1, R + 1, R), 2 * xin.numel() + 4 * n,
               ev(lambda: L.call("icg_sg2_fir_act_fwd", xin, f, y, y2, d, noise, HW, strength, bias, N, C, R + 1, R + 1, 4, 4, 1, 1, 1, 1, 0, 4.0, R, R, 3,
                                 0.2, 1.414, 256.0, 1)))
        report("sg2 blur (fir without epilogue) [%d,%d,%d,%d]->%d fp16" % (N, C, R + 1, R + 1, R), 2 * xin.numel() + 2 * n,
               ev(lambda: L.call("icg_sg2_fir_act_fwd", xin, f, None, y2, None, None, 0, None, None, N, C, R + 1, R + 1, 4, 4, 1, 1, 1, 1, 0, 4.0, R, R, 1,
                                 0.2, 1.0, -1.0, 1)))
        del x, y, y2, xin
    # weight preparation of cfg4's generator (13 modulated 3x3 layers, 512 ... 64 channels) in one call: reads W, writes 2 fp16 layouts + wsq
    from ic_gan_amd.stylegan_ops import fused_layers as FL
    chans = [(512, 512)] * 7 + [(256, 512), (256, 256), (128, 256), (128, 128), (64, 128), (64, 64)]
    items, nbytes = [], 0
    for (O, I) in chans:
        w = torch.randn(O, I, 3, 3, device=dev)
        items.append(dict(w=w, w_fwd=torch.empty(O, 3, 3, I, device=dev, dtype=torch.float16), w_adj=torch.empty(I, 3, 3, O, device=dev, dtype=torch.float16),
                          wsq=torch.empty(O, I, device=dev), wscale=torch.empty(O, device=dev), warg=torch.empty(O, device=dev, dtype=torch.int32),
                          prenorm=True, gain=0.02, flip=False))
        nbytes += w.numel() * (4 + 4 + 2 + 2) + O * I * 4          # (the rows kernel and the layout kernel each read W once)
    report("sg2_weight_prep_multi, 13 layers of cfg4's G (two launches)", nbytes, ev(lambda: ops.sg2_weight_prep_multi(items)))
    del items

    # bias gradient out of the Winograd dy transform (replaces the colsum pass): the pass itself, with and without the sums
    B, C, H = 64, 96, 256
    dy = cl(B, C, H, H)
    T = B * (H // 4) * (H // 4)
    v = torch.empty(36 * T * C, device=dev)
    dw, db = torch.empty(9 * C * C, device=dev), torch.empty(C, device=dev)
    nb = L.query("icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes", B, H, H, C, C, 36); ws = ops._bytes(nb, dev)
    t1 = ev(lambda: L.call("icg_conv2d_wino4_wgrad_from_v", v, dy, dw, B, H, H, C, C, 36, 0, 1.0, ws, nb), 5, 2)
    t2 = ev(lambda: L.call("icg_conv2d_wino4_wgrad_from_v_db", v, dy, dw, db, B, H, H, C, C, 36, 0, 1.0, ws, nb), 5, 2)
    print(f"wino4 wgrad from V 96->96@256 B64: {t1 * 1e3:.3f} ms; with the bias-gradient sums in the dy transform: {t2 * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
