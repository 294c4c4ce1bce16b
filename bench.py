#!/usr/bin/env python
"""bench.py — images/sec of the IC-GAN BigGAN G+D training step on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one call of ic_gan_amd.train_fns.GAN_training_function.train on one synthetic batch that is already
resident in HBM: 1 D update (G forward no-grad on B, D forward/backward on fake++real 2B, Adam) + 1 G update
(G forward, D forward, backward through D and G, Adam) + EMA — exactly reference train_fns.py:40-191 with
num_D_steps = num_D_accumulations = num_G_accumulations = 1.  Workload = BASELINE.json configs[2] ("cfg3"): class +
instance conditional BigGAN 256x256, ch 96, B = 64 per GPU (the configuration the metric is quoted on; it fits one
288 GB MI355X).  Weights: the reference's orthogonal init; data: synthetic.

Output: ONE JSON line on rank 0 (see the driver contract) with two extra objects:
  roofline     — for the dominant kernel by time (an fp32-MFMA GEMM): FLOPs the kernel EXECUTED per launch / its average
                 launch duration (HIP events on the launch stream, inside the timed region) / the 157.3 TFLOP/s fp32-MFMA
                 peak = `frac`.  The same time priced on the reference op graph's FLOPs is `algorithmic_tflops` (above the
                 peak where Winograd / resample-fused identities remove multiply-adds: `algorithmic_speedup`).
                 `roofline.step` prices the whole step: executed TFLOP at the MFMA peak vs measured HBM GB at 8 TB/s.
  cpu_baseline — the CPU oracle ("port": oracle/biggan_oracle.py, pinned to reference-generated goldens incl. the real
                 widths) timed on this box's host cores on a bounded sample of the same workload; the unmodified reference
                 cannot be timed here (/root/reference does not exist on the GPU box)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: fp16 / bf16 MFMA, dense (StyleGAN2's fp16 blocks: v_mfma_f32_16x16x32_f16)
PEAK_HBM_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s achievable by a float4 copy)

WORKLOADS = {
    # name: (config overrides, per-GPU batch)
    "cfg3": (dict(resolution=256, G_ch=96, D_ch=96, class_cond=True, instance_cond=True, G_attn="64", D_attn="64"), 64),
    "cfg2": (dict(resolution=128, G_ch=96, D_ch=96, class_cond=False, instance_cond=True, G_attn="64", D_attn="64"), 64),
    "cfg1": (dict(resolution=64, G_ch=64, D_ch=64, class_cond=False, instance_cond=True, G_attn="32", D_attn="32"), 8),
    # BASELINE.json configs[4]: IC-GAN BigGAN-deep 256x256 ch=128 bs=128/GPU, attention at 64 -- the instance-conditioned
    # extension of ic_gan_amd/BigGANdeep.py (the reference's BigGANdeep.py cannot train IC-GAN) driven by the same
    # train_fns.GAN_training_function step (1 D + 1 G update, Adam x2, EMA)
    "cfg5": (dict(model="BigGANdeep", resolution=256, G_ch=128, D_ch=128, G_depth=2, D_depth=2, dim_z=128, shared_dim=128,
                  class_cond=False, instance_cond=True, G_attn="64", D_attn="64", G_lr=5e-5, D_lr=2e-4), 128),
}
BASE_CFG = dict(dim_z=120, shared_dim=128, shared_dim_feat=512, G_shared=True, G_shared_feat=True, hier=True,
                n_classes=1000, SN_eps=1e-6, BN_eps=1e-5, adam_eps=1e-6, G_lr=4e-5, D_lr=1e-4, G_B1=0.0, G_B2=0.999,
                D_B1=0.0, D_B2=0.999, ema=True, ema_decay=0.9999, ema_start=20000, toggle_grads=True, num_D_steps=1,
                num_D_accumulations=1, num_G_accumulations=1, split_D=False, DiffAugment="", DA=False, D_ortho=0.0,
                G_ortho=0.0, G_init="ortho", D_init="ortho")


class KernelTimer:
    """HIP-event timing of every convolution launch (fprop and dgrad go through icg_conv2d_fprop, wgrad through
    icg_conv2d_wgrad, the resample-fused forms through icg_conv2d_up_* / icg_conv2d_down_*) on the stream the kernels
    are launched on (torch's current stream).  Records are keyed by the kernel's rocprofv3 name, obtained from the
    library (icg_gemm_last_variant), so they can be compared line by line with profiles/*_kernel_stats.csv."""

    # entry point -> (index of B in the argument list, kind)
    SPEC = {"icg_conv2d_fprop": (8, "conv"), "icg_conv2d_fprop_ws": (8, "conv"), "icg_conv2d_wino_fprop": (8, "wino"), "icg_conv2d_wino4_fprop": (8, "wino4"), "icg_conv2d_wino_wgrad": (6, "wino"), "icg_conv2d_wino4_wgrad": (6, "wino4"), "icg_conv2d_wino4_wgrad_from_v": (3, "from_v"), "icg_conv2d_wino4_wgrad_from_v_db": (4, "from_v"),
            "icg_conv2d_up_wino_fprop": (7, "rs_up"), "icg_conv2d_up_wino_dgrad": (3, "rs_up"), "icg_conv2d_up_wino_wgrad": (6, "rs_up"),
            "icg_conv2d_down_wino_fprop": (5, "rs_down"), "icg_conv2d_down_wino_dgrad": (3, "rs_down"), "icg_conv2d_down_wino_dgrad_relu": (4, "rs_down"), "icg_conv2d_down_wino_wgrad": (3, "rs_down"),
            "icg_conv2d_wgrad": (6, "conv"), "icg_conv2d_up_fprop": (7, "up"),
            "icg_conv2d_up_dgrad": (3, "up"), "icg_conv2d_up_wgrad": (6, "up"), "icg_conv2d_down_fprop": (5, "up"),
            "icg_conv2d_down_dgrad": (3, "up"), "icg_conv2d_down_dgrad_relu": (4, "up"), "icg_conv2d_down_wgrad": (3, "up"),
            # StyleGAN2 (cfg4): general-geometry convolutions and the two HBM-bound plugins
            "icg_conv2d_g_fprop": (4, "gconv"), "icg_conv2d_g_fprop_ws": (4, "gconv"), "icg_conv2d_tr2_fprop": (4, "tr2"),
            "icg_conv2d_g_wgrad": (3, "gconv"), "icg_conv2d_g_fprop_f16": (3, "hconv"), "icg_conv2d_g_wgrad_f16": (3, "hwgrad"), "icg_bias_act": (6, "bias_act"), "icg_bias_act_typed": (6, "bias_act"),
            "icg_upfirdn2d": (3, "upfirdn2d"), "icg_upfirdn2d_nhwc": (3, "upfirdn2d"), "icg_upfirdn2d_typed": (3, "upfirdn2d"),
            # fused StyleGAN2 layers (round 5): the fp16 convolution with the layer epilogue on its accumulators / the style scale on its
            # A fragments, and the HBM-bound row kernels of csrc/sg2_fused.hip (algorithmic bytes = one pass over each activation operand)
            "icg_conv2d_g_fprop_f16_act": (13, "hconv"), "icg_modconv2d_f16": (14, "hconv"),
            "icg_sg2_act_fwd": (7, "sg2_rows"), "icg_sg2_modulate": (3, "sg2_rows"), "icg_sg2_act_bwd": (9, "sg2_rows"),
            "icg_sg2_modulate_bwd": (5, "sg2_rows"), "icg_sg2_torgb_fwd": (8, "sg2_rows"), "icg_sg2_torgb_bwd": (10, "sg2_rows"),
            "icg_sg2_fromrgb_fwd": (4, "sg2_rows"), "icg_sg2_fromrgb_bwd": (6, "sg2_rows"), "icg_sg2_fir_act_fwd": (9, "sg2_fir")}

    def __init__(self, period=4):
        # Stratified sampling: every launch is COUNTED per (entry point, shape); every `period`-th launch of such a key is bracketed
        # by HIP events.  A key's launches do identical work, so its time is (mean bracketed launch) x (launch count); period 1
        # brackets every launch.  An event pair drains the stream's launch pipeline for a few microseconds -- bracketing all of the
        # ~1300 launches of a cfg3 step costs ~2.7 ms/step of the very time being measured (`--timer-period 1` to see it).
        self.period = max(int(period), 1)
        self.keys = {}         # (entry point, shape ints) -> [kernel name, alg flops, exe flops, alg bytes, launches, [(start, end), ...]]
        self.hbm_keys = {}     # the same for the HBM-bound StyleGAN2 plugins: -> [op name, bytes, launches, [(start, end), ...]]
        self.enabled = False

    @property
    def records(self):
        """[(kernel name, alg flops, exe flops, alg bytes, seconds)] per counted launch (seconds: the key's mean bracketed launch)"""
        out = []
        for kname, alg, exe, byt, n, ev in self.keys.values():
            if not ev:
                continue
            mean = sum(s.elapsed_time(e) for s, e in ev) * 1e-3 / len(ev)
            out.extend([(kname, alg, exe, byt, mean)] * n)
        return out

    @property
    def hbm_records(self):
        out = []
        for name, byt, n, ev in self.hbm_keys.values():
            if ev:
                mean = sum(s.elapsed_time(e) for s, e in ev) * 1e-3 / len(ev)
                out.extend([(name, byt, mean)] * n)
        return out

    def install(self):
        import ctypes
        import ic_gan_amd._lib as L
        raw = L.call
        timer = self
        last = (ctypes.c_int * 4)()
        query = L.lib().icg_gemm_last_variant

        def timed_call(name, *args):
            if not timer.enabled or name not in timer.SPEC:
                return raw(name, *args)
            sl, mode = timer.SPEC[name]
            if mode in ("bias_act", "upfirdn2d", "sg2_rows", "sg2_fir"):          # HBM-bound plugins: algorithmic bytes / HIP-event time
                if mode == "sg2_rows":       # N, HW, C at args[sl:sl + 3]; storage type = the activation tensors'; one pass per activation operand
                    N, HW, C = args[sl:sl + 3]
                    acts = [a for a in args[:sl] if isinstance(a, torch.Tensor) and a.numel() >= N * HW * min(C, 3)]
                    byt = float(sum(a.numel() * a.element_size() for a in acts))
                elif mode == "sg2_fir":
                    N, C, H, W = args[sl:sl + 4]
                    oh, ow = args[sl + 12], args[sl + 13]
                    esz = 2 if args[-1] == 1 else 4
                    byt = float(N) * C * (H * W + (2 if args[2] is not None else 1) * oh * ow) * esz
                elif mode == "bias_act":
                    n = args[sl]
                    esz = {1: 2, 2: 8}.get(args[-1], 4) if name.endswith("typed") else 4
                    byt = float(n) * esz * (2 + sum(1 for a in args[2:5] if a is not None))     # x, y (+ xref / yref / dy)
                else:
                    N, C, H, W = args[sl:sl + 4]
                    oh, ow = args[sl + 16:sl + 18]
                    esz = {1: 2, 2: 8}.get(args[-2], 4) if name.endswith("typed") else 4
                    byt = float(N) * C * (H * W + oh * ow) * esz
                key = (name,) + tuple(a for a in args[sl:] if isinstance(a, int))
                rec = timer.hbm_keys.setdefault(key, [mode + " (" + name + ")", byt, 0, []])
                rec[2] += 1
                if (rec[2] - 1) % timer.period:
                    return raw(name, *args)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                raw(name, *args)
                e.record()
                rec[3].append((s, e))
                return
            if mode == "hconv":      # the same gather on fp16 operands (csrc/hconv.hip); zero-inserted sources run as four tap phases
                B, Hin, Win, Cin, Hout, Wout, Cout, R = args[sl:sl + 8]
                zins = args[sl + 10] if len(args) > sl + 10 else 0
                alg = exe = 2.0 * B * Hout * Wout * Cout * Cin * R * R / (4.0 if zins == 2 else 1.0)   # phased: R R / 4 taps per output
                outs = 1 if name == "icg_conv2d_g_fprop_f16" else sum(1 for a in (args[3 if name == "icg_modconv2d_f16" else 2], args[4 if name == "icg_modconv2d_f16" else 3]) if a is not None)
                byt = 2.0 * (B * (Hin * Win * Cin + outs * Hout * Wout * Cout) + Cout * Cin * R * R)
            elif mode == "hwgrad":   # fp16 weight gradient (csrc/hwgrad.hip): executed on 128 x 128 tiles of the [R R Cin][Cout] result
                B, Hin, Win, Cin, Hout, Wout, Cout, R = args[sl:sl + 8]
                alg = 2.0 * B * Hout * Wout * Cout * Cin * R * R
                exe = 2.0 * B * Hout * (32 * ((Wout + 31) // 32)) * (128 * ((Cout + 127) // 128)) * (128 * ((R * R * Cin + 127) // 128))
                byt = 2.0 * B * (Hin * Win * Cin + Hout * Wout * Cout) + 4.0 * Cout * Cin * R * R
            elif mode == "gconv":    # out[b,oy,ox,co] = sum src(...)*w: one multiply-add per (output, Cin, tap)
                B, Hin, Win, Cin, Hout, Wout, Cout, R = args[sl:sl + 8]
                alg = exe = 2.0 * B * Hout * Wout * Cout * Cin * R * R
                byt = 4.0 * (B * (Hin * Win * Cin + Hout * Wout * Cout) + Cout * Cin * R * R)
            elif mode == "tr2":      # stride-2 transposed 3x3 in phase form: 16 tap slots per 4 outputs, 9 of them non-zero
                B, Hin, Win, Cin, Hout, Wout, Cout = args[sl:sl + 7]
                alg = 2.0 * B * Hout * Wout * Cout * Cin * 9 / 4
                exe = 2.0 * B * Hout * Wout * Cout * Cin * 4
                byt = 4.0 * (B * (Hin * Win * Cin + Hout * Wout * Cout) + 16 * Cout * Cin)
            elif mode == "conv":
                B, H, W, Cin, Cout, R = args[sl:sl + 6]
                alg = exe = 2.0 * B * H * W * Cout * Cin * R * R
                byt = 4.0 * (B * H * W * (Cin + Cout) + Cout * Cin * R * R)
            elif mode in ("wino", "wino4"):   # Winograd: 16 GEMMs over 1/4 of the pixels (16/36 of the direct MACs) or
                B, H, W, Cin, Cout = args[sl:sl + 5]          # F(4x4,3x3): 36 GEMMs over 1/16 of the pixels (9/36)
                alg = 2.0 * B * H * W * Cout * Cin * 9
                exe = alg * (9.0 if mode == "wino4" else 16.0) / 36.0
                byt = 4.0 * (B * H * W * (Cin + Cout) + Cout * Cin * 9)
            elif mode == "from_v":                  # weight gradient from the forward pass's V planes (H, W: full resolution)
                B, H, W, Cin, Cout, planes = args[sl:sl + 6]
                alg = 2.0 * B * H * W * Cout * Cin * 9
                exe = alg * planes / 144.0
                byt = 4.0 * (B * H * W * (Cin + Cout) + Cout * Cin * 9)
            elif mode in ("rs_up", "rs_down"):      # resample-fused layer in the 25-plane F(4x4,3x3) domain: 25 GEMMs over 1/16 of
                B, Hs, Ws, Cin, Cout = args[sl:sl + 5]      # the full-resolution pixels = 25/144 of the reference graph's MACs
                alg = 2.0 * B * (4 * Hs * Ws) * Cout * Cin * 9
                exe = 2.0 * B * (Hs * Ws // 4) * Cout * Cin * 25
                lo, hi = (Cin, Cout) if mode == "rs_up" else (Cout, Cin)
                byt = 4.0 * (B * Hs * Ws * (lo + 4 * hi) + 9 * Cout * Cin)
            else:       # upsample- / avgpool-fused conv (2x2-phase or 4x4-stride-2 form): executed MACs are 16/36 of the
                        # reference op graph's (3x3 at the HIGH resolution); tensors: low-res one side, high-res the other
                B, Hs, Ws, Cin, Cout = args[sl:sl + 5]
                alg = 2.0 * B * (4 * Hs * Ws) * Cout * Cin * 9
                exe = 2.0 * B * Hs * Ws * Cout * Cin * 16
                lo, hi = (Cin, Cout) if "_up_" in name else (Cout, Cin)
                byt = 4.0 * (B * Hs * Ws * (lo + 4 * hi) + 16 * Cout * Cin)
            key = (name,) + tuple(a for a in args[sl:] if isinstance(a, int))
            rec = timer.keys.get(key)
            if rec is not None:
                rec[4] += 1
                if (rec[4] - 1) % timer.period:
                    return raw(name, *args)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            raw(name, *args)
            e.record()
            if rec is not None:
                rec[5].append((s, e))
                return
            query(last)
            pk = "icg_gemm_planes1_kernel" if last[3] == 4 else "icg_gemm_planes_kernel"     # single- / two-level chains

            def fwd_gemm():      # the forward / data-gradient plane GEMM that ran: second-generation kernel (pgemm.hip) or the first
                if last[0] == 2:
                    return "icg_pgemm_nn_kernel<%d, %d>" % (last[2], 1 if last[3] == 4 else 2)
                return "%s<0, 0, %d>" % (pk, last[2])

            def wg_gemm():       # ... and the weight-gradient one
                if last[0] == 3:
                    return "icg_pgemm_tn_kernel<%d, %d>" % (last[2], 1 if last[3] == 4 else 2)
                return "%s<1, 1, %d>" % (pk, last[2])
            if last[0] == 5 and mode in ("wino4", "rs_up", "rs_down"):
                # narrow layer on the fused F(4x4,3x3) kernel (csrc/fwino.hip): ONE kernel (+ a 1.3 MB weight re-layout), comparable
                # with its rocprofv3 row; executed FLOPs = the 36 / 25 plane GEMMs
                kname = "void icg_fwino_kernel<%d, %d, %d>(FwinoP)" % (last[1], last[2], last[3])
            elif mode == "hwgrad":
                kname = "icg_hwgrad_kernel(HwgradP)"
            elif mode == "hconv":
                cout = args[sl + 6]
                ep = 0 if name == "icg_conv2d_g_fprop_f16" else (1 if name == "icg_conv2d_g_fprop_f16_act" else int(args[4] is not None))
                mod = int(name == "icg_modconv2d_f16" and args[1] is not None)
                kname = "void icg_hconv_kernel<%d, %d, %d>(HconvP)" % (4 if cout % 128 == 0 else (3 if cout % 96 == 0 else 2), ep, mod)
            elif mode == "from_v":
                kname = "composite: weight gradient from the saved V planes (wino4_dy_kernel + %d batched split-K %s GEMMs + reduce + wino4_dw_kernel)" % (args[sl + 5], wg_gemm())
            elif mode in ("rs_up", "rs_down"):
                kind = (name[:-5] if name.endswith("_relu") else name).rsplit("_", 1)[1]
                kname = "composite: %s-fused conv %s in the 25-plane F(4x4,3x3) domain (transforms + 25 batched %s GEMMs)" % (
                    "upsample" if mode == "rs_up" else "avgpool", kind,
                    wg_gemm() if kind == "wgrad" else fwd_gemm())
            elif mode in ("wino", "wino4"):     # three kernels behind one entry point: not comparable with a single rocprof row
                kname = (("composite: wino4_input_kernel + wino4_dy_kernel + %s (36 batched split-K GEMMs) + reduce + wino4_dw_kernel" % wg_gemm()
                          if mode == "wino4" else
                          "composite: wino_input_kernel + wino_dy_kernel + %s (16 batched split-K GEMMs) + reduce + wino_dw_kernel" % wg_gemm())
                         if name.endswith("wgrad") else
                         ("composite: wino4_input_kernel + %s (36 batched GEMMs) + wino4_output_kernel" % fwd_gemm()
                          if mode == "wino4" else
                          "composite: wino_input_kernel + %s (16 batched GEMMs) + wino_output_kernel" % fwd_gemm()))
            elif last[0] == -4:      # thin-input 3x3 kernels (narrow_conv.hip): {-4, fprop/wgrad, Cout, Cin}
                if 9 * last[3] <= 32 and last[2] % 32 == 0 and last[2] <= 128:      # MFMA form (narrow_conv.hip: thin_mfma_ok)
                    kname = ("void thin_wgrad_mfma_kernel<%d, %d, 0>" if last[1] else "void thin_fprop_mfma_kernel<%d, %d>") % (last[3], last[2] // 32)
                else:
                    lp = 4
                    while lp < last[2] // 4:
                        lp *= 2
                    kname = "void thin_%s_kernel<%d, %d>" % ("wgrad" if last[1] else "fprop", last[3], lp)
            elif last[0] == -3:      # skinny linear kernels (narrow_conv.hip): {-3, fprop/wgrad, Cout, Cin}
                kname = "skinny_wgrad_kernel" if last[1] else "void smallm_nt_kernel<16, %d>" % (4 if last[3] % 4 == 0 else 1)
            elif last[0] == -2:      # direct narrow-output kernels (narrow_conv.hip): {-2, fprop/wgrad, Cout, Cin}
                if last[2] <= 3 and last[3] % 32 == 0 and last[3] <= 128:      # MFMA forms (narrow_conv.hip: narrow_*_mfma_ok)
                    kname = ("void thin_wgrad_mfma_kernel<%d, %d, 1>" if last[1] else "void narrow_fprop_mfma_kernel<%d, %d>") % (last[2], last[3] // 32)
                else:
                    lp = 1
                    while lp < last[3] // 4:
                        lp *= 2
                    kname = "void narrow_%s_kernel<%d, %d>" % ("wgrad" if last[1] else "fprop", last[2], lp)
            elif last[0] == -5:      # 1x1 convolution with <= 3 input channels (gemm_conv.hip: conv1x1_k3_kernel), write-bound
                kname = "conv1x1_k3_kernel"
            elif last[0] == 3:       # weight gradient of a prologue-free 1x1 convolution on the second-generation TN plane GEMM (one plane)
                kname = "void icg_pgemm_tn_kernel<%d, %d>(PgemmTnP)" % (last[2], 1 if last[3] == 4 else 2)
            elif last[0] == 4:       # second-generation implicit-GEMM convolution (pgemm.hip): {4, ReLU prologue, NT, levels}
                kname = "void icg_pconv_kernel<%d, %d, %d>(PconvP)" % (last[2], last[1], last[3])
            elif last[3] == 4:       # prologue-free 1x1 convolution on the persistent plain-GEMM body (single-level chains)
                kname = "void icg_gemm_planes1_kernel<%d, %d, %d>(GemmP)" % (last[0], last[1], last[2])
            else:
                kname = "void icg_gemm_kernel<%d, %d, %d, %d>(GemmP)" % tuple(last)
            timer.keys[key] = [kname, alg, exe, byt, 1, [(s, e)]]

        L.call = timed_call

    @staticmethod
    def planes(enable):
        """the Winograd-plane GEMMs run inside the composite entry points: the library brackets them with HIP events on the
        launch stream itself (icg_planes_timing)"""
        import ic_gan_amd._lib as L
        L.lib().icg_planes_timing(int(enable))      # 0: off; P: count every launch per shape, bracket every P-th

    @staticmethod
    def planes_summary():
        """{rocprofv3 kernel name: [algorithmic flops, seconds, launches, executed flops, operand bytes]} of those GEMMs."""
        import ctypes
        import ic_gan_amd._lib as L
        buf = (ctypes.c_double * (7 * 64))()
        rows = L.lib().icg_planes_timing_drain(ctypes.cast(buf, ctypes.c_void_p), 64)
        out = {}
        for r in range(max(rows, 0)):
            amode, tn, planes, n, ms, flops, byt = buf[7 * r: 7 * r + 7]
            mode = "1, 1" if int(amode) == 1 else "0, 0"
            # executed MACs -> MACs of the reference op graph: 36 of 144 (F(4x4,3x3)), 25 of 144 (resample-fused), 16 of 36 (F(2x2,3x3))
            alg = flops * {36: 144 / 36, 25: 144 / 25, 16: 36 / 16}.get(int(planes), 1.0)
            kern = "icg_gemm_planes1_kernel" if int(tn) >= 10 else "icg_gemm_planes_kernel"      # single- / two-level chains
            kname = "void %s<%s, %d>(GemmP)" % (kern, mode, int(tn) % 10)
            if int(amode) == 2:          # second-generation plane GEMM (pgemm.hip): <16-column tiles per wave, accumulation levels>
                kname = "void icg_pgemm_nn_kernel<%d, %d>(PgemmP)" % (int(tn) % 10, 1 if int(tn) >= 10 else 2)
            elif int(amode) == 3:
                kname = "void icg_pgemm_tn_kernel<%d, %d>(PgemmTnP)" % (int(tn) % 10, 1 if int(tn) >= 10 else 2)
            a = out.setdefault(kname, [0.0, 0.0, 0, 0.0, 0.0])
            a[0] += alg; a[1] += ms * 1e-3; a[2] += int(n); a[3] += flops; a[4] += byt
        return out

    def hbm_roofline(self):
        """roofline object of the HBM-bound op with the largest total time (None when the workload launched none)."""
        agg = {}
        for name, byt, secs in self.hbm_records:
            a = agg.setdefault(name, [0.0, 0.0, 0])
            a[0] += byt; a[1] += secs; a[2] += 1
        if not agg:
            return None
        name, (byt, secs, n) = max(agg.items(), key=lambda kv: kv[1][1])
        gbs = byt / secs / 1e9
        return {"bound": "hbm", "kernel": name, "achieved": round(gbs, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBPS, 4), "traffic": None, "launches": n, "avg_launch_ms": round(secs / n * 1e3, 4),
                "algorithmic_bytes_per_launch": round(byt / n),
                "all_hbm_ops": {k: {"GB/s": round(v[0] / v[1] / 1e9, 1), "frac": round(v[0] / v[1] / 1e9 / PEAK_HBM_GBPS, 4),
                                    "launches": v[2], "total_ms": round(v[1] * 1e3, 3)} for k, v in agg.items()}}

    def freeze_planes(self):
        """drain the library's plane-GEMM records now (icg_planes_timing(0) discards them): called at the end of the instrumented
        timed region, before the un-instrumented one switches the brackets off"""
        self._planes = self.planes_summary()

    def summary(self):
        agg = {}
        for kname, alg, exe, byt, secs in self.records:
            a = agg.setdefault(kname, [0.0, 0.0, 0, 0.0, 0.0])
            a[0] += alg
            a[1] += secs
            a[2] += 1
            a[3] += exe
            a[4] += byt
        return agg


def csrc_sha256():
    """content hash of the kernel sources (ic_gan_amd/csrc/*.hip, *.h): tools/pmc_hbm.py stores it in the traffic file, the bench
    line compares it with the tree it runs from (`traffic_stale`) -- the GPU box has no .git to ask"""
    import glob
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ic_gan_amd", "csrc")
    # the traffic file is taken on the cfg3 step: sources none of whose kernels that step launches (StyleGAN2's fp16 convolutions and
    # plugins, the kNN build) do not invalidate it
    other = {"hconv.hip", "hwgrad.hip", "stylegan_ops.hip", "stylegan_ops_typed.hip", "knn.hip", "sg2_fused.hip"}
    for path in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        if os.path.basename(path) in other:
            continue
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(kname):
    """HBM bytes per launch of `kname` from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE
    / --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 correction; tools/pmc_hbm.py writes the
    summary).  PMC collection cannot run inside the timed process, hence the file; None when it is absent.
    -> (bytes per launch of kname, source string, HBM GB per training step over ALL kernels, commit the file was taken at)"""
    import glob
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_hbm_traffic.json")))
    try:
        with open(paths[-1]) as f:
            table = json.load(f)
    except (OSError, ValueError, IndexError):
        return None, None, None, None
    measured_traffic.stale = table.get("csrc_sha256") != csrc_sha256()      # kernels changed since the PMC passes were taken
    per_launch = table.get("kernels", {}).get(kname, {}).get("hbm_bytes_per_launch")
    steps = table.get("steps_profiled", 3)          # r01 file: `--steps 2 --warmup 1` = 3 steps in the profiled process
    total = sum(k["launches"] * k["hbm_bytes_per_launch"] for k in table.get("kernels", {}).values())
    src = "%s [%s, measured at commit %s]" % (table.get("source"), os.path.basename(paths[-1]), table.get("commit", "23abb1c (round 1)"))
    return per_launch, src, round(total / steps / 1e9, 1), table.get("commit", "23abb1c")


def assemble_roofline(timer, steps, elapsed, with_step_traffic=True):
    """roofline object (see the module docstring) from the HIP-event records of the timed region; None without records."""
    agg = timer.summary()
    # inner GEMMs of the composites (their time is part of the composite rows too)
    agg.update(timer._planes if getattr(timer, "_planes", None) is not None else timer.planes_summary())
    cands = [(k, v) for k, v in agg.items() if not k.startswith("composite:")]
    if not cands:
        return None
    variant, (flops, secs, n, exe, byt) = max(cands, key=lambda kv: kv[1][1])
    traffic, traffic_src, step_hbm_gb, _ = measured_traffic(variant) if with_step_traffic else (None, None, None, None)
    alg_tf, exe_tf = flops / secs / 1e12, exe / secs / 1e12
    # Roofline of the dominant kernel = what the MFMA pipe EXECUTED / its HIP-event time / the fp32 MFMA peak.  The same
    # time priced on the reference op graph's FLOPs (3x3 conv counted directly, on the upsampled tensor where the
    # reference upsamples first) is `algorithmic_tflops`; their ratio is the algebraic saving (Winograd F(4x4,3x3): 144/36,
    # resample-fused 25-plane form: 144/25, 2x2-phase / 4x4-stride-2 forms: 36/16).
    step_exe = sum(r[2] for r in timer.records) / steps              # entry-point level: no double counting
    step_ms = elapsed / steps * 1e3
    t_mfma = sum(r[2] / ((PEAK_F16_MFMA_TFLOPS if ("icg_hconv_kernel" in r[0] or "icg_hwgrad_kernel" in r[0]) else PEAK_F32_MFMA_TFLOPS) * 1e12)
                 for r in timer.records) / steps * 1e3                # every launch against the MFMA roof of its own operand type
    t_hbm = (step_hbm_gb / PEAK_HBM_GBPS * 1e3) if step_hbm_gb else None
    peak = PEAK_F16_MFMA_TFLOPS if ("icg_hconv_kernel" in variant or "icg_hwgrad_kernel" in variant) else PEAK_F32_MFMA_TFLOPS       # the dominant kernel's own MFMA roof
    return {"bound": "mfma", "kernel": variant, "achieved": round(exe_tf, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(exe_tf / peak, 4), "traffic": traffic,
            "traffic_source": traffic_src, "traffic_stale": (getattr(measured_traffic, "stale", None) if traffic_src else None),
            "algorithmic_bytes_per_launch": round(byt / n),
            "algorithmic_tflops": round(alg_tf, 2), "algorithmic_speedup": round(alg_tf / exe_tf, 3),
            "launches": n, "avg_launch_ms": round(secs / n * 1e3, 4),
            "executed_gflop_per_launch_avg": round(exe / n / 1e9, 3),
            # whole step against its own binding roof: executed conv/GEMM work at the MFMA peak vs measured HBM traffic
            # (all kernels, PMC passes of the same command) at 8 TB/s
            "step": {"executed_tflop": round(step_exe / 1e12, 3), "hbm_gb_measured": step_hbm_gb,
                     "t_mfma_ms": round(t_mfma, 2), "t_hbm_ms": (round(t_hbm, 1) if t_hbm else None),
                     "ms_per_step": round(step_ms, 2),
                     "frac": round(max(t_mfma, t_hbm or 0.0) / step_ms, 4),
                     "frac_if_serial": round((t_mfma + (t_hbm or 0.0)) / step_ms, 4)},
            "all_conv_kernels": {k: {"algorithmic_tflops": round(v[0] / v[1] / 1e12, 2),
                                     "executed_tflops": round(v[3] / v[1] / 1e12, 2),
                                     "ms_per_step": round(v[1] / steps * 1e3, 2),
                                     "avg_launch_ms": round(v[1] / v[2] * 1e3, 4),
                                     "launches_per_step": v[2] // steps} for k, v in agg.items()}}


def build_models(cfg, device, init_mode="ortho"):
    import importlib
    M = importlib.import_module("ic_gan_amd." + cfg.get("model", "BigGAN"))      # the reference's plugin seam (trainer.py:122)
    from ic_gan_amd import utils
    from ic_gan_amd.optim import FusedAdam
    G = M.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False, "no_optim": True}).to(device)
    D = M.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False}).to(device)
    try:                       # reference init (orthogonal) on the device: QR of up to 1536 x 13824 matrices
        if init_mode != "ortho":
            raise RuntimeError("--init %s requested" % init_mode)
        G.init_weights(); D.init_weights()
        init = "ortho"
    except Exception as exc:   # noqa: BLE001  (rocSOLVER unavailable -> documented fall back to N(0, 0.02))
        print(f"[bench] orthogonal init unavailable ({exc}); using N02", file=sys.stderr)
        G.init, D.init = "N02", "N02"
        G.init_weights(); D.init_weights()
        init = "N02"
    G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True}).to(device)
    ema = utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    opt_d = FusedAdam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), eps=cfg["adam_eps"])
    opt_g = FusedAdam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), eps=cfg["adam_eps"])
    return M, G, D, G_ema, ema, opt_g, opt_d, init


def synthetic_batch(cfg, batch, seed):
    """Real-side batch of the step: x uint8 uniform -> ((u8/255)-0.5)*2 (datasets_common.py:505-507),
    labels uniform, unit-L2 2048-d features (datasets_common.py:662,678)."""
    rs = np.random.RandomState(seed)
    r = cfg["resolution"]
    x = torch.from_numpy(rs.randint(0, 256, size=(batch, 3, r, r)).astype(np.float32)).div_(255.0).sub_(0.5).mul_(2.0)
    y = torch.from_numpy(rs.randint(0, cfg["n_classes"], size=(batch,)).astype(np.int64))
    f = rs.standard_normal((batch, 2048))
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    return x, (y if cfg["class_cond"] else None), (torch.from_numpy(f).float() if cfg["instance_cond"] else None)


def conditioning_sampler(cfg, dim_z, batch, device, seed, n_table=10000, k_nn=50):
    """`sample_conditionings()` of train_fns.py:70,135 = functools.partial(sample_conditioning_values, ...) exactly as
    trainer.py:373-385 builds it, over a synthetic conditioning table (N=10 000 unit-norm 2048-d rows,
    labels, exact L2 k=50 neighbourhoods) held resident in HBM by ic_gan_amd.data_utils.ConditioningStore."""
    import functools
    from ic_gan_amd import data_utils
    rs = np.random.RandomState(seed)
    feats = rs.standard_normal((n_table, 2048)).astype(np.float32)
    labels = rs.randint(0, cfg["n_classes"], size=n_table).astype(np.int64)
    store = data_utils.ConditioningStore(labels=labels, feats=feats, load_in_mem_feats=True, k_nn=k_nn,
                                         load_labels=cfg["class_cond"], load_features=cfg["instance_cond"],
                                         device=device)
    np.random.seed(seed)                       # the sampler draws from numpy's global state, like the reference
    z_, y_ = data_utils.prepare_z_y(batch, dim_z, cfg["n_classes"], device=device)
    return functools.partial(data_utils.sample_conditioning_values, z_=z_, y_=y_, dataset=store, batch_size=batch,
                             weights_sampling=None, ddp=True, constant_conditioning=False,
                             class_cond=cfg["class_cond"], instance_cond=cfg["instance_cond"],
                             nn_sampling_strategy="instance_balance")


def bench_stylegan2(args, device, rank, world, local_rank, use_ddp):
    """Secondary workload (BASELINE.json configs[3]): IC-GAN StyleGAN2 256x256 `cfg=auto` on 1 GPU
    (stylegan2_ada_pytorch/train.py:291-372 for res 256, 1 GPU: batch 16, fmaps 0.5, 2 mapping layers, lr 0.0025,
    gamma 0.8192, mbstd 4, ema 5 kimg), instance-conditioned (h_dim 2048), fp32 (num_fp16_res = 0).  A step is one
    training iteration; the timed region must span whole lazy-regularisation cycles (16 iterations) to be an average."""
    import copy
    from ic_gan_amd.stylegan2 import networks as N
    from ic_gan_amd.stylegan2.training_step import TrainingStep
    res, b = 256, (args.batch or 16)
    torch.manual_seed(rank)
    np.random.seed(rank)
    # --fp16: the reference's own cfg=auto block precision (train.py:297-310: num_fp16_res=4, conv_clamp=256): the 32x32 ... 256x256
    # blocks store fp16 (bias_act / upfirdn2d / modulation glue move half the bytes), convolutions compute in fp32 on MFMA
    common = dict(channel_base=16384, channel_max=512, num_fp16_res=(4 if args.fp16 else 0),
                  conv_clamp=(256 if args.fp16 else None))
    G = N.Generator(z_dim=512, c_dim=0, h_dim=2048, w_dim=512, img_resolution=res, img_channels=3,
                    mapping_kwargs=dict(num_layers=2), synthesis_kwargs=common).train().requires_grad_(False).to(device)
    D = N.Discriminator(c_dim=0, h_dim=2048, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=2),
                        epilogue_kwargs=dict(mbstd_group_size=4), **common).train().requires_grad_(False).to(device)
    G_ema = copy.deepcopy(G).eval()
    mods = None
    if use_ddp:      # training_loop.py:293-310: mapping, synthesis and D are wrapped separately, no buffer broadcast
        from torch.nn.parallel import DistributedDataParallel as DDP
        mods = {}
        for name, m in (("G_mapping", G.mapping), ("G_synthesis", G.synthesis), ("D", D)):
            m.requires_grad_(True)
            mods[name] = DDP(m, device_ids=[local_rank], broadcast_buffers=False)
            m.requires_grad_(False)
    adam = dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)
    step = TrainingStep(G, D, G_ema, device, batch_size=b * world, batch_gpu=b, num_gpus=world,
                        loss_kwargs=dict(r1_gamma=0.0002 * res ** 2 / (b * world)), G_opt_kwargs=adam, D_opt_kwargs=adam,
                        ema_kimg=b * world * 10 / 32, ema_rampup=0.05, ddp_modules=mods)
    rs = np.random.RandomState(7 + rank)
    img = torch.from_numpy((rs.randint(0, 256, size=(b, 3, res, res)) / 127.5 - 1).astype(np.float32)).to(device)

    def unit(n):
        h = rs.standard_normal((n, 2048)).astype(np.float32)
        return torch.from_numpy(h / np.linalg.norm(h, axis=1, keepdims=True)).to(device)

    real_h, real_c = unit(b), torch.empty([b, 0], device=device)
    n_ph = len(step.phases)
    gen_h, gen_c = unit(n_ph * b), torch.empty([n_ph * b, 0], device=device)

    def one_step():
        return step(img, real_c, real_h, torch.randn([n_ph * b, 512], device=device), gen_c, gen_h)

    timer = KernelTimer(args.timer_period)
    timer.install()
    for _ in range(args.warmup):
        one_step()
    step.batch_idx = 0                        # the timed region starts at a cycle boundary
    if args.steps % 16:
        print(f"[bench] cfg4: --steps {args.steps} is not a multiple of 16 (the lazy-regularisation cycle: Greg every 4, Dreg "
              f"every 16 iterations): the mean is not a whole-cycle average", file=sys.stderr)
    if use_ddp:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    timer.planes(timer.period)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if use_ddp:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if use_ddp:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        dist.destroy_process_group()
    if rank == 0:
        roof = assemble_roofline(timer, args.steps, elapsed, with_step_traffic=False)
        print(json.dumps({
            "metric": "images/sec training iteration, IC-GAN StyleGAN2 256^2 (cfg4, secondary workload)",
            "value": round(b * world * args.steps / elapsed, 3), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": ("f16 (fp16 blocks: fp16 operands, fp32 accumulation, as the reference) / f32" if args.fp16 else "f32"), "data": "synthetic",
            "config": {"workload": "cfg4: IC-GAN StyleGAN2 256x256 cfg=auto, h_dim 2048, " + ("fp16 blocks (num_fp16_res=4, conv_clamp=256), fp32 MFMA arithmetic" if args.fp16 else "fp32 (num_fp16_res=0)") + "; Gmain+Dmain every iteration, "
                                   "Greg every 4, Dreg every 16 (steps should be a multiple of 16)",
                       "batch_per_gpu": b, "global_batch": b * world, "parallelism": f"dp{world}"},
            # dominant MFMA kernel of the iteration, and the dominant HBM-bound plugin (bias_act / upfirdn2d: algorithmic bytes
            # of SURVEY 8(d) -- 8 B per element, 4 (in + out) B -- over HIP-event time, against the 8 TB/s HBM peak)
            "roofline": roof, "roofline_hbm": timer.hbm_roofline(),
            "cpu_baseline": (run_cpu_baseline_child("cfg4") if world == 1 and not args.no_cpu_baseline else None)}),
            flush=True)


def bench_sampling(args, device, rank, world):
    """Secondary workload: the sampling engine (inference/utils.py:176-269) — G_ema in eval mode (stored BN statistics, no
    SN update), class + instance conditioning drawn by the HBM-resident sampler, cfg3 generator (256x256, ch 96).  A step is
    one `inference.sample` call of `--batch` images (default 64); data-parallel replicas, no collective."""
    from ic_gan_amd import inference, utils
    b = args.batch or 64
    cfg = dict(BASE_CFG)
    cfg.update(WORKLOADS["cfg3"][0])
    utils.seed_rng(rank)
    import ic_gan_amd.BigGAN as M
    G = M.Generator(**{**cfg, "skip_init": True, "no_optim": True, "G_init": "N02"}).to(device)
    G.init = "N02"
    G.init_weights()
    G.eval()
    from ic_gan_amd import layers as _layers
    _layers.enable_sn_eval_cache(G)          # frozen sampling weights: W/sigma computed once (inference.load_model_inference does the same)
    sampler = conditioning_sampler(cfg, G.dim_z, b, device, seed=1000 + rank)
    if args.graph:
        G = inference.GraphedGenerator(G, b, class_cond=True, instance_cond=True, device=device, static_weights=True)
    timer = KernelTimer(args.timer_period)
    if not args.graph:                 # (a HIP-graph replay has no per-launch events: roofline null there)
        timer.install()
    for _ in range(args.warmup):
        inference.sample(G, sampler, cfg, class_cond=True, instance_cond=True, device=device)
    torch.cuda.synchronize()
    timer.enabled = not args.graph
    if not args.graph:
        timer.planes(timer.period)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img, _, _ = inference.sample(G, sampler, cfg, class_cond=True, instance_cond=True, device=device)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec sampling, IC-GAN BigGAN 256^2 generator (secondary workload)",
            "value": round(b * world * args.steps / elapsed, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "sample: cfg3 generator in eval mode, class + instance conditioning from the resident table",
                       "batch_per_gpu": b, "hip_graph": bool(args.graph), "finite": bool(torch.isfinite(img).all())},
            "roofline": (assemble_roofline(timer, args.steps, elapsed, with_step_traffic=False) if not args.graph else None),
            "cpu_baseline": (run_cpu_baseline_child("sample") if world == 1 and not args.no_cpu_baseline else None)}),
            flush=True)


def cpu_baseline(cfg, name, budget_s=25.0):
    """CPU oracle (restatement of the reference step, pinned to reference goldens) on this box's host cores."""
    from oracle import biggan_oracle as O, synth
    import ic_gan_amd.BigGAN as M
    # Thread count: with all 256 hardware threads of the GPU box PyTorch's CPU convolutions oversubscribe (measured
    # 375 s/step); 8 threads in the build container gave 7.6 s/step.  We cap at 32 and report that as `cores`.
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    b = 2 if cfg["resolution"] >= 128 else 8
    G = M.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = M.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    gsd = synth.synth_state(synth.spec_of(G.state_dict()), 11)
    dsd = synth.synth_state(synth.spec_of(D.state_dict()), 22)
    dim_z = G.dim_z
    del G, D
    ema_sd = {k: v.clone() for k, v in gsd.items()}
    og = O.AdamState(O.param_names(gsd), cfg["G_lr"], 0.0, 0.999, 1e-6)
    od = O.AdamState(O.param_names(dsd), cfg["D_lr"], 0.0, 0.999, 1e-6)
    samp = synth.CondSampler(cfg, dim_z, b, 3)
    x, y, f = synth.synth_batch(cfg, b, seed=4)
    times = []
    t_end = time.time() + budget_s
    while len(times) < 3 and (not times or time.time() + times[-1] < t_end):
        t0 = time.time()
        O.train_step(gsd, dsd, ema_sd, cfg, og, od, x, y, f, samp, len(times) + 1, b)
        times.append(time.time() - t0)
    t = min(times)
    return {"value": round(b / t, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "reference_unavailable_on_gpu_box": True,
            "sample": f"{name} shape, batch {b}" + (" (full batch)" if name == "cfg1" else " (reduced batch, images/sec scale ~linearly)")
                      + f", best of {len(times)} step(s) of oracle.biggan_oracle.train_step (PyTorch CPU fp32), {t:.2f} s/step"}


class _Patch:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


def cpu_baseline_emulated(cfg, name, budget_s=25.0):
    """CPU baseline for workloads without a stand-alone oracle module (cfg5: the instance-conditioned BigGAN-deep has no
    reference model; cfg4: StyleGAN2): the product's host code with every C-ABI call served by oracle/kernel_ref.py -- the
    plain-PyTorch CPU restatement of each kernel that the parity tests check the HIP kernels against (`kind` "port")."""
    import importlib
    from oracle import kernel_ref, synth
    import ic_gan_amd.ops as ops
    kernel_ref.install(_Patch())
    ops.disable_winograd()                       # the CPU restatement of a direct convolution is one F.conv2d
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    if name == "cfg4":
        return _cpu_baseline_stylegan2(budget_s)
    from ic_gan_amd import train_fns, utils
    from ic_gan_amd.optim import FusedAdam
    M = importlib.import_module("ic_gan_amd." + cfg.get("model", "BigGAN"))
    b = 2
    G = M.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    D = M.Discriminator(**{**cfg, "skip_init": True})
    G.load_state_dict(synth.synth_state(synth.spec_of(G.state_dict()), 11))
    D.load_state_dict(synth.synth_state(synth.spec_of(D.state_dict()), 22))
    G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    ema = utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    opt_d = FusedAdam(D.parameters(), lr=cfg["D_lr"], betas=(0.0, 0.999), eps=1e-6)
    opt_g = FusedAdam(G.parameters(), lr=cfg["G_lr"], betas=(0.0, 0.999), eps=1e-6)
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    train = train_fns.GAN_training_function(G, D, GD, ema, {"itr": 1}, cfg, synth.CondSampler(cfg, G.dim_z, b, 3),
                                            embedded_optimizers=False, device="cpu", batch_size=b)
    x, y, f = synth.synth_batch(cfg, b, seed=4)
    G.train(); D.train()
    times, t_end = [], time.time() + budget_s
    while len(times) < 3 and (not times or time.time() + times[-1] < t_end):
        t0 = time.time()
        train(x, y, f)
        times.append(time.time() - t0)
    t = min(times)
    return {"value": round(b / t, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "reference_unavailable_on_gpu_box": True,
            "sample": f"{name} shape, batch {b} (reduced batch), best of {len(times)} step(s) of the product host code over "
                      f"oracle/kernel_ref.py (PyTorch CPU fp32 restatement of every kernel), {t:.2f} s/step"}


def _cpu_baseline_stylegan2(budget_s):
    import copy
    from ic_gan_amd.stylegan2 import networks as N
    from ic_gan_amd.stylegan2.training_step import TrainingStep
    res, b = 256, 2
    common = dict(channel_base=16384, channel_max=512, num_fp16_res=0, conv_clamp=None)
    G = N.Generator(z_dim=512, c_dim=0, h_dim=2048, w_dim=512, img_resolution=res, img_channels=3,
                    mapping_kwargs=dict(num_layers=2), synthesis_kwargs=common).train().requires_grad_(False)
    D = N.Discriminator(c_dim=0, h_dim=2048, img_resolution=res, img_channels=3, mapping_kwargs=dict(num_layers=2),
                        epilogue_kwargs=dict(mbstd_group_size=2), **common).train().requires_grad_(False)
    adam = dict(lr=0.0025, betas=[0, 0.99], eps=1e-8)
    step = TrainingStep(G, D, copy.deepcopy(G).eval(), "cpu", batch_size=b, batch_gpu=b, num_gpus=1,
                        loss_kwargs=dict(r1_gamma=0.0002 * res ** 2 / 16), G_opt_kwargs=adam, D_opt_kwargs=adam,
                        ema_kimg=5.0, ema_rampup=0.05)
    rs = np.random.RandomState(7)
    img = torch.from_numpy((rs.randint(0, 256, size=(b, 3, res, res)) / 127.5 - 1).astype(np.float32))
    unit = lambda n: torch.from_numpy((lambda h: h / np.linalg.norm(h, axis=1, keepdims=True))(rs.standard_normal((n, 2048)).astype(np.float32)))
    n_ph = len(step.phases)
    times, t_end = [], time.time() + budget_s
    step.batch_idx = 1                           # iterations 1, 2, ...: Gmain + Dmain only (the lazy regularisers run every 4 / 16)
    while len(times) < 3 and (not times or time.time() + times[-1] < t_end):
        t0 = time.time()
        step(img, torch.empty([b, 0]), unit(b), torch.randn([n_ph * b, 512]), torch.empty([n_ph * b, 0]), unit(n_ph * b))
        times.append(time.time() - t0)
    t = min(times)
    return {"value": round(b / t, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "reference_unavailable_on_gpu_box": True,
            "sample": f"cfg4 shape, batch {b} (reduced batch), best of {len(times)} Gmain + Dmain iteration(s) of the product host "
                      f"code over oracle/kernel_ref.py (PyTorch CPU fp32 restatement of every kernel), {t:.2f} s/iteration"}


def cpu_baseline_sampling(cfg, b=2):
    """sampling engine: the oracle's eval-mode generator forward on the host cores"""
    from oracle import biggan_oracle as O, synth
    import ic_gan_amd.BigGAN as M
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    G = M.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    gsd = synth.synth_state(synth.spec_of(G.state_dict()), 11)
    z, lab, fg = synth.CondSampler(cfg, G.dim_z, b, 3)()
    times = []
    with torch.no_grad():
        for _ in range(3):
            t0 = time.time()
            O.generator_forward(gsd, cfg, z, lab, fg, False)
            times.append(time.time() - t0)
    t = min(times)
    return {"value": round(b / t, 4), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "reference_unavailable_on_gpu_box": True,
            "sample": f"cfg3 generator, eval mode, batch {b}, best of 3 calls of oracle.biggan_oracle.generator_forward, {t:.2f} s/call"}


def run_cpu_baseline_child(workload):
    """the CPU leg runs in a separate process with a hard wall-clock bound: the default bench run must finish in minutes"""
    import subprocess
    try:
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload],
                             capture_output=True, text=True, timeout=240,
                             env={**os.environ, "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
        line = [l for l in res.stdout.splitlines() if l.startswith("CPU_BASELINE ")][-1]
        return json.loads(line[len("CPU_BASELINE "):])
    except Exception as exc:   # noqa: BLE001
        return {"value": None, "unit": "images/sec", "cores": min(32, os.cpu_count() or 1), "kind": "port",
                "sample": f"not completed within 240 s ({type(exc).__name__})"}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_command(n, argv=None, port=None):
    """the command line `python bench.py --gpus N ...` re-executes itself as (one rank per GPU of this node, rendezvous on 127.0.0.1)"""
    argv = list(sys.argv[1:] if argv is None else argv)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + argv


def self_launch(n):
    import subprocess
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    rc = subprocess.call(self_launch_command(n), env=env)
    if rc != 0:
        raise SystemExit(rc)


def rank_host_resources(local_rank, local_world):
    """One rank per GPU shares the node's cores with its siblings: give each an equal, disjoint slice of the cores this process may
    run on (affinity) and as many intra-op threads -- under torchrun every rank otherwise inherits OMP_NUM_THREADS=1 (its default)
    or, unset, all cores (N ranks x all cores oversubscribe the host-side conditioning sampler).  `self_launch` sets the thread count
    in the environment for the ranks it starts; this covers the driver's torchrun form as well.  ICG_NO_AFFINITY=1: leave both alone."""
    if os.environ.get("ICG_NO_AFFINITY") == "1" or local_world <= 1:
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    per = max(1, len(cores) // local_world)
    mine = cores[local_rank * per:(local_rank + 1) * per] or cores
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        mine = cores
    torch.set_num_threads(max(1, min(len(mine), 16)))
    return {"cores": len(mine), "first_core": mine[0], "intra_op_threads": torch.get_num_threads()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS) + ["cfg4", "sample"])
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--init", default="ortho", choices=["ortho", "N02"],
                    help="weight init; N02 skips the rocSOLVER QR (which crashes under rocprofv3 --pmc); timings are init-independent")
    ap.add_argument("--graph", action="store_true", help="sample workload: replay the generator forward from a HIP graph")
    ap.add_argument("--no-winograd", action="store_true",
                    help="implicit-GEMM / phase / 4x4-stride-2 kernels only (ops.disable_winograd): the strict-parity route")
    ap.add_argument("--no-fuse-relu-backward", action="store_true",
                    help="ablation: ReLU backward of D's layers as a separate pass instead of the data-gradient epilogue")
    ap.add_argument("--no-fused-attention", action="store_true",
                    help="ablation: attention scores as GEMM + stand-alone softmax instead of the fused kernel (ops.FUSED_ATTENTION_SCORES)")
    ap.add_argument("--wgrad-stream", action="store_true",
                    help="opt-in: weight gradients on a side HIP stream, concurrently with the data gradient (ops.WGRAD_SIDE_STREAM)")
    ap.add_argument("--no-prefetch-next-step", action="store_true",
                    help="issue every operation where the reference does instead of queueing the next step's first conditioning draw and "
                         "generator forward behind the current step's EMA, before the loss read-back (train_fns.PREFETCH_NEXT_STEP)")
    ap.add_argument("--reference-comm", action="store_true",
                    help="N > 1: keep the reference's communication pattern (an all-reduce per accumulation round, D's reducer armed in the "
                         "G phase, a buffer broadcast per forward) instead of train_fns.COMM_SAVINGS")
    ap.add_argument("--sync-bn", action="store_true", help="cross-replica BN statistics over RCCL (cfg3 variant)")
    ap.add_argument("--fp16", action="store_true", help="cfg4: the reference's cfg=auto precision (num_fp16_res=4, conv_clamp=256)")
    ap.add_argument("--accumulate", type=int, default=1,
                    help="gradient accumulation rounds per phase (num_D_accumulations = num_G_accumulations); `--accumulate 4 --batch 16` is "
                         "the shipped cfg3 schedule (cc_icgan_res256.json:22-24,40): 64 images per GPU and step as 4 x 16")
    ap.add_argument("--no-kernel-timer", action="store_true",
                    help="no HIP-event instrumentation at all (no roofline object): the un-instrumented step time as the value")
    ap.add_argument("--timer-period", type=int, default=4,
                    help="HIP-event brackets on every P-th launch of each (entry point, shape); all launches are counted (1: bracket all)")
    ap.add_argument("--no-uninstrumented-leg", action="store_true",
                    help="skip the second, un-instrumented timed region the N = 1 run reports beside the instrumented one")
    args = ap.parse_args()

    if args.cpu_baseline_only:          # child process of the N=1 run: bounded CPU sample, prints one JSON object
        cfg = dict(BASE_CFG)
        cfg.update(WORKLOADS.get(args.workload, WORKLOADS["cfg3"])[0])
        if args.workload in ("cfg4", "cfg5"):
            res = cpu_baseline_emulated(cfg, args.workload)
        elif args.workload == "sample":
            res = cpu_baseline_sampling(cfg)
        else:
            res = cpu_baseline(cfg, args.workload)
            if args.workload != "cfg1":      # cfg1 (the reference's CPU-runnable configuration) at its full batch of 8
                c1 = dict(BASE_CFG)
                c1.update(WORKLOADS["cfg1"][0])
                res["cfg1_full_batch"] = cpu_baseline(c1, "cfg1", budget_s=8.0)
        print("CPU_BASELINE " + json.dumps(res), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks here, as the reference's trainer does with
        # mp.spawn(train, nprocs=world_size) (BigGAN_PyTorch/trainer.py:70-75).  Re-executing this script under
        # torch.distributed.run gives the ranks exactly the environment of the driver's torchrun form (which keeps working:
        # WORLD_SIZE is set there and this branch is not taken).  Rank 0's JSON line goes to this process' stdout.
        return self_launch(args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_ddp = world > 1 or os.environ.get("ICG_FORCE_DDP") == "1"   # (the override exercises the RCCL/DDP wiring on 1 GPU)
    host = rank_host_resources(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else None
    if use_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL on ROCm.  (ICG_BENCH_SHARED_GPU=1, tests only: all ranks on GPU 0 over gloo -- RCCL refuses two ranks per device --
        # so that the N > 1 code path of this script can be exercised on a 1-GPU box; never a measurement)
        shared = os.environ.get("ICG_BENCH_SHARED_GPU") == "1"
        dist.init_process_group(backend="gloo" if shared else "nccl")
        if shared:
            local_rank = 0
    assert world == max(args.gpus, 1), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if args.no_winograd:
        import ic_gan_amd.ops as _ops
        _ops.disable_winograd()
    if args.no_fuse_relu_backward:
        import ic_gan_amd.ops as _ops
        _ops.FUSE_RELU_BACKWARD = False
    if args.wgrad_stream:
        import ic_gan_amd.ops as _ops
        _ops.WGRAD_SIDE_STREAM = True
    if args.no_fused_attention:
        import ic_gan_amd.ops as _ops
        _ops.FUSED_ATTENTION_SCORES = False

    if args.workload == "sample":
        return bench_sampling(args, device, rank, world)
    if args.workload == "cfg4":
        return bench_stylegan2(args, device, rank, world, local_rank, use_ddp)
    over, batch = WORKLOADS[args.workload]
    batch = args.batch or batch
    cfg = dict(BASE_CFG)
    cfg.update(over)
    if args.sync_bn:
        cfg["sync_bn"] = True
    acc = max(args.accumulate, 1)
    cfg["num_D_accumulations"] = cfg["num_G_accumulations"] = acc
    from ic_gan_amd import train_fns, utils
    # the library's default is the reference's communication pattern; the bench measures the trimmed one (same losses and parameters,
    # tests/test_ddp_gloo_cpu.py) unless --reference-comm, and reports which in `comm.comm_savings`
    train_fns.COMM_SAVINGS = not args.reference_comm
    train_fns.PREFETCH_NEXT_STEP = not args.no_prefetch_next_step
    utils.seed_rng(0 + rank)
    M, G, D, G_ema, ema, opt_g, opt_d, init = build_models(cfg, device, args.init)
    dim_z = G.dim_z
    if use_ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP
        # reference wiring: trainer.py:196-210 (separate wrappers, find_unused_parameters, buffer broadcast on)
        G = DDP(G, device_ids=[local_rank], output_device=local_rank, find_unused_parameters=True)
        D = DDP(D, device_ids=[local_rank], output_device=local_rank, find_unused_parameters=True)
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 0}
    sampler = conditioning_sampler(cfg, dim_z, batch, device, seed=1000 + rank)
    train = train_fns.GAN_training_function(G, D, GD, ema, state, cfg, sampler, embedded_optimizers=False,
                                            device=device, batch_size=batch)
    x, y, f = synthetic_batch(cfg, batch * acc, seed=7 + rank)      # train() consumes batch x num_D_accumulations real images
    x, y, f = x.to(device), (y.to(device) if y is not None else None), (f.to(device) if f is not None else None)

    timer = KernelTimer(args.timer_period)
    if not args.no_kernel_timer:
        timer.install()
    comm = None
    if use_ddp:
        # what the step all-reduces: a counting communication hook on both wrappers (bytes, buckets); the hook otherwise does
        # what the default one does (average).  `exposed_ms_per_step` below = step time with - without the collectives.
        class _Comm:
            def __init__(self):
                self.bytes = self.buckets = 0

        def _hook(state, bucket):
            buf = bucket.buffer()
            state.bytes += buf.numel() * buf.element_size()
            state.buckets += 1
            buf.div_(dist.get_world_size())
            return dist.all_reduce(buf, async_op=True).get_future().then(lambda fut: fut.value()[0])

        comm = {"G": _Comm(), "D": _Comm()}
        G.register_comm_hook(comm["G"], _hook)
        D.register_comm_hook(comm["D"], _hook)

    def one_step():
        state["itr"] += 1
        G.train(); D.train(); G_ema.train()
        return train(x, y, f)

    for _ in range(args.warmup):
        one_step()
    if use_ddp:
        dist.barrier()
    torch.cuda.synchronize()
    if comm:
        comm["G"].bytes = comm["G"].buckets = comm["D"].bytes = comm["D"].buckets = 0
    timer.enabled = not args.no_kernel_timer
    timer.planes(0 if args.no_kernel_timer else timer.period)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = one_step()
    torch.cuda.synchronize()
    if use_ddp:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if use_ddp:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    def timed_region(n):
        """a further timed region of n steps, bracketed like the one above (barrier + synchronize, maximum over the ranks)"""
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            one_step()
        torch.cuda.synchronize()
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        if use_ddp:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # what the K timed steps all-reduced (the legs below run the hooks again)
    comm_counts = {k: (v.bytes, v.buckets) for k, v in comm.items()} if comm else None
    uninstr = None
    if not args.no_kernel_timer and not args.no_uninstrumented_leg:
        # the figure above is measured with the HIP-event brackets of KernelTimer / icg_planes_timing active; the same K steps again
        # without any instrumentation, reported beside it
        timer.freeze_planes()
        timer.planes(False)
        uninstr = timed_region(args.steps)
    comm_report = None
    if comm:
        per_step = {k: {"allreduce_bytes_per_step": v[0] // args.steps, "buckets_per_step": v[1] // args.steps}
                    for k, v in comm_counts.items()}
        # exposed (non-overlapped) communication: the same steps with every collective of the two wrappers suppressed (no_sync:
        # each rank then trains on its own gradients -- a timing leg only, run after the measurement)
        import contextlib
        with G.no_sync(), D.no_sync():
            t_nosync = timed_region(args.steps)
        base = uninstr if uninstr is not None else elapsed
        comm_report = {**per_step, "ms_per_step_without_collectives": round(t_nosync / args.steps * 1e3, 3),
                       "exposed_ms_per_step": round((base - t_nosync) / args.steps * 1e3, 3),
                       "comm_savings": bool(train_fns.COMM_SAVINGS),
                       "hook": "counting comm hook (default averaging all-reduce); train_fns.COMM_SAVINGS=%s" % train_fns.COMM_SAVINGS}
        if not args.sync_bn and world > 1:
            # the same steps with cross-replica BN statistics (north_star: "SyncBatchNorm stats overlapped with backward"): the layers
            # read their `sync_bn` attribute per call, so the leg toggles it on the built networks -- one extra figure in the same run
            bns = [m for net in (G, G_ema) for m in net.modules() if hasattr(m, "sync_bn")]
            for m in bns:
                m.sync_bn = True
            one_step()
            t_sync = timed_region(args.steps)
            for m in bns:
                m.sync_bn = False
            comm_report["sync_bn_leg"] = {"ms_per_step": round(t_sync / args.steps * 1e3, 3),
                                          "images_per_sec": round(batch * acc * world * args.steps / t_sync, 3), "bn_layers": len(bns)}

    if rank == 0:
        roof = None if args.no_kernel_timer else assemble_roofline(timer, args.steps, elapsed,
                                                                    with_step_traffic=(args.workload == "cfg3" and acc == 1))
        out = {
            "metric": "images/sec G+D train step, IC-GAN BigGAN 256^2 bs=64/GPU" if args.workload == "cfg3"
            else ("images/sec G+D train step, IC-GAN BigGAN-deep 256^2 ch=128 bs=128/GPU (cfg5, secondary workload)"
                  if args.workload == "cfg5" else f"images/sec G+D train step, IC-GAN BigGAN ({args.workload})"),
            "value": round(batch * acc * world * args.steps / elapsed, 3), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: IC-GAN {cfg.get('model', 'BigGAN')} {cfg['resolution']}x{cfg['resolution']} ch={cfg['G_ch']}"
                                   f" class_cond={cfg['class_cond']} instance_cond={cfg['instance_cond']} hier attn@{cfg['G_attn']},"
                                   f" 1 D step + 1 G step + Adam x2 + EMA, fp32 exact MFMA"
                                   + (f", {acc} accumulation rounds of {batch} images per phase" if acc > 1 else "")
                                   + ("; parity: UNPINNED (the reference has no instance-conditioned BigGAN-deep, SURVEY F5)" if args.workload == "cfg5" else ""),
                       "batch_per_gpu": batch * acc, "micro_batch": batch, "accumulations": acc, "global_batch": batch * acc * world,
                       "parallelism": f"dp{world}",
                       "instrumented": not args.no_kernel_timer,
                       "timer_period": (None if args.no_kernel_timer else timer.period),   # HIP-event brackets on every P-th launch per (entry point, shape)
                       "uninstrumented_ms_per_step": (round(uninstr / args.steps * 1e3, 3) if uninstr is not None else None),
                       "uninstrumented_images_per_sec": (round(batch * acc * world * args.steps / uninstr, 3) if uninstr is not None else None),
                       "comm": comm_report, "rank_host_resources": host,
                       "rccl_world_size": (dist.get_world_size() if use_ddp else 1), "init": init, "sync_bn": bool(args.sync_bn), "winograd": not args.no_winograd, "wgrad_side_stream": bool(args.wgrad_stream),
                       "prefetch_next_step": bool(train_fns.PREFETCH_NEXT_STEP),   # the next step's first draw + generator forward queued before the loss read-back
                       "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "losses_last_step": metrics},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline_child(args.workload)
        print(json.dumps(out), flush=True)
    if use_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
