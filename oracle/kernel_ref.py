"""TEST INFRASTRUCTURE: plain-PyTorch fp32 (CPU) reference of every entry point of the C-ABI in
include/icgan_hip.h, one function per ``icg_*`` symbol, same argument order (minus the stream).

Two uses, both in tests only:
  * per-kernel parity on the GPU: run the HIP kernel on device buffers, this reference on host copies
  * `install(monkeypatch)`: route ``ic_gan_amd._lib.call`` to these functions so the *host-side* logic of the
    product (autograd wiring, layouts, schedules) can be exercised on a CPU-only box.  The product itself
    never imports this module and has no such route.

Arguments arrive as the torch tensors the product passes (their memory is interpreted exactly as the
kernels do: NHWC for activations), Python scalars, or None for absent pointers.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

PRE_RELU, PRE_AFFINE, UPSAMPLE2X, RES_UPSAMPLE2X, RES_RELU_MASK = 1, 2, 4, 8, 16


def mem(t: torch.Tensor) -> torch.Tensor:
    """Flat 1-D *view* of a tensor in memory order (contiguous or channels-last 4-D)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)
    raise ValueError("tensor is neither contiguous nor channels-last")


def _nhwc(t, b, h, w, c):
    return mem(t)[: b * h * w * c].view(b, h, w, c)


def _act(x, scale, shift, ss_bstride, flags, b, hs, ws, c):
    a = _nhwc(x, b, hs, ws, c)
    if flags & PRE_AFFINE:
        rows = b if ss_bstride else 1
        sc = mem(scale)[: rows * c].view(rows, 1, 1, c)
        sh = mem(shift)[: rows * c].view(rows, 1, 1, c)
        a = a * sc + sh
    if flags & PRE_RELU:
        a = F.relu(a)
    a = a.permute(0, 3, 1, 2)
    if flags & UPSAMPLE2X:
        a = F.interpolate(a, scale_factor=2)
    return a


def icg_conv2d_fprop(x, w, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, R, flags, alpha):
    up = 1 if flags & UPSAMPLE2X else 0
    a = _act(x, scale, shift, ss_bstride, flags, B, H >> up, W >> up, Cin)
    wt = mem(w)[: Cout * R * R * Cin].view(Cout, R, R, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(a, wt, None, 1, R // 2) * alpha
    if bias is not None:
        y = y + mem(bias)[:Cout].view(1, -1, 1, 1)
    if flags & RES_RELU_MASK:
        assert residual is not None and not flags & RES_UPSAMPLE2X
        r = _nhwc(residual, B, H, W, Cout).permute(0, 3, 1, 2)
        y = torch.where(r > 0, y, torch.zeros_like(y))
    elif residual is not None:
        if flags & RES_UPSAMPLE2X:
            r = _nhwc(residual, B, H // 2, W // 2, Cout).permute(0, 3, 1, 2)
            r = F.interpolate(r, scale_factor=2)
        else:
            r = _nhwc(residual, B, H, W, Cout).permute(0, 3, 1, 2)
        y = y + r
    mem(out)[: B * H * W * Cout].copy_(y.permute(0, 2, 3, 1).reshape(-1))


def icg_conv2d_fprop_workspace_bytes(B, H, W, Cin, Cout, R, flags):
    return 0


def icg_conv2d_fprop_ws(x, w, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, R, flags, alpha, workspace,
                        workspace_bytes):
    icg_conv2d_fprop(x, w, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, R, flags, alpha)


_WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)


def icg_wino_weight_transform(w, U, N, K):
    g = mem(w)[: N * 9 * K].view(N, 3, 3, K).double()
    u = torch.einsum("ar,nrsk,bs->abnk", _WINO_G, g, _WINO_G)          # [4][4][N][K]
    mem(U)[: 16 * N * K].copy_(u.reshape(-1).float())


def icg_conv2d_wino_workspace_bytes(B, H, W, Cin, Cout):
    return 16 * B * (H // 2) * (W // 2) * (Cin + Cout) * 4


def icg_conv2d_wino_fprop(x, U, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, flags, alpha, workspace,
                          workspace_bytes):
    u = mem(U)[: 16 * Cout * Cin].view(4, 4, Cout, Cin).double()
    inv = torch.tensor([[1.0, 0, 0, 0], [0, 1.0, -1.0, 0], [0, 0, 0, 1.0]], dtype=torch.float64)   # g = inv @ (G g)
    g = torch.einsum("ra,abnk,sb->nrsk", inv, u, inv).float().contiguous()                          # [Cout][3][3][Cin]
    icg_conv2d_fprop(x, g, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, 3, flags, alpha)


_WINO4_G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
                         [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)


def icg_wino4_weight_transform(w, U, N, K):
    g = mem(w)[: N * 9 * K].view(N, 3, 3, K).double()
    u = torch.einsum("ar,nrsk,bs->abnk", _WINO4_G, g, _WINO4_G)
    mem(U)[: 36 * N * K].copy_(u.reshape(-1).float())


def _fwino_ws_extra(planes, H, W, Cin, Cout):
    """room for the fragment-major weight copy of the fused narrow-layer kernel (csrc/fwino.hip)"""
    return planes * Cin * Cout * 4 if (Cin % 32 == 0 and Cout % 96 == 0 and H % 16 == 0 and W % 16 == 0) else 0


def icg_conv2d_wino4_workspace_bytes(B, H, W, Cin, Cout):
    return 36 * B * (H // 4) * (W // 4) * (Cin + Cout) * 4 + _fwino_ws_extra(36, H, W, Cin, Cout)


def icg_conv2d_wino4_fprop(x, U, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, flags, alpha, workspace,
                           workspace_bytes):
    u = mem(U)[: 36 * Cout * Cin].view(6, 6, Cout, Cin).double()
    inv = torch.tensor([[4.0, 0, 0, 0, 0, 0], [0, -3.0, 3.0, 0, 0, 0], [0, 0, 0, 0, 0, 1.0]], dtype=torch.float64)
    g = torch.einsum("ra,abnk,sb->nrsk", inv, u, inv).float().contiguous()
    icg_conv2d_fprop(x, g, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, 3, flags, alpha)
    _w4_store_v(workspace, _act(x, scale, shift, ss_bstride, flags, B, H, W, Cin), 36)


def icg_conv2d_wino_wgrad_workspace_bytes(B, H, W, Cin, Cout):
    return 16


def icg_conv2d_wino_wgrad(x, dy, dw, scale, shift, ss_bstride, B, H, W, Cin, Cout, flags, workspace, workspace_bytes):
    icg_conv2d_wgrad(x, dy, dw, scale, shift, ss_bstride, B, H, W, Cin, Cout, 3, flags, None, 0)


def icg_conv2d_wino4_wgrad_workspace_bytes(B, H, W, Cin, Cout):
    return 16


def icg_conv2d_wino4_wgrad(x, dy, dw, scale, shift, ss_bstride, B, H, W, Cin, Cout, flags, workspace, workspace_bytes):
    icg_conv2d_wgrad(x, dy, dw, scale, shift, ss_bstride, B, H, W, Cin, Cout, 3, flags, None, 0)


def icg_conv2d_wino4_wgrad_from_v_workspace_bytes(B, H, W, Cin, Cout, planes):
    return 16


_W4_BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0],
                       [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)


def _w4_store_v(workspace, a, planes):
    """what the HIP forward entries leave at the start of their workspace: V[i][j][tile][c] = (B^T d B)[i][j] of the activated
    (and, for the upsample-fused layer, upsampled) input a [B, C, H, W]"""
    Bn, C, H, W = a.shape
    th, tw = H // 4, W // 4
    ap = F.pad(a.double(), (1, 1, 1, 1)).permute(0, 2, 3, 1)                   # [B, H+2, W+2, C]
    d = torch.stack([torch.stack([ap[:, r: r + 4 * th: 4, s_: s_ + 4 * tw: 4, :] for s_ in range(6)], 0) for r in range(6)], 0)
    v = torch.einsum("ir,rsbyxc,js->ijbyxc", _W4_BT, d, _W4_BT)
    if planes == 25:
        idx = torch.tensor(_W4R)
        v = v[idx[:, None], idx[None, :]]
    flat = v.reshape(-1).float()
    workspace.view(-1)[: flat.numel() * 4].view(torch.float32).copy_(flat)


_W4_AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)


def icg_conv2d_wino4_wgrad_from_v(V, dy, dw, B, H, W, Cin, Cout, planes, dy_up, dy_alpha, workspace, workspace_bytes):
    """reference in the Winograd domain (fp64): dw = G^T [ sum_tiles (A dy A^T) .* V ] G over the stored planes"""
    th, tw = H // 4, W // 4
    T = B * th * tw
    comps = list(range(6)) if planes == 36 else _W4R
    n = len(comps)
    v = mem(V)[: planes * T * Cin].view(n, n, T, Cin).double()
    if dy_up:
        g = F.interpolate(_nhwc(dy, B, H // 2, W // 2, Cout).permute(0, 3, 1, 2), scale_factor=2)
    else:
        g = _nhwc(dy, B, H, W, Cout).permute(0, 3, 1, 2)
    g = (g.double() * dy_alpha).permute(0, 2, 3, 1).reshape(B, th, 4, tw, 4, Cout)          # [b, y, a, x, c, co]
    A = _W4_AT.t()[comps]                                                                   # [n][4]
    DY = torch.einsum("ia,byaxco,jc->ijbyxo", A, g, A).reshape(n, n, T, Cout)
    dU = torch.einsum("ijtc,ijto->ijco", v, DY)
    G = _WINO4_G[comps]                                                                     # [n][3]
    gw = torch.einsum("ir,ijco,js->rsco", G, dU, G)                                         # HWIO
    mem(dw)[: 9 * Cin * Cout].copy_(gw.reshape(-1).float())


def icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes(B, H, W, Cin, Cout, planes):
    return 16


def icg_conv2d_wino4_wgrad_from_v_db(V, dy, dw, dbias, B, H, W, Cin, Cout, planes, dy_up, dy_alpha, workspace, workspace_bytes):
    """... and dbias = column sums of dy (dy at the pooled resolution when dy_up)"""
    icg_conv2d_wino4_wgrad_from_v(V, dy, dw, B, H, W, Cin, Cout, planes, dy_up, dy_alpha, workspace, workspace_bytes)
    hh, ww = (H // 2, W // 2) if dy_up else (H, W)
    mem(dbias)[:Cout].copy_(_nhwc(dy, B, hh, ww, Cout).double().sum((0, 1, 2)).float())


def icg_gemm_tn_batched_workspace_bytes(M, N, K, batch):
    return 16


def icg_gemm_tn_batched(A, B, C, M, N, K, strideA, strideB, strideC, batch, workspace, workspace_bytes):
    for b in range(batch):
        a = mem(A)[b * strideA: b * strideA + K * M].view(K, M).double()
        bb = mem(B)[b * strideB: b * strideB + K * N].view(K, N).double()
        mem(C)[b * strideC: b * strideC + M * N].copy_((a.t() @ bb).float().reshape(-1))


def icg_conv2d_wgrad_workspace_bytes(B, H, W, Cin, Cout, R):
    return 16


def icg_conv2d_wgrad(x, dy, dw, scale, shift, ss_bstride, B, H, W, Cin, Cout, R, flags, workspace, workspace_bytes):
    up = 1 if flags & UPSAMPLE2X else 0
    a = _act(x, scale, shift, ss_bstride, flags, B, H >> up, W >> up, Cin)
    g = _nhwc(dy, B, H, W, Cout).permute(0, 3, 1, 2)
    gw = torch.nn.grad.conv2d_weight(a.contiguous(), (Cout, Cin, R, R), g.contiguous(), padding=R // 2)
    mem(dw)[: R * R * Cin * Cout].copy_(gw.permute(2, 3, 1, 0).reshape(-1))       # HWIO


# tap sets of the 4-phase form of (nearest x2 upsample -> 3x3 conv): phase al, source tap u  ->  3x3 taps r
_S = {(0, 0): [0], (0, 1): [1, 2], (1, 0): [0, 1], (1, 1): [2]}
_T = {0: [2], 1: [1, 2], 2: [0, 1], 3: [0]}          # 4x4 stride-2 data-gradient taps


def icg_conv2d_up_fprop(x, wp, bias, out, scale, shift, ss_bstride, B, Hs, Ws, Cin, Cout, flags):
    a = _act(x, scale, shift, ss_bstride, flags, B, Hs, Ws, Cin)               # [B,Cin,Hs,Ws]
    w = mem(wp)[: 16 * Cout * Cin].view(4, Cout, 2, 2, Cin)
    y = torch.empty(B, Cout, 2 * Hs, 2 * Ws)
    for al in range(2):
        for be in range(2):
            ap = F.pad(a, (1 - be, be, 1 - al, al))
            y[:, :, al::2, be::2] = F.conv2d(ap, w[al * 2 + be].permute(0, 3, 1, 2))
    if bias is not None:
        y = y + mem(bias)[:Cout].view(1, -1, 1, 1)
    mem(out)[: B * 4 * Hs * Ws * Cout].copy_(y.permute(0, 2, 3, 1).reshape(-1))


def icg_conv2d_up_dgrad(dy, vd, da, B, Hs, Ws, Cin, Cout):
    g = _nhwc(dy, B, 2 * Hs, 2 * Ws, Cout).permute(0, 3, 1, 2)
    w = mem(vd)[: 16 * Cout * Cin].view(Cin, 4, 4, Cout).permute(0, 3, 1, 2)     # [Cin][Cout][4][4]
    r = F.conv2d(g, w, None, stride=2, padding=1)
    mem(da)[: B * Hs * Ws * Cin].copy_(r.permute(0, 2, 3, 1).reshape(-1))


def icg_conv2d_up_wgrad_workspace_bytes(B, Hs, Ws, Cin, Cout):
    return 16


def icg_conv2d_up_wgrad(x, dy, dwp, scale, shift, ss_bstride, B, Hs, Ws, Cin, Cout, flags, workspace, workspace_bytes):
    a = _act(x, scale, shift, ss_bstride, flags, B, Hs, Ws, Cin)
    g = _nhwc(dy, B, 2 * Hs, 2 * Ws, Cout).permute(0, 3, 1, 2)
    o = torch.empty(4, 2, 2, Cin, Cout)
    for al in range(2):
        for be in range(2):
            ap = F.pad(a, (1 - be, be, 1 - al, al)).contiguous()
            gw = torch.nn.grad.conv2d_weight(ap, (Cout, Cin, 2, 2), g[:, :, al::2, be::2].contiguous())
            o[al * 2 + be] = gw.permute(2, 3, 1, 0)
    mem(dwp)[: 16 * Cin * Cout].copy_(o.reshape(-1))


def icg_conv2d_down_fprop(x, vdn, bias, residual, out, B, Hp, Wp, Cin, Cout, flags):
    a = _act(x, None, None, 0, flags, B, 2 * Hp, 2 * Wp, Cin)
    w = mem(vdn)[: 16 * Cout * Cin].view(Cout, 4, 4, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(a, w, None, stride=2, padding=1)
    if bias is not None:
        y = y + mem(bias)[:Cout].view(1, -1, 1, 1)
    if residual is not None:
        y = y + _nhwc(residual, B, Hp, Wp, Cout).permute(0, 3, 1, 2)
    mem(out)[: B * Hp * Wp * Cout].copy_(y.permute(0, 2, 3, 1).reshape(-1))


def icg_conv2d_down_dgrad(dy, wq, da, B, Hp, Wp, Cin, Cout):
    g = _nhwc(dy, B, Hp, Wp, Cout).permute(0, 3, 1, 2)
    w = mem(wq)[: 16 * Cout * Cin].view(4, Cin, 2, 2, Cout)
    r = torch.empty(B, Cin, 2 * Hp, 2 * Wp)
    for al in range(2):
        for be in range(2):
            gp = F.pad(g, (1 - be, be, 1 - al, al))
            r[:, :, al::2, be::2] = F.conv2d(gp, w[al * 2 + be].permute(0, 3, 1, 2))
    mem(da)[: B * 4 * Hp * Wp * Cin].copy_(r.permute(0, 2, 3, 1).reshape(-1))


def _relu_mask_inplace(dx, relu_in, n):
    d, r = mem(dx)[:n], mem(relu_in)[:n]
    d.copy_(torch.where(r > 0, d, torch.zeros_like(d)))


def icg_conv2d_down_dgrad_relu(dy, wq, relu_in, dx, B, Hp, Wp, Cin, Cout):
    icg_conv2d_down_dgrad(dy, wq, dx, B, Hp, Wp, Cin, Cout)
    _relu_mask_inplace(dx, relu_in, B * 4 * Hp * Wp * Cin)


def icg_conv2d_down_wgrad_workspace_bytes(B, Hp, Wp, Cin, Cout):
    return 16


def icg_conv2d_down_wgrad(x, dy, dvdn, B, Hp, Wp, Cin, Cout, flags, workspace, workspace_bytes):
    a = _act(x, None, None, 0, flags, B, 2 * Hp, 2 * Wp, Cin)
    g = _nhwc(dy, B, Hp, Wp, Cout).permute(0, 3, 1, 2)
    gw = torch.nn.grad.conv2d_weight(a.contiguous(), (Cout, Cin, 4, 4), g.contiguous(), stride=2, padding=1)
    mem(dvdn)[: 16 * Cin * Cout].copy_(gw.permute(2, 3, 1, 0).reshape(-1))       # [P][Q][ci][co]


# ---- resample-fused layers in the 25-plane F(4x4,3x3) domain: recover the 3x3 kernel from U and evaluate the op graph itself
_W4R = [0, 1, 3, 4, 5]                              # transform components kept in the 25-plane layout


def icg_wino4r_weight_transform(w, U, N, K):
    g = mem(w)[: N * 9 * K].view(N, 3, 3, K).double()
    u = torch.einsum("ar,nrsk,bs->abnk", _WINO4_G[_W4R], g, _WINO4_G[_W4R])
    mem(U)[: 25 * N * K].copy_(u.reshape(-1).float())


def _w4r_kernel(U, N, K):
    """[N][3][3][K] from U [5][5][N][K]:  g0 = 4 u0,  g2 = u5,  g1 = -6 u1 - g0 - g2"""
    u = mem(U)[: 25 * N * K].view(5, 5, N, K).double()
    inv = torch.tensor([[4.0, 0, 0, 0, 0], [-4.0, -6.0, 0, 0, -1.0], [0, 0, 0, 0, 1.0]], dtype=torch.float64)
    return torch.einsum("ra,abnk,sb->nrsk", inv, u, inv).float().contiguous()


def icg_conv2d_rs_wino_workspace_bytes(B, H, W, Cin, Cout):
    return 25 * B * (H // 4) * (W // 4) * (Cin + Cout) * 4 + _fwino_ws_extra(25, H, W, Cin, Cout)


def icg_conv2d_rs_wino_wgrad_workspace_bytes(B, H, W, Cin, Cout):
    return 16


def icg_conv2d_up_wino_fprop(x, U, bias, out, scale, shift, ss_bstride, B, Hs, Ws, Cin, Cout, flags, workspace, workspace_bytes):
    g = _w4r_kernel(U, Cout, Cin)
    icg_conv2d_fprop(x, g, bias, None, out, scale, shift, ss_bstride, B, 2 * Hs, 2 * Ws, Cin, Cout, 3, flags | UPSAMPLE2X, 1.0)
    _w4_store_v(workspace, _act(x, scale, shift, ss_bstride, flags | UPSAMPLE2X, B, Hs, Ws, Cin), 25)


def icg_conv2d_up_wino_dgrad(dy, U, da, B, Hs, Ws, Cin, Cout, workspace, workspace_bytes):
    wd = _w4r_kernel(U, Cin, Cout)                                       # dgrad layout [Cin][3][3][Cout], taps flipped
    g = _nhwc(dy, B, 2 * Hs, 2 * Ws, Cout).permute(0, 3, 1, 2)
    r = F.conv2d(g, wd.permute(0, 3, 1, 2), None, 1, 1)                  # gradient at the upsampled resolution
    r = F.avg_pool2d(r, 2) * 4.0                                         # adjoint of the nearest upsample
    mem(da)[: B * Hs * Ws * Cin].copy_(r.permute(0, 2, 3, 1).reshape(-1))


def icg_conv2d_up_wino_wgrad(x, dy, dw, scale, shift, ss_bstride, B, Hs, Ws, Cin, Cout, flags, workspace, workspace_bytes):
    icg_conv2d_wgrad(x, dy, dw, scale, shift, ss_bstride, B, 2 * Hs, 2 * Ws, Cin, Cout, 3, flags | UPSAMPLE2X, None, 0)


def icg_conv2d_down_wino_fprop(x, U, bias, residual, out, B, Hp, Wp, Cin, Cout, flags, workspace, workspace_bytes):
    a = _act(x, None, None, 0, flags, B, 2 * Hp, 2 * Wp, Cin)
    g = _w4r_kernel(U, Cout, Cin)
    y = F.avg_pool2d(F.conv2d(a, g.permute(0, 3, 1, 2), None, 1, 1), 2)
    if bias is not None:
        y = y + mem(bias)[:Cout].view(1, -1, 1, 1)
    if residual is not None:
        y = y + _nhwc(residual, B, Hp, Wp, Cout).permute(0, 3, 1, 2)
    mem(out)[: B * Hp * Wp * Cout].copy_(y.permute(0, 2, 3, 1).reshape(-1))
    _w4_store_v(workspace, a, 25)


def icg_conv2d_down_wino_dgrad(dy, U, da, B, Hp, Wp, Cin, Cout, workspace, workspace_bytes):
    wd = _w4r_kernel(U, Cin, Cout)
    g = F.interpolate(_nhwc(dy, B, Hp, Wp, Cout).permute(0, 3, 1, 2), scale_factor=2) * 0.25
    r = F.conv2d(g, wd.permute(0, 3, 1, 2), None, 1, 1)
    mem(da)[: B * 4 * Hp * Wp * Cin].copy_(r.permute(0, 2, 3, 1).reshape(-1))


def icg_conv2d_down_wino_dgrad_relu(dy, U, relu_in, dx, B, Hp, Wp, Cin, Cout, workspace, workspace_bytes):
    icg_conv2d_down_wino_dgrad(dy, U, dx, B, Hp, Wp, Cin, Cout, workspace, workspace_bytes)
    _relu_mask_inplace(dx, relu_in, B * 4 * Hp * Wp * Cin)


def icg_conv2d_down_wino_wgrad(x, dy, dw, B, Hp, Wp, Cin, Cout, flags, workspace, workspace_bytes):
    a = _act(x, None, None, 0, flags, B, 2 * Hp, 2 * Wp, Cin)
    g = F.interpolate(_nhwc(dy, B, Hp, Wp, Cout).permute(0, 3, 1, 2), scale_factor=2) * 0.25
    gw = torch.nn.grad.conv2d_weight(a.contiguous(), (Cout, Cin, 3, 3), g.contiguous(), padding=1)
    mem(dw)[: 9 * Cin * Cout].copy_(gw.permute(2, 3, 1, 0).reshape(-1))       # HWIO


# ---- fused F(4x4,3x3) kernel for the narrow layers (csrc/fwino.hip): the same op graph, evaluated directly
def icg_fwino_applies(B, H, W, Cin, Cout):
    import os
    env = lambda k, d: int(os.environ.get(k, d))
    if not env("ICG_FWINO", 1):
        return 0
    if Cin % 32 or Cin > env("ICG_FWINO_MAXK", 192) or Cout % 96 or Cout > env("ICG_FWINO_MAXN", 192) or H % 16 or W % 16:
        return 0
    if B * H * W * Cin * 4 >= 4278190080 or B * H * W * Cout * 4 >= 4278190080:      # 32-bit byte offsets into x / out
        return 0
    wgs = B * (H // 16) * (W // 16) * (Cout // 96)
    return 1 if env("ICG_FWINO_MIN_WGS", 512) <= wgs < 0x7FFFFFFF else 0


def icg_fwino_weight_bytes(planes, Cin, Cout):
    return planes * Cin * Cout * 4


def icg_fwino_pack_weights(U, Uf, planes, Cin, Cout):
    """Uf[((p NT + jt) KG + kg) 64 + l][e] = U[p][16 jt + (l & 15)][16 kg + 4 (l >> 4) + e]"""
    u = mem(U)[: planes * Cout * Cin].view(planes, Cout // 16, 16, Cin // 16, 4, 4)          # [p][jt][n][kg][kq][e]
    mem(Uf)[: planes * Cout * Cin].copy_(u.permute(0, 1, 3, 4, 2, 5).reshape(-1))           # [p][jt][kg][kq][n][e]


def _fwino_unpack(Uf, planes, Cin, Cout):
    uf = mem(Uf)[: planes * Cout * Cin].view(planes, Cout // 16, Cin // 16, 4, 16, 4)        # [p][jt][kg][kq][n][e]
    return uf.permute(0, 1, 4, 2, 3, 5).reshape(planes * Cout * Cin).contiguous()


def icg_fwino_conv(x, Uf, bias, residual, out, scale, shift, ss_bstride, B, H, W, Cin, Cout, flags, alpha, in_up, out_pool, V):
    planes = 25 if (in_up or out_pool) else 36
    U = _fwino_unpack(Uf, planes, Cin, Cout)
    if planes == 25:
        g = _w4r_kernel(U, Cout, Cin)
    else:
        u = U.view(6, 6, Cout, Cin).double()
        inv = torch.tensor([[4.0, 0, 0, 0, 0, 0], [0, -3.0, 3.0, 0, 0, 0], [0, 0, 0, 0, 0, 1.0]], dtype=torch.float64)
        g = torch.einsum("ra,abnk,sb->nrsk", inv, u, inv).float().contiguous()
    a = _act(x, scale, shift, ss_bstride, flags | (UPSAMPLE2X if in_up else 0), B, H >> (1 if in_up else 0), W >> (1 if in_up else 0), Cin)
    y = F.conv2d(a, g.permute(0, 3, 1, 2), None, 1, 1)
    Ho, Wo = H, W
    if out_pool:
        y = F.avg_pool2d(y, 2) * 4.0
        Ho, Wo = H // 2, W // 2
    y = y * alpha
    if bias is not None:
        y = y + mem(bias)[:Cout].view(1, -1, 1, 1)
    if flags & RES_RELU_MASK:
        r = _nhwc(residual, B, Ho, Wo, Cout).permute(0, 3, 1, 2)
        y = torch.where(r > 0, y, torch.zeros((), dtype=y.dtype))
    elif residual is not None:
        if flags & RES_UPSAMPLE2X:
            y = y + F.interpolate(_nhwc(residual, B, Ho // 2, Wo // 2, Cout).permute(0, 3, 1, 2), scale_factor=2)
        else:
            y = y + _nhwc(residual, B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    mem(out)[: B * Ho * Wo * Cout].copy_(y.permute(0, 2, 3, 1).reshape(-1))
    if V is not None:
        n = planes * B * (H // 4) * (W // 4) * Cin
        tmp = torch.empty(n * 4, dtype=torch.uint8)
        _w4_store_v(tmp, a, planes)
        mem(V)[:n].copy_(tmp.view(torch.float32))


def icg_gemm_batched(A, Bm, C, M, N, K, transA, transB, strideA, strideB, strideC, batch, alpha):
    a, b, c = mem(A), mem(Bm), mem(C)
    for z in range(batch):
        az = a[z * strideA: z * strideA + M * K]
        bz = b[z * strideB: z * strideB + N * K]
        am = az.view(K, M).t() if transA else az.view(M, K)
        bm = bz.view(N, K).t() if transB else bz.view(K, N)
        c[z * strideC: z * strideC + M * N].copy_((alpha * (am @ bm)).reshape(-1))


def icg_plane_gemm(A, Bm, C, M, N, K, planes, alpha):
    icg_gemm_batched(A, Bm, C, M, N, K, 0, 1, M * K, N * K, M * N, planes, alpha)


def icg_plane_gemm_tn_workspace_bytes(M, N, K, planes):
    return 16


def icg_plane_gemm_tn(A, Bm, C, M, N, K, planes, workspace, workspace_bytes):
    icg_gemm_batched(A, Bm, C, M, N, K, 1, 0, K * M, K * N, M * N, planes, 1.0)


# ---------------------------------------------------------------- BN
def icg_bn_workspace_bytes(rows, C):
    return 2 * C * 4


def icg_bn_partial_stats(x, shift_k, rows, C, workspace, workspace_bytes):
    v = mem(x)[: rows * C].view(rows, C).double()
    k = mem(shift_k)[:C].double() if shift_k is not None else torch.zeros(C, dtype=torch.float64)
    d = v - k
    ws = workspace.view(torch.float32)
    ws[:C] = d.sum(0).float()
    ws[C: 2 * C] = (d * d).sum(0).float()
    workspace._emu_sums = torch.cat([d.sum(0), (d * d).sum(0)])    # keep full precision for the next stage


def icg_bn_reduce_partials(workspace, rows, C, sums):
    sums[: 2 * C].copy_(workspace._emu_sums)


def icg_bn_finalize(sums, shift_k, count, running_mean, running_var, momentum, eps, training, gain, bias, gb_rows,
                    gain_offset, C, mean, invstd, scale, shift):
    if training:
        k = mem(shift_k)[:C].double() if shift_k is not None else torch.zeros(C, dtype=torch.float64)
        if count <= 0:
            count = float(sums[2 * C])          # packed cross-replica payload (icg_bn_sync_pack)
        m1 = sums[:C] / count
        var = (sums[C: 2 * C] / count - m1 * m1).clamp_min(0.0)
        mu = k + m1
        istd = 1.0 / torch.sqrt(var + eps)
        if running_mean is not None:
            unb = var * count / (count - 1.0) if count > 1 else var
            running_mean.copy_(((1 - momentum) * running_mean.double() + momentum * mu).float())
            running_var.copy_(((1 - momentum) * running_var.double() + momentum * unb).float())
    else:
        mu = running_mean.double()
        istd = 1.0 / torch.sqrt(running_var.double() + eps)
    mean.copy_(mu.float())
    invstd.copy_(istd.float())
    g = gain_offset + (mem(gain)[: gb_rows * C].view(gb_rows, C) if gain is not None else torch.zeros(gb_rows, C))
    be = mem(bias)[: gb_rows * C].view(gb_rows, C) if bias is not None else torch.zeros(gb_rows, C)
    sc = invstd.view(1, C) * g
    mem(scale)[: gb_rows * C].copy_(sc.reshape(-1))
    mem(shift)[: gb_rows * C].copy_((be - mean.view(1, C) * sc).reshape(-1))


def icg_bn_reduce_finalize(workspace, rows, C, shift_k, running_mean, running_var, momentum, eps, gain, bias, gb_rows, gain_offset,
                           mean, invstd, scale, shift):
    sums = torch.empty(2 * C, dtype=torch.float64)
    icg_bn_reduce_partials(workspace, rows, C, sums)
    icg_bn_finalize(sums, shift_k, float(rows), running_mean, running_var, momentum, eps, 1, gain, bias, gb_rows, gain_offset, C, mean,
                    invstd, scale, shift)


def icg_bn_sync_pack(sums, shift_k, local_count, C, payload):
    k = mem(shift_k)[:C].double() if shift_k is not None else torch.zeros(C, dtype=torch.float64)
    s1, s2 = sums[:C], sums[C: 2 * C]
    payload[:C] = s1 + local_count * k
    payload[C: 2 * C] = s2 + 2.0 * k * s1 + local_count * k * k
    payload[2 * C] = local_count


def icg_bn_apply(x, scale, shift, ss_bstride, B, HW, C, flags, y):
    a = _act(x, scale, shift, ss_bstride, flags | PRE_AFFINE, B, HW, 1, C)
    mem(y)[: B * HW * C].copy_(a.permute(0, 2, 3, 1).reshape(-1))


def _bwd_dy(x, da, scale, shift, ss_bstride, B, Hs, Ws, C, flags):
    xv = _nhwc(x, B, Hs, Ws, C)
    if flags & UPSAMPLE2X:
        d = _nhwc(da, B, 2 * Hs, 2 * Ws, C).view(B, Hs, 2, Ws, 2, C).sum((2, 4))
    else:
        d = _nhwc(da, B, Hs, Ws, C)
    sc = sh = None
    if flags & PRE_AFFINE:
        rows = B if ss_bstride else 1
        sc = mem(scale)[: rows * C].view(rows, 1, 1, C)
        sh = mem(shift)[: rows * C].view(rows, 1, 1, C)
    if flags & PRE_RELU:
        y = xv * sc + sh if sc is not None else xv
        d = d * (y > 0)
    return xv, d, sc


def icg_bn_bwd_workspace_bytes(B, Hs, Ws, C):
    return 16


def icg_bn_bwd_reduce(x, da, scale, shift, ss_bstride, mean, B, Hs, Ws, C, flags, workspace, workspace_bytes, sum_dy,
                      sum_dyx):
    xv, d, _ = _bwd_dy(x, da, scale, shift, ss_bstride, B, Hs, Ws, C, flags)
    mu = mem(mean)[:C].view(1, 1, 1, C) if mean is not None else 0.0
    mem(sum_dy)[: B * C].copy_(d.double().sum((1, 2)).float().reshape(-1))
    mem(sum_dyx)[: B * C].copy_((d.double() * (xv - mu).double()).sum((1, 2)).float().reshape(-1))


def icg_bn_bwd_channel_sums(sum_dy, sum_dyx, gain, gb_rows, gain_offset, invstd, B, C, chan_sums):
    g = gain_offset + (mem(gain)[: gb_rows * C].view(gb_rows, C).double() if gain is not None
                       else torch.zeros(gb_rows, C, dtype=torch.float64))
    sd = mem(sum_dy)[: B * C].view(B, C).double()
    sx = mem(sum_dyx)[: B * C].view(B, C).double()
    chan_sums[:C] = (g * sd).sum(0)
    chan_sums[C: 2 * C] = (g * invstd.double().view(1, C) * sx).sum(0)


def icg_bn_bwd_coefs(sum_dy, sum_dyx, chan_sums, invstd, count, batch_stats, gb_rows, B, C, dgain, dbias, coefA, coefB):
    sd = mem(sum_dy)[: B * C].view(B, C).double()
    sx = mem(sum_dyx)[: B * C].view(B, C).double()
    istd = invstd.double().view(1, C)
    dg, db = istd * sx, sd
    if gb_rows == 1:
        dg, db = dg.sum(0, keepdim=True), db.sum(0, keepdim=True)
    if dgain is not None:
        mem(dgain)[: gb_rows * C].copy_(dg.float().reshape(-1))
    if dbias is not None:
        mem(dbias)[: gb_rows * C].copy_(db.float().reshape(-1))
    if batch_stats:
        if count <= 0:
            count = float(chan_sums[2 * C])
        coefA.copy_((invstd.double() * chan_sums[:C] / count).float())
        coefB.copy_((invstd.double() ** 2 * chan_sums[C: 2 * C] / count).float())
    else:
        coefA.zero_()
        coefB.zero_()


def icg_bn_bwd_apply(x, da, scale, shift, ss_bstride, mean, coefA, coefB, B, Hs, Ws, C, flags, dx):
    xv, d, sc = _bwd_dy(x, da, scale, shift, ss_bstride, B, Hs, Ws, C, flags)
    o = d * sc if sc is not None else d
    if coefA is not None:
        o = o - coefA.view(1, 1, 1, C) - coefB.view(1, 1, 1, C) * (xv - mem(mean)[:C].view(1, 1, 1, C))
    mem(dx)[: B * Hs * Ws * C].copy_(o.reshape(-1))


# ---------------------------------------------------------------- spectral norm
def icg_sn_scratch_bytes(rows, Cin, R):
    return 16


_C = {0: [0], 1: [0, 1], 2: [1, 2], 3: [2]}          # 4x4 pooled-kernel taps:  c[P] = sum_{a+r=P} w[r]
_PD = {(0, 0): 3, (0, 1): 1, (1, 0): 2, (1, 1): 0}    # down-dgrad phase (al,u) -> P


def icg_sn_forward(w, u, sv, rows, Cin, R, eps, training, v_out, u_out, sigma_out, w_ohwi, w_dgrad, w_up_fprop,
                   w_up_dgrad, w_down_fprop, w_down_dgrad, scratch, scratch_bytes):
    wm = mem(w)[: rows * Cin * R * R].view(rows, -1)
    uu = mem(u)[:rows].view(1, rows)
    v = F.normalize(uu @ wm, eps=eps)
    s = v @ wm.t()
    un = F.normalize(s, eps=eps)
    sigma = (s * un).sum()
    v_out.copy_(v.view(-1))
    u_out.copy_(un.view(-1))
    sigma_out.fill_(float(sigma))
    if training:
        mem(u)[:rows].copy_(un.view(-1))
        if sv is not None:
            sv.fill_(float(sigma))
    w4 = wm.view(rows, Cin, R, R) / sigma
    w_ohwi.copy_(w4.permute(0, 2, 3, 1).reshape(-1))
    if w_dgrad is not None:
        w_dgrad.copy_(w4.flip(2, 3).permute(1, 2, 3, 0).reshape(-1))
    if w_up_fprop is not None:
        wp = torch.zeros(4, rows, 2, 2, Cin)
        for al in range(2):
            for be in range(2):
                for uu in range(2):
                    for vv in range(2):
                        acc = 0
                        for r in _S[(al, uu)]:
                            for c in _S[(be, vv)]:
                                acc = acc + w4[:, :, r, c]
                        wp[al * 2 + be, :, uu, vv, :] = acc
        w_up_fprop.copy_(wp.reshape(-1))
    if w_up_dgrad is not None:
        vd = torch.zeros(Cin, 4, 4, rows)
        for P in range(4):
            for Q in range(4):
                acc = 0
                for r in _T[P]:
                    for c in _T[Q]:
                        acc = acc + w4[:, :, r, c]
                vd[:, P, Q, :] = acc.t()
        w_up_dgrad.copy_(vd.reshape(-1))
    if w_down_fprop is not None or w_down_dgrad is not None:
        vdn = torch.zeros(rows, 4, 4, Cin)
        for P in range(4):
            for Q in range(4):
                acc = 0
                for r in _C[P]:
                    for c in _C[Q]:
                        acc = acc + w4[:, :, r, c]
                vdn[:, P, Q, :] = 0.25 * acc
        if w_down_fprop is not None:
            w_down_fprop.copy_(vdn.reshape(-1))
        if w_down_dgrad is not None:
            wq = torch.zeros(4, Cin, 2, 2, rows)
            for (al, uu), P in _PD.items():
                for (be, vv), Q in _PD.items():
                    wq[al * 2 + be, :, uu, vv, :] = vdn[:, P, Q, :].t()
            w_down_dgrad.copy_(wq.reshape(-1))


def icg_sn_backward_scratch_bytes(rows, Cin, R):
    return 256 * 8 + rows * Cin * R * R * 4


def icg_sn_backward(dw_hwio, dw_ohwi, dw_up, dw_down, w_ohwi, u_saved, v_saved, sigma, rows, Cin, R, dw, accumulate,
                    scratch, scratch_bytes):
    g = torch.zeros(rows, Cin, R, R)
    if dw_hwio is not None:
        g = g + mem(dw_hwio)[: rows * Cin * R * R].view(R, R, Cin, rows).permute(3, 2, 0, 1)
    if dw_ohwi is not None:
        g = g + mem(dw_ohwi)[: rows * Cin * R * R].view(rows, R, R, Cin).permute(0, 3, 1, 2)
    if dw_up is not None:       # adjoint of the phase-weight construction in icg_sn_forward
        d = mem(dw_up)[: 16 * rows * Cin].view(4, 2, 2, Cin, rows)
        g = g.clone()
        for al in range(2):
            for be in range(2):
                for uu in range(2):
                    for vv in range(2):
                        for r in _S[(al, uu)]:
                            for c in _S[(be, vv)]:
                                g[:, :, r, c] += d[al * 2 + be, uu, vv].t()
    if dw_down is not None:     # adjoint of vdn = 0.25 * c (x) c
        d = mem(dw_down)[: 16 * rows * Cin].view(4, 4, Cin, rows)
        g = g.clone()
        for P in range(4):
            for Q in range(4):
                for r in _C[P]:
                    for c in _C[Q]:
                        g[:, :, r, c] += 0.25 * d[P, Q].t()
    w_ = w_ohwi.view(rows, R, R, Cin).permute(0, 3, 1, 2)
    dot = (g.double() * w_.double()).sum().float()
    corr = 0.0
    if u_saved is not None and v_saved is not None:
        corr = dot * (u_saved.view(rows, 1) * v_saved.view(1, -1)).view(rows, Cin, R, R)
    val = (g - corr) / sigma
    d = mem(dw)[: rows * Cin * R * R]
    d.copy_(d + val.reshape(-1) if accumulate else val.reshape(-1))


# ---------------------------------------------------------------- pointwise
def icg_nchw_to_nhwc(x, y, B, C, H, W):
    mem(y)[: B * C * H * W].copy_(mem(x)[: B * C * H * W].view(B, C, H * W).transpose(1, 2).reshape(-1))


def icg_nhwc_to_nchw(x, y, B, C, H, W):
    mem(y)[: B * C * H * W].copy_(mem(x)[: B * C * H * W].view(B, H * W, C).transpose(1, 2).reshape(-1))


def icg_tanh_fwd(x, y, n):
    mem(y)[:n].copy_(torch.tanh(mem(x)[:n]))


def icg_tanh_bwd(y, dy, dx, n):
    yy = mem(y)[:n]
    mem(dx)[:n].copy_(mem(dy)[:n] * (1 - yy * yy))


def icg_relu_fwd(x, y, n):
    mem(y)[:n].copy_(F.relu(mem(x)[:n]))


def icg_relu_bwd(x, dy, dx, n):
    mem(dx)[:n].copy_(mem(dy)[:n] * (mem(x)[:n] > 0))


def icg_add(a, b, y, n):
    mem(y)[:n].copy_(mem(a)[:n] + mem(b)[:n])


def _pool_view(t, B, H, W, C):
    return _nhwc(t, B, H, W, C).view(B, H // 2, 2, W // 2, 2, C)


def icg_avgpool2_fwd(x, add, y, B, H, W, C):
    v = _pool_view(x, B, H, W, C).sum((2, 4)) * 0.25
    if add is not None:
        v = v + _nhwc(add, B, H // 2, W // 2, C)
    mem(y)[: B * H * W * C // 4].copy_(v.reshape(-1))


def icg_sumpool2_fwd(x, y, B, H, W, C):
    mem(y)[: B * H * W * C // 4].copy_(_pool_view(x, B, H, W, C).sum((2, 4)).reshape(-1))


def icg_avgpool2_bwd(dy, dx, B, H, W, C):
    g = _nhwc(dy, B, H // 2, W // 2, C) * 0.25
    mem(dx)[: B * H * W * C].copy_(g.view(B, H // 2, 1, W // 2, 1, C).expand(B, H // 2, 2, W // 2, 2, C).reshape(-1))


def icg_avgpool2_bwd_add(dy, carry, dx, B, H, W, C):
    g = _nhwc(dy, B, H // 2, W // 2, C) * 0.25
    full = g.view(B, H // 2, 1, W // 2, 1, C).expand(B, H // 2, 2, W // 2, 2, C).reshape(-1)
    mem(dx)[: B * H * W * C].copy_(full + mem(carry)[: B * H * W * C])


def icg_maxpool2_fwd(x, y, B, H, W, C):
    xn = _nhwc(x, B, H, W, C).permute(0, 3, 1, 2)
    mem(y)[: B * H * W * C // 4].copy_(F.max_pool2d(xn, 2).permute(0, 2, 3, 1).reshape(-1))


def icg_maxpool2_bwd(x, dy, dx, B, H, W, C):
    with torch.enable_grad():
        xn = _nhwc(x, B, H, W, C).permute(0, 3, 1, 2).detach().clone().requires_grad_(True)
        out = F.max_pool2d(xn, 2)
        g = _nhwc(dy, B, H // 2, W // 2, C).permute(0, 3, 1, 2)
        (gx,) = torch.autograd.grad(out, xn, g)
    mem(dx)[: B * H * W * C].copy_(gx.permute(0, 2, 3, 1).reshape(-1))


def icg_attn_split_pool(y, theta, phi_p, g_p, B, H, W, d, dv):
    yy = _nhwc(y, B, H, W, 2 * d + dv)
    mem(theta)[: B * H * W * d].copy_(yy[..., :d].reshape(-1))
    for dst, sl, c in ((phi_p, slice(d, 2 * d), d), (g_p, slice(2 * d, 2 * d + dv), dv)):
        p = F.max_pool2d(yy[..., sl].permute(0, 3, 1, 2), 2)
        mem(dst)[: B * (H // 2) * (W // 2) * c].copy_(p.permute(0, 2, 3, 1).reshape(-1))


def icg_attn_split_pool_bwd(y, dtheta, dphi_p, dg_p, dy, B, H, W, d, dv):
    C = 2 * d + dv
    yy = _nhwc(y, B, H, W, C)
    out = torch.empty(B, H, W, C)
    out[..., :d] = mem(dtheta)[: B * H * W * d].view(B, H, W, d)
    for src, sl, c in ((dphi_p, slice(d, 2 * d), d), (dg_p, slice(2 * d, C), dv)):
        with torch.enable_grad():
            xn = yy[..., sl].permute(0, 3, 1, 2).detach().clone().requires_grad_(True)
            o = F.max_pool2d(xn, 2)
            g = mem(src)[: B * (H // 2) * (W // 2) * c].view(B, H // 2, W // 2, c).permute(0, 3, 1, 2)
            (gx,) = torch.autograd.grad(o, xn, g)
        out[..., sl] = gx.permute(0, 2, 3, 1)
    mem(dy)[: B * H * W * C].copy_(out.reshape(-1))


def icg_attn_gamma_scale(gamma, w_a, ws_a, w_b, ws_b, n):
    g = mem(gamma)[0]
    mem(ws_a)[:n].copy_(g * mem(w_a)[:n])
    if w_b is not None:
        mem(ws_b)[:n].copy_(g * mem(w_b)[:n])


def icg_attn_gamma_bwd(gamma, dws, w, dw, dgamma, n):
    g = mem(gamma)[0]
    mem(dgamma)[0] = float((mem(dws)[:n].double() * mem(w)[:n].double()).sum())
    mem(dw)[:n].copy_(g * mem(dws)[:n])


def icg_softmax_fwd(x, y, rows, cols):
    mem(y)[: rows * cols].copy_(F.softmax(mem(x)[: rows * cols].view(rows, cols), -1).reshape(-1))


def icg_attn_scores_softmax_applies(n, m, d):
    return int(n >= 32 and n % 32 == 0 and m >= 128 and m % 128 == 0 and m <= 1024 and d in (8, 16, 24, 32, 48, 64))


def icg_attn_scores_softmax(theta, phi, beta, B, n, m, d):
    q = mem(theta)[: B * n * d].view(B, n, d)
    k = mem(phi)[: B * m * d].view(B, m, d)
    mem(beta)[: B * n * m].copy_(F.softmax(torch.bmm(q, k.transpose(1, 2)), -1).reshape(-1))


def icg_softmax_bwd(y, dy, dx, rows, cols):
    yy = mem(y)[: rows * cols].view(rows, cols)
    g = mem(dy)[: rows * cols].view(rows, cols)
    mem(dx)[: rows * cols].copy_((yy * (g - (yy * g).sum(-1, keepdim=True))).reshape(-1))


def icg_attn_dscores_applies(n, m, dv):
    return int(n >= 32 and n % 32 == 0 and m >= 128 and m % 128 == 0 and m <= 1024 and dv in (96, 192))


def icg_attn_dscores(dO, V, beta, dS, B, n, m, dv):
    do = mem(dO)[: B * n * dv].view(B, n, dv)
    v = mem(V)[: B * m * dv].view(B, m, dv)
    bt = mem(beta)[: B * n * m].view(B, n, m)
    g = torch.bmm(do, v.transpose(1, 2))
    mem(dS)[: B * n * m].copy_((bt * (g - (bt * g).sum(-1, keepdim=True))).reshape(-1))


def icg_relu_sumpool_fwd(x, y, B, HW, C):
    mem(y)[: B * C].copy_(F.relu(mem(x)[: B * HW * C].view(B, HW, C)).sum(1).reshape(-1))


def icg_relu_sumpool_bwd(x, dy, dx, B, HW, C):
    xv = mem(x)[: B * HW * C].view(B, HW, C)
    mem(dx)[: B * HW * C].copy_(((xv > 0) * mem(dy)[: B * C].view(B, 1, C)).reshape(-1))


def icg_scale_add_fwd(gamma, o, x, out, n):
    mem(out)[:n].copy_(gamma.view(-1)[0] * mem(o)[:n] + mem(x)[:n])


def icg_scale_add_bwd(gamma, o, dout, d_o, dgamma, n, scratch, scratch_bytes):
    mem(d_o)[:n].copy_(gamma.view(-1)[0] * mem(dout)[:n])
    dgamma.fill_(float((mem(dout)[:n].double() * mem(o)[:n].double()).sum()))


def icg_colsum_workspace_bytes(rows, C):
    return 16


def icg_colsum(x, rows, C, out, workspace, workspace_bytes):
    mem(out)[:C].copy_(mem(x)[: rows * C].view(rows, C).double().sum(0).float())


# ---------------------------------------------------------------- optimiser / EMA (tensor-level references of
# icg_adam_multi / icg_ema_multi; the C entry points take descriptor arrays of raw pointers)
def adam_multi_ref(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step):
    """torch.optim.Adam single-tensor rule (weight_decay=0, amsgrad=False)."""
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        m.lerp_(g, 1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)


def ema_multi_ref(targets, sources, decay):
    for t, s in zip(targets, sources):
        t.copy_(t * decay + s * (1 - decay))


# ---------------------------------------------------------------- StyleGAN2 ops
_ACTS = {1: lambda x, a: x, 2: lambda x, a: F.relu(x), 3: lambda x, a: F.leaky_relu(x, a), 4: lambda x, a: torch.tanh(x),
         5: lambda x, a: torch.sigmoid(x), 6: lambda x, a: F.elu(x), 7: lambda x, a: F.selu(x),
         8: lambda x, a: F.softplus(x), 9: lambda x, a: torch.sigmoid(x) * x}


def icg_bias_act(x, b, xref, yref, dy, y, n, step_b, size_b, grad, act, alpha, gain, clamp):
    """grad 0 only follows _bias_act_ref (bias_act.py:177-207); grad 1/2 are derived by autograd from it.  Works in the dtype
    of x (fp32 here; icg_bias_act_typed passes fp64 buffers for fp64 storage)."""
    xv = mem(x)[:n]
    ct = xv.dtype
    idx = (torch.arange(n) // step_b) % size_b if b is not None else None
    bias = mem(b)[idx] if b is not None else 0.0
    up = mem(dy)[:n] if dy is not None else 1.0

    def fwd(inp):
        o = _ACTS[act](inp + bias, alpha) * gain
        return o.clamp(-clamp, clamp) if clamp >= 0 else o

    if grad == 0:
        mem(y)[:n].copy_(fwd(xv) * up if dy is not None else fwd(xv))
        return
    xr = mem(xref)[:n].detach().clone().double().requires_grad_(True) if xref is not None else None
    if xr is None:       # piecewise-linear activations: the derivative is a function of the sign of y (bias_act.cu)
        names = {1: "linear", 2: "relu", 3: "lrelu"}
        if act not in names or yref is None and (act != 1 or clamp >= 0):
            raise NotImplementedError("kernel_ref.icg_bias_act(grad>0) without xref: linear / relu / lrelu only")
        if grad == 2:
            mem(y)[:n].zero_()
            return
        yv = mem(yref)[:n] if yref is not None else None
        one = torch.ones(n, dtype=ct)
        slope = one if act == 1 else torch.where(yv > 0, one, torch.full((n,), alpha if act == 3 else 0.0, dtype=ct))
        g = xv * slope * gain
        if clamp >= 0:
            g = torch.where(yv.abs() < clamp, g, torch.zeros_like(g))
        mem(y)[:n].copy_(g)
        return
    bias = bias.double() if b is not None else 0.0
    with torch.enable_grad():
        o = fwd(xr)
        (g1,) = torch.autograd.grad(o.sum(), xr, create_graph=True)
        if grad == 1:
            mem(y)[:n].copy_((xv.double() * g1 * up).detach().to(ct))
        else:
            g2 = None
            if g1.requires_grad:
                (g2,) = torch.autograd.grad(g1.sum(), xr, allow_unused=True)
            g2 = torch.zeros_like(xr) if g2 is None else g2
            mem(y)[:n].copy_((xv.double() * g2 * up).detach().to(ct))


_DT = {0: torch.float32, 1: torch.float16, 2: torch.float64}


def _up(t, dtype):
    """storage dtype -> the plugin's internal compute type (bias_act.cu:18-21: half -> float)"""
    if t is None:
        return None
    return t.double() if dtype == 2 else t.float()


def icg_colsum_f16_applies(C):
    v = C // 8
    return int(C >= 8 and C % 8 == 0 and v <= 256 and (v & (v - 1)) == 0)


def icg_colsum_f16_workspace_bytes(rows, C):
    return 16


def icg_colsum_f16(x, rows, C, out, workspace, workspace_bytes):
    assert x.dtype == torch.float16 and out.dtype == torch.float32
    mem(out)[:C].copy_(mem(x)[: rows * C].view(rows, C).double().sum(0).float())


def icg_bias_act_typed(x, b, xref, yref, dy, y, n, step_b, size_b, grad, act, alpha, gain, clamp, dtype):
    """fp16 / fp64 storage: compute in the internal type, round once into y (bias_act.cu:26-150)."""
    if dtype == 0:
        return icg_bias_act(x, b, xref, yref, dy, y, n, step_b, size_b, grad, act, alpha, gain, clamp)
    assert x.dtype == _DT[dtype] and y.dtype == _DT[dtype]
    if dtype == 2:          # fp64 storage = fp64 arithmetic: the same routine on the fp64 buffers
        return icg_bias_act(x, b, xref, yref, dy, y, n, step_b, size_b, grad, act, alpha, gain, clamp)
    tmp = torch.empty(n, dtype=torch.float32)
    f = lambda t: None if t is None else mem(t)[: (size_b if t is b else n)].float()
    icg_bias_act(f(x), f(b), f(xref), f(yref), f(dy), tmp, n, step_b, size_b, grad, act, alpha, gain, clamp)
    mem(y)[:n].copy_(tmp.to(_DT[dtype]))


def icg_upfirdn2d_typed(x, f, y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, outH,
                        outW, dtype, channels_last):
    """accumulate in the internal type, round once (upfirdn2d.cu:32-95); channels_last: x / y are [N][H][W][C] in memory"""
    ct = torch.float64 if dtype == 2 else torch.float32
    xin = mem(x)[: N * C * H * W]
    xin = xin.view(N, H, W, C).permute(0, 3, 1, 2).contiguous() if channels_last else xin.view(N, C, H, W)
    tmp = torch.empty(N * C * outH * outW, dtype=ct)
    if dtype == 2:
        icg_upfirdn2d(xin.double().contiguous(), f, tmp, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
                      flip, gain, outH, outW)
    else:
        icg_upfirdn2d(xin.float().contiguous(), f, tmp, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1,
                      flip, gain, outH, outW)
    out = tmp.view(N, C, outH, outW)
    if channels_last:
        out = out.permute(0, 2, 3, 1).contiguous()
    mem(y)[: N * C * outH * outW].copy_(out.reshape(-1).to(_DT[dtype]))


def icg_upfirdn2d(x, f, y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, outH, outW):
    """_upfirdn2d_ref (upfirdn2d.py:199-246) restated for a 2-D filter."""
    xv = mem(x)[: N * C * H * W].view(N, C, H, 1, W, 1)
    xv = F.pad(xv, [0, upx - 1, 0, 0, 0, upy - 1]).reshape(N, C, H * upy, W * upx)
    xv = F.pad(xv, [max(padx0, 0), max(padx1, 0), max(pady0, 0), max(pady1, 0)])
    xv = xv[:, :, max(-pady0, 0): xv.shape[2] - max(-pady1, 0), max(-padx0, 0): xv.shape[3] - max(-padx1, 0)]
    ff = mem(f)[: fh * fw].view(fh, fw).to(xv.dtype) * gain
    if not flip:
        ff = ff.flip([0, 1])
    out = F.conv2d(xv, ff[None, None].repeat(C, 1, 1, 1), groups=C)[:, :, ::downy, ::downx]
    assert out.shape[2] == outH and out.shape[3] == outW, (out.shape, outH, outW)
    mem(y)[: N * C * outH * outW].copy_(out.reshape(-1))


def icg_upfirdn2d_nhwc(x, f, y, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, outH, outW):
    xn = _nhwc(x, N, H, W, C).permute(0, 3, 1, 2).contiguous()
    yn = torch.empty(N, C, outH, outW)
    icg_upfirdn2d(xn, f, yn, N, C, H, W, fh, fw, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain, outH, outW)
    mem(y)[: N * C * outH * outW].copy_(yn.permute(0, 2, 3, 1).reshape(-1))


def _gather_src(x, B, Hin, Win, Cin, zero_insert):
    a = _nhwc(x, B, Hin, Win, Cin).permute(0, 3, 1, 2)
    if zero_insert:
        z = zero_insert
        up = torch.zeros(B, Cin, (Hin - 1) * z + 1, (Win - 1) * z + 1, dtype=a.dtype)
        up[:, :, ::z, ::z] = a
        a = up
    return a


def _fit(y, Hout, Wout):
    """crop / zero-extend the conv output to the requested grid (positions beyond the source read zeros)."""
    y = y[:, :, :Hout, :Wout]
    return F.pad(y, (0, Wout - y.shape[3], 0, Hout - y.shape[2]))


def icg_conv2d_g_fprop(x, w, bias, out, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert):
    a = _gather_src(x, B, Hin, Win, Cin, zero_insert)
    wt = mem(w)[: Cout * R * R * Cin].view(Cout, R, R, Cin).permute(0, 3, 1, 2)
    # enough trailing zeros that every requested output position exists
    need_h = (Hout - 1) * stride + R - pad - a.shape[2]
    need_w = (Wout - 1) * stride + R - pad - a.shape[3]
    a = F.pad(a, (pad, max(need_w, 0), pad, max(need_h, 0)))
    y = _fit(F.conv2d(a, wt, None, stride, 0), Hout, Wout)
    if bias is not None:
        y = y + mem(bias)[:Cout].view(1, -1, 1, 1)
    mem(out)[: B * Hout * Wout * Cout].copy_(y.permute(0, 2, 3, 1).reshape(-1))


def icg_conv2d_g_fprop_f16_applies(Cin, Cout, R, stride, zero_insert):
    return int(Cin >= 32 and Cin % 32 == 0 and Cout >= 64 and (Cout % 64 == 0 or Cout % 96 == 0) and 1 <= R <= 7 and stride >= 1
               and zero_insert in (0, 1, 2) and not (zero_insert == 2 and stride != 1))


def icg_conv2d_g_fprop_f16(x, w, out, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert):
    """fp16 storage, exact products, wide accumulation, one rounding (the kernel accumulates in fp32; fp64 here)."""
    assert x.dtype == torch.float16 and w.dtype == torch.float16 and out.dtype == torch.float16
    y = torch.empty(B * Hout * Wout * Cout, dtype=torch.float64)
    icg_conv2d_g_fprop(x.double(), w.double(), None, y, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert)
    mem(out)[: y.numel()].copy_(y.to(torch.float16))


def icg_conv2d_g_fprop_workspace_bytes(B, Hout, Wout, Cin, Cout, R, zero_insert):
    return 0


def icg_conv2d_g_fprop_ws(x, w, bias, out, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert, workspace,
                          workspace_bytes):
    icg_conv2d_g_fprop(x, w, bias, out, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert)


def icg_conv2d_tr2_fprop(x, wp, bias, out, B, Hin, Win, Cin, Hout, Wout, Cout):
    a = _nhwc(x, B, Hin, Win, Cin).permute(0, 3, 1, 2)
    w = mem(wp)[: 16 * Cout * Cin].view(2, 2, Cout, 2, 2, Cin)
    big = torch.zeros(B, Cout, 2 * Hin + 2, 2 * Win + 2)
    for al in range(2):
        for be in range(2):
            ap = F.pad(a, (1 - be, 1 + be, 1 - al, 1 + al))          # rows m-1+al+u for m in [0, Hin], u in {0,1}
            big[:, :, al::2, be::2] = F.conv2d(ap, w[al, be].permute(0, 3, 1, 2))
    y = big[:, :, :Hout, :Wout]
    if bias is not None:
        y = y + mem(bias)[:Cout].view(1, -1, 1, 1)
    mem(out)[: B * Hout * Wout * Cout].copy_(y.permute(0, 2, 3, 1).reshape(-1))


def icg_conv2d_g_wgrad_f16_applies(Cin, Cout, R, stride):
    return int(Cin >= 32 and Cin % 32 == 0 and Cout >= 32 and Cout % 32 == 0 and 1 <= R <= 7 and 1 <= stride <= 4)


def icg_conv2d_g_wgrad_f16_workspace_bytes(B, Hout, Wout, Cin, Cout, R):
    return 16


def icg_conv2d_g_wgrad_f16(x, dy, dw, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, workspace, workspace_bytes):
    assert x.dtype == torch.float16 and dy.dtype == torch.float16 and dw.dtype == torch.float32
    icg_conv2d_g_wgrad(x.float(), dy.float(), dw, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, workspace, workspace_bytes)


def icg_conv2d_g_wgrad_workspace_bytes(B, Hout, Wout, Cin, Cout, R):
    return 16


def icg_conv2d_g_wgrad(x, dy, dw, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, workspace, workspace_bytes):
    a = _nhwc(x, B, Hin, Win, Cin).double()
    g = _nhwc(dy, B, Hout, Wout, Cout).double()
    need_h = (Hout - 1) * stride + R - pad - Hin
    need_w = (Wout - 1) * stride + R - pad - Win
    ap = F.pad(a, (0, 0, pad, max(need_w, 0), pad, max(need_h, 0)))
    o = torch.empty(R, R, Cin, Cout, dtype=torch.float64)
    for r in range(R):
        for s_ in range(R):
            win = ap[:, r: r + (Hout - 1) * stride + 1: stride, s_: s_ + (Wout - 1) * stride + 1: stride, :]
            o[r, s_] = torch.einsum("bhwi,bhwo->io", win, g)
    mem(dw)[: R * R * Cin * Cout].copy_(o.float().reshape(-1))


# ---------------------------------------------------------------- host-logic test harness
def icg_knn_l2_workspace_bytes(N, D):
    return 16


def icg_knn_l2(feats, N, D, k, idx, d2, workspace, workspace_bytes):
    """exact L2 top-k of every row against the table (fp64 distances; stable sort = ties to the lower index; self first)"""
    f = mem(feats)[: N * D].view(N, D).double()
    sq = (f * f).sum(1)
    dist = (sq[:, None] + sq[None, :] - 2.0 * (f @ f.t())).clamp_(min=0)
    dist[torch.arange(N), torch.arange(N)] = -1.0
    order = torch.sort(dist, dim=1, stable=True)
    idx.view(-1)[: N * k].copy_(order.indices[:, :k].reshape(-1))
    mem(d2)[: N * k].copy_(order.values[:, :k].clamp(min=0).float().reshape(-1))


# ---------------------------------------------------------------- fused StyleGAN2 layers (csrc/sg2_fused.hip)
def _rt(v, dtype):
    """round to the storage type and come back to fp32 (dtype 1 = fp16)"""
    return v.half().float() if dtype == 1 else v


def sg2_weight_prep_ref(items):
    """ops.sg2_weight_prep_multi (icg_sg2_weight_prep_multi): items = dicts with w, w_fwd, w_adj, wsq, wscale, warg, prenorm, gain, flip."""
    for it in items:
        w = it["w"].detach().float()
        O, I, R, _ = w.shape
        if it["prenorm"]:
            flat = w.reshape(O, -1).abs()
            m, arg = flat.max(dim=1)
            scale = (1.0 / m) * torch.tensor(it["gain"], dtype=torch.float32)
            it["warg"].copy_(arg.to(torch.int32))
        else:
            scale = torch.full((O,), it["gain"], dtype=torch.float32)
        it["wscale"].copy_(scale)
        wn = w * scale.view(O, 1, 1, 1)
        if it["wsq"] is not None:
            it["wsq"].copy_(wn.square().sum(dim=[2, 3]))
        g = wn.permute(0, 2, 3, 1)                         # [O][R][R][I]
        if it["flip"]:
            g = g.flip(1, 2)
        mem(it["w_fwd"])[: g.numel()].copy_(g.reshape(-1).to(it["w_fwd"].dtype))
        if it["w_adj"] is not None:
            a = g.flip(1, 2).permute(3, 1, 2, 0)           # [I][R][R][O], taps reversed
            mem(it["w_adj"])[: a.numel()].copy_(a.reshape(-1).to(it["w_adj"].dtype))


def icg_sg2_style_prep(lin, bias, bias_gain, post_gain, wsq, N, I, O, prenorm, s, smax, sarg, d):
    s0 = mem(lin)[: N * I].view(N, I)
    if bias is not None:
        s0 = s0 + mem(bias)[:I] * bias_gain
    s0 = s0 * post_gain
    sv = s0
    if prenorm:
        m, arg = s0.abs().max(dim=1)
        mem(smax)[:N].copy_(s0.gather(1, arg[:, None])[:, 0])
        mem(sarg)[:N].copy_(arg.to(torch.int32))
        sv = s0 / m[:, None]
    mem(s)[: N * I].copy_(sv.reshape(-1))
    if wsq is not None:
        q = sv.square() @ mem(wsq)[: O * I].view(O, I).t()
        mem(d)[: N * O].copy_((q + 1e-8).rsqrt().reshape(-1))


def icg_sg2_rows_applies(C, dtype):
    vec = 8 if dtype == 1 else 4
    if dtype not in (0, 1) or C < vec or C % vec:
        return 0
    v = C // vec
    return int(v <= 256 and (v & (v - 1)) == 0)


def icg_sg2_modulate(x, s, xs, N, HW, C, dtype):
    xv = mem(x)[: N * HW * C].view(N, HW, C).float()
    sv = _rt(mem(s)[: N * C].view(N, 1, C), dtype)
    mem(xs)[: N * HW * C].copy_((xv * sv).reshape(-1).to(xs.dtype))


def _sg2_act(v, act, alpha):
    return torch.where(v < 0, v * alpha, v) if act == 3 else v


def icg_sg2_act_fwd(c, d, noise, noise_bstride, strength, bias, y, N, HW, O, act, alpha, gain, clamp, dtype):
    z = mem(c)[: N * HW * O].view(N, HW, O).float()
    nz = None
    if noise is not None:
        st = mem(strength)[0] if strength is not None else 1.0
        nb = mem(noise)
        nz = torch.stack([nb[n * noise_bstride: n * noise_bstride + HW] for n in range(N)]) * st
        nz = _rt(nz, dtype).view(N, HW, 1)
    if d is not None:
        dv = _rt(mem(d)[: N * O].view(N, 1, O), dtype)
        z = _rt(z * dv + (nz if nz is not None else 0.0), dtype)
    elif nz is not None:
        z = _rt(z + nz, dtype)
    if bias is not None:
        z = z + _rt(mem(bias)[:O], dtype)
    o = _sg2_act(z, act, alpha) * gain
    if clamp >= 0:
        o = o.clamp(-clamp, clamp)
    mem(y)[: N * HW * O].copy_(o.reshape(-1).to(y.dtype))


def icg_sg2_fir_act_fwd(x, f, c, y, d, noise, noise_bstride, strength, bias, N, C, H, W, fh, fw, padx0, padx1, pady0, pady1, flip, fgain, outH,
                        outW, act, alpha, gain, clamp, dtype):
    """icg_upfirdn2d_typed (up = down = 1, channels-last) followed by icg_sg2_act_fwd"""
    tmp = torch.empty(N * outH * outW * C, dtype=x.dtype)
    if dtype == 1:
        icg_upfirdn2d_typed(x, f, tmp, N, C, H, W, fh, fw, 1, 1, 1, 1, padx0, padx1, pady0, pady1, flip, fgain, outH, outW, 1, 1)
    else:
        icg_upfirdn2d_nhwc(x, f, tmp, N, C, H, W, fh, fw, 1, 1, 1, 1, padx0, padx1, pady0, pady1, flip, fgain, outH, outW)
    if c is not None:
        mem(c)[: tmp.numel()].copy_(tmp)
    icg_sg2_act_fwd(tmp, d, noise, noise_bstride, strength, bias, y, N, outH * outW, C, act, alpha, gain, clamp, dtype)


def icg_conv2d_g_fprop_f16_act(x, w, c, y, d, noise, noise_bstride, strength, bias, act, alpha, gain, clamp, B, Hin, Win, Cin, Hout, Wout, Cout,
                               R, stride, pad):
    """icg_conv2d_g_fprop_f16 followed by icg_sg2_act_fwd"""
    tmp = torch.empty(B * Hout * Wout * Cout, dtype=torch.float16)
    icg_conv2d_g_fprop_f16(x, w, tmp, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, 0)
    if c is not None:
        mem(c)[: tmp.numel()].copy_(tmp)
    icg_sg2_act_fwd(tmp, d, noise, noise_bstride, strength, bias, y, B, Hout * Wout, Cout, act, alpha, gain, clamp, 1)


def icg_modconv2d_f16_applies(Cin, Cout, R, stride, zero_insert, Hout, Wout):
    px = (Hout // 2) * (Wout // 2) if zero_insert == 2 else Hout * Wout
    return int(bool(icg_conv2d_g_fprop_f16_applies(Cin, Cout, R, stride, zero_insert)) and Cin <= 1024 and px >= 127)


def icg_modconv2d_f16(x, style, w, c, y, d, noise, noise_bstride, strength, bias, act, alpha, gain, clamp, B, Hin, Win, Cin, Hout, Wout, Cout,
                      R, stride, pad, zero_insert):
    """icg_sg2_modulate, icg_conv2d_g_fprop_f16 and (y given) icg_sg2_act_fwd"""
    xs = x
    if style is not None:
        xs = torch.empty(B * Hin * Win * Cin, dtype=torch.float16)
        icg_sg2_modulate(x, style, xs, B, Hin * Win, Cin, 1)
    tmp = torch.empty(B * Hout * Wout * Cout, dtype=torch.float16)
    icg_conv2d_g_fprop_f16(xs, w, tmp, B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert)
    if c is not None:
        mem(c)[: tmp.numel()].copy_(tmp)
    if y is not None:
        icg_sg2_act_fwd(tmp, d, noise, noise_bstride, strength, bias, y, B, Hout * Wout, Cout, act, alpha, gain, clamp, 1)


def _rows_geometry(HW, V):
    nrl = 256 // V
    r = max(-(-HW // 64), nrl)
    r = -(-r // nrl) * nrl
    return r, -(-HW // r)


def icg_sg2_rows_workspace_bytes(N, HW, C, ncols, dtype):
    if not icg_sg2_rows_applies(C, dtype):
        return 0
    _, chunks = _rows_geometry(HW, C // (8 if dtype == 1 else 4))
    return N * chunks * ncols * 4


def icg_sg2_act_bwd(dy, y, c, d, noise, noise_bstride, dc, sums, tot, N, HW, O, act, alpha, gain, clamp, dtype, workspace, workspace_bytes):
    g = mem(dy)[: N * HW * O].view(N, HW, O).float()
    yv = mem(y)[: N * HW * O].view(N, HW, O).float()
    slope = torch.where(yv > 0, torch.ones_like(yv), torch.full_like(yv, alpha)) if act == 3 else torch.ones_like(yv)
    dz = g * (gain * slope)
    if clamp >= 0:
        dz = torch.where((yv > -clamp) & (yv < clamp), dz, torch.zeros_like(dz))
    dz = _rt(dz, dtype)
    out = torch.zeros(N, 2 * O + 1, dtype=torch.float64)
    out[:, :O] = dz.double().sum(1)
    if c is not None:
        out[:, O:2 * O] = (dz.double() * mem(c)[: N * HW * O].view(N, HW, O).double()).sum(1)
    if noise is not None:
        nb = mem(noise)
        nz = torch.stack([nb[n * noise_bstride: n * noise_bstride + HW] for n in range(N)]).double()
        out[:, 2 * O] = (dz.double().sum(2) * nz).sum(1)
    if sums is not None:
        mem(sums)[: N * (2 * O + 1)].copy_(out.float().reshape(-1))
    if tot is not None:
        mem(tot)[: 2 * O + 1].copy_(out.sum(0).float())
    if dc is not None:
        o = _rt(dz * _rt(mem(d)[: N * O].view(N, 1, O), dtype), dtype) if d is not None else dz
        mem(dc)[: N * HW * O].copy_(o.reshape(-1).to(dc.dtype))


def icg_sg2_modulate_bwd(dxs, x, s, dx, ds, N, HW, C, dtype, workspace, workspace_bytes):
    g = mem(dxs)[: N * HW * C].view(N, HW, C).float()
    xv = mem(x)[: N * HW * C].view(N, HW, C).float()
    mem(ds)[: N * C].copy_((g.double() * xv.double()).sum(1).float().reshape(-1))
    if dx is not None:
        sv = _rt(mem(s)[: N * C].view(N, 1, C), dtype)
        mem(dx)[: N * HW * C].copy_((g * sv).reshape(-1).to(dx.dtype))


def _raw(t):
    """the memory behind a pointer argument that is a strided view (row n at p + n * stride): flat from the view's first element"""
    n = t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()
    return t.as_strided((n,), (1,))


def icg_sg2_mod2(x, a, g, b, u, N, HW, C, dtype):
    o = _rt(mem(x)[: N * HW * C].view(N, HW, C).float() * _rt(mem(a)[: N * C].view(N, 1, C), dtype), dtype)
    if g is not None:
        o = o + _rt(mem(g)[: N * HW * C].view(N, HW, C).float() * _rt(mem(b)[: N * C].view(N, 1, C), dtype), dtype)
    mem(u)[: N * HW * C].copy_(o.reshape(-1).to(u.dtype))


def icg_sg2_act_bwd2(dy, y, c, cdc, d, cdd, cdy, cc, sums, N, HW, O, act, alpha, gain, clamp, dtype, workspace, workspace_bytes):
    g = mem(dy)[: N * HW * O].view(N, HW, O).float()
    yv = mem(y)[: N * HW * O].view(N, HW, O).float()
    m = gain * (torch.where(yv > 0, torch.ones_like(yv), torch.full_like(yv, alpha)) if act == 3 else torch.ones_like(yv))
    if clamp >= 0:
        m = torch.where((yv > -clamp) & (yv < clamp), m, torch.zeros_like(m))
    dz = _rt(g * m, dtype)
    kv = mem(cdc)[: N * HW * O].view(N, HW, O).float()
    dv = _rt(mem(d)[: N * O].view(N, 1, O), dtype) if d is not None else 1.0
    o1 = kv * dv
    if cdd is not None:
        ev = mem(cdd)[: N * O].view(N, 1, O)
        o1 = o1 + ev * mem(c)[: N * HW * O].view(N, HW, O).float()
        if cc is not None:
            mem(cc)[: N * HW * O].copy_((ev * dz).reshape(-1).to(cc.dtype))
    elif cc is not None:
        mem(cc)[: N * HW * O].zero_()
    mem(cdy)[: N * HW * O].copy_((o1 * m).reshape(-1).to(cdy.dtype))
    mem(sums)[: N * O].copy_((kv.double() * dz.double()).sum(1).float().reshape(-1))


def icg_sg2_weight_bwd_q(dw_conv, layout, t, s, N, Q, w, wscale, warg, prenorm, c0, round_f16, dw, O, I, R, workspace, workspace_bytes):
    RR = R * R
    dc = mem(dw_conv)[: RR * I * O]
    dc = dc.view(RR, I, O).permute(2, 1, 0) if layout == 0 else dc.view(RR, O, I).permute(1, 2, 0)      # -> [O][I][RR]
    if round_f16:
        dc = dc.half().float()
    wv = mem(w)[: O * I * RR].view(O, I, RR)
    sc = mem(wscale)[:O].view(O, 1, 1)
    g = dc
    if t is not None:
        q = mem(t)[: N * O].view(N, O).t() @ mem(s)[: N * I].view(N, I).square()                 # [O][I]
        g = g + wv * sc * q[:, :, None]
    if Q is not None:
        g = g + wv * sc * mem(Q)[: O * I].view(O, I)[:, :, None]
    out = g * sc
    if prenorm:
        D = (g * wv).sum(dim=[1, 2])
        arg = mem(warg)[:O].long()
        flat = out.reshape(O, -1).clone()
        wa = wv.reshape(O, -1)[torch.arange(O), arg]
        flat[torch.arange(O), arg] -= torch.sign(wa) * D * sc.view(O) * sc.view(O) / c0
        out = flat
    mem(dw)[: O * I * RR].copy_(out.reshape(-1))


def icg_sg2_torgb_bwd2(dimg, y, x, s, w, a, cdx, cim, clamp, mask_clamp, cdimg, cx, sums, tot, N, HW, C, dtype, workspace, workspace_bytes):
    mk = torch.ones(N, HW, 3)
    if mask_clamp and clamp >= 0:
        yv = mem(y)[: N * HW * 3].view(N, HW, 3).float()
        mk = ((yv > -clamp) & (yv < clamp)).float()
    dz = _rt(mem(dimg)[: N * 3 * HW].view(N, 3, HW).permute(0, 2, 1), dtype) * mk                 # [N][HW][3]
    xv = mem(x)[: N * HW * C].view(N, HW, C).float()
    sv = _rt(mem(s)[: N * C].view(N, 1, C), dtype)
    wv = _rt(mem(w)[: 3 * C].view(3, C), dtype)
    av = _rt(mem(a)[: N * C].view(N, 1, C), dtype)
    dxs = _rt(dz @ wv, dtype)
    u = _rt(xv * av, dtype)
    per = torch.zeros(N, 4 * C, dtype=torch.float64)
    if cdx is not None:
        gv = mem(cdx)[: N * HW * C].view(N, HW, C).float()
        u = _rt(u + _rt(gv * sv, dtype), dtype)
        per[:, :C] = (gv.double() * dxs.double()).sum(1)
    per[:, C:] = torch.einsum("npo,npc->noc", dz.double(), u.double()).reshape(N, 3 * C)
    out = (u @ wv.t()) * mk                                                                           # [N][HW][3]
    out = out.permute(0, 2, 1)
    if cim is not None:
        out = out + mem(cim)[: N * 3 * HW].view(N, 3, HW)
    mem(cdimg)[: N * 3 * HW].copy_(out.reshape(-1))
    if cx is not None:
        mem(cx)[: N * HW * C].copy_((dxs * av).reshape(-1).to(cx.dtype))
    mem(sums)[: N * 4 * C].copy_(per.float().reshape(-1))
    mem(tot)[: 4 * C].copy_(per.sum(0).float())


def icg_sg2_style_bwd(ds_mod, ds_stride, dd, dd_stride, d, s, wsq, N, I, O, g, pdot, t):
    dsm = torch.stack([_raw(ds_mod)[n * ds_stride: n * ds_stride + I] for n in range(N)])
    sv = mem(s)[: N * I].view(N, I)
    gv = dsm
    if dd is not None:
        ddv = torch.stack([_raw(dd)[n * dd_stride: n * dd_stride + O] for n in range(N)])
        dv = mem(d)[: N * O].view(N, O)
        tv = -ddv * dv * dv * dv
        mem(t)[: N * O].copy_(tv.reshape(-1))
        gv = dsm + sv * (tv @ mem(wsq)[: O * I].view(O, I))
    mem(g)[: N * I].copy_(gv.reshape(-1))
    nb = -(-I // 64)
    pad = F.pad(gv * sv, (0, nb * 64 - I)).view(N, nb, 64).sum(2)
    mem(pdot)[: N * nb].copy_(pad.reshape(-1))


def _sg2_dlin(g, smax, sarg, pdot, npdot, post_gain, N, I):
    gv = mem(g)[: N * I].view(N, I).clone()
    if smax is not None:
        sm = mem(smax)[:N]
        P = mem(pdot)[: N * npdot].view(N, npdot).sum(1)
        arg = mem(sarg)[:N].long()
        gv[torch.arange(N), arg] -= torch.sign(sm) * P
        gv = gv / sm.abs()[:, None]
    return gv * post_gain


def icg_sg2_fc_bwd(g, smax, sarg, pdot, npdot, post_gain, x, W, N, I, K, wgain, bias_gain, dW, db, dx):
    dl = _sg2_dlin(g, smax, sarg, pdot, npdot, post_gain, N, I)
    if dW is not None:
        mem(dW)[: I * K].copy_((dl.t() @ mem(x)[: N * K].view(N, K) * wgain).reshape(-1))
    if db is not None:
        mem(db)[:I].copy_(dl.sum(0) * bias_gain)
    if dx is not None:
        mem(dx)[: N * K].copy_((dl @ mem(W)[: I * K].view(I, K) * wgain).reshape(-1))


def icg_sg2_weight_bwd_workspace_bytes(O, I):
    return O * (-(-I // 32)) * 4


def icg_sg2_weight_bwd(dw_conv, layout, t, s, N, w, wscale, warg, prenorm, c0, round_f16, dw, O, I, R, workspace, workspace_bytes):
    icg_sg2_weight_bwd_q(dw_conv, layout, t, s, N, None, w, wscale, warg, prenorm, c0, round_f16, dw, O, I, R, workspace, workspace_bytes)


def icg_sg2_fromrgb_applies(O, dtype):
    return int(bool(icg_sg2_rows_applies(O, dtype)) and O // (8 if dtype == 1 else 4) <= 64)


def icg_sg2_fromrgb_fwd(x, w, bias, y, N, HW, O, act, alpha, gain, clamp, dtype):
    xv = mem(x)[: N * 3 * HW].view(N, 3, HW).float()
    wv = mem(w)[: O * 3].view(O, 3).float()
    a = _rt(torch.einsum("ncp,oc->npo", xv, wv), dtype)
    if bias is not None:
        a = a + _rt(mem(bias)[:O], dtype)
    o = _sg2_act(a, act, alpha) * gain
    if clamp >= 0:
        o = o.clamp(-clamp, clamp)
    mem(y)[: N * HW * O].copy_(o.reshape(-1).to(y.dtype))


def icg_sg2_fromrgb_bwd(dy, y, x, w, dimg, tot, N, HW, O, act, alpha, gain, clamp, dtype, workspace, workspace_bytes):
    g = mem(dy)[: N * HW * O].view(N, HW, O).float()
    yv = mem(y)[: N * HW * O].view(N, HW, O).float()
    slope = torch.where(yv > 0, torch.ones_like(yv), torch.full_like(yv, alpha)) if act == 3 else torch.ones_like(yv)
    dz = g * (gain * slope)
    if clamp >= 0:
        dz = torch.where((yv > -clamp) & (yv < clamp), dz, torch.zeros_like(dz))
    dz = _rt(dz, dtype).double()
    xv = mem(x)[: N * 3 * HW].view(N, 3, HW).double()
    out = torch.zeros(O, 4, dtype=torch.float64)
    out[:, :3] = torch.einsum("npo,ncp->oc", dz, xv)
    out[:, 3] = dz.sum(dim=[0, 1])
    mem(tot)[: 4 * O].copy_(out.float().reshape(-1))
    if dimg is not None:
        di = torch.einsum("npo,oc->ncp", dz, mem(w)[: O * 3].view(O, 3).double())
        mem(dimg)[: N * 3 * HW].copy_(di.float().reshape(-1).to(dimg.dtype))


def icg_sg2_torgb_applies(C, dtype):
    vec = 8 if dtype == 1 else 4
    if dtype not in (0, 1) or C < vec or C % vec:
        return 0
    v = C // vec
    return int((v & (v - 1)) == 0 and (v <= 64 or v in (128, 256)))


def icg_sg2_torgb_fwd(x, s, w, bias, clamp, img_in, img_out, y, N, HW, C, dtype):
    xv = mem(x)[: N * HW * C].view(N, HW, C).float()
    xs = _rt(xv * _rt(mem(s)[: N * C].view(N, 1, C), dtype), dtype)
    o = _rt(xs @ _rt(mem(w)[: 3 * C].view(3, C), dtype).t(), dtype)
    if bias is not None:
        o = o + _rt(mem(bias)[:3], dtype)
    if clamp >= 0:
        o = o.clamp(-clamp, clamp)
    o = _rt(o, dtype)
    mem(y)[: N * HW * 3].copy_(o.reshape(-1).to(y.dtype))
    im = o.permute(0, 2, 1)                                # [N][3][HW]
    if img_in is not None:
        im = im + mem(img_in)[: N * 3 * HW].view(N, 3, HW)
    mem(img_out)[: N * 3 * HW].copy_(im.reshape(-1))


def icg_sg2_torgb_bwd_workspace_bytes(N, HW, C, dtype):
    return 16 if icg_sg2_torgb_applies(C, dtype) else 0


def icg_sg2_torgb_bwd(dimg, y, x, s, w, clamp, mask_clamp, dx, sums, tot, N, HW, C, dtype, workspace, workspace_bytes):
    dz = _rt(mem(dimg)[: N * 3 * HW].view(N, 3, HW).permute(0, 2, 1), dtype)                      # [N][HW][3]
    if mask_clamp and clamp >= 0:
        yv = mem(y)[: N * HW * 3].view(N, HW, 3).float()
        dz = torch.where((yv > -clamp) & (yv < clamp), dz, torch.zeros_like(dz))
    xv = mem(x)[: N * HW * C].view(N, HW, C).float()
    sv = _rt(mem(s)[: N * C].view(N, 1, C), dtype)
    wv = _rt(mem(w)[: 3 * C].view(3, C), dtype)
    dxs = _rt(dz @ wv, dtype)
    xs = _rt(xv * sv, dtype)
    per = torch.zeros(N, 4 * C + 3, dtype=torch.float64)
    per[:, :C] = (dxs.double() * xv.double()).sum(1)
    per[:, C:4 * C] = torch.einsum("npo,npc->noc", dz.double(), xs.double()).reshape(N, 3 * C)
    per[:, 4 * C:] = dz.double().sum(1)
    mem(sums)[: N * (4 * C + 3)].copy_(per.float().reshape(-1))
    mem(tot)[: 4 * C + 3].copy_(per.sum(0).float())
    if dx is not None:
        mem(dx)[: N * HW * C].copy_(_rt(dxs * sv, dtype).reshape(-1).to(dx.dtype))


def nan_to_num_multi_ref(tensors, nan=0.0, posinf=None, neginf=None):
    """ops.nan_to_num_multi (icg_nan_to_num_multi) as the per-tensor torch call it batches (training_loop.py:511-515)"""
    for t in tensors:
        if t is not None and t.numel():
            torch.nan_to_num(t, nan=nan, posinf=posinf, neginf=neginf, out=t)


def install(monkeypatch):
    """Route ic_gan_amd._lib.call / query to this module (CPU host-logic tests only)."""
    import ic_gan_amd._lib as L
    import ic_gan_amd.ops as ops
    g = globals()

    def call(name, *args):
        g[name](*args)

    def query(name, *args):
        return int(g[name](*args))

    monkeypatch.setattr(L, "call", call)
    monkeypatch.setattr(L, "query", query)
    monkeypatch.setattr(ops, "_require_gpu", lambda t: None)
    monkeypatch.setattr(ops, "adam_multi", adam_multi_ref)
    monkeypatch.setattr(ops, "ema_multi", ema_multi_ref)
    monkeypatch.setattr(ops, "nan_to_num_multi", nan_to_num_multi_ref)
    monkeypatch.setattr(ops, "sg2_weight_prep_multi", sg2_weight_prep_ref)
    import ic_gan_amd.stylegan_ops.fused_layers as FL
    monkeypatch.setattr(FL, "_EMULATED", True)
    def sn_backward_many_ref(items):
        out = []
        for raw, form, sn, like in items:
            forms = [None] * 4
            forms[form] = raw.contiguous()
            out.append(ops._sn_backward(forms[0], forms[1], sn, like, dw_up=forms[2], dw_down=forms[3]))      # (sn.handle is None here:
            #                                                          the node holds handle-free copies of the states)
        return out

    def linear_group_ref(mode, M, K, items):
        """ops._linear_group (icg_linear_group): the per-item dense products it batches"""
        if mode == 2:
            acc = torch.zeros(M, K)
            for _, w, dy, _, n in items:
                acc += mem(dy)[: M * n].view(M, n) @ mem(w)[: K * n].view(K, n).t()
            mem(items[0][3])[: M * K].copy_(acc.reshape(-1))
            return
        for x, w, dy, out, n in items:
            xv = mem(x)[: M * K].view(M, K)
            if mode == 0:
                mem(out)[: M * n].copy_((xv @ mem(w)[: n * K].view(n, K).t()).reshape(-1))
            else:
                mem(out)[: K * n].copy_((xv.t() @ mem(dy)[: M * n].view(M, n)).reshape(-1))

    monkeypatch.setattr(ops, "_linear_group", linear_group_ref)
    monkeypatch.setattr(ops, "sn_backward_many", sn_backward_many_ref)
    monkeypatch.setattr(ops, "sn_prepare_many", lambda items, eps, training: [
        ops.sn_prepare(w, u, sv, eps, training, nd, up, dn, *rest) for (w, u, sv, nd, up, dn, *rest) in items])
