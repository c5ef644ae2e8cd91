"""Helpers for the -m gpu parity tests: build one-op engine programs around torch tensors."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from yololite_amd.nn.engine import HipEngine
from yololite_amd.nn.lower import BF16, F32, OP_ATTENTION, OP_CONV, OP_DWCONV3, OP_SPPF_POOL, OP_UPSAMPLE, Program, TView, _Arena


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).float()


class Mini:
    """A hand-built program: allocate buffers, add ops, upload NCHW tensors, run, read back."""

    def __init__(self, batch: int, device):
        self.batch = batch
        self.device = device
        self.ar = _Arena(batch)
        self.prog = Program(batch, 32, 32, 80)
        self.prog.input_view = self.ar.alloc(32, 32, 8)
        self.fills = []
        self.eng: Optional[HipEngine] = None

    def buf(self, H, W, C, dtype=BF16) -> TView:
        return self.ar.alloc(H, W, C, dtype)

    def fill(self, view: TView, x_nchw: torch.Tensor):
        self.fills.append((view, x_nchw))

    def op(self, **kw):
        kw.setdefault("res", None)
        self.prog.ops.append(kw)

    def run(self):
        self.prog.ws_bytes = self.ar.top + 256
        self.eng = HipEngine(self.prog, self.device)
        ws = self.eng.workspace  # already cleared by the engine (guard pixels)
        for v, x in self.fills:
            if v.dtype == F32:
                n = self.batch * v.H * v.W * v.pitch
                t = ws[v.offset: v.offset + 4 * n].view(torch.float32).view(self.batch, v.H, v.W, v.pitch)
                t[..., v.coff: v.coff + v.C] = x.permute(0, 2, 3, 1).float().to(self.device)
            else:
                self.eng.write_view(v, x)
        im = torch.zeros(self.batch, 3, 32, 32, device=self.device)
        self.eng.forward(im, decode=False)
        torch.cuda.synchronize()

    def read(self, view: TView) -> torch.Tensor:
        if view.dtype != F32:  # layout invariant: kernels never write guard pixels
            assert self.eng.guards_are_zero(view), "a kernel wrote into the guard pixels"
        return self.eng.view_tensor(view).cpu()


def run_conv(device, x, w, b, k, s, act, res=None, pitch_extra=(0, 0), out_f32=False, cin_weight=0,
             expect_kernel=None):
    """x (B,Cin,H,W) fp32 (already bf16-representable), w OIHW, b (Cout,) -> HIP conv output (fp32 NCHW)."""
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    m = Mini(B, device)
    xin = m.buf(H, W, Cin + pitch_extra[0]).slice(pitch_extra[0], Cin) if pitch_extra[0] else m.buf(H, W, Cin)
    yb = m.buf(Ho, Wo, Cout + pitch_extra[1], F32 if out_f32 else BF16)
    yv = yb.slice(pitch_extra[1], Cout) if pitch_extra[1] else yb
    m.fill(xin, x)
    rv = None
    if res is not None:
        rv = m.buf(Ho, Wo, Cout)
        m.fill(rv, res)
    m.op(kind=OP_CONV, name="t", k=k, stride=s, act=int(act), inp=xin, out=yv, res=rv,
         weight=np.ascontiguousarray(w.numpy()), bias=np.ascontiguousarray(b.numpy()), cin_weight=cin_weight)
    m.run()
    if expect_kernel is not None:
        assert m.eng.op_kernel(0) == expect_kernel, (m.eng.op_kernel(0), expect_kernel)
    return m.read(yv)


# ------------------------------------------------------------------------------------------------ fp8 (e4m3) convs
F8_MAX = 448.0


def e4m3_round(t: torch.Tensor) -> torch.Tensor:
    """Saturating round-to-nearest-even onto the OCP e4m3 grid (what the kernels' v_cvt_pk_fp8_f32 + clamp do)."""
    return t.float().clamp(-F8_MAX, F8_MAX).to(torch.float8_e4m3fn).float()


def quantize_filters_e4m3(w: torch.Tensor):
    """Per-output-channel e4m3 quantisation as conv_f8_pack_weights does it: scale = absmax / 448."""
    m = w.abs().flatten(1).amax(1)
    s = torch.where(m > 0, m / F8_MAX, torch.ones_like(m))
    return e4m3_round(w / s.view(-1, 1, 1, 1)), s


def run_conv_f8(device, xq, in_scale, w, b, k, s, act, res=None, res_scale=0.0, out_scale=0.0, expect_kernel=None):
    """xq (B,Cin,H,W): e4m3-representable values (the real input is xq * in_scale); w fp32 OIHW; res: bf16-representable
    tensor, or e4m3-representable values when res_scale > 0; out_scale > 0: fp8 output (returned as raw e4m3 values)."""
    from yololite_amd.nn.lower import FP8

    B, Cin, H, W = xq.shape
    Cout = w.shape[0]
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    m = Mini(B, device)
    xin = m.buf(H, W, Cin, FP8)
    yv = m.buf(Ho, Wo, Cout, FP8 if out_scale > 0 else BF16)
    m.fill(xin, xq)
    rv = None
    if res is not None:
        rv = m.buf(Ho, Wo, Cout, FP8 if res_scale > 0 else BF16)
        m.fill(rv, res)
    m.op(kind=OP_CONV, name="t8", k=k, stride=s, act=int(act), inp=xin, out=yv, res=rv, in_scale=in_scale,
         out_scale=out_scale, res_scale=res_scale,
         weight=np.ascontiguousarray(w.numpy()), bias=np.ascontiguousarray(b.numpy()))
    m.run()
    if expect_kernel is not None:
        assert m.eng.op_kernel(0) == expect_kernel, (m.eng.op_kernel(0), expect_kernel)
    return m.read(yv)
