"""Helpers for the -m gpu tests: move NC[D]HW CPU tensors into channels-last device buffers and back, and build
the ctypes structs of include/biu.h around them.  Uses torch only for memory and layout shuffling."""
import ctypes as C

import torch

from bio_image_unet_amd._lib import BIU_BF16, BIU_F32, biu_act, biu_xform, check, lib  # noqa: F401

DT = {"f32": (torch.float32, BIU_F32), "bf16": (torch.bfloat16, BIU_BF16)}


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def as5d(x):
    """NCHW -> NC1HW (2-D tensors are D = 1 volumes)."""
    return x.unsqueeze(2) if x.dim() == 4 else x


class Dev:
    """A channels-last device buffer [N,D,H,W,pitch] holding `x` (NCDHW, CPU) in channels c0..c0+C."""

    def __init__(self, x=None, *, shape=None, dtype="f32", pitch=None, c0=0, fill=None):
        tdt, _ = DT[dtype]
        if x is not None:
            x = as5d(x)
            n, c, d, h, w = x.shape
        else:
            n, c, d, h, w = shape
        pitch = pitch or c
        self.n, self.c, self.d, self.h, self.w, self.pitch, self.c0 = n, c, d, h, w, pitch, c0
        self.buf = torch.full((n, d, h, w, pitch), float("nan") if fill is None else fill, dtype=tdt, device="cuda")
        if x is not None:
            self.buf[..., c0:c0 + c] = x.permute(0, 2, 3, 4, 1).to(device="cuda", dtype=tdt)
        self.act = biu_act(self.buf.data_ptr() + c0 * self.buf.element_size(), n, d, h, w, c, pitch)

    def a(self):
        return C.byref(self.act)

    def get(self, squeeze2d=False):
        t = self.buf[..., self.c0:self.c0 + self.c].float().cpu().permute(0, 4, 1, 2, 3).contiguous()
        return t.squeeze(2) if squeeze2d else t

    def ref(self):
        """What the device actually holds (after rounding to the storage dtype), as NCDHW fp32 on the CPU."""
        return self.get()


class XF:
    def __init__(self, c, seed=0, identity=False):
        g = torch.Generator().manual_seed(seed)
        if identity:
            self.scale, self.shift, self.slope = torch.ones(c), torch.zeros(c), torch.ones(c)
        else:
            self.scale = torch.randn(c, generator=g) * 0.5 + 1.0
            self.scale[::3] *= -1          # negative BN scales must be handled (max-pool is not monotone then)
            self.shift = torch.randn(c, generator=g) * 0.3
            self.slope = torch.full((c,), 0.1)
        self.d = [t.cuda() for t in (self.scale, self.shift, self.slope)]
        self.s = biu_xform(*[t.data_ptr() for t in self.d])

    def x(self):
        return C.byref(self.s)

    def apply(self, x):      # x NCDHW cpu fp32
        shp = (1, -1) + (1,) * (x.dim() - 2)
        t = x * self.scale.view(shp) + self.shift.view(shp)
        return torch.where(t > 0, t, t * self.slope.view(shp))


def tol(dtype):
    # fp32: 1e-3 rel is the north-star bound; per-op results are far tighter.  bf16: 8-bit mantissa storage.
    return dict(rtol=2e-4, atol=2e-5) if dtype == "f32" else dict(rtol=3e-2, atol=3e-2)


def assert_close(got, want, dtype, what="", scale_atol=True):
    t = tol(dtype)
    atol = t["atol"] * (float(want.abs().max()) + 1e-12 if scale_atol else 1.0)
    torch.testing.assert_close(got, want, rtol=t["rtol"], atol=atol, msg=lambda m: f"{what}: {m}")
