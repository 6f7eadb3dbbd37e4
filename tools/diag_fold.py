"""Diagnostic build (tools/build_variant.sh diag -DBIU_DIAG=1): where the cycles of an item of the folded FORWARD go (k_conv_pipe<T, 2, 2, 1, ...>,
wave 0 of every block), at cfg4's decode5 / decode3 and cfg5's up3_conv shapes.      python tools/diag_fold.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_image_unet_amd._lib import SIGNATURES, biu_act, biu_xform  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "variants", "libbiu_diag.so"))
for name, (res, args) in SIGNATURES.items():
    getattr(lib, name).restype = res
    getattr(lib, name).argtypes = args
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
diag = torch.zeros(10, dtype=torch.int64, device="cuda")
C.c_void_p.in_dll(lib, "biu_diag_buffer").value = diag.data_ptr()
names = ["loop", "issue", "mfma", "epilogue", "barrier1", "commit", "barrier2"]
for name, n, cin, cout, (d, h, w), accumulate_onto in (("decode5 fold", 4, 64, 32, (64, 64, 64), True), ("decode3 fold", 4, 128, 64, (32, 32, 32), True),
                                                       ("up3_conv", 1, 128, 128, (64, 128, 128), False)):
    x = torch.randn(n, d, h, w, cin, device="cuda").to(torch.bfloat16)
    y = torch.randn(n, 2 * d, 2 * h, 2 * w, cout, device="cuda").to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.02
    xfv = [torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"), torch.full((cin,), 0.1, device="cuda")]
    xf = biu_xform(*[t.data_ptr() for t in xfv])
    ax, ay = biu_act(x.data_ptr(), n, d, h, w, cin, cin), biu_act(y.data_ptr(), n, 2 * d, 2 * h, 2 * w, cout, cout)
    pf = torch.empty(lib.biu_upconv_packed_bytes(0, cin, cout, 1), dtype=torch.uint8, device="cuda")
    assert lib.biu_upconv_pack(0, P(wt), cin, cout, 1, P(pf), st) == 0
    stat = torch.empty(lib.biu_upconv_fwd_stats_floats(C.byref(ax), C.byref(ay)), device="cuda")
    nblk = C.c_int(0)
    f = lambda: lib.biu_upconv_fwd(C.byref(ax), C.byref(xf), P(pf), None, C.byref(ay), P(stat), stat.numel(), C.byref(nblk), 1, st)  # noqa: E731
    assert f() == 0
    torch.cuda.synchronize()
    diag.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    dv = diag.cpu().tolist()
    nb = max(dv[7], 1)
    tot = sum(dv[:7])
    print(f"{name} {cin} -> {cout} coarse {(d, h, w)}: {e0.elapsed_time(e1):.3f} ms, clock {dv[8] / max(dv[9], 1) * 0.1:.2f} GHz, items {nb}, cycles/item {tot / nb:.0f}: "
          + ", ".join(f"{nm} {dv[i] / nb:.0f} ({100 * dv[i] / tot:.0f}%)" for i, nm in enumerate(names)), flush=True)
