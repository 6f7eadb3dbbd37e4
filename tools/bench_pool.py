"""Micro-benchmark of the max-pool kernels through the C ABI: dense tensor vs a 32-channel slice of a 96-channel buffer
(the layout the skip connections have inside their concat buffer).   python tools/bench_pool.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bio_image_unet_amd._lib import biu_act, biu_xform, check, lib  # noqa: E402

st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())


def timeit(f, reps=10):
    for _ in range(3):
        check(f())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        check(f())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, (n, d, h, w), c in (("3d-L1", (4, 128, 128, 128), 32), ("3d-L2", (4, 64, 64, 64), 64), ("2d-L1", (16, 1, 512, 512), 32)):
    for pitch, c0 in ((c, 0), (3 * c, 2 * c)):
        x = torch.randn(n, d, h, w, pitch, device="cuda").bfloat16()
        dx = torch.randn(n, d, h, w, pitch, device="cuda").bfloat16()
        pd = 2 if d > 1 else 1
        out = torch.empty(n, d // pd, h // 2, w // 2, c, device="cuda", dtype=torch.bfloat16)
        dout = torch.randn_like(out)
        ax = biu_act(x.data_ptr() + 2 * c0, n, d, h, w, c, pitch)
        adx = biu_act(dx.data_ptr() + 2 * c0, n, d, h, w, c, pitch)
        ao = biu_act(out.data_ptr(), n, d // pd, h // 2, w // 2, c, c)
        ado = biu_act(dout.data_ptr(), n, d // pd, h // 2, w // 2, c, c)
        vec = [torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1, torch.full((c,), 0.1, device="cuda")]
        xf = biu_xform(*[t.data_ptr() for t in vec])
        mean, invstd = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        part = torch.empty(8192 * c * 2, device="cuda")
        nblk = C.c_int(0)
        vox = n * d * h * w
        tf = timeit(lambda: lib.biu_maxpool_fwd(C.byref(ax), C.byref(xf), C.byref(ao), 1, st))
        tb = timeit(lambda: lib.biu_maxpool_bwd(C.byref(ax), C.byref(xf), C.byref(ado), C.byref(adx), 1, 1, st))
        tr = timeit(lambda: lib.biu_maxpool_bwd_bnred(C.byref(ax), C.byref(xf), C.byref(ado), C.byref(adx), 1, P(mean), P(invstd), P(part),
                                                      part.numel(), C.byref(nblk), 1, st))
        gb_f = vox * c * 2 * (1 + 1.0 / (4 * pd)) / 1e9
        gb_b = vox * c * 2 * (3 + 1.0 / (4 * pd)) / 1e9
        print(f"{name} C={c} pitch={pitch}: fwd {tf:.3f} ms ({gb_f / tf * 1e3:.0f} GB/s) | bwd(acc) {tb:.3f} ms ({gb_b / tb * 1e3:.0f} GB/s) | "
              f"bwd+bnred {tr:.3f} ms ({gb_b / tr * 1e3:.0f} GB/s)", flush=True)
