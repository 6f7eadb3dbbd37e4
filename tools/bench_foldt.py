"""ConvTranspose + concat + conv of a decoder level as one folded op (biu_foldt_*) at cfg4's decode5 / decode3 shapes, per call, against the
three separate ops it replaces.      python tools/bench_foldt.py [bf16|f32] [decode5|decode3|decode1]        (BENCH_LEGS=fwd,dg,wg,wg1; BENCH_REPS=n)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bio_image_unet_amd._lib import biu_act, biu_xform, check, lib  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
only = sys.argv[2] if len(sys.argv) > 2 else None
legs = (os.environ.get("BENCH_LEGS") or "fwd,dg,wg,ref").split(",")
reps = int(os.environ.get("BENCH_REPS", "5"))
tdt, code = (torch.bfloat16, 1) if dt == "bf16" else (torch.float32, 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def timed(f):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, n, cl, cup, cs, cout, (d, h, w) in (("decode5", 4, 64, 64, 32, 32, (64, 64, 64)), ("decode3", 4, 128, 128, 64, 64, (32, 32, 32)),
                                               ("decode1", 4, 256, 256, 128, 128, (16, 16, 16))):
    if only and only != name:
        continue
    xl = torch.randn(n, d, h, w, cl, device="cuda").to(tdt)
    sk = torch.randn(n, 2 * d, 2 * h, 2 * w, cs, device="cuda").to(tdt)
    y = torch.empty(n, 2 * d, 2 * h, 2 * w, cout, device="cuda", dtype=tdt)
    dy = torch.randn(n, 2 * d, 2 * h, 2 * w, cout, device="cuda").to(tdt)
    dxl, dsk = torch.empty_like(xl), torch.empty_like(sk)
    wc = torch.randn(cout, cup + cs, 3, 3, 3, device="cuda") * 0.02
    bc = torch.randn(cout, device="cuda")
    wt = torch.randn(cl, cup, 2, 2, 2, device="cuda") * 0.05
    bt = torch.randn(cup, device="cuda")
    one = lambda c: biu_xform(torch.ones(c, device="cuda").data_ptr(), torch.zeros(c, device="cuda").data_ptr(), torch.full((c,), 0.1, device="cuda").data_ptr())  # noqa: E731
    keep = [torch.ones(cl, device="cuda"), torch.zeros(cl, device="cuda"), torch.full((cl,), 0.1, device="cuda"), torch.ones(cs, device="cuda"),
            torch.zeros(cs, device="cuda"), torch.full((cs,), 0.1, device="cuda")]
    xfl, xfs = biu_xform(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()), biu_xform(keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr())
    A = lambda t, dd, hh, ww, c: biu_act(t.data_ptr(), n, dd, hh, ww, c, c)  # noqa: E731
    axl, ask, ay, ady = A(xl, d, h, w, cl), A(sk, 2 * d, 2 * h, 2 * w, cs), A(y, 2 * d, 2 * h, 2 * w, cout), A(dy, 2 * d, 2 * h, 2 * w, cout)
    adxl, adsk = A(dxl, d, h, w, cl), A(dsk, 2 * d, 2 * h, 2 * w, cs)
    pk = torch.empty(lib.biu_foldt_packed_bytes(cl, cs, cout, code), dtype=torch.uint8, device="cuda")
    stat = torch.empty(lib.biu_foldt_fwd_stats_floats(C.byref(axl), C.byref(ay)), device="cuda")
    ws = torch.empty(lib.biu_foldt_bwd_weight_workspace(cl, cs, cout, code), dtype=torch.uint8, device="cuda")
    kv = [torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda"), torch.full((cout,), 0.1, device="cuda"), torch.ones(cout, device="cuda"),
          torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")]
    dwc, dwt, dbt = torch.empty_like(wc), torch.empty_like(wt), torch.empty_like(bt)
    nblk = C.c_int(0)
    fl = 2.0 * n * 8 * d * h * w * cout * (27 * cs + 8 * cl)
    out = []
    t_pk = timed(lambda: check(lib.biu_foldt_pack(P(wc), P(bc), P(wt), P(bt), cl, cup, cs, cout, code, P(pk), st)))
    out.append(f"pack {t_pk:.3f} ms")
    if "fwd" in legs:
        t = timed(lambda: check(lib.biu_foldt_fwd(C.byref(axl), C.byref(xfl), C.byref(ask), C.byref(xfs), P(pk), C.byref(ay), P(stat), stat.numel(), C.byref(nblk), code, st)))
        out.append(f"fwd {t:.3f} ms {fl / t / 1e9:.0f} TF/s")
    if "dg" in legs:
        t = timed(lambda: check(lib.biu_foldt_bwd_data(C.byref(ady), P(pk), C.byref(adxl), 0, C.byref(adsk), 0, None, None, None, None, None, None, None, 0, None,
                                                       P(ws), ws.numel(), code, st)))
        out.append(f"dgrad {t:.3f} ms {fl / t / 1e9:.0f} TF/s")
    if "wg" in legs:
        t = timed(lambda: check(lib.biu_foldt_bwd_weight_bn(C.byref(axl), C.byref(xfl), C.byref(ask), C.byref(xfs), C.byref(ady), C.byref(ay), P(kv[0]), P(kv[1]), P(kv[2]),
                                                            P(kv[3]), P(kv[4]), P(kv[5]), None, P(wc), P(wt), P(bt), cup, P(dwc), P(dwt), P(dbt), P(ws), ws.numel(), code, st)))
        out.append(f"wgrad_bn {t:.3f} ms {fl / t / 1e9:.0f} TF/s")
    if "wg1" in legs:                                    # the main-stream part (skip half with da -> dy, G): wgrad_bn minus this = border sums + chain rule
        t = timed(lambda: check(lib.biu_foldt_bwd_weight_bn_phase(C.byref(axl), C.byref(xfl), C.byref(ask), C.byref(xfs), C.byref(ady), C.byref(ay), P(kv[0]), P(kv[1]),
                                                                  P(kv[2]), P(kv[3]), P(kv[4]), P(kv[5]), None, P(wc), P(wt), P(bt), cup, P(dwc), P(dwt), P(dbt), P(ws),
                                                                  ws.numel(), code, 1 | 4, st)))
        out.append(f"wgrad_bn(phases 1 | 4) {t:.3f} ms")
    print(f"{name} x_low {cl} ch @{(d, h, w)}, up {cup} | skip {cs} -> {cout} @{(2 * d, 2 * h, 2 * w)}: " + " | ".join(out), flush=True)
