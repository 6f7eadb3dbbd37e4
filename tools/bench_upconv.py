"""Nearest-neighbour up-sampling + 3x3x3 conv (forward with BatchNorm statistics) at cfg5's upN_conv shapes: the folded kernel
(biu_upconv_fwd) against biu_nearest_up_fwd + biu_conv_fwd_stats.      python tools/bench_upconv.py [bf16|f32]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bio_image_unet_amd._lib import biu_act, biu_xform, check, lib  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tdt, code = (torch.bfloat16, 1) if dt == "bf16" else (torch.float32, 0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


def timed(f, reps=5):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, n, c, (d, h, w) in (("up3_conv", 1, 128, (64, 128, 128)), ("up2_conv", 1, 256, (32, 64, 64)), ("up1_conv", 1, 512, (16, 32, 32))):
    x = torch.randn(n, d, h, w, c, device="cuda").to(tdt)
    u = torch.empty(n, 2 * d, 2 * h, 2 * w, c, device="cuda", dtype=tdt)
    y = torch.empty(n, 2 * d, 2 * h, 2 * w, c, device="cuda", dtype=tdt)
    wt = torch.randn(c, c, 3, 3, 3, device="cuda") * 0.02
    bias = torch.randn(c, device="cuda")
    xs, xb, xl = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.full((c,), 0.1, device="cuda")
    xf = biu_xform(xs.data_ptr(), xb.data_ptr(), xl.data_ptr())
    ax, au, ay = biu_act(x.data_ptr(), n, d, h, w, c, c), biu_act(u.data_ptr(), n, 2 * d, 2 * h, 2 * w, c, c), biu_act(y.data_ptr(), n, 2 * d, 2 * h, 2 * w, c, c)
    pk = torch.empty(lib.biu_conv_packed_bytes(0, c, c, 3, 3, 3, 1, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(0, P(wt), c, c, 3, 3, 3, code, P(pk), st))
    pf = torch.empty(lib.biu_upconv_packed_bytes(0, c, c, code), dtype=torch.uint8, device="cuda")
    stat = torch.empty(max(lib.biu_conv_fwd_stats_floats(C.byref(ay), 3), lib.biu_upconv_fwd_stats_floats(C.byref(ax), C.byref(ay))), device="cuda")
    nblk = C.c_int(0)
    t_up = timed(lambda: check(lib.biu_nearest_up_fwd(C.byref(ax), C.byref(xf), C.byref(au), code, st)))
    t_cv = timed(lambda: check(lib.biu_conv_fwd_stats(C.byref(au), None, P(wt), P(pk), P(bias), 3, 3, 3, 1, C.byref(ay), P(stat), stat.numel(), C.byref(nblk), None, 0, code, st)))
    t_pk = timed(lambda: check(lib.biu_upconv_pack(0, P(wt), c, c, code, P(pf), st)))
    t_fd = timed(lambda: check(lib.biu_upconv_fwd(C.byref(ax), C.byref(xf), P(pf), P(bias), C.byref(ay), P(stat), stat.numel(), C.byref(nblk), code, st)))
    yref = y.clone()                                          # (the timed loop above left the unfolded result in y before the folded one overwrote it?  recompute both)
    check(lib.biu_conv_fwd_stats(C.byref(au), None, P(wt), P(pk), P(bias), 3, 3, 3, 1, C.byref(ay), P(stat), stat.numel(), C.byref(nblk), None, 0, code, st))
    yref = y.float().clone()
    check(lib.biu_upconv_fwd(C.byref(ax), C.byref(xf), P(pf), P(bias), C.byref(ay), P(stat), stat.numel(), C.byref(nblk), code, st))
    yf = y.float()
    diff = (yf - yref).abs()
    print(f"   folded vs unfolded: max |diff| {float(diff.max()):.4f}, mean {float(diff.mean()):.5f}, max |y| {float(yref.abs().max()):.3f}, rms y {float(yref.pow(2).mean().sqrt()):.3f}; "
          f"worst voxel {tuple(int(v) for v in torch.unravel_index(diff.amax(-1).argmax(), diff.shape[:-1]))}", flush=True)
    # backward: data gradient (+ nearest_up_bwd) and BatchNorm-fused weight gradient, unfolded on the up-sampled tensor against folded
    dy = torch.randn(n, 2 * d, 2 * h, 2 * w, c, device="cuda").to(tdt)
    du = torch.empty_like(u)
    dx = torch.empty_like(x)
    ady, adu, adx = biu_act(dy.data_ptr(), n, 2 * d, 2 * h, 2 * w, c, c), biu_act(du.data_ptr(), n, 2 * d, 2 * h, 2 * w, c, c), biu_act(dx.data_ptr(), n, d, h, w, c, c)
    pk1 = torch.empty(lib.biu_conv_packed_bytes(1, c, c, 3, 3, 3, 1, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(1, P(wt), c, c, 3, 3, 3, code, P(pk1), st))
    pf1 = torch.empty(lib.biu_upconv_packed_bytes(1, c, c, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_upconv_pack(1, P(wt), c, c, code, P(pf1), st))
    t_dg = timed(lambda: check(lib.biu_conv_bwd_data(C.byref(ady), P(wt), P(pk1), 3, 3, 3, 1, C.byref(adu), 0, None, 0, code, st)))
    t_ub = timed(lambda: check(lib.biu_nearest_up_bwd(C.byref(adu), C.byref(adx), 0, code, st)))
    t_fdg = timed(lambda: check(lib.biu_upconv_bwd_data(C.byref(ady), P(pf1), C.byref(adx), 0, code, st)))
    kv = [torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.full((c,), 0.1, device="cuda"), torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")]
    ws = torch.empty(max(lib.biu_conv_bwd_weight_workspace(c, c, 3, 3, 3, code), lib.biu_upconv_bwd_weight_workspace(c, c, code)), dtype=torch.uint8, device="cuda")
    dw = torch.empty_like(wt)
    t_wg = timed(lambda: check(lib.biu_conv_bwd_weight_bn(C.byref(au), None, C.byref(ady), C.byref(ay), P(kv[0]), P(kv[1]), P(kv[2]), P(kv[3]), P(kv[4]), P(kv[5]), 3, 3, 3, 1, P(dw), P(ws), ws.numel(), code, st)))
    t_fwg = timed(lambda: check(lib.biu_upconv_bwd_weight_bn(C.byref(ax), C.byref(xf), C.byref(ady), C.byref(ay), P(kv[0]), P(kv[1]), P(kv[2]), P(kv[3]), P(kv[4]), P(kv[5]), P(dw), P(ws), ws.numel(), code, st)))
    print(f"   data gradient: conv {t_dg:.3f} + nearest_up_bwd {t_ub:.3f} ms  |  folded {t_fdg:.3f} ms      weight gradient (BatchNorm-fused): unfolded {t_wg:.3f} ms  |  folded {t_fwg:.3f} ms", flush=True)
    fl27 = 2.0 * n * 8 * d * h * w * 27 * c * c
    print(f"{name} {c}->{c} coarse {(d, h, w)}: up-sample {t_up:.3f} ms + conv {t_cv:.3f} ms ({fl27 / t_cv / 1e9:.0f} TF/s)  |  folded {t_fd:.3f} ms "
          f"({fl27 * 8 / 27 / t_fd / 1e9:.0f} TF/s of its own 8-tap work) + fold/pack {t_pk:.3f} ms", flush=True)
