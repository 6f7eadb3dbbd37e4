"""Micro-benchmark of the MFMA conv kernels through the C ABI at the layer shapes of the bench workloads.

    python tools/bench_conv.py [cfg4|cfg2] [bf16|f32]

Prints HIP-event time and algorithmic TFLOP/s (2*vox*taps*Cin*Cout) for forward, data-gradient and weight-gradient."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bio_image_unet_amd._lib import biu_act, biu_xform, check, lib  # noqa: E402

LAYERS = {
    "cfg4": [("encode2", 3, 4, 16, 32, (128, 128, 128)), ("encode4", 3, 4, 32, 64, (64, 64, 64)), ("decode1", 3, 4, 384, 128, (32, 32, 32)),
             ("decode3", 3, 4, 192, 64, (64, 64, 64)), ("decode4", 3, 4, 64, 64, (64, 64, 64)), ("decode5", 3, 4, 96, 32, (128, 128, 128)),
             ("decode6", 3, 4, 32, 16, (128, 128, 128)), ("middle2", 3, 4, 128, 256, (16, 16, 16)),
             ("decode5b", 3, 4, 32, 32, (128, 128, 128)), ("encode3", 3, 4, 32, 64, (64, 64, 64))],
    "cfg1": [("encode6", 2, 2, 128, 128, (64, 64)), ("encode7", 2, 2, 128, 256, (32, 32)), ("encode8", 2, 2, 256, 256, (32, 32)), ("middle1", 2, 2, 256, 512, (16, 16)),
             ("middle2", 2, 2, 512, 512, (16, 16)), ("decode1", 2, 2, 512, 256, (32, 32))],
    "cfg3": [("encode2", 2, 32, 32, 32, (512, 512)), ("encode4", 2, 32, 64, 64, (256, 256)), ("decode7", 2, 16, 64, 32, (512, 512)),
             ("encode6", 2, 32, 128, 128, (128, 128)), ("encode8", 2, 32, 256, 256, (64, 64))],
    "cfg2": [("encode2", 2, 16, 64, 64, (512, 512)), ("decode7", 2, 16, 128, 64, (512, 512)), ("decode5", 2, 16, 256, 128, (256, 256)),
             ("middle2", 2, 16, 1024, 1024, (32, 32))],
}


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    dt = sys.argv[2] if len(sys.argv) > 2 else ("bf16" if cfg in ("cfg4", "cfg3") else "f32")
    only = sys.argv[3] if len(sys.argv) > 3 else None
    tdt, code = (torch.bfloat16, 1) if dt == "bf16" else (torch.float32, 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    keep = []
    for name, nd, n, cin, cout, sp in LAYERS[cfg]:
        if only and only != name:
            continue
        kd = 3 if nd == 3 else 1
        d, h, w = (sp if nd == 3 else (1,) + tuple(sp))
        x = torch.randn(n, d, h, w, cin, device="cuda").to(tdt)
        y = torch.empty(n, d, h, w, cout, device="cuda", dtype=tdt)
        dy = torch.randn(n, d, h, w, cout, device="cuda").to(tdt)
        dx = torch.empty_like(x)
        if os.environ.get("BENCH_ZERO") == "1":          # power probe: all-zero operands toggle no datapath bits
            x.zero_(); dy.zero_()
        wt = torch.randn((cout, cin) + (3,) * nd, device="cuda") * 0.05
        if os.environ.get("BENCH_ZERO") == "1":
            wt.zero_()
        bias = torch.randn(cout, device="cuda")
        xs, xb, xl = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda"), torch.full((cin,), 0.1, device="cuda")
        xf = biu_xform(xs.data_ptr(), xb.data_ptr(), xl.data_ptr())
        ax = biu_act(x.data_ptr(), n, d, h, w, cin, cin)
        ay = biu_act(y.data_ptr(), n, d, h, w, cout, cout)
        ady = biu_act(dy.data_ptr(), n, d, h, w, cout, cout)
        adx = biu_act(dx.data_ptr(), n, d, h, w, cin, cin)
        pk0 = torch.empty(max(lib.biu_conv_packed_bytes(0, cin, cout, kd, 3, 3, 1, code), 16), dtype=torch.uint8, device="cuda")
        pk1 = torch.empty(max(lib.biu_conv_packed_bytes(1, cin, cout, kd, 3, 3, 1, code), 16), dtype=torch.uint8, device="cuda")
        check(lib.biu_conv_pack(0, P(wt), cin, cout, kd, 3, 3, code, P(pk0), st))
        check(lib.biu_conv_pack(1, P(wt), cin, cout, kd, 3, 3, code, P(pk1), st))
        ws = torch.empty(max(lib.biu_conv_bwd_weight_workspace(cin, cout, kd, 3, 3, code), 16), dtype=torch.uint8, device="cuda")
        dw = torch.empty_like(wt)
        fl = 2.0 * n * d * h * w * (27 if nd == 3 else 9) * cin * cout
        noxf = os.environ.get("BENCH_NOXF") == "1"
        kvec = [torch.rand(cout, device="cuda") * 0.1 + 0.95 for _ in range(1)] + [torch.zeros(cout, device="cuda"), torch.full((cout,), 0.1, device="cuda"),
                torch.full((cout,), 1.0, device="cuda"), torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")]
        stat = torch.empty(lib.biu_conv_fwd_stats_floats(C.byref(ay), kd), device="cuda")
        nblk = C.c_int(0)
        calls = {
            "fwd": lambda: lib.biu_conv_fwd(C.byref(ax), None if noxf else C.byref(xf), P(wt), P(pk0), P(bias), kd, 3, 3, 1, C.byref(ay), None, 0, code, st),
            "dgrad": lambda: lib.biu_conv_bwd_data(C.byref(ady), P(wt), P(pk1), kd, 3, 3, 1, C.byref(adx), 0, None, 0, code, st),
            "wgrad": lambda: lib.biu_conv_bwd_weight(C.byref(ax), C.byref(xf), C.byref(ady), kd, 3, 3, 1, P(dw), None, P(ws), ws.numel(), code, st),
            # the training step's forms: forward + BatchNorm statistics; weight gradient with BatchNorm backward in its loader
            # (cA = 1, cB = cC = 0 keeps dy bounded over the repeats)
            "fwd_st": lambda: lib.biu_conv_fwd_stats(C.byref(ax), None if noxf else C.byref(xf), P(wt), P(pk0), P(bias), kd, 3, 3, 1, C.byref(ay),
                                                     P(stat), stat.numel(), C.byref(nblk), None, 0, code, st),
            "wg_bn": lambda: lib.biu_conv_bwd_weight_bn(C.byref(ax), C.byref(xf), C.byref(ady), C.byref(ay), P(kvec[0]), P(kvec[1]), P(kvec[2]),
                                                        P(kvec[3]), P(kvec[4]), P(kvec[5]), kd, 3, 3, 1, P(dw), P(ws), ws.numel(), code, st),
        }
        # the data gradient with the upstream block's BatchNorm-backward sums in its epilogue (what a train step runs for most layers)
        yup = torch.randn(n, d, h, w, cin, device="cuda").to(tdt)
        ayup = biu_act(yup.data_ptr(), n, d, h, w, cin, cin)
        uvec = [torch.rand(cin, device="cuda") + 0.5, torch.zeros(cin, device="cuda"), torch.full((cin,), 0.1, device="cuda"), torch.zeros(cin, device="cuda"),
                torch.ones(cin, device="cuda")]
        part = torch.empty(lib.biu_bwd_data_bnred_floats(C.byref(adx), kd, 0), device="cuda")
        nb2 = C.c_int(0)
        keep.extend([yup, uvec, part])
        calls["dg_red"] = lambda: lib.biu_conv_bwd_data_bnred(C.byref(ady), P(wt), P(pk1), kd, 3, 3, 1, C.byref(adx), C.byref(ayup), P(uvec[0]), P(uvec[1]), P(uvec[2]),
                                                              P(uvec[3]), P(uvec[4]), P(part), part.numel(), C.byref(nb2), None, 0, code, st)
        # two-source forms (what the decoder's first conv runs): x = concat(x0 | x1) held in two dense tensors, split 2:1 like up(2F) | skip(F)
        c0 = (2 * cin // 3) // 32 * 32
        if cin % 96 == 0 and c0 > 0 and lib.biu_conv_cat_ok is not None:
            x0 = x[..., :c0].contiguous()
            x1 = x[..., c0:].contiguous()
            ax0 = biu_act(x0.data_ptr(), n, d, h, w, c0, c0)
            ax1 = biu_act(x1.data_ptr(), n, d, h, w, cin - c0, cin - c0)
            xs0 = biu_xform(xs.data_ptr(), xb.data_ptr(), xl.data_ptr())
            xs1 = biu_xform(xs[c0:].data_ptr(), xb[c0:].data_ptr(), xl[c0:].data_ptr())
            if lib.biu_conv_cat_ok(C.byref(ax0), C.byref(ax1), C.byref(ay), kd, 3, 3, 1, code):
                keep.extend([x0, x1])
                calls["fwd_cat"] = lambda: lib.biu_conv_fwd_cat(C.byref(ax0), C.byref(xs0), C.byref(ax1), C.byref(xs1), P(wt), P(pk0), P(bias), kd, 3, 3, 1,
                                                                 C.byref(ay), P(stat), stat.numel(), C.byref(nblk), None, 0, code, st)
                calls["wg_cat"] = lambda: lib.biu_conv_bwd_weight_cat(C.byref(ax0), C.byref(xs0), C.byref(ax1), C.byref(xs1), C.byref(ady), C.byref(ay),
                                                                      P(kvec[0]), P(kvec[1]), P(kvec[2]), P(kvec[3]), P(kvec[4]), P(kvec[5]), kd, 3, 3, 1,
                                                                      P(dw), P(ws), ws.numel(), code, st)
                dx0, dx1 = torch.empty_like(x0), torch.empty_like(x1)
                adx0 = biu_act(dx0.data_ptr(), n, d, h, w, c0, c0)
                adx1 = biu_act(dx1.data_ptr(), n, d, h, w, cin - c0, cin - c0)
                keep.extend([dx0, dx1])
                calls["dg_cat"] = lambda: lib.biu_conv_bwd_data_cat(C.byref(ady), P(wt), P(pk1), kd, 3, 3, 1, C.byref(adx0), 0, C.byref(adx1), 0, None, 0, code, st)
        legs = os.environ.get("BENCH_LEGS")
        if legs:
            calls = {k: v for k, v in calls.items() if k in legs.split(",")}
        out = []
        for k, f in calls.items():
            for _ in range(2):
                check(f())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                check(f())
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out.append(f"{k} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF/s")
        print(f"{name:8s} {cin:4d}->{cout:4d} @{sp}  " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
