"""Diagnostic build (BIU_DIAG): where a step of the rolling-window convolution (biu_conv_roll.hip) spends its cycles -- wave 0 of every block.
    bash tools/build_variant.sh diag -DBIU_DIAG=1 && BIU_ROLL=always python tools/diag_roll.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_image_unet_amd._lib import biu_act, biu_xform, SIGNATURES
lib = C.CDLL(os.path.join(ROOT, "tools", "variants", os.environ.get("DIAG_LIB", "libbiu_diag.so")))
for name, (res, args) in SIGNATURES.items():
    getattr(lib, name).restype = res; getattr(lib, name).argtypes = args
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
diag = torch.zeros(10, dtype=torch.int64, device="cuda")
C.c_void_p.in_dll(lib, "biu_diag_buffer").value = diag.data_ptr()
n, (d, h, w) = 4, (128, 128, 128)
names = ["loop", "wait fetch", "mfma phase", "epilogue", "barrier"]
for cin, cout in ((16, 32), (32, 16), (32, 32)):
    xfv = [torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1, torch.full((cin,), 0.1, device="cuda")]
    xfs = biu_xform(*[t.data_ptr() for t in xfv])
    x = torch.randn(n, d, h, w, cin, device="cuda").to(torch.bfloat16)
    y = torch.empty(n, d, h, w, cout, device="cuda", dtype=torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.05
    pk = torch.empty(lib.biu_conv_packed_bytes(0, cin, cout, 3, 3, 3, 1, 1), dtype=torch.uint8, device="cuda")
    lib.biu_conv_pack(0, P(wt), cin, cout, 3, 3, 3, 1, P(pk), st)
    ax = biu_act(x.data_ptr(), n, d, h, w, cin, cin); ay = biu_act(y.data_ptr(), n, d, h, w, cout, cout)
    stat = torch.empty(lib.biu_conv_fwd_stats_floats(C.byref(ay), 3), device="cuda"); nblk = C.c_int(0)
    yup = torch.randn(n, d, h, w, cout, device="cuda").to(torch.bfloat16); ayup = biu_act(yup.data_ptr(), n, d, h, w, cout, cout)
    uv = [torch.rand(cout, device="cuda") + 0.5, torch.zeros(cout, device="cuda"), torch.full((cout,), 0.1, device="cuda"), torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")]
    part = torch.empty(lib.biu_bwd_data_bnred_floats(C.byref(ay), 3, 0), device="cuda"); nb2 = C.c_int(0)
    calls = {
        "plain, no transform": lambda: lib.biu_conv_fwd(C.byref(ax), None, P(wt), P(pk), None, 3, 3, 3, 1, C.byref(ay), None, 0, 1, st),
        "transform": lambda: lib.biu_conv_fwd(C.byref(ax), C.byref(xfs), P(wt), P(pk), None, 3, 3, 3, 1, C.byref(ay), None, 0, 1, st),
        "transform + statistics": lambda: lib.biu_conv_fwd_stats(C.byref(ax), C.byref(xfs), P(wt), P(pk), None, 3, 3, 3, 1, C.byref(ay), P(stat), stat.numel(), C.byref(nblk), None, 0, 1, st),
        "BatchNorm-backward sums (no transform)": lambda: lib.biu_conv_bwd_data_bnred(C.byref(ax), P(wt), P(pk), 3, 3, 3, 1, C.byref(ay), C.byref(ayup), P(uv[0]), P(uv[1]), P(uv[2]), P(uv[3]), P(uv[4]), P(part), part.numel(), C.byref(nb2), None, 0, 1, st),
    }
    for tag, f in calls.items():
        f(); torch.cuda.synchronize(); diag.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        dv = diag.cpu().tolist(); nb = max(dv[7], 1); tot = sum(dv[:5])
        print(f"{cin}->{cout} {tag}: {e0.elapsed_time(e1):.3f} ms, clock {dv[8] / max(dv[9], 1) * 0.1:.2f} GHz, steps {nb}, cycles/step {tot / nb:.0f}: " +
              ", ".join(f"{nm} {dv[i] / nb:.0f} ({100 * dv[i] / tot:.0f}%)" for i, nm in enumerate(names)), flush=True)
