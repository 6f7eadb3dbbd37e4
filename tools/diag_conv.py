"""Diagnostic build (BIU_DIAG): per-phase cycle shares of the MFMA conv kernel (thread 0 of every block)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_image_unet_amd._lib import biu_act, biu_xform, SIGNATURES
lib = C.CDLL(os.path.join(ROOT, "tools", "variants", "libbiu_diag.so"))
for name, (res, args) in SIGNATURES.items():
    getattr(lib, name).restype = res; getattr(lib, name).argtypes = args
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
diag = torch.zeros(10, dtype=torch.int64, device="cuda")
C.c_void_p.in_dll(lib, "biu_diag_buffer").value = diag.data_ptr()
shapes = [("encode2", 16, 32, (128,128,128)), ("decode5", 96, 32, (128,128,128)), ("decode3", 192, 64, (64,64,64)), ("decode6", 32, 16, (128,128,128))]
n = 4
use_xf = os.environ.get("DIAG_XF", "0") == "1"
for name, cin, cout, (d,h,w) in shapes:
    xfv = [torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1, torch.full((cin,), 0.1, device="cuda")]
    xfs = biu_xform(*[t.data_ptr() for t in xfv])
    XF = C.byref(xfs) if use_xf else None
    x = torch.randn(n,d,h,w,cin, device="cuda").to(torch.bfloat16); y = torch.empty(n,d,h,w,cout, device="cuda", dtype=torch.bfloat16)
    wt = torch.randn(cout,cin,3,3,3, device="cuda")*0.05
    pk = torch.empty(lib.biu_conv_packed_bytes(0,cin,cout,3,3,3,1,1), dtype=torch.uint8, device="cuda")
    lib.biu_conv_pack(0,P(wt),cin,cout,3,3,3,1,P(pk),st)
    ax = biu_act(x.data_ptr(),n,d,h,w,cin,cin); ay = biu_act(y.data_ptr(),n,d,h,w,cout,cout)
    stat = torch.empty(lib.biu_conv_fwd_stats_floats(C.byref(ay), 3), device="cuda"); nblk = C.c_int(0)
    fwd = (lambda: lib.biu_conv_fwd_stats(C.byref(ax),XF,P(wt),P(pk),None,3,3,3,1,C.byref(ay),P(stat),stat.numel(),C.byref(nblk), None,0,1,st)) if os.environ.get("DIAG_STATS") == "1" else (lambda: lib.biu_conv_fwd(C.byref(ax),XF,P(wt),P(pk),None,3,3,3,1,C.byref(ay), None,0,1,st))
    fwd(); torch.cuda.synchronize()
    diag.zero_()
    e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); fwd(); e1.record(); torch.cuda.synchronize()
    dv = diag.cpu().tolist(); nb = max(dv[7],1)
    names = ["loop", "issue", "mfma", "epilogue", "barrier1", "commit", "barrier2"]
    tot = sum(dv[:7])
    print(f"{name}: {e0.elapsed_time(e1):.3f} ms, clock {dv[8]/max(dv[9],1)*0.1:.2f} GHz, items {nb}, cycles/item {tot/nb:.0f}: " + ", ".join(f"{nm} {dv[i]/nb:.0f} ({100*dv[i]/tot:.0f}%)" for i,nm in enumerate(names)), flush=True)

# ---- weight gradient (plain and with the fused BatchNorm backward) ------------------------------------------------------
names = ["loop", "mfma+loads", "barrier1", "commit", "barrier2"]
for name, cin, cout, (d,h,w) in shapes:
    x = torch.randn(n,d,h,w,cin, device="cuda").to(torch.bfloat16); y = torch.randn(n,d,h,w,cout, device="cuda").to(torch.bfloat16)
    dy = torch.randn(n,d,h,w,cout, device="cuda").to(torch.bfloat16)
    xfv = [torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1, torch.full((cin,), 0.1, device="cuda")]
    xfs = biu_xform(*[t.data_ptr() for t in xfv])
    kv = [torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda"), torch.full((cout,), 0.1, device="cuda"), torch.ones(cout, device="cuda"),
          torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")]
    ax = biu_act(x.data_ptr(),n,d,h,w,cin,cin); ay = biu_act(y.data_ptr(),n,d,h,w,cout,cout); ady = biu_act(dy.data_ptr(),n,d,h,w,cout,cout)
    ws = torch.empty(lib.biu_conv_bwd_weight_workspace(cin,cout,3,3,3,1), dtype=torch.uint8, device="cuda")
    dw = torch.empty(cout,cin,3,3,3, device="cuda")
    for tag, call in (("wgrad", lambda: lib.biu_conv_bwd_weight(C.byref(ax),C.byref(xfs),C.byref(ady),3,3,3,1,P(dw),None,P(ws),ws.numel(),1,st)),
                      ("wgrad_bn", lambda: lib.biu_conv_bwd_weight_bn(C.byref(ax),C.byref(xfs),C.byref(ady),C.byref(ay),P(kv[0]),P(kv[1]),P(kv[2]),P(kv[3]),P(kv[4]),P(kv[5]),3,3,3,1,P(dw),P(ws),ws.numel(),1,st))):
        call(); torch.cuda.synchronize(); diag.zero_()
        e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        dv = diag.cpu().tolist(); nb = max(dv[7],1); tot = sum(dv[:5])
        print(f"{name} {tag}: {e0.elapsed_time(e1):.3f} ms, clock {dv[8]/max(dv[9],1)*0.1:.2f} GHz, bricks/block-sum {nb}, cycles/brick {tot/nb:.0f}: " + ", ".join(f"{nm} {dv[i]/nb:.0f} ({100*dv[i]/tot:.0f}%)" for i,nm in enumerate(names)), flush=True)
