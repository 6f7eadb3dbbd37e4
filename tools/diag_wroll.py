"""BIU_DIAG build: where a step of the rolling-window weight gradient (k_wgrad_roll) spends its cycles, seen by one wave of every block
(-DBIU_DIAG_WAVE=w picks the wave).   python tools/diag_wroll.py [lib]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_image_unet_amd._lib import biu_act, biu_xform, SIGNATURES
lib = C.CDLL(os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "tools/variants/libbiu_diag.so"))
for name, (res, args) in SIGNATURES.items():
    getattr(lib, name).restype = res; getattr(lib, name).argtypes = args
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
diag = torch.zeros(10, dtype=torch.int64, device="cuda")
C.c_void_p.in_dll(lib, "biu_diag_buffer").value = diag.data_ptr()
names = ["loop", "fetch issue", "mfma", "wait vmcnt", "finish", "barrier"]
for name, cin, cout, (d, h, w) in [("decode5", 96, 32, (128, 128, 128)), ("decode3", 192, 64, (64, 64, 64)), ("decode6", 32, 16, (128, 128, 128))]:
    n = 4
    x = torch.randn(n, d, h, w, cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(n, d, h, w, cout, device="cuda").to(torch.bfloat16)
    ax = biu_act(x.data_ptr(), n, d, h, w, cin, cin); ady = biu_act(dy.data_ptr(), n, d, h, w, cout, cout)
    ws = torch.empty(lib.biu_conv_bwd_weight_workspace(cin, cout, 3, 3, 3, 1), dtype=torch.uint8, device="cuda")
    dw = torch.empty(cout, cin, 3, 3, 3, device="cuda")
    call = lambda: lib.biu_conv_bwd_weight(C.byref(ax), None, C.byref(ady), 3, 3, 3, 1, P(dw), None, P(ws), ws.numel(), 1, st)
    call(); torch.cuda.synchronize(); diag.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    dv = diag.cpu().tolist(); nb = max(dv[7], 1); tot = sum(dv[:6])
    print(f"{name} wgrad (no transform): {e0.elapsed_time(e1):.3f} ms, clock {dv[8] / max(dv[9], 1) * 0.1:.2f} GHz, steps {nb}, cycles/step {tot / nb:.0f}: " +
          ", ".join(f"{nm} {dv[i] / nb:.0f} ({100 * dv[i] / tot:.0f}%)" for i, nm in enumerate(names)), flush=True)
