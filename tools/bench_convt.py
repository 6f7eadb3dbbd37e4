"""ConvTranspose k2 s2 legs at the cfg4 shapes (up3: 64->64 @64^3 -> 128^3, up2: 128->128 @32^3, up1: 256->256 @16^3), bf16, HIP events."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bio_image_unet_amd._lib import BIU_BF16, biu_act, check, lib
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
for name, n, s, c in (("up3", 4, 64, 64), ("up2", 4, 32, 128), ("up1", 4, 16, 256)):
    x = torch.randn(n, s, s, s, c, device="cuda").bfloat16()
    y = torch.empty(n, 2 * s, 2 * s, 2 * s, c, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(n, 2 * s, 2 * s, 2 * s, c, device="cuda").bfloat16()
    dx = torch.empty_like(x)
    wt = torch.randn(c, c, 2, 2, 2, device="cuda") * 0.05; b = torch.randn(c, device="cuda")
    pk0 = torch.empty(lib.biu_convt_packed_bytes(0, c, c, 2, BIU_BF16), dtype=torch.uint8, device="cuda")
    pk1 = torch.empty(lib.biu_convt_packed_bytes(1, c, c, 2, BIU_BF16), dtype=torch.uint8, device="cuda")
    check(lib.biu_convt_pack(0, P(wt), c, c, 2, BIU_BF16, P(pk0), st)); check(lib.biu_convt_pack(1, P(wt), c, c, 2, BIU_BF16, P(pk1), st))
    ax, ay = biu_act(x.data_ptr(), n, s, s, s, c, c), biu_act(y.data_ptr(), n, 2 * s, 2 * s, 2 * s, c, c)
    ady, adx = biu_act(dy.data_ptr(), n, 2 * s, 2 * s, 2 * s, c, c), biu_act(dx.data_ptr(), n, s, s, s, c, c)
    ws = torch.empty(max(lib.biu_convt_bwd_weight_workspace(c, c, 2, BIU_BF16), 16), dtype=torch.uint8, device="cuda")
    dw, db = torch.empty_like(wt), torch.empty_like(b)
    legs = {"fwd": lambda: lib.biu_convt_fwd(C.byref(ax), None, P(wt), P(pk0), P(b), 2, C.byref(ay), BIU_BF16, st),
            "dgrad": lambda: lib.biu_convt_bwd_data(C.byref(ady), P(wt), P(pk1), 2, C.byref(adx), 0, BIU_BF16, st),
            "wgrad": lambda: lib.biu_convt_bwd_weight(C.byref(ax), None, C.byref(ady), 2, P(dw), P(db), P(ws), ws.numel(), BIU_BF16, st)}
    gb = (x.numel() + y.numel()) * 2 / 1e9
    out = []
    for k, f in legs.items():
        for _ in range(2): check(f())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): check(f())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append(f"{k} {ms:.3f} ms ({gb / ms:.2f} TB/s)")
    print(f"{name} {c}->{c} @{s}^3: " + " | ".join(out), flush=True)
