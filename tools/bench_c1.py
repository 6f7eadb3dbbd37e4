"""Micro-benchmark of the Cin = 1 first-layer kernels at cfg4's extent (4 x 128^3, 16 output channels, bf16)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bio_image_unet_amd._lib import BIU_BF16, biu_act, check, lib
n, d, h, w, co = 4, 128, 128, 128, 16
x = torch.rand(n, d, h, w, 1, device="cuda").bfloat16()
dy = torch.randn(n, d, h, w, co, device="cuda").bfloat16()
y = torch.empty_like(dy)
wt = torch.randn(co, 1, 3, 3, 3, device="cuda"); b = torch.randn(co, device="cuda"); dw = torch.empty_like(wt)
ax, ady, ay = biu_act(x.data_ptr(), n, d, h, w, 1, 1), biu_act(dy.data_ptr(), n, d, h, w, co, co), biu_act(y.data_ptr(), n, d, h, w, co, co)
wsz = lib.biu_conv_bwd_weight_workspace(1, co, 3, 3, 3, BIU_BF16)
ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fw = lambda: check(lib.biu_conv_fwd(C.byref(ax), None, p(wt), None, p(b), 3, 3, 3, 1, C.byref(ay), None, 0, BIU_BF16, st), "fwd")
wg = lambda: check(lib.biu_conv_bwd_weight(C.byref(ax), None, C.byref(ady), 3, 3, 3, 1, p(dw), None, p(ws), wsz, BIU_BF16, st), "wgrad")
bytes_f = n * d * h * w * (1 + co) * 2
print(f"c1 fwd   {timeit(fw):.3f} ms  ({bytes_f / timeit(fw) / 1e6:.0f} GB/s)   wgrad {timeit(wg):.3f} ms ({bytes_f / timeit(wg) / 1e6:.0f} GB/s)  BIU_C1_TPW={os.environ.get('BIU_C1_TPW')}")
