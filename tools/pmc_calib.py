"""FETCH_SIZE calibration on a known byte count in OUR access pattern (16-byte pieces per lane, 32 contiguous bytes per voxel row
and instruction): the bf16 ConvTranspose forward kernel reads its coarse input exactly once (one column block) and writes the 8x
larger output exactly once.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bio_image_unet_amd._lib import BIU_BF16, biu_act, check, lib
n, d, h, w, c = 4, 64, 64, 64, 64
x = torch.randn(n, d, h, w, c, device="cuda").bfloat16()
y = torch.empty(n, 2 * d, 2 * h, 2 * w, c, device="cuda", dtype=torch.bfloat16)
wt = torch.randn(c, c, 2, 2, 2, device="cuda") * 0.05; b = torch.randn(c, device="cuda")
pk = torch.empty(lib.biu_convt_packed_bytes(0, c, c, 2, BIU_BF16), dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
check(lib.biu_convt_pack(0, P(wt), c, c, 2, BIU_BF16, P(pk), st))
ax, ay = biu_act(x.data_ptr(), n, d, h, w, c, c), biu_act(y.data_ptr(), n, 2 * d, 2 * h, 2 * w, c, c)
for _ in range(5):
    check(lib.biu_convt_fwd(C.byref(ax), None, P(wt), P(pk), P(b), 2, C.byref(ay), BIU_BF16, st))
torch.cuda.synchronize()
print(f"convt fwd: reads {x.numel() * 2 / 1e9:.4f} GB, writes {y.numel() * 2 / 1e9:.4f} GB per launch (k_convt_all)")
