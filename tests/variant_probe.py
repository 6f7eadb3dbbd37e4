"""Helper of test_gpu_variants: one train-mode forward/backward of a small network, parameter gradients dumped to a file.
Run as a subprocess so that BIU_DISABLE (read once per process by the library) can differ between runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bio_image_unet_amd as B  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402

which, out = sys.argv[1], sys.argv[2]
torch.manual_seed(0)
g = torch.Generator().manual_seed(5)
if which in ("unet3d_bf16", "unet3d_f32"):   # UNet3D(n_filter=32): decode6 forward and encode2 data gradient take the 16-row MFMA kernel
    m = B.UNet3D(1, 1, 32).cuda()
    m.load_state_dict(O.init_unet3d(1, 1, 32, seed=7))
    if which.endswith("bf16"):
        m.set_compute_dtype(torch.bfloat16)
    shape = (2, 1, 16, 32, 32)
elif which == "unet2d_bf16_n8":       # Unet(n_filter=32) bf16 on 8 images: the 64^2 and 32^2 levels take the 2-D form of the rolling-window weight gradient
    m = B.Unet(1, 1, 32).cuda()
    m.load_state_dict(O.init_unet2d(1, 1, 32, seed=3))
    m.set_compute_dtype(torch.bfloat16)
    shape = (8, 1, 64, 64)
elif which in ("mo3d_interp_bf16", "mo3d_interp_f32"):   # MultiOutputUnet3D(use_interpolation=True): nearest up-sampling + upN_conv, forward folded onto the coarse tensor
    heads = {"seg": {"channels": 1, "activation": None}}
    m = B.MultiOutputUnet3D(1, heads, 32, True).cuda()
    m.load_state_dict(O.init_mo3d(1, heads, 32, True, seed=9))
    if which.endswith("bf16"):
        m.set_compute_dtype(torch.bfloat16)
    shape = (2, 1, 32, 64, 64)        # (512 voxels per channel at the bottleneck: its BatchNorm statistics are not pure noise)
else:                                 # Unet(n_filter=32) fp32, 2 x 64 x 64: the 8x8 / 4x4 layers split over their input channels
    m = B.Unet(1, 1, 32).cuda()
    m.load_state_dict(O.init_unet2d(1, 1, 32, seed=3))
    shape = (2, 1, 64, 64)
m.train()
x = torch.rand(*shape, generator=g).cuda()
y = (torch.rand(*shape, generator=g) > 0.5).float().cuda()
outs = m(x)
if isinstance(outs, dict):            # single un-activated head: (activated, logits) like the other networks
    outs = (outs["seg"], outs["seg"])
loss = O.bce_dice_loss(outs[1], y)
loss.backward()
torch.save({"loss": float(loss.detach()), "logits": outs[1].detach().cpu(), **{k: p.grad.cpu() for k, p in m.named_parameters()}}, out)
