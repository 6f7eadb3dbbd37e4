"""GPU diagnostic: per-parameter gradient error of the HIP engine against the oracle.

  fp32: HIP vs the fp64 oracle under several BIU_DISABLE settings (which kernel family moves the error?)
  bf16: HIP vs the bf16-storage emulation of the oracle (same rounding points) and vs fp64
Usage: python tools/diag_grad.py [unet2d|unet3d] > gpurun_out/diag_grad.txt
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bio_image_unet_amd as B
from oracle import unet_oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "unet2d"
torch.manual_seed(0)
if kind == "unet2d":
    shape, nf = (2, 1, 128, 128), 16
    sd = O.init_unet2d(1, 1, nf, seed=3); fwd = O.unet2d_forward; mk = lambda: B.Unet(1, 1, nf)
else:
    shape, nf = (2, 1, 16, 32, 32), 32
    sd = O.init_unet3d(1, 1, nf, seed=3); fwd = O.unet3d_forward; mk = lambda: B.UNet3D(1, 1, nf)
x = torch.rand(*shape); y = (torch.rand(*shape) > 0.5).float()


def oracle(dt, emu):
    osd = O.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}, requires_grad=True)
    with O.emulate_bf16(emu):
        _, logits = fwd(osd, x.to(dt), training=True)
        loss = O.bce_dice_loss(logits, y.to(dt))
        g = O.grads_of(loss, osd)
    return logits.detach(), g


def hip(dtype):
    m = mk().cuda()
    m.load_state_dict(sd)
    if dtype == "bf16":
        m.set_compute_dtype(torch.bfloat16)
    m.train()
    _, logits = m(x.cuda())
    loss = O.bce_dice_loss(logits, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    return logits.detach().cpu(), {k: p.grad.cpu() for k, p in m.named_parameters()}


def errs(g, truth):
    gs = max(float(v.abs().max()) for v in truth.values())
    out = {}
    for k, w in truth.items():
        if k.endswith(".0.bias") and not k.startswith("final"):
            continue
        a = g[k].double(); w = w.double()
        out[k] = (float((a - w).abs().max()) / (float(w.abs().max()) + 1e-2 * gs), float((a - w).norm() / (w.norm() + 1e-300)),
                  float((a * w).sum() / (a.norm() * w.norm() + 1e-300)))
    return out


def show(tag, e, n=6):
    top = sorted(e.items(), key=lambda kv: -kv[1][0])[:n]
    print(f"{tag}: worst max-err " + ", ".join(f"{k} {v[0]:.2e}" for k, v in top))


lt, gt = oracle(torch.float64, False)
l32, g32 = oracle(torch.float32, False)
show("CPU fp32 oracle vs fp64", errs(g32, gt))
for dis in ["", "fused_stats", "dgrad_bnred", "wgrad_bn", "conv_fwd", "conv_dgrad", "conv_wgrad", "convt_fwd,convt_dgrad,convt_wgrad",
            "conv_fwd,conv_dgrad", "fused_stats,dgrad_bnred,wgrad_bn"]:
    os.environ["BIU_DISABLE"] = dis
    lh, gh = hip("f32")
    print(f"[fp32 BIU_DISABLE='{dis}'] logits err {float((lh.double() - lt).abs().max() / lt.abs().max()):.2e}")
    show("   HIP fp32 vs fp64", errs(gh, gt))
os.environ["BIU_DISABLE"] = ""
le, ge = oracle(torch.float32, True)
lh, gh = hip("bf16")
print(f"[bf16] logits: HIP vs emulation {float((lh - le).abs().max() / le.abs().max()):.2e}; emulation vs fp64 {float((le.double() - lt).abs().max() / lt.abs().max()):.2e}")
print(f"[bf16] masks differ from emulation at {int(((lh > 0) != (le > 0)).sum())} of {lh.numel()} voxels")
show("   HIP bf16 vs emulation", errs(gh, ge), 10)
show("   HIP bf16 vs fp64     ", errs(gh, gt), 4)
show("   emulation vs fp64    ", errs(ge, gt), 4)
e = errs(gh, ge)
print("   HIP bf16 vs emulation: max l2 %.3e, min cos %.6f" % (max(v[1] for v in e.values()), min(v[2] for v in e.values())))
