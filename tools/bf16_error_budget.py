"""Where does the bf16 engine's gradient error come from?  CPU-only experiment with the oracle.

Compares, per parameter tensor, against the fp64 gradient of the same network:
  (a) the bf16-storage emulation of the oracle (rounds exactly where the engine stores bf16 / packs MFMA operands);
  (b) variants with one class of rounding switched off, to attribute the error.
Run:  python tools/bf16_error_budget.py [unet2d|unet3d]
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_oracle as O

kind = sys.argv[1] if len(sys.argv) > 1 else "unet2d"
torch.manual_seed(0)
if kind == "unet2d":
    shape, nf = (2, 1, 128, 128), 16
    sd = O.init_unet2d(1, 1, nf, seed=3); fwd = O.unet2d_forward
else:
    shape, nf = (2, 1, 16, 32, 32), 32
    sd = O.init_unet3d(1, 1, nf, seed=3); fwd = O.unet3d_forward
x = torch.rand(*shape); y = (torch.rand(*shape) > 0.5).float()

def run(dt, emu):
    osd = O.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}, requires_grad=True)
    with O.emulate_bf16(emu):
        _, logits = fwd(osd, x.to(dt), training=True)
        loss = O.bce_dice_loss(logits, y.to(dt))
        g = O.grads_of(loss, osd)
    return logits.detach(), g

def errs(g, truth):
    gs = max(float(v.abs().max()) for v in truth.values())
    out = {}
    for k, w in truth.items():
        a = g[k].double(); w = w.double()
        e = float((a - w).abs().max()) / (float(w.abs().max()) + 1e-2 * gs)
        l2 = float((a - w).norm() / (w.norm() + 1e-300))
        cos = float((a * w).sum() / (a.norm() * w.norm() + 1e-300))
        out[k] = (e, l2, cos)
    return out

lt, gt = run(torch.float64, False)
l32, g32 = run(torch.float32, False)
le, ge = run(torch.float32, True)
print("logits: emu-vs-fp64 rel max err", float((le.double() - lt).abs().max() / lt.abs().max()))
ee = errs(ge, gt)
print("%-28s %9s %9s %9s" % ("param", "maxerr", "l2err", "cos"))
for k, (e, l2, c) in ee.items():
    if k.endswith(".0.bias") and not k.startswith("final"):
        continue
    print("%-28s %9.4f %9.4f %9.5f" % (k, e, l2, c))

def summary(tag, g):
    e = errs(g, gt)
    ks = [k for k in e if not (k.endswith(".0.bias") and not k.startswith("final"))]
    worst = max(ks, key=lambda k: e[k][1])
    import statistics
    print("%-34s median l2 %.4f  worst l2 %.4f (%s, cos %.4f)" % (tag, statistics.median(e[k][1] for k in ks), e[worst][1], worst, e[worst][2]))

print()
summary("fp32 CPU oracle", g32)
summary("bf16 emulation (engine)", ge)
O._Emu.fwd, O._Emu.bwd = True, False
summary("  values only (fp32 gradients)", run(torch.float32, True)[1])
O._Emu.fwd, O._Emu.bwd = False, True
summary("  gradients only (fp32 values)", run(torch.float32, True)[1])
O._Emu.fwd, O._Emu.bwd = True, True
# conditioning: fp64 arithmetic, every conv weight perturbed by 2^-9 relative noise (one bf16 rounding of the weights alone)
osd = {k: v.clone() for k, v in sd.items()}
g_ = torch.Generator().manual_seed(11)
for k in osd:
    if k.endswith(".0.weight"):
        osd[k] = osd[k] * (1 + 2.0 ** -9 * (torch.rand(osd[k].shape, generator=g_) * 2 - 1))
sd_keep, sd = sd, osd
summary("fp64, conv weights * (1 + U(-2^-9, 2^-9))", run(torch.float64, False)[1])
sd = sd_keep
