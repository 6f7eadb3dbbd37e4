"""Run-to-run check of the eager step: two models with the same weights take the same batch; all gradients together must agree to
rounding.  Counts the steps where they do not (python tools/diag_repro.py [bf16|f32] [steps] [graph])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bio_image_unet_amd as B
from bio_image_unet_amd.losses import BCEDiceLoss
from bio_image_unet_amd.optim import Adam
from oracle import unet_oracle as O

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
with_graph = len(sys.argv) > 3
sd = O.init_unet2d(1, 1, 16, seed=4)
crit = BCEDiceLoss(0.5, 0.5)
def make():
    m = B.Unet(1, 1, 16).cuda(); m.load_state_dict(sd)
    if dtype == "bf16":
        m.set_compute_dtype(torch.bfloat16)
    return m.train()
ta, tb = make(), make()
opt = Adam(ta.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(11)
if with_graph:                      # a captured graph replayed in between, as in tests/test_gpu_graph.py
    from bio_image_unet_amd.graph import GraphedTrainStep
    mg = make()
    og = Adam(mg.parameters(), lr=1e-3)
    x0 = torch.rand(2, 1, 64, 64, generator=g).cuda(); y0 = (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float().cuda()
    gstep = GraphedTrainStep(mg, lambda outs, y: crit(outs[1], y), og, [x0], [y0])
bad = 0
for i in range(steps):
    x = torch.rand(2, 1, 64, 64, generator=g).cuda(); y = (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float().cuda()
    tb.load_state_dict(ta.state_dict())
    if with_graph:
        float(gstep([x], [y]))
    la = crit(ta(x)[1], y); ta.zero_grad(set_to_none=True); la.backward()
    lb = crit(tb(x)[1], y); tb.zero_grad(set_to_none=True); lb.backward()
    num = den = 0.0
    pb = dict(tb.named_parameters())
    for n, p in ta.named_parameters():
        num += float((p.grad - pb[n].grad).double().pow(2).sum()); den += float(pb[n].grad.double().pow(2).sum())
    d = (num / den) ** 0.5
    if d > 1e-2:
        bad += 1
        print(f"step {i}: loss A {float(la):.6f} B {float(lb):.6f} all-gradient difference {d:.3e}")
    opt.step()
print(f"{dtype} BIU_DISABLE={os.environ.get('BIU_DISABLE', '')} graph={with_graph}: {bad} of {steps} steps differ by more than 1e-2")
