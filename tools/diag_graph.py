"""Debug aid: a graphed step against eager twins, per step (see tests/test_gpu_graph.py):  python tools/diag_graph.py [bf16|f32]
Prints the relative difference of all gradients together, graph vs eager twin A and eager twin A vs eager twin B (the run-to-run floor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bio_image_unet_amd as B
from bio_image_unet_amd.graph import GraphedTrainStep
from bio_image_unet_amd.losses import BCEDiceLoss
from bio_image_unet_amd.optim import Adam
from oracle import unet_oracle as O

dtypes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16"]
def run(dtype):
    sd = O.init_unet2d(1, 1, 16, seed=4)
    crit = BCEDiceLoss(0.5, 0.5)
    def make():
        m = B.Unet(1, 1, 16).cuda(); m.load_state_dict(sd)
        if dtype == "bf16":
            m.set_compute_dtype(torch.bfloat16)
        return m.train()
    m, ta, tb = make(), make(), make()
    opt = Adam(m.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(11)
    data = [(torch.rand(2, 1, 64, 64, generator=g).cuda(), (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float().cuda()) for _ in range(5)]
    lossf = (lambda outs, y: crit(outs[1][0], y[0]) + crit(outs[1][1], y[1])) if os.environ.get("DIAG_INDEXED") else (lambda outs, y: crit(outs[1], y))
    gstep = GraphedTrainStep(m, lossf, opt, [data[0][0]], [data[0][1]])
    print("node kinds:", gstep.node_kinds)
    def eager(t, x, y):
        l = lossf(t(x), y); t.zero_grad(set_to_none=True); l.backward(); return float(l)
    def dist(a, b):
        num = den = 0.0
        pb = dict(b.named_parameters())
        for n, p in a.named_parameters():
            if ".0.bias" in n and "final" not in n:
                continue
            num += float((p.grad - pb[n].grad).double().pow(2).sum()); den += float(pb[n].grad.double().pow(2).sum())
        return (num / den) ** 0.5
    for i, (x, y) in enumerate(data):
        ta.load_state_dict(m.state_dict()); tb.load_state_dict(m.state_dict())
        lg = float(gstep([x], [y]))
        la, lb = eager(ta, x, y), eager(tb, x, y)
        if dist(m, ta) > 1e-2 or dist(ta, tb) > 1e-2:
            pa, pb = dict(ta.named_parameters()), dict(tb.named_parameters())
            for n, p in m.named_parameters():
                r = lambda u, v: float((u - v).norm() / (v.norm() + 1e-30))
                ga, gb, ab = r(p.grad, pa[n].grad), r(p.grad, pb[n].grad), r(pa[n].grad, pb[n].grad)
                if max(ga, gb, ab) > 1e-2:
                    print(f"      {n:26s} graph-vs-A {ga:.2e} graph-vs-B {gb:.2e} A-vs-B {ab:.2e}")
        print(f"step {i}: loss graph {lg:.6f} eagerA {la:.6f} eagerB {lb:.6f} | grads graph-vs-A {dist(m, ta):.3e}  A-vs-B {dist(ta, tb):.3e}")

for dt in dtypes:
    print("====", dt)
    run(dt)
