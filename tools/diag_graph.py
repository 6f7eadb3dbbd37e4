"""Debug aid: graphed step against an eager twin, per step and parameter (see tests/test_gpu_graph.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bio_image_unet_amd as B
from bio_image_unet_amd.graph import GraphedTrainStep
from bio_image_unet_amd.losses import BCEDiceLoss
from bio_image_unet_amd.optim import Adam
from oracle import unet_oracle as O

sd = O.init_unet2d(1, 1, 16, seed=4)
crit = BCEDiceLoss(0.5, 0.5)
def make():
    m = B.Unet(1, 1, 16).cuda(); m.load_state_dict(sd); return m.train()
m, twin = make(), make()
opt = Adam(m.parameters(), lr=1e-3)
g = torch.Generator().manual_seed(11)
data = [(torch.rand(2, 1, 64, 64, generator=g).cuda(), (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float().cuda()) for _ in range(3)]
gstep = GraphedTrainStep(m, lambda outs, y: crit(outs[1], y), opt, [data[0][0]], [data[0][1]])
for i, (x, y) in enumerate(data):
    twin.load_state_dict(m.state_dict())
    lg = float(gstep([x], [y]))
    torch.cuda.synchronize()
    snap = {n: p.grad.clone() for n, p in m.named_parameters()}
    print("   after replay: max |grad|", max(float(v.abs().max()) for v in snap.values()))
    out_t = twin(x)[1]
    torch.cuda.synchronize()
    ch = [n for n, p in m.named_parameters() if not torch.equal(p.grad, snap[n])]
    print("   grads of the graphed model changed by the twin's FORWARD:", ch)
    le = crit(out_t, y); twin.zero_grad(set_to_none=True); le.backward()
    torch.cuda.synchronize()
    ch = [n for n, p in m.named_parameters() if not torch.equal(p.grad, snap[n])]
    print("   ... after the twin's BACKWARD:", ch)
    segs = torch.cuda.memory_snapshot()
    for n in ch[:3]:
        a = dict(m.named_parameters())[n].grad.data_ptr()
        for sg in segs:
            if sg["address"] <= a < sg["address"] + sg["total_size"]:
                print(f"   {n}: grad ptr {a:x} in segment {sg['address']:x} size {sg['total_size']} pool {sg.get('segment_pool_id')} stream {sg.get('stream')}")
    print(f"step {i}: loss graph {lg:.6f} eager {float(le):.6f}")
    pt = dict(twin.named_parameters())
    for n, p in m.named_parameters():
        a, b = p.grad, pt[n].grad
        fa, fb = bool(torch.isfinite(a).all()), bool(torch.isfinite(b).all())
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        if not (fa and fb) or rel > 1e-3:
            print(f"   {n:28s} finite graph {fa} eager {fb}  |g| {float(a.abs().max()):.3e} |e| {float(b.abs().max()):.3e} rel {rel:.3e} shape {tuple(a.shape)} ptr {a.data_ptr():x}")

print("---- replays only, a fresh model")
m2 = make()
opt2 = Adam(m2.parameters(), lr=1e-3)
g2 = GraphedTrainStep(m2, lambda outs, y: crit(outs[1], y), opt2, [data[0][0]], [data[0][1]])
for i in range(6):
    x, y = data[i % 3]
    l = float(g2([x], [y]))
    worst = max(((float(p.grad.abs().max()), n) for n, p in m2.named_parameters()), key=lambda t: t[0])
    print(f"replay {i}: loss {l:.6f} max |grad| {worst[0]:.3e} ({worst[1]})")
    if i == 2:
        junk = [torch.randn(1 << 20, device="cuda") * 1e9 for _ in range(64)]      # eager allocations between replays
        torch.cuda.synchronize()
        del junk
