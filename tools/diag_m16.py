"""Per-parameter gradient of one bf16 train step of Unet(1,1,16) at batch 4, dumped to a file: run twice (BIU_DISABLE=m16 / not) and diff."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bio_image_unet_amd as B
from oracle import unet_oracle as O
out = sys.argv[1]
torch.manual_seed(0)
m = B.Unet(1, 1, 16).cuda()
m.load_state_dict(O.init_unet2d(1, 1, 16, seed=3))
m.set_compute_dtype(torch.bfloat16)
m.train()
g = torch.Generator().manual_seed(5)
x = torch.rand(4, 1, 64, 64, generator=g).cuda()
y = (torch.rand(4, 1, 64, 64, generator=g) > 0.5).float().cuda()
loss = O.bce_dice_loss(m(x)[1], y)
loss.backward()
torch.save({"loss": float(loss), **{k: p.grad.cpu() for k, p in m.named_parameters()}}, out)
if len(sys.argv) > 2:
    a, b = torch.load(sys.argv[2]), torch.load(out)
    print("loss", a["loss"], b["loss"])
    for k in a:
        if k == "loss":
            continue
        d = float((a[k] - b[k]).norm() / (a[k].norm() + 1e-30))
        if d > float(os.environ.get("DIFF_THR", "1e-3")):
            print(f"{k:40s} rel diff {d:.4f}")
