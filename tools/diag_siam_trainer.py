import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bio_image_unet_amd as B
from oracle import unet_oracle as O
torch.manual_seed(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "max"
m = B.Siam_UNet(8, mode).cuda()
sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(0)
x, px = torch.rand(2, 1, 32, 32, generator=g), torch.rand(2, 1, 32, 32, generator=g)
y = (torch.rand(2, 1, 32, 32, generator=g) > 0.5).float()
osd = O.clone_state(sd0, requires_grad=True)
_, ol = O.siam_forward(osd, x, px, mode=mode, training=True)
for lossname in ("siam", "plain"):
    for v in osd.values():
        if v.requires_grad: v.grad = None
    _, ol = O.siam_forward(O.clone_state(sd0, requires_grad=False) if False else osd, x, px, mode=mode, training=True)
    ol_loss = O.siam_bce_dice_loss(ol, y, 1.0, 1.0) if lossname == "siam" else O.bce_dice_loss(ol, y, 1.0, 1.0)
    ol_loss.backward()
    m.load_state_dict(sd0); m.zero_grad(); m.train()
    _, l = m(x.cuda(), px.cuda())
    from bio_image_unet_amd.losses import BCEDiceLossSiam, BCEDiceLoss
    loss = (BCEDiceLossSiam(1, 1) if lossname == "siam" else BCEDiceLoss(1, 1))(l, y.cuda())
    loss.backward()
    print(lossname, "loss", float(loss), float(ol_loss), "logits maxdiff", float((l.detach().cpu() - ol.detach()).abs().max()))
    worst = []
    for k, p in m.named_parameters():
        if k.endswith(".0.bias") and not k.startswith("final"): continue
        a, b_ = p.grad.cpu(), osd[k].grad
        worst.append((float((a - b_).abs().max() / (b_.abs().max() + 1e-12)), k))
    worst.sort(reverse=True)
    print("   worst grad rel err:", worst[:6])
