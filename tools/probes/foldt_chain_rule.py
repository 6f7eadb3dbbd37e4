"""CPU derivation check (fp64) of the chain rule of the folded ConvTranspose + concat + conv op: dW_conv, dW_T from the per-parity G[p][t],\nand the ConvT-bias term b_T[c] S_k of dW_conv, against torch autograd.      python tools/probes/foldt_chain_rule.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
torch.manual_seed(0)
n, cl, cup, cs, cout, sp = 1, 32, 32, 16, 32, (2, 3, 4)
hi = tuple(2*v for v in sp)
xa = torch.randn(n, cl, *sp, dtype=torch.float64, requires_grad=True)
sa = torch.randn(n, cs, *hi, dtype=torch.float64, requires_grad=True)
wt = torch.randn(cl, cup, 2, 2, 2, dtype=torch.float64, requires_grad=True)
bt = torch.randn(cup, dtype=torch.float64, requires_grad=True)
wc = torch.randn(cout, cup+cs, 3, 3, 3, dtype=torch.float64, requires_grad=True)
up = F.conv_transpose3d(xa, wt, bt, stride=2)
y = F.conv3d(torch.cat([up, sa], 1), wc, None, padding=1)
dy = torch.randn_like(y); dy = dy - dy.mean(dim=(0,2,3,4), keepdim=True)
y.backward(dy)
# G[p][t][co][ci] = sum_v dy[2v+p][co] x[v+t-1+p][ci]
xp = F.pad(xa.detach(), (1,1,1,1,1,1))
G = torch.zeros(8, 8, cout, cl, dtype=torch.float64)
D,H,W = sp
for p in range(8):
    pd,ph,pw = (p>>2)&1,(p>>1)&1,p&1
    dyp = dy[:, :, pd::2, ph::2, pw::2]
    for t in range(8):
        td,th,tw = (t>>2)&1,(t>>1)&1,t&1
        xs = xp[:, :, td+pd:td+pd+D, th+ph:th+ph+H, tw+pw:tw+pw+W]
        G[p,t] = torch.einsum('nodhw,ncdhw->oc', dyp, xs)
def tq(p,k):
    kk = (k//9, (k//3)%3, k%3); t=q=0
    for ax in range(3):
        pa = (p>>(2-ax))&1
        ta = (1 if kk[ax]>=1 else 0) if pa==0 else (1 if kk[ax]==2 else 0)
        t |= ta<<(2-ax); q |= ((pa+kk[ax]+1)&1)<<(2-ax)
    return t,q
dwc = torch.zeros(cout, cup, 27, dtype=torch.float64)
wtd = wt.detach().reshape(cl, cup, 8)
for k in range(27):
    for p in range(8):
        t,q = tq(p,k)
        dwc[:, :, k] += torch.einsum('ic,oi->oc', wtd[:, :, q], G[p,t])
ref = wc.grad[:, :cup].reshape(cout, cup, 27)
print("chain dW_conv up-half rel err:", float((dwc-ref).norm()/ref.norm()))
dwt = torch.zeros(cl, cup, 8, dtype=torch.float64)
wcd = wc.detach()[:, :cup].reshape(cout, cup, 27)
for p in range(8):
    for k in range(27):
        t,q = tq(p,k)
        dwt[:, :, q] += torch.einsum('oc,oi->ic', wcd[:, :, k], G[p,t])
print("chain dW_T rel err:", float((dwt-wt.grad.reshape(cl,cup,8)).norm()/wt.grad.norm()))
# bias term: dW_conv[co][c][k] += b_T[c] * S_k[co], S_k = sum of dy over voxels where tap k stays inside
D2,H2,W2 = hi
S = torch.zeros(27, cout, dtype=torch.float64)
for k in range(27):
    kd,kh,kw = k//9,(k//3)%3,k%3
    d0,d1 = (1 if kd==0 else 0), (D2-1 if kd==2 else D2)
    h0,h1 = (1 if kh==0 else 0), (H2-1 if kh==2 else H2)
    w0,w1 = (1 if kw==0 else 0), (W2-1 if kw==2 else W2)
    S[k] = dy[:, :, d0:d1, h0:h1, w0:w1].sum(dim=(0,2,3,4))
dwc2 = dwc + torch.einsum('c,ko->ock', bt.detach(), S)
print("with bias term:", float((dwc2-ref).norm()/ref.norm()))
