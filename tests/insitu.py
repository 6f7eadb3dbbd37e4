"""Teacher-forced ("in situ") verification of every kernel call of a whole-network step.

A whole-network comparison against the oracle cannot be sharp: LeakyReLU / max-pool decisions at |t| ~ 1e-7 flip between
any two fp32 implementations (one flipped voxel moves a bottleneck weight gradient by 1e-3 .. 1e-2), and in bf16 mode the
rounding itself is chaotic -- two correct bf16 implementations with different fp32 summation orders decorrelate to the
bf16 noise level within a few layers (tools/bf16_error_budget.py, profiles/r02_bf16_error_budget.md).

So this checker follows the engine node by node: BEFORE a node runs it snapshots the node's actual operands as the engine
holds them in HBM (stored activations, their consumer transforms, stored gradients), AFTER the node ran it recomputes the
node's outputs from exactly those operands with torch CPU ops (float64) -- rounding to bf16 where the engine stores or packs
bf16 -- and compares them with what the kernel wrote.  Every conv / ConvT / pool / head / join kernel, every fused
BatchNorm-statistics, BatchNorm-backward-sums and BatchNorm-backward-apply path and both members of every two-source
(concat) call is checked on the shapes and fusions of the real network, to ~1 ulp of the storage type.

Test infrastructure only (uses Engine.trace, a hook the product never sets).
"""
import math

import torch
import torch.nn.functional as F

from bio_image_unet_amd import engine as E
from bio_image_unet_amd._lib import lib

LRELU = E.LRELU_SLOPE


def _r(t, bf16):
    return t.to(torch.bfloat16).to(t.dtype) if bf16 else t


def stored(act_or_buf_t, c0=None, c=None):
    """[N,D,H,W,C] device tensor slice -> NCDHW float64 on the CPU."""
    t = act_or_buf_t
    if c0 is not None:
        t = t[..., c0:c0 + c]
    return t.detach().float().cpu().permute(0, 4, 1, 2, 3).contiguous().double()


def act_raw(a):
    return stored(a.buf.t, a.c0, a.c)


def act_grad(a):
    return stored(a.buf.grad(), a.c0, a.c)


def xf_vectors(a):
    """(scale, shift, slope) of an Act as float32 CPU vectors, or None when the transform is the identity."""
    if not a.is_lazy():
        return None
    return tuple(a.vec(n).detach().float().cpu() for n in ("scale", "shift", "slope"))


def apply_xf(raw, xf):
    """T(v) = lrelu_slope(scale*v + shift) evaluated like the kernels do: fp32 fused multiply-add, then the select."""
    if xf is None:
        return raw
    sc, sh, sl = (v.view(1, -1, 1, 1, 1).double() for v in xf)
    t = (sc * raw + sh).float().double()                # one fp32 rounding of the fma result (operands are exact in fp64)
    return torch.where(t > 0, t, (sl * t).float().double())


def act_T(a):
    return apply_xf(act_raw(a), xf_vectors(a))


def dact(raw, xf):
    """T'(v): 1 where scale*v+shift > 0 else slope."""
    sc, sh, sl = (v.view(1, -1, 1, 1, 1).double() for v in xf)
    t = (sc * raw + sh).float().double()
    return torch.where(t > 0, torch.ones_like(t), sl.expand_as(t)), t


class Report:
    def __init__(self):
        self.rows = []          # (label, what, worst normalised deviation, fraction within tolerance)
        self.n_out = 0          # elements further than 1x their tolerance, over all element-wise checks
        self.n_elem = 0         # elements compared
        self.outliers = []      # (label, what, count beyond 1x, count, worst) of the checks that had any

    def add(self, label, what, dev, frac, n_out=0, n=0):
        self.rows.append((label, what, dev, frac))
        self.n_out += n_out
        self.n_elem += n
        if n_out:
            self.outliers.append((label, what, n_out, n, dev))

    def worst(self):
        return max(self.rows, key=lambda r: r[2]) if self.rows else None

    def table(self):
        return "\n".join(f"{l:28s} {w:22s} worst {d:9.3e}   within tol {f:8.6f}" for l, w, d, f in self.rows)


class InSitu:
    """eng.trace callback.  ``bf16``: the engine stores activations / gradients in bf16."""

    def __init__(self, eng, bf16: bool, strict_frac=0.999):
        self.eng, self.bf16, self.rep = eng, bf16, Report()
        self.snap = {}
        self.failures = []
        self.strict_frac = strict_frac
        # one storage ulp: 2^-8 relative for bf16 (value = 1.xxxxxxx * 2^e, 7 fraction bits), fp32 accumulation noise otherwise
        self.rel = 2.0 ** -7 if bf16 else 2e-5
        eng.trace = self

    # ---- comparison ------------------------------------------------------------------------------------------------
    def close(self, label, what, got, want, rel=None, abs_frac=None, ignore=None, frac=None):
        """|got - want| <= rel*|want| + abs_frac*rms(want) for at least `frac` of the elements (default strict_frac), and no
        element further than 64x that (a 1-ulp flip after a different fp32 summation order is inside the tolerance; the
        rare element whose LeakyReLU / max decision sits exactly on the boundary is inside `frac`).  `ignore`: mask of
        elements excluded (decision boundary closer than the kernel's own rounding)."""
        rel = self.rel if rel is None else rel
        got, want = got.double(), want.double()
        rms = float(want.pow(2).mean().sqrt()) + 1e-30
        abs_ = (abs_frac if abs_frac is not None else (2.0 ** -8 if self.bf16 else 1e-5)) * rms
        tol = rel * want.abs() + abs_
        dev = (got - want).abs() / tol
        if ignore is not None:
            dev = torch.where(ignore, torch.zeros_like(dev), dev)
        frac_ok = float((dev <= 1.0).double().mean())
        worst = float(dev.max()) if dev.numel() else 0.0
        self.rep.add(label, what, worst, frac_ok, int((dev > 1.0).sum()), dev.numel())
        need = self.strict_frac if frac is None else frac
        if not (frac_ok >= need and worst <= 64.0) or not torch.isfinite(got).all():
            self.failures.append(f"{label}: {what}: only {frac_ok:.6f} of the elements within tolerance (need {need}), worst {worst:.1f}x")

    def vec_close(self, label, what, got, want, rel=1e-4, floor=1e-6):
        got, want = got.double().flatten(), want.double().flatten()
        scale = float(want.abs().max()) + 1e-30
        dev = float(((got - want).abs() / (rel * want.abs() + floor * scale + 1e-30)).max())
        self.rep.add(label, what, dev, 1.0 if dev <= 1 else 0.0)
        if dev > 1.0 or not torch.isfinite(got).all():
            self.failures.append(f"{label}: {what}: deviation {dev:.2f}x tolerance (rel {rel})")

    # ---- helpers ---------------------------------------------------------------------------------------------------
    def _mfma(self, cin, cout, dil=1, k=3):
        if cin == 1 and self.bf16 and dil == 1 and k == 3 and cout >= 16 and cout % 16 == 0:
            return True                     # first layer through the im2col MFMA kernels (csrc/biu_c1.hip): bf16 weights
        return dil == 1 and k != 1 and cin >= 16 and cin % (16 if self.bf16 else 8) == 0 and cout >= 16 and cout % 8 == 0

    def _conv(self, x, w, b, dil, transposed=False):
        nd3 = w.dim() == 5
        if not nd3:
            x = x.squeeze(2)
        if transposed:
            y = (F.conv_transpose3d if nd3 else F.conv_transpose2d)(x, w, b, stride=2)
        else:
            y = (F.conv3d if nd3 else F.conv2d)(x, w, b, padding=dil if w.shape[-1] == 3 else 0, dilation=dil)
        return y if nd3 else y.unsqueeze(2)

    def _fold_conv(self, xc, w32, b):
        """The folded up-conv as the kernel computes it (include/biu.h: biu_upconv_fwd): per output parity p the taps of the 3x3x3 kernel
        that read the same coarse voxel are summed in fp32 (ascending tap order, as k_pack_upconv does), rounded to the compute dtype,
        and applied as a 2x2x2 kernel to the zero-padded coarse tensor xc: y[2v + p] = sum_t W'[p][t] xc[v + t - 1 + p]."""
        n, c, d, h, w_ = xc.shape
        cout = w32.shape[0]
        xp = F.pad(xc, (1, 1, 1, 1, 1, 1))
        y = torch.zeros(n, cout, 2 * d, 2 * h, 2 * w_, dtype=torch.float64)
        cls = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}            # (parity, coarse tap) -> fine taps of one axis
        for pd in range(2):
            for ph in range(2):
                for pw in range(2):
                    k = torch.zeros(cout, c, 2, 2, 2, dtype=torch.float32)
                    for td in range(2):
                        for th in range(2):
                            for tw in range(2):
                                acc = torch.zeros(cout, c, dtype=torch.float32)
                                for kd in cls[(pd, td)]:
                                    for kh in cls[(ph, th)]:
                                        for kw in cls[(pw, tw)]:
                                            acc = acc + w32[:, :, kd, kh, kw]
                                k[:, :, td, th, tw] = acc
                    k = _r(k, self.bf16).double()
                    y[:, :, pd::2, ph::2, pw::2] = F.conv3d(xp[:, :, pd:pd + d + 1, ph:ph + h + 1, pw:pw + w_ + 1], k, b)
        return y

    def _foldt_fwd(self, nd):
        """ConvTranspose + concat + conv as biu_foldt_fwd computes it: composed weights W'[p][t] = sum_k sum_c W_conv[.][c][k] W_T[.][c][q(p,k)]
        (fp32, rounded to the compute dtype once), the skip half + biases first (stored), the border shell corrected (stored), the fold
        accumulated on top (stored)."""
        ct = nd.foldt
        lo, skip = ct.xin, nd.xin.parts[1]
        cup = ct.y.c
        wc = nd.conv.weight.detach().cpu().float()
        wt = ct.up.weight.detach().cpu().float()
        bt = ct.up.bias.detach().cpu().double()
        bc = nd.conv.bias.detach().cpu().double() if nd.conv.bias is not None else torch.zeros(wc.shape[0], dtype=torch.float64)
        xl = _r(act_T(lo).float(), self.bf16).double()
        xs = _r(act_T(skip).float(), self.bf16).double()
        cout, cl = wc.shape[0], wt.shape[0]
        wb = torch.einsum("ocdhw,c->odhw", wc[:, :cup].double(), bt)                       # Wb[co][k]
        form = lib.biu_foldt_fwd_form(lo.a(), skip.a(), nd.y.a(), nd.y.buf.eng.dtype)     # 1: rolling-window kernels (the fold is stored first)
        y1 = F.conv3d(xs, _r(wc[:, cup:], self.bf16).double(), None if form == 1 else (bc + wb.sum(dim=(1, 2, 3))).float().double(), padding=1)
        if form != 1:
            y1 = _r(y1.float(), self.bf16).double()
        # taps outside the tensor carry no ConvT bias: subtract them on the border shell
        n, _, d2, h2, w2 = y1.shape
        ones = torch.ones(1, 1, d2, h2, w2, dtype=torch.float64)
        inside = torch.stack([F.conv3d(ones, torch.eye(27, dtype=torch.float64)[k].view(1, 1, 3, 3, 3), padding=1)[0, 0] for k in range(27)])   # [27][d][h][w]
        fix = torch.einsum("ok,kdhw->odhw", wb.reshape(cout, 27), 1.0 - inside)
        if form != 1:
            y1 = _r((y1 - fix.unsqueeze(0)).float(), self.bf16).double()
        cls = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
        xp = F.pad(xl, (1, 1, 1, 1, 1, 1))
        d, h, w_ = xl.shape[2:]
        fold = torch.zeros_like(y1)
        wtv = wt.reshape(cl, cup, 8)
        for pd in range(2):
            for ph in range(2):
                for pw in range(2):
                    k8 = torch.zeros(cout, cl, 2, 2, 2, dtype=torch.float32)
                    for td in range(2):
                        for th in range(2):
                            for tw in range(2):
                                acc = torch.zeros(cout, cl, dtype=torch.float32)
                                for kd in cls[(pd, td)]:
                                    for kh in cls[(ph, th)]:
                                        for kw in cls[(pw, tw)]:
                                            q = ((pd + kd + 1) & 1) * 4 + ((ph + kh + 1) & 1) * 2 + ((pw + kw + 1) & 1)
                                            acc = acc + wc[:, :cup, kd, kh, kw] @ wtv[:, :, q].t()
                                k8[:, :, td, th, tw] = acc
                    fold[:, :, pd::2, ph::2, pw::2] = F.conv3d(xp[:, :, pd:pd + d + 1, ph:ph + h + 1, pw:pw + w_ + 1], _r(k8, self.bf16).double())
        if form == 1:
            # the fold with the border-state bias (an fp32 table: b_conv + sum_k Wb[k] - the Wb[k] of the taps outside) is the stored
            # intermediate; the skip half is added in fp32 and the sum rounded once
            base = (bc + wb.sum(dim=(1, 2, 3))).float().double().view(1, -1, 1, 1, 1) - fix.float().double().unsqueeze(0)
            return _r((fold + base).float(), self.bf16).double() + y1
        return y1 + fold

    def _foldt_cat(self, nd, round_inputs):
        """The concat the folded op never materialises: (convT(T(x_low)) + b_T | T(skip)), the up half NOT rounded to the storage type."""
        ct = nd.foldt
        xl, xs = act_T(ct.xin), act_T(nd.xin.parts[1])
        if round_inputs:
            xl, xs = _r(xl.float(), True).double(), _r(xs.float(), True).double()
        return xl, xs

    def _parts(self, xin):
        return list(xin.parts) if isinstance(xin, E.CatAct) else [xin]

    def _T_cat(self, xin):
        return torch.cat([act_T(p) for p in self._parts(xin)], 1)

    def _gsnap(self, a):
        """gradient of an Act before a node accumulates into it (None when nothing has been written yet)."""
        return act_grad(a) if a.g_written() else None

    # ---- dispatcher ------------------------------------------------------------------------------------------------
    def __call__(self, phase, node, when):
        with torch.enable_grad():           # (autograd's backward runs with grad mode off; the checker differentiates torch ops)
            self._dispatch(phase, node, when)

    def _dispatch(self, phase, node, when):
        torch.cuda.synchronize()
        key = (phase, id(node) if not isinstance(node, tuple) else "heads")
        name = type(node).__name__ if not isinstance(node, tuple) else "Heads"
        fn = getattr(self, f"{phase}_{name}_{when}", None)
        if fn is None:
            return
        if when == "pre":
            self.snap[key] = fn(node)
        else:
            fn(node, self.snap.pop(key, None))

    # ================================================================================================================
    # forward
    # ================================================================================================================
    def fwd_ConvBlockNode_pre(self, nd):
        bn = nd.bn
        return dict(rm=bn.running_mean.detach().cpu().clone() if bn.running_mean is not None else None,
                    rv=bn.running_var.detach().cpu().clone() if bn.running_var is not None else None)

    def fwd_ConvBlockNode_post(self, nd, s):
        lab = nd.label + ":fwd"
        if getattr(nd, "foldt", None) is not None:
            want = self._foldt_fwd(nd)
            if not self.bf16:            # the folded op is the same function as ConvT -> concat -> conv
                ct = nd.foldt
                up = F.conv_transpose3d(act_T(ct.xin), ct.up.weight.detach().cpu().double(), ct.up.bias.detach().cpu().double(), stride=2)
                ref = F.conv3d(torch.cat([up, act_T(nd.xin.parts[1])], 1), nd.conv.weight.detach().cpu().double(),
                               nd.conv.bias.detach().cpu().double() if nd.conv.bias is not None else None, padding=1)
                assert float((want - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), "foldt emulation != ConvT -> concat -> conv"
            y = act_raw(nd.y)
            self.close(lab, "conv output y (ConvT + concat + conv folded)", y, want)
            self._check_bn_fwd(nd, s, y, lab)
            return
        a = F.interpolate(act_T(nd.fold_src), scale_factor=2, mode="nearest") if getattr(nd, "fold_all", False) else self._T_cat(nd.xin)
        w = nd.conv.weight.detach().cpu().double()
        b = nd.conv.bias.detach().cpu().double() if nd.conv.bias is not None else None
        mf = self._mfma(nd.xin.c, nd.y.c, nd.dil, nd.kw)
        if getattr(nd, "fold_src", None) is not None:
            # forward folded onto the coarse tensor: same function of the fp32 weights, but the weights are rounded AFTER the fold
            ac = act_T(nd.fold_src)
            want = self._fold_conv(_r(ac.float(), self.bf16).double(), nd.conv.weight.detach().cpu().float(), b)
            if not self.bf16:
                assert float((want - self._conv(a, w, b, nd.dil)).abs().max()) <= 1e-5 * float(want.abs().max()), "fold emulation != conv of the up-sampled tensor"
        else:
            if self.bf16 and mf:
                a, w = _r(a.float(), True).double(), _r(w.float(), True).double()
            want = self._conv(a, w, b, nd.dil)
        y = act_raw(nd.y)
        self.close(lab, "conv output y", y, want)
        self._check_bn_fwd(nd, s, y, lab)

    def _check_bn_fwd(self, nd, s, y, lab):
        bn = nd.bn
        cnt = y.numel() / y.shape[1]
        if nd.batch_stats:
            mean = y.mean(dim=(0, 2, 3, 4))
            var = y.var(dim=(0, 2, 3, 4), unbiased=False)
            invstd = 1.0 / torch.sqrt(var + bn.eps)
            g, be = bn.weight.detach().cpu().double(), bn.bias.detach().cpu().double()
            self.vec_close(lab, "BN scale", nd.y.vec("scale").cpu(), g * invstd, rel=2e-5)
            self.vec_close(lab, "BN shift", nd.y.vec("shift").cpu(), be - mean * g * invstd, rel=2e-5, floor=2e-5)
            self.vec_close(lab, "save_mean", nd.save_mean.cpu(), mean, rel=2e-5, floor=2e-6)
            self.vec_close(lab, "save_invstd", nd.save_invstd.cpu(), invstd, rel=2e-5)
            if s["rm"] is not None and bn.track_running_stats:
                mom = bn.momentum if bn.momentum is not None else 0.1
                self.vec_close(lab, "running_mean", bn.running_mean.cpu(), (1 - mom) * s["rm"].double() + mom * mean, rel=2e-5, floor=2e-6)
                self.vec_close(lab, "running_var", bn.running_var.cpu(), (1 - mom) * s["rv"].double() + mom * var * cnt / max(cnt - 1, 1), rel=2e-5)
        else:
            invstd = 1.0 / torch.sqrt(bn.running_var.detach().cpu().double() + bn.eps)
            g, be = bn.weight.detach().cpu().double(), bn.bias.detach().cpu().double()
            self.vec_close(lab, "BN scale (eval)", nd.y.vec("scale").cpu(), g * invstd, rel=2e-6)
            self.vec_close(lab, "BN shift (eval)", nd.y.vec("shift").cpu(), be - bn.running_mean.detach().cpu().double() * g * invstd, rel=2e-6, floor=2e-6)

    def fwd_ConvTNode_post(self, nd, s):
        if getattr(nd, "folded_into", None) is not None:       # ConvT + concat + conv run as one op (biu_foldt_*): checked at that conv block
            return
        a = act_T(nd.xin)
        w = nd.up.weight.detach().cpu().double()
        if self.bf16 and self._mfma(nd.xin.c, nd.y.c):
            a, w = _r(a.float(), True).double(), _r(w.float(), True).double()
        want = self._conv(a, w, nd.up.bias.detach().cpu().double(), 1, transposed=True)
        self.close(nd.label + ":fwd", "ConvT output", act_raw(nd.y), want)

    def _resample(self, kind, a, out_space):
        if kind == "maxpool":
            return F.max_pool3d(a, (2 if a.shape[2] > 1 else 1, 2, 2))
        if kind == "down":
            return a[:, :, ::2 if a.shape[2] > 1 else 1, ::2, ::2]
        sf = (2 if out_space[1] == 2 * a.shape[2] else 1, 2, 2)
        if kind == "up":
            return F.interpolate(a, scale_factor=sf, mode="nearest")
        if a.shape[2] == 1 and sf[0] == 1:
            return F.interpolate(a.squeeze(2), scale_factor=2, mode="bilinear", align_corners=False).unsqueeze(2)
        return F.interpolate(a, scale_factor=sf, mode="trilinear", align_corners=False)

    def fwd_ResampleNode_post(self, nd, s):
        if getattr(nd, "skip", False):             # its reader folds forward and backward onto the coarse tensor: nothing was computed
            return
        want = self._resample(nd.kind, act_T(nd.xin), nd.y.space)
        # stored result: one rounding of the exact value (the pool compares unrounded fp32 values)
        self.close(nd.label + ":fwd", nd.kind, act_raw(nd.y), want, rel=2.0 ** -8 if self.bf16 else 1e-6)

    def fwd_MaxJoinNode_post(self, nd, s):
        self.close(nd.label + ":fwd", "max join", act_raw(nd.y), torch.maximum(act_T(nd.a_), act_T(nd.b_)), rel=2.0 ** -8 if self.bf16 else 1e-6)

    def fwd_CopyNode_post(self, nd, s):
        self.close(nd.label + ":fwd", "copy", act_raw(nd.y), act_T(nd.xin), rel=2.0 ** -8 if self.bf16 else 1e-6)

    def fwd_XCorrNode_post(self, nd, s):
        cur, prev = act_T(nd.a_).squeeze(2), act_T(nd.b_).squeeze(2)
        b, c, h, w = prev.shape
        out = F.conv2d(cur.reshape(1, b * c, h, w), prev.reshape(b * c, 1, h, w), groups=b * c, padding="same").view(b, c, h, w)
        self.close(nd.label + ":fwd", "xcorr", act_raw(nd.y), out.unsqueeze(2))

    def fwd_HeadNode_post(self, nd, s):
        a = act_T(nd.xin)
        w = nd.conv.weight.detach().cpu().double().reshape(nd.cout, nd.xin.c)
        logits = torch.einsum("ncdhw,oc->nodhw", a, w) + nd.conv.bias.detach().cpu().double().view(1, -1, 1, 1, 1)
        if self.eng.nd == 2:
            logits = logits.squeeze(2)
        if nd.logits is not None:
            self.close(nd.label + ":fwd", "logits", nd.logits.cpu(), logits, rel=2e-5, abs_frac=1e-5)
        if nd.activated is not None:
            act = {0: lambda t: t, 1: torch.sigmoid, 2: torch.tanh, 3: F.relu}[nd.act](logits)
            self.close(nd.label + ":fwd", "activated", nd.activated.cpu(), act, rel=2e-5, abs_frac=1e-5)

    # ================================================================================================================
    # backward
    # ================================================================================================================
    def bwd_Heads_pre(self, hg):
        _, head_grads = hg
        live = []
        for h, g in zip(self.eng.heads, head_grads):
            if g is None:
                continue
            gl, ga, a = (None if t is None else t.detach().float().cpu().double() for t in g)
            dl = gl if gl is not None else 0.0
            if ga is not None:                     # the activation's derivative from the activated output (engine.HeadNode.dlogits)
                d = {0: lambda a: torch.ones_like(ga), 1: lambda a: a * (1 - a), 2: lambda a: 1 - a * a, 3: lambda a: (a > 0).double()}[h.act](a)
                dl = dl + ga * d
            live.append((h, dl))
        return dict(live=live)

    def bwd_Heads_post(self, hg, s):
        if not s["live"]:
            return
        x = s["live"][0][0].xin
        a = act_T(x)
        dx = torch.zeros_like(a)
        for h, dl in s["live"]:
            lab = h.label + ":bwd"
            if self.eng.nd == 2:
                dl = dl.unsqueeze(2)
            w = h.conv.weight.detach().cpu().double().reshape(h.cout, x.c)
            dx += torch.einsum("nodhw,oc->ncdhw", dl, w)
            self.vec_close(lab, "head dW", self.eng.grads[h.conv.weight].cpu().reshape(h.cout, x.c), torch.einsum("nodhw,ncdhw->oc", dl, a), rel=1e-4, floor=1e-5)
            self.vec_close(lab, "head dbias", self.eng.grads[h.conv.bias].cpu(), dl.sum(dim=(0, 2, 3, 4)), rel=1e-4, floor=1e-5)
        self.close("heads:bwd", "d trunk output", act_grad(x), dx)
        self._check_red(x, "heads:bwd")

    def _check_red(self, xin_act, lab):
        """If the node just handed the producer of `xin_act` its BatchNorm-backward partial sums, check them against the
        gradient tensor as stored."""
        if isinstance(xin_act, E.CatAct) or len(xin_act.leaves) != 1:
            return
        up = xin_act.buf.producer.get(xin_act.leaves[0])
        if up is None or not getattr(up, "red_nblk", 0):
            return
        c = up.y.c
        part = up.red_partial[:up.red_nblk * c * 2].detach().cpu().double().view(up.red_nblk, c, 2).sum(0)
        da, y = act_grad(up.y), act_raw(up.y)
        fac, _ = dact(y, xf_vectors(up.y))
        dz = da * fac
        mean, invstd = up.save_mean.cpu().double().view(1, -1, 1, 1, 1), up.save_invstd.cpu().double().view(1, -1, 1, 1, 1)
        s1, s2 = dz.sum(dim=(0, 2, 3, 4)), (dz * (y - mean) * invstd).sum(dim=(0, 2, 3, 4))
        norm = float(dz.abs().sum(dim=(0, 2, 3, 4)).max()) + 1e-30          # sums of signed terms: tolerance relative to sum |dz|
        for k, want in ((0, s1), (1, s2)):
            dev = float((part[:, k] - want).abs().max()) / (2e-5 * norm * (float(((y - mean) * invstd).abs().max()) if k else 1.0))
            self.rep.add(lab, f"fused BN-bwd sum S{k + 1} -> {up.label}", dev, 1.0 if dev <= 1 else 0.0)
            if dev > 1.0:
                self.failures.append(f"{lab}: fused BatchNorm-backward sum S{k + 1} for {up.label}: {dev:.2f}x tolerance")

    def bwd_ConvBlockNode_pre(self, nd):
        if not nd.y.g_written():
            return None
        parts = self._parts(nd.xin)
        extra = list(nd.foldt.params) if getattr(nd, "foldt", None) is not None else []
        if extra:                   # folded ConvT + concat + conv: gradients go to the ConvT's coarse input and to the skip tensor
            return dict(da=act_grad(nd.y), gx=[self._gsnap(nd.foldt.xin), self._gsnap(parts[1])],
                        pg={p: (self.eng.grads[p].detach().cpu().double().clone() if p in self.eng.grads else None) for p in list(nd.params) + extra})
        return dict(da=act_grad(nd.y), gx=[self._gsnap(p) for p in parts],
                    gfold=self._gsnap(nd.fold_src) if getattr(nd, "fold_dg_slot", None) is not None else None,
                    pg={p: (self.eng.grads[p].detach().cpu().double().clone() if p in self.eng.grads else None) for p in nd.params})

    def _pgrad(self, nd, s, p):
        g = self.eng.grads[p].detach().cpu().double()
        return g - s["pg"][p] if s["pg"][p] is not None else g

    def bwd_ConvBlockNode_post(self, nd, s):
        if s is None:
            return
        lab = nd.label + ":bwd"
        y, da = act_raw(nd.y), s["da"]
        xf = xf_vectors(nd.y)
        fac, t = dact(y, xf)
        near = t.abs() < 1e-6 * float(t.abs().max())            # LeakyReLU decision closer than fp32 rounding: either branch is right
        dz = da * fac
        mean, invstd = nd.save_mean.cpu().double().view(1, -1, 1, 1, 1), nd.save_invstd.cpu().double().view(1, -1, 1, 1, 1)
        yhat = (y - mean) * invstd
        m = y.numel() / y.shape[1]
        s1, s2 = dz.sum(dim=(0, 2, 3, 4)), (dz * yhat).sum(dim=(0, 2, 3, 4))
        self.vec_close(lab, "dbeta", self._pgrad(nd, s, nd.bn.bias), s1, rel=1e-4, floor=3e-5)
        self.vec_close(lab, "dgamma", self._pgrad(nd, s, nd.bn.weight), s2, rel=1e-4, floor=3e-5)
        gis = (nd.bn.weight.detach().cpu().double() * nd.save_invstd.cpu().double()).view(1, -1, 1, 1, 1)
        dy = gis * (dz - s1.view(1, -1, 1, 1, 1) / m - yhat * s2.view(1, -1, 1, 1, 1) / m)
        dy_got = act_grad(nd.y)
        # the result is a difference of nearly equal terms where |dy| << |dz|: tolerance relative to the operands
        self.close(lab, "dy (BN+LReLU bwd)", dy_got, dy, abs_frac=(2.0 ** -8 if self.bf16 else 1e-5) * float(dz.pow(2).mean().sqrt() * gis.abs().max() / (dy.pow(2).mean().sqrt() + 1e-30) + 1.0), ignore=near)
        if getattr(nd, "foldt", None) is not None:
            self._bwd_foldt_check(nd, s, lab, dy_got)
            return
        # weight gradient from the dy the engine actually stored
        if getattr(nd, "fold_all", False):          # the up-sampled tensor was never materialised: rebuild it from the coarse one
            a = F.interpolate(act_T(nd.fold_src), scale_factor=2, mode="nearest")
        else:
            a = self._T_cat(nd.xin)
        wgrad_mfma = nd.xin.c >= 16 and nd.xin.c % 8 == 0 and nd.y.c >= 16 and nd.y.c % 8 == 0 and nd.dil == 1 and nd.kw == 3
        if self.bf16 and wgrad_mfma:
            a = _r(a.float(), True).double()
        w = nd.conv.weight.detach().cpu().double()
        wv = w.clone().requires_grad_(True)
        av = a.clone().requires_grad_(True)
        wq = _r(w.float(), True).double() if (self.bf16 and self._mfma(nd.y.c, nd.xin.c, nd.dil, nd.kw)) else w     # dgrad: K = Cout
        out = self._conv(av, wv, None, nd.dil)
        (gw,) = torch.autograd.grad(out, wv, dy_got, retain_graph=False)
        self.vec_close(lab, "dW", self._pgrad(nd, s, nd.conv.weight), gw, rel=2e-4, floor=3e-5)
        # data gradient(s)
        parts = self._parts(nd.xin)
        if getattr(nd, "fold_dg_slot", None) is not None:
            # folded: the gradient goes straight to the coarse tensor (biu_upconv_bwd_data); the kernel's weights are the folded ones
            xc = _r(act_T(nd.fold_src).float(), self.bf16).double().requires_grad_(True)
            (gc,) = torch.autograd.grad(self._fold_conv(xc, nd.conv.weight.detach().cpu().float(), None), xc, dy_got)
            if s["gfold"] is not None:
                gc = gc + s["gfold"]
            self.close(lab, "dx (coarse, folded)" + (" (accumulated)" if s["gfold"] is not None else ""), act_grad(nd.fold_src), gc)
            assert not nd.xin.g_written(), "the up-sampled tensor's gradient must stay unwritten when the data gradient is folded"
        elif any(self.eng.wants_grad(p) if isinstance(p, E.Act) else True for p in parts):
            out2 = self._conv(av, wq, None, nd.dil)
            (gx,) = torch.autograd.grad(out2, av, dy_got)
            o = 0
            for p, before in zip(parts, s["gx"]):
                if isinstance(p, E.Act) and not self.eng.wants_grad(p):
                    o += p.c
                    continue
                want = gx[:, o:o + p.c]
                if before is not None:
                    want = want + before
                self.close(lab, f"dx[{o}:{o + p.c}]" + (" (accumulated)" if before is not None else ""), act_grad(p), want)
                self._check_red(p, lab)
                o += p.c

    def _bwd_foldt_check(self, nd, s, lab, dy_got):
        """Gradients of the folded ConvT + concat + conv from the dy the engine stored: autograd through the unfolded expression with the
        operands as the kernels see them (inputs in the storage type, the up-sampled tensor never rounded, weights fp32 in the chain rule;
        the data gradients use the packed -- rounded -- weights)."""
        ct = nd.foldt
        lo, skip = ct.xin, nd.xin.parts[1]
        cup = ct.y.c
        xl, xs = self._foldt_cat(nd, self.bf16)
        xl, xs = xl.clone().requires_grad_(True), xs.clone().requires_grad_(True)
        wc = nd.conv.weight.detach().cpu().double().requires_grad_(True)
        wt = ct.up.weight.detach().cpu().double().requires_grad_(True)
        bt = ct.up.bias.detach().cpu().double().requires_grad_(True)
        out = F.conv3d(torch.cat([F.conv_transpose3d(xl, wt, bt, stride=2), xs], 1), wc, None, padding=1)
        gwc, gwt, gbt = torch.autograd.grad(out, (wc, wt, bt), dy_got)
        # the ConvT bias reaches dW_conv (b_T[c] S_k) and db_T (W_conv . S_k) through S_k = sum of dy over the voxels whose tap k stays inside,
        # which the kernel takes as MINUS the border sums: sum_v dy = 0 exactly behind a train-mode BatchNorm.  The STORED dy carries its
        # rounding (bf16: |sum_v dy| ~ sqrt(N) 2^-9 |dy|), which autograd on the stored dy includes and the kernel -- closer to the fp32
        # reference there -- does not: that residue is allowed for
        tot = float(dy_got.sum(dim=(0, 2, 3, 4)).abs().max())
        resid_w = tot * float(bt.detach().abs().max())
        self.vec_close(lab, "dW (conv, folded)", self._pgrad(nd, s, nd.conv.weight), gwc, rel=2e-4, floor=3e-5 + resid_w / (float(gwc.abs().max()) + 1e-30))
        self.vec_close(lab, "dW (ConvT, folded)", self._pgrad(nd, s, ct.up.weight), gwt, rel=2e-4, floor=3e-5)
        resid = tot * float(wc.detach()[:, :cup].abs().sum(dim=(0, 2, 3, 4)).max())
        self.vec_close(lab, "dbias (ConvT, folded)", self._pgrad(nd, s, ct.up.bias), gbt, rel=2e-4, floor=3e-5 + resid / (float(gbt.abs().max()) + 1e-30))
        # data gradients: the packed (rounded) composed / sliced weights
        out2 = self._foldt_fwd_lin(nd, xl, xs)
        gl, gs = torch.autograd.grad(out2, (xl, xs), dy_got)
        for act, g, before, name in ((lo, gl, s["gx"][0], "dx (coarse, folded)"), (skip, gs, s["gx"][1], "dx (skip)")):
            want = g + before if before is not None else g
            self.close(lab, name + (" (accumulated)" if before is not None else ""), act_grad(act), want)
            self._check_red(act, lab)
        assert not ct.y.g_written(), "the ConvT output's gradient must stay unwritten when the op is folded"

    def _foldt_fwd_lin(self, nd, xl, xs):
        """Linear part of _foldt_fwd (no biases, no intermediate rounding) with the rounded weights, differentiable in (xl, xs)."""
        ct = nd.foldt
        cup = ct.y.c
        wc = nd.conv.weight.detach().cpu().float()
        wtv = ct.up.weight.detach().cpu().float().reshape(ct.xin.c, cup, 8)
        cout, cl = wc.shape[0], wtv.shape[0]
        y = F.conv3d(xs, _r(wc[:, cup:], self.bf16).double(), None, padding=1)
        cls = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
        xp = F.pad(xl, (1, 1, 1, 1, 1, 1))
        d, h, w_ = xl.shape[2:]
        parts = {}
        for pd in range(2):
            for ph in range(2):
                for pw in range(2):
                    k8 = torch.zeros(cout, cl, 2, 2, 2, dtype=torch.float32)
                    for td in range(2):
                        for th in range(2):
                            for tw in range(2):
                                acc = torch.zeros(cout, cl, dtype=torch.float32)
                                for kd in cls[(pd, td)]:
                                    for kh in cls[(ph, th)]:
                                        for kw in cls[(pw, tw)]:
                                            q = ((pd + kd + 1) & 1) * 4 + ((ph + kh + 1) & 1) * 2 + ((pw + kw + 1) & 1)
                                            acc = acc + wc[:, :cup, kd, kh, kw] @ wtv[:, :, q].t()
                                k8[:, :, td, th, tw] = acc
                    parts[(pd, ph, pw)] = F.conv3d(xp[:, :, pd:pd + d + 1, ph:ph + h + 1, pw:pw + w_ + 1], _r(k8, self.bf16).double())
        n = xl.shape[0]
        # interleave the 8 parity classes: [n][c][d][2][h][2][w][2]
        st = torch.stack([torch.stack([torch.stack([parts[(pd, ph, pw)] for pw in range(2)], -1) for ph in range(2)], -3) for pd in range(2)], -5)
        fold = st.reshape(n, cout, 2 * d, 2 * h, 2 * w_)
        return y + fold

    def bwd_ConvTNode_pre(self, nd):
        if getattr(nd, "folded_into", None) is not None or not nd.y.g_written():
            return None
        return dict(gx=self._gsnap(nd.xin), pg={p: (self.eng.grads[p].detach().cpu().double().clone() if p in self.eng.grads else None) for p in nd.params})

    def bwd_ConvTNode_post(self, nd, s):
        if s is None:
            return
        lab = nd.label + ":bwd"
        dy = act_grad(nd.y)
        a = act_T(nd.xin)
        if self.bf16 and nd.xin.c >= 16 and nd.xin.c % 8 == 0 and nd.y.c >= 16 and nd.y.c % 8 == 0:
            a = _r(a.float(), True).double()
        w = nd.up.weight.detach().cpu().double()
        wv, av = w.clone().requires_grad_(True), a.clone().requires_grad_(True)
        out = self._conv(av, wv, None, 1, transposed=True)
        (gw,) = torch.autograd.grad(out, wv, dy)
        self.vec_close(lab, "dW", self._pgrad(nd, s, nd.up.weight), gw, rel=2e-4, floor=3e-5)
        self.vec_close(lab, "dbias", self._pgrad(nd, s, nd.up.bias), dy.sum(dim=(0, 2, 3, 4)), rel=2e-4, floor=3e-5)
        if self.eng.wants_grad(nd.xin):
            wq = _r(w.float(), True).double() if (self.bf16 and self._mfma(nd.y.c, nd.xin.c)) else w
            (gx,) = torch.autograd.grad(self._conv(av, wq, None, 1, transposed=True), av, dy)
            if s["gx"] is not None:
                gx = gx + s["gx"]
            self.close(lab, "dx", act_grad(nd.xin), gx)
            self._check_red(nd.xin, lab)

    def bwd_ResampleNode_pre(self, nd):
        if not nd.y.g_written() or not self.eng.wants_grad(nd.xin):
            return None
        return dict(gx=self._gsnap(nd.xin))

    def bwd_ResampleNode_post(self, nd, s):
        if s is None:
            return
        lab = nd.label + ":bwd"
        a = act_T(nd.xin).clone().requires_grad_(True)
        out = self._resample(nd.kind, a, nd.y.space)
        (gx,) = torch.autograd.grad(out, a, act_grad(nd.y))
        if s["gx"] is not None:
            gx = gx + s["gx"]
        self.close(lab, f"{nd.kind} dx" + (" (accumulated)" if s["gx"] is not None else ""), act_grad(nd.xin), gx, rel=2.0 ** -8 if self.bf16 else 1e-6)
        self._check_red(nd.xin, lab)

    def bwd_MaxJoinNode_pre(self, nd):
        if not nd.y.g_written():
            return None
        return dict(ga=self._gsnap(nd.a_), gb=self._gsnap(nd.b_))

    def bwd_MaxJoinNode_post(self, nd, s):
        if s is None:
            return
        ta, tb = act_T(nd.a_), act_T(nd.b_)
        g = act_grad(nd.y)
        # torch.maximum's backward: the winner takes the gradient, an exact tie splits it evenly (frequent in bf16)
        wa = torch.where(ta > tb, g, torch.where(ta == tb, 0.5 * g, torch.zeros_like(g)))
        wb = torch.where(tb > ta, g, torch.where(ta == tb, 0.5 * g, torch.zeros_like(g)))
        if s["ga"] is not None:
            wa, wb = wa + s["ga"], wb + s["gb"]
        self.close(nd.label + ":bwd", "max join da", act_grad(nd.a_), wa, rel=2.0 ** -8 if self.bf16 else 1e-6)
        self.close(nd.label + ":bwd", "max join db", act_grad(nd.b_), wb, rel=2.0 ** -8 if self.bf16 else 1e-6)


def _nc(t):
    return t.detach().cpu().double()[:, :, None, None, None]


def _fwd_DropoutNode_post(self, nd, s):
    self.close(nd.label + ":fwd", "dropout", act_raw(nd.y), act_T(nd.xin) * _nc(nd.factor), rel=2.0 ** -8 if self.bf16 else 1e-6)


def _bwd_DropoutNode_post(self, nd, s):
    if nd.y.g_written():
        self.close(nd.label + ":bwd", "dropout dx", act_grad(nd.xin), act_grad(nd.y) * _nc(nd.factor), rel=2.0 ** -8 if self.bf16 else 1e-6)
        self._check_red(nd.xin, nd.label + ":bwd")


def _fwd_AddReluNode_post(self, nd, s):
    self.close(nd.label + ":fwd", "relu(a + b)", act_raw(nd.y), F.relu(act_T(nd.a_) + act_T(nd.b_)), rel=2.0 ** -8 if self.bf16 else 1e-6)


def _bwd_AddReluNode_post(self, nd, s):
    if nd.y.g_written():
        g = torch.where(act_raw(nd.y) > 0, act_grad(nd.y), torch.zeros_like(act_grad(nd.y)))
        for who, a in (("da", nd.a_), ("db", nd.b_)):
            self.close(nd.label + ":bwd", "add_relu " + who, act_grad(a), g, rel=2.0 ** -8 if self.bf16 else 1e-6)


def _bwd_GateNode_pre(self, nd):
    return dict(ge=self._gsnap(nd.e_)) if nd.y.g_written() else None


def _fwd_GateNode_post(self, nd, s):
    self.close(nd.label + ":fwd", "skip * sigmoid(psi)", act_raw(nd.y), act_T(nd.e_) * torch.sigmoid(act_T(nd.psi_)))


def _bwd_GateNode_post(self, nd, s):
    if s is None:
        return
    sg, te, g = torch.sigmoid(act_T(nd.psi_)), act_T(nd.e_), act_grad(nd.y)
    de = g * sg
    if s["ge"] is not None:
        de = de + s["ge"]
    self.close(nd.label + ":bwd", "gate de", act_grad(nd.e_), de)
    dpsi = (g * te).sum(1, keepdim=True) * sg * (1 - sg)
    self.close(nd.label + ":bwd", "gate dpsi", act_grad(nd.psi_), dpsi, abs_frac=(2.0 ** -8 if self.bf16 else 1e-5) * float((g * te).abs().sum(1).mean() / (dpsi.pow(2).mean().sqrt() + 1e-30) + 1.0))


for _n, _f in list(globals().items()):
    if _n.startswith(("_fwd_", "_bwd_")) and callable(_f):
        setattr(InSitu, _n[1:], _f)


def attach(model, xs, bf16):
    """Build (or fetch) the engine for these inputs and hook the checker in; returns the InSitu object."""
    eng = model._engine_for(*xs)
    return InSitu(eng, bf16)


def extract_decisions(eng):
    """The discrete decisions the engine's last forward took, in execution order, in the form ``oracle.forced_decisions``
    consumes: LeakyReLU branch masks (sign of fma(scale, y, shift) on the stored y -- the expression every kernel evaluates),
    max-pool argmax indices (first maximum in scan order) and the comparison sign (+1 / 0 / -1) of the Siam 'max' join."""
    q = {"lrelu": [], "pool": [], "max": []}
    sq = (lambda t: t) if eng.nd == 3 else (lambda t: t.squeeze(2))
    for nd in eng.nodes:
        if isinstance(nd, E.ConvBlockNode):
            _, t = dact(act_raw(nd.y), xf_vectors(nd.y))
            q["lrelu"].append(sq(t > 0))
        elif isinstance(nd, E.ResampleNode) and nd.kind == "maxpool":
            a = sq(act_T(nd.xin))
            _, idx = (F.max_pool3d if eng.nd == 3 else F.max_pool2d)(a, 2, 2, return_indices=True)
            q["pool"].append(idx)
        elif isinstance(nd, E.MaxJoinNode):
            q["max"].append(sq(torch.sign(act_T(nd.a_) - act_T(nd.b_))))
        elif isinstance(nd, E.AddReluNode):
            q.setdefault("relu", []).append(sq(act_raw(nd.y) > 0))
    return q


def decision_mismatch(q_engine, q_oracle):
    """{kind: (decisions that differ, decisions taken)} between the engine's decisions (``extract_decisions``) and the ones a
    free-running oracle forward recorded (``oracle.record_decisions``).  Blocks without an activation (slope 1: the branch is
    meaningless) are counted too -- the sign of t is still a property of the stored tensor."""
    out = {}
    for k, mine in q_engine.items():
        theirs = q_oracle.get(k, [])
        assert len(mine) == len(theirs), f"{k}: engine took {len(mine)} decisions, oracle {len(theirs)}"
        diff = tot = 0
        for a, b in zip(mine, theirs):
            assert a.shape == b.shape, (k, a.shape, b.shape)
            diff += int((a != b).sum())
            tot += a.numel()
        if tot:
            out[k] = (diff, tot)
    return out
