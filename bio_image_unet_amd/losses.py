"""Loss criteria of the reference trainers (the scalar the backward pass starts from), kept in PyTorch: they are
reductions over the fp32 logits tensor the head kernel emits.

Mirrors ``bio_image_unet/unet/losses.py``: BCELoss2d (:5-37), SoftDiceLoss (:40-75), BCEDiceLoss (:78-112),
logcoshDiceLoss (:115-142), TverskyLoss (:145-191), logcoshTverskyLoss (:194-239) -- same class names, constructor
arguments and forward(logits, targets) contract.
"""
import torch
from torch import nn


class BCELoss2d(nn.Module):
    def __init__(self, weight=None, size_average=True):
        super().__init__()
        self.bce_loss = nn.BCEWithLogitsLoss(weight=weight, reduction="mean" if size_average else "sum")

    def forward(self, logits, targets):
        return self.bce_loss(logits, targets)


class SoftDiceLoss(nn.Module):
    def __init__(self, smooth=1.0):
        super().__init__()
        self.smooth = smooth

    def forward(self, logits, targets):
        n = targets.size(0)
        p = torch.sigmoid(logits).view(n, -1)
        t = targets.view(n, -1)
        score = 2.0 * ((p * t).sum(1) + self.smooth) / (p.sum(1) + t.sum(1) + self.smooth)
        return 1 - score.mean()


def _fusable(logits, targets):
    return (logits.is_cuda and logits.dtype == torch.float32 and targets.dtype == torch.float32 and logits.shape == targets.shape
            and logits.is_contiguous() and targets.is_contiguous() and logits.numel() > 0)


class _FusedSegLoss(torch.autograd.Function):
    """One autograd node, one pass over (logits, targets) each way (``biu_bce_dice_fwd/bwd``, ``biu_pair_smooth_l1_fwd/bwd``) for

        a_bce * BCEWithLogits(mean) + a_dice * (1 - mean_n[2 (I_n + s) / (P_n + T_n + s)])          (unet/losses.py:78-112)
      + a_tv * (1 - Tversky) | a_tv * log cosh(1 - Tversky),  Tversky = (TP + s) / (TP + al FP + be FN + s)   (:145-239)
      + w_time * SmoothL1(logits[1:], logits[:-1])                                                   (unet3d/train.py:140-145)

    instead of the eager graph's ~15 full-tensor element-wise / reduction kernels.  ``cfg`` selects the terms."""

    @staticmethod
    def _scalars(cfg):
        tv = cfg.get("tversky")
        return (float(cfg.get("bce", 0.0)), float(cfg.get("dice", 0.0)), float(cfg.get("smooth", 1.0)), 1 if tv else 0,
                float(tv[0]) if tv else 0.0, float(tv[1]) if tv else 0.0, float(tv[2]) if tv else 0.0, 1 if (tv and tv[3]) else 0,
                float(cfg.get("time", 0.0)))

    @staticmethod
    def forward(ctx, logits, targets, cfg):
        import ctypes as C
        from ._lib import check, lib
        n = targets.size(0)
        per = logits.numel() // n
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        nb = lib.biu_bce_dice_blocks(per)
        partial = torch.empty((n, nb, 4), dtype=torch.float32, device=logits.device)
        check(lib.biu_bce_dice_fwd(C.c_void_p(logits.data_ptr()), C.c_void_p(targets.data_ptr()), n, per,
                                   C.c_void_p(partial.data_ptr()), st), "bce_dice_fwd")
        sc = _FusedSegLoss._scalars(cfg)
        w_time = sc[8]
        pt, nbt = None, 0
        if w_time and n > 1:
            nbt = lib.biu_pair_smooth_l1_blocks((n - 1) * per)
            pt = torch.empty(nbt, dtype=torch.float32, device=logits.device)
            check(lib.biu_pair_smooth_l1_fwd(C.c_void_p(logits.data_ptr()), n, per, C.c_void_p(pt.data_ptr()), st), "pair_smooth_l1_fwd")
        # the per-block sums -> the loss value, in one launch (biu_seg_loss_finish); saved = {loss, sums, den, Tversky terms}
        saved = torch.empty(1 + 5 * n + 3, dtype=torch.float32, device=logits.device)
        check(lib.biu_seg_loss_finish(C.c_void_p(partial.data_ptr()), n, nb, per, C.c_void_p(pt.data_ptr()) if pt is not None else None, nbt,
                                      *sc, C.c_void_p(saved.data_ptr()), st), "seg_loss_finish")
        loss = saved[0]
        if w_time and n <= 1:            # a batch of one: nn.SmoothL1Loss over empty slices is nan in the reference, and so here
            loss = loss + float("nan")
        ctx.save_for_backward(logits, targets)
        ctx.saved, ctx.cfg, ctx.np = saved, cfg, (n, per)
        return loss

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        from ._lib import check, lib
        logits, targets = ctx.saved_tensors
        cfg, saved, (n, per) = ctx.cfg, ctx.saved, ctx.np
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        sc = _FusedSegLoss._scalars(cfg)
        w_time = sc[8]
        g = g.reshape(1).to(torch.float32).contiguous()
        # d/dl_i = c0 * (p_i - t_i) + (c1 + c2 * t_i) * p_i (1 - p_i), per sample: coefficients in one launch (biu_seg_loss_coef)
        coef = torch.empty((n, 3), dtype=torch.float32, device=logits.device)
        ct = torch.empty(1, dtype=torch.float32, device=logits.device) if (w_time and n > 1) else None
        check(lib.biu_seg_loss_coef(C.c_void_p(g.data_ptr()), C.c_void_p(saved.data_ptr()), n, per, *sc, C.c_void_p(coef.data_ptr()),
                                    C.c_void_p(ct.data_ptr()) if ct is not None else None, st), "seg_loss_coef")
        dl = torch.empty_like(logits)
        check(lib.biu_bce_dice_bwd(C.c_void_p(logits.data_ptr()), C.c_void_p(targets.data_ptr()), n, per,
                                   C.c_void_p(coef.data_ptr()), C.c_void_p(dl.data_ptr()), 0, st), "bce_dice_bwd")
        if ct is not None:
            check(lib.biu_pair_smooth_l1_bwd(C.c_void_p(logits.data_ptr()), n, per, C.c_void_p(ct.data_ptr()), C.c_void_p(dl.data_ptr()), 1, st),
                  "pair_smooth_l1_bwd")
        return dl, None, None


class BCEDiceLoss(nn.Module):
    def __init__(self, alpha, beta):
        super().__init__()
        self.bce, self.dice, self.alpha, self.beta = BCELoss2d(), SoftDiceLoss(), alpha, beta

    def forward(self, logits, targets, time_weight: float = 0.0):
        """``time_weight`` adds the 3-D trainer's ``SmoothL1(logits[1:], logits[:-1]) * time_weight`` to the same fused pass."""
        if _fusable(logits, targets):
            return _FusedSegLoss.apply(logits, targets, dict(bce=float(self.alpha), dice=float(self.beta), smooth=float(self.dice.smooth),
                                                             time=float(time_weight)))
        loss = self.alpha * self.bce(logits, targets) + self.beta * self.dice(logits, targets)
        if time_weight:
            loss = loss + nn.functional.smooth_l1_loss(logits[1:], logits[:-1]) * time_weight
        return loss


class logcoshDiceLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.dice = SoftDiceLoss()

    def forward(self, logits, targets):
        x = self.dice(logits, targets)
        return torch.log((torch.exp(x) + torch.exp(-x)) / 2)


def _tversky_index(logits, targets, alpha, beta, smooth):
    p = torch.sigmoid(logits).view(-1)
    t = targets.view(-1)
    tp = (p * t).sum()
    fp = ((1 - t) * p).sum()
    fn = (t * (1 - p)).sum()
    return (tp + smooth) / (tp + alpha * fp + beta * fn + smooth)


class TverskyLoss(nn.Module):
    def __init__(self, alpha=0.5, beta=0.5, smooth=1):
        super().__init__()
        self.alpha, self.beta, self.smooth = alpha, beta, smooth

    _logcosh = False

    def forward(self, inputs, targets, time_weight: float = 0.0):
        if _fusable(inputs, targets):
            return _FusedSegLoss.apply(inputs, targets, dict(tversky=(float(self.alpha), float(self.beta), float(self.smooth), self._logcosh),
                                                             time=float(time_weight)))
        x = 1 - _tversky_index(inputs, targets, self.alpha, self.beta, self.smooth)
        loss = torch.log(torch.cosh(x)) if self._logcosh else x
        if time_weight:
            loss = loss + nn.functional.smooth_l1_loss(inputs[1:], inputs[:-1]) * time_weight
        return loss


class logcoshTverskyLoss(TverskyLoss):
    _logcosh = True


class weightedBCELoss(nn.Module):
    """BCE on sigmoid(logits) with weight alpha where target >= 0.5, beta elsewhere, mean over elements
    (``siam_unet/losses.py:109-148``)."""

    def __init__(self, alpha=1, beta=0.1):
        super().__init__()
        self.alpha, self.beta = alpha, beta

    def forward(self, logits, targets):
        probs = torch.sigmoid(logits)
        weights = torch.where(targets >= 0.5, torch.full_like(targets, self.alpha), torch.full_like(targets, self.beta))
        return torch.mean(nn.functional.binary_cross_entropy(probs, targets, reduction="none") * weights)


class BCELoss2dProb(nn.Module):
    """The Siam package's ``BCELoss2d`` (``siam_unet/losses.py:73-105``): ``nn.BCELoss`` on ``sigmoid(logits)`` -- NOT
    ``BCEWithLogitsLoss``.  Same value in the bulk; where fp32 ``sigmoid`` saturates (|logit| > ~17) its log terms clamp at
    -100 and the gradient vanishes, which ``BCEWithLogits`` does not do, so the reference arithmetic is kept verbatim."""

    def __init__(self, weight=None, reduction="mean", **kwargs):
        super().__init__()
        self.bce_loss = nn.BCELoss(weight, reduction=reduction)

    def forward(self, logits, targets):
        return self.bce_loss(torch.sigmoid(logits).view(-1), targets.view(-1))


class BCEDiceLossSiam(nn.Module):
    """``siam_unet.BCEDiceLoss`` (``siam_unet/losses.py:5-39``): alpha * BCELoss(sigmoid) + beta * SoftDice
    (``score.sum() / num``, the same number as the 2-D package's ``score.mean()``)."""

    def __init__(self, alpha, beta):
        super().__init__()
        self.bce, self.dice, self.alpha, self.beta = BCELoss2dProb(), SoftDiceLoss(), alpha, beta

    def forward(self, logits, targets):
        return self.alpha * self.bce(logits, targets) + self.beta * self.dice(logits, targets)


class TemporalConsistencyLoss(nn.Module):
    """L1 between consecutive slices of axis 2 of a (B, C, Z, X, Y) prediction (``multi_output_unet3d/losses.py``)."""

    def forward(self, predictions):
        return nn.functional.l1_loss(predictions[:, :, 1:, :, :], predictions[:, :, :-1, :, :])


class BCEDiceTemporalLoss(nn.Module):
    """``w0 * BCEDice(1, 1) + w1 * TemporalConsistency`` (``multi_output_unet3d/losses.py``, default weights (1.0, 0.1))."""

    def __init__(self, loss_params=(1.0, 0.1)):
        super().__init__()
        self.bce_dice_loss = BCEDiceLoss(1, 1)
        self.temporal_consistency_loss = TemporalConsistencyLoss()
        self.loss_params = loss_params

    def forward(self, predictions, targets):
        return (self.loss_params[0] * self.bce_dice_loss(predictions, targets) +
                self.loss_params[1] * self.temporal_consistency_loss(predictions))
