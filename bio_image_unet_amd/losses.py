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


class BCEDiceLoss(nn.Module):
    def __init__(self, alpha, beta):
        super().__init__()
        self.bce, self.dice, self.alpha, self.beta = BCELoss2d(), SoftDiceLoss(), alpha, beta

    def forward(self, logits, targets):
        return self.alpha * self.bce(logits, targets) + self.beta * self.dice(logits, targets)


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

    def forward(self, inputs, targets):
        return 1 - _tversky_index(inputs, targets, self.alpha, self.beta, self.smooth)


class logcoshTverskyLoss(TverskyLoss):
    def forward(self, inputs, targets):
        return torch.log(torch.cosh(1 - _tversky_index(inputs, targets, self.alpha, self.beta, self.smooth)))


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
