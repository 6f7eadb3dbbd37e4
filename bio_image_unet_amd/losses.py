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
