"""CPU oracle for the U-Net hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a *functional* restatement (plain PyTorch CPU fp32, ``torch.nn.functional`` only) of the
forward dataflow of the four reference models and of the loss / train-step arithmetic that sits on the
hot path.  It exists so that the HIP engine can be checked on the GPU box, where ``/root/reference`` does
not exist.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package (``bio_image_unet_amd``) never does.

Pinned by: ``tests/golden/*.npz`` -- vectors produced by importing the reference model files by path
(``tests/golden/make_golden.py``) -- and asserted equal in ``tests/test_oracle_golden.py``.

Every model is expressed over a flat ``dict[str, Tensor]`` that uses the reference's ``state_dict`` key
schema, so a reference checkpoint, a golden fixture and the product's ``state_dict()`` are all
interchangeable inputs.

Reference citations (relative to /root/reference/bio_image_unet):
  * conv->BN->LeakyReLU(0.1)->Dropout(0) block ........ unet/unet.py:54-60, unet3d/unet3d.py:52-58,
                                                        siam_unet/siam_unet.py:59-65,
                                                        multi_output_unet3d/multi_output_unet3d.py:84-92
  * Unet.forward ...................................... unet/unet.py:69-104
  * UNet3D.forward .................................... unet3d/unet3d.py:63-99
  * Siam_UNet.forward / depthwise_xcorr ............... siam_unet/siam_unet.py:85-148, 75-83
  * MultiOutputUnet3D.forward / apply_activation ...... multi_output_unet3d/multi_output_unet3d.py:106-170, 97-104
  * BCELoss2d / SoftDiceLoss / BCEDiceLoss / Tversky .. unet/losses.py:5-37, 40-75, 78-112, 145-191
  * 2D Trainer loss expression (batch-axis quirk) ..... unet/train.py:133-134
  * 3D Trainer loss (SmoothL1 "time" term) ............ unet3d/train.py:140-145
  * mo3d Trainer: per-head loss menu, weighted sum,
    clip_grad_norm_(1.0), Adam ........................ multi_output_unet3d/train.py:149-162, 183-201
  * mo3d criteria (temporal L1 along Z) ............... multi_output_unet3d/losses.py:81-117, 250-298
  * the optimisation step of every Trainer ............ unet/train.py:102,137-139 (torch.optim.Adam defaults)
  * init_weights ...................................... utils/utils.py:76-78
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm default (unet/unet.py:57 uses defaults)
BN_MOMENTUM = 0.1
LRELU_SLOPE = 0.1    # unet/unet.py:58

State = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------------
# bf16-storage emulation (checker for the engine's bf16 mode; the reference itself is fp32 only)
# --------------------------------------------------------------------------------------------------
# With ``emulate_bf16()`` active the forwards below round to bfloat16 at exactly the points where the HIP engine
# stores a tensor in HBM as bf16 or packs an MFMA operand, and their autograd graphs round the activation
# gradients where the engine stores those: arithmetic stays fp32 (the engine accumulates in fp32).  It is what the
# engine's bf16 mode computes, restated with torch CPU ops; the unemulated functions stay the reference's arithmetic.
#   * a raw convolution / ConvTranspose output y is stored in bf16; its gradient dy likewise;
#   * BatchNorm statistics are those of the stored y; BatchNorm-affine + LeakyReLU run in fp32 on it;
#   * MFMA layers (Cin a multiple of 16, Cout >= 16) take operands T(y) and weights rounded to bf16; the Cin = 1
#     first layer, odd widths and the 1x1 heads multiply the unrounded fp32 T(y) with fp32 weights;
#   * max-pool / nearest / trilinear compare and interpolate the unrounded T(y) and store the result in bf16;
#   * an activation gradient is rounded when a kernel writes it and once more after a second writer accumulates.
class _Emu:
    on = False
    fwd = True      # round values (ablation switches for tools/bf16_error_budget.py; both True = the engine)
    bwd = True      # round activation gradients


class emulate_bf16:
    def __init__(self, on: bool = True):
        self.on = on

    def __enter__(self):
        self.prev, _Emu.on = _Emu.on, self.on
        return self

    def __exit__(self, *exc):
        _Emu.on = self.prev
        return False


def _r(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(t.dtype)


def _rf(t):
    return _r(t) if _Emu.fwd else t.view_as(t)


def _rb(t):
    return _r(t) if _Emu.bwd else t


class _RoundBoth(torch.autograd.Function):      # a tensor stored as bf16 whose gradient twin is stored as bf16 too
    @staticmethod
    def forward(ctx, x):
        return _rf(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


class _RoundFwd(torch.autograd.Function):       # an MFMA weight operand: rounded copy, fp32 gradient
    @staticmethod
    def forward(ctx, x):
        return _rf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):       # fp32 view of a stored tensor: only its gradient twin is bf16
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


# --------------------------------------------------------------------------------------------------
# forced decisions (checker for whole-network gradients)
# --------------------------------------------------------------------------------------------------
# LeakyReLU's branch and max-pool's / torch.maximum's winner are discontinuities of the gradient: an element within
# fp32 rounding of the boundary falls either way in any fp32 implementation -- the reference's CPU path included -- and
# one such voxel moves a bottleneck weight gradient by 1e-3 .. 1e-2 (tests/test_gpu_models.py).  With
# ``forced_decisions(q)`` the forwards below take these decisions from ``q`` -- lists, in execution order, of the masks /
# argmax indices an implementation under test actually took -- instead of re-deciding, so its gradients can be compared
# with fp64 arithmetic on the SAME piecewise-linear branch of the network.
class _Forced:
    q = None


class forced_decisions:
    def __init__(self, q):
        self.q = {k: list(v) for k, v in q.items()} if q is not None else None

    def __enter__(self):
        self.prev, _Forced.q = _Forced.q, self.q
        return self

    def __exit__(self, *exc):
        left = {k: len(v) for k, v in (_Forced.q or {}).items() if v}
        _Forced.q = self.prev
        if exc[0] is None and left:
            raise AssertionError(f"forced decisions not consumed: {left}")
        return False


class _Rec:
    q = None


class record_decisions:
    """Free-running counterpart of ``forced_decisions``: the forwards below append the decisions THEY take (same keys, same
    order, same encoding) to the dict this context yields, so a test can count how many of an implementation's decisions
    differ from the reference arithmetic's own (expected: the few elements within fp32 rounding of a boundary)."""

    def __enter__(self):
        self.prev, _Rec.q = _Rec.q, {"lrelu": [], "pool": [], "max": [], "relu": []}
        return _Rec.q

    def __exit__(self, *exc):
        _Rec.q = self.prev
        return False


def _lrelu(t, slope=None):
    slope = LRELU_SLOPE if slope is None else slope
    if _Rec.q is not None:
        _Rec.q["lrelu"].append((t > 0).detach())
    if slope == 1.0:
        if _Forced.q is not None:
            _Forced.q["lrelu"].pop(0)            # the engine records a (meaningless) branch for every conv block
        return t
    if _Forced.q is not None:
        m = _Forced.q["lrelu"].pop(0)
        return torch.where(m, t, slope * t)
    return F.leaky_relu(t, slope) if slope else F.relu(t)


def _relu(t):
    if _Rec.q is not None:
        _Rec.q["relu"].append((t > 0).detach())
    if _Forced.q is not None:
        return torch.where(_Forced.q["relu"].pop(0), t, torch.zeros_like(t))
    return F.relu(t)


def _maxpool(t):
    nd3 = t.dim() == 5
    if _Rec.q is not None:
        _Rec.q["pool"].append((F.max_pool3d if nd3 else F.max_pool2d)(t.detach(), 2, 2, return_indices=True)[1])
    if _Forced.q is not None:
        idx = _Forced.q["pool"].pop(0)                       # flat index into the (D*)H*W plane of each (n, c), as max_pool returns
        return t.flatten(2).gather(2, idx.flatten(2)).view(idx.shape)
    return F.max_pool3d(t, 2, 2) if nd3 else F.max_pool2d(t, 2, 2)


def _maximum(a, b):
    if _Rec.q is not None:
        _Rec.q["max"].append(torch.sign(a - b).detach())
    if _Forced.q is not None:               # sign(a - b) as the implementation under test saw it; a tie splits the gradient
        sgn = _Forced.q["max"].pop(0)
        return torch.where(sgn > 0, a, torch.where(sgn < 0, b, 0.5 * (a + b)))
    return torch.maximum(a, b)


def st(x):
    """Tensor as stored in HBM (value and gradient)."""
    return _RoundBoth.apply(x) if _Emu.on else x


def st_grad(x):
    """fp32 value, stored gradient."""
    return _RoundBwd.apply(x) if _Emu.on else x


def emu_input(x):
    """Network input as the engine holds it (biu_from_nchw converts to the storage type)."""
    return _rf(x) if _Emu.on else x


def mfma_layer(cin: int, cout: int, dilation: int = 1, ksize: int = 3) -> bool:
    """Which conv / ConvTranspose layers run on the MFMA kernels in bf16 mode (csrc/biu_conv_mfma.hip: chan_ok); 1x1 convs
    (attention gates) and dilated ones take the any-shape kernels, which multiply unrounded fp32 operands."""
    if cin == 1 and dilation == 1 and ksize == 3 and cout >= 16 and cout % 16 == 0:
        return True                             # first layer: im2col MFMA kernels (csrc/biu_c1.hip), weights packed as bf16
    return dilation == 1 and ksize != 1 and cin >= 16 and cin % 16 == 0 and cout >= 16 and cout % 8 == 0


def _operands(x, w, cin, cout, dilation=1, ksize=3):
    if not _Emu.on:
        return x, w
    if mfma_layer(cin, cout, dilation, ksize):
        return _RoundBoth.apply(x), _RoundFwd.apply(w)
    return _RoundBwd.apply(x), w


# --------------------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------------------
def conv_block(sd: State, name: str, x: torch.Tensor, *, training: bool, dilation: int = 1, slope=None,
               dropout_factor: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``nn.Sequential(ConvNd(k=3, padding=d, dilation=d), BatchNormNd, LeakyReLU(0.1), Dropout(0))``.

    Train mode: batch statistics (biased var) normalise, running stats updated with unbiased var and
    momentum 0.1, ``num_batches_tracked += 1`` -- exactly what ``F.batch_norm`` does when handed the
    running buffers.  Eval mode: running statistics.

    Variants of the other ``bio_image_unet.unet`` networks: ``slope`` = 0 for the ReLU blocks of Unet_v0 / BabyUnet, 1 for the
    attention gate's activation-free ``Conv2d(k=1) -> BatchNorm2d`` pairs (a 1x1 kernel takes padding 0);
    ``dropout_factor`` = the [N, C] multiplier (0 or 1/(1-p)) ``Dropout2d`` applied, when it is active.
    """
    w = sd[f"{name}.0.weight"]
    b = sd[f"{name}.0.bias"]
    conv = F.conv3d if w.dim() == 5 else F.conv2d
    k = w.shape[-1]
    xo, wo = _operands(x, w, w.shape[1], w.shape[0], dilation, k)
    y = st(conv(xo, wo, b, padding=dilation if k == 3 else 0, dilation=dilation))
    rm, rv = sd[f"{name}.1.running_mean"], sd[f"{name}.1.running_var"]
    if training:
        nbt = sd.get(f"{name}.1.num_batches_tracked")
        if nbt is not None:
            nbt += 1
    y = F.batch_norm(y, rm, rv, sd[f"{name}.1.weight"], sd[f"{name}.1.bias"],
                     training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    a = st_grad(_lrelu(y, slope))
    if dropout_factor is not None:
        a = st(a * dropout_factor.view(dropout_factor.shape + (1,) * (a.dim() - 2)))
    return a


def up_conv_t(sd: State, name: str, x: torch.Tensor) -> torch.Tensor:
    """``nn.ConvTranspose{2,3}d(k=2, stride=2)`` (unet/unet.py:38, unet3d/unet3d.py:40)."""
    w = sd[f"{name}.weight"]
    f = F.conv_transpose3d if w.dim() == 5 else F.conv_transpose2d
    xo, wo = _operands(x, w, w.shape[0], w.shape[1])
    return st(f(xo, wo, sd[f"{name}.bias"], stride=2))


def checked_concat(x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """unet/unet.py:62-67 -- order is (upsampled, skip); mismatch raises ValueError."""
    if x1.shape != x2.shape:
        raise ValueError("concatenation failed: wrong dimensions")
    return torch.cat((x1, x2), 1)


def head_activation(x: torch.Tensor, activation: Optional[str]) -> torch.Tensor:
    """multi_output_unet3d/multi_output_unet3d.py:97-104."""
    if activation == "sigmoid":
        return torch.sigmoid(x)
    if activation == "tanh":
        return torch.tanh(x)
    if activation == "relu":
        return F.relu(x)
    return x


# --------------------------------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------------------------------
def _decoder2d(sd: State, mid2, skips, training: bool):
    """Shared 2-D decoder of Unet and Siam_UNet (unet/unet.py:87-104)."""
    e2, e4, e6, e8 = skips
    t = mid2
    for lvl, skip in zip((1, 2, 3, 4), (e8, e6, e4, e2)):
        t = checked_concat(up_conv_t(sd, f"up{lvl}", t), skip)
        t = conv_block(sd, f"decode{2 * lvl - 1}", t, training=training)
        t = conv_block(sd, f"decode{2 * lvl}", t, training=training)
    logits = F.conv2d(t, sd["final.0.weight"], sd["final.0.bias"])
    return torch.sigmoid(logits), logits


def _encoder2d(sd: State, x, training: bool, dilation: int):
    """encode1..8 with 2x2 max-pools; returns (pooled bottleneck input, skips)."""
    skips = []
    t = x
    for lvl in range(4):
        t = conv_block(sd, f"encode{2 * lvl + 1}", t, training=training, dilation=dilation)
        t = conv_block(sd, f"encode{2 * lvl + 2}", t, training=training, dilation=dilation)
        skips.append(t)
        t = st(_maxpool(t))
    return t, skips


def unet2d_forward(sd: State, x: torch.Tensor, *, dilation: int = 1, training: bool = True):
    """``Unet.forward`` -> (sigmoid(logits), logits)."""
    m4, skips = _encoder2d(sd, emu_input(x), training, dilation)
    mid = conv_block(sd, "middle_conv1", m4, training=training, dilation=dilation)
    mid = conv_block(sd, "middle_conv2", mid, training=training, dilation=dilation)
    return _decoder2d(sd, mid, skips, training)


def depthwise_xcorr(cur: torch.Tensor, prev: torch.Tensor) -> torch.Tensor:
    """siam_unet/siam_unet.py:75-83: per-sample, per-channel correlation with padding='same'."""
    b, c, h, w = prev.shape
    out = F.conv2d(cur.reshape(1, b * c, h, w), prev.reshape(b * c, 1, h, w), groups=b * c, padding="same")
    return out.view(b, c, out.size(2), out.size(3))


def siam_forward(sd: State, x: torch.Tensor, prev_x: torch.Tensor, *, mode: str = "concat",
                 training: bool = True):
    """``Siam_UNet.forward``: weight-shared encoder applied to x then prev_x (BN stats per call)."""
    m4, skips = _encoder2d(sd, emu_input(x), training, 1)
    mm4, _ = _encoder2d(sd, emu_input(prev_x), training, 1)
    if mode == "corr":
        join = st(depthwise_xcorr(m4, mm4))
    elif mode == "max":
        join = st(_maximum(m4, mm4))
    elif mode == "concat":
        join = conv_block(sd, "conv_concat", checked_concat(m4, mm4), training=training)
    elif mode == "control":
        join = m4
    else:
        raise NotImplementedError("Unknown mode: {}".format(mode))
    mid = conv_block(sd, "middle_conv1", join, training=training)
    mid = conv_block(sd, "middle_conv2", mid, training=training)
    return _decoder2d(sd, mid, skips, training)


def legacy_unet_forward(sd: State, x: torch.Tensor, *, levels: int = 4, training: bool = True,
                        dropout_factor: Optional[torch.Tensor] = None):
    """``Unet_v0.forward`` (unet/unet_v0.py:69-106, levels=4) / ``BabyUnet.forward`` (unet/baby_unet.py:66-93, levels=3): ReLU
    blocks, skips from the FIRST conv of each level, ``Dropout2d(0.5)`` behind ``middle_conv2`` (``dropout_factor`` [N, C] =
    the multiplier it applied in this call; None = inactive / eval), a final ``conv(F -> 1)`` block and a 1x1 head."""
    skips = []
    t = emu_input(x)
    for lvl in range(levels):
        e = conv_block(sd, f"encode{2 * lvl + 1}", t, training=training, slope=0.0)
        skips.append(e)
        t = conv_block(sd, f"encode{2 * lvl + 2}", e, training=training, slope=0.0)
        t = st(_maxpool(t))
    t = conv_block(sd, "middle_conv1", t, training=training, slope=0.0)
    t = conv_block(sd, "middle_conv2", t, training=training, slope=0.0, dropout_factor=dropout_factor)
    if dropout_factor is None:
        t = st(t)                                   # the engine materialises the bottleneck behind the (inactive) dropout
    for lvl, skip in zip(range(1, levels + 1), reversed(skips)):
        t = checked_concat(up_conv_t(sd, f"up{lvl}", t), skip)
        t = conv_block(sd, f"decode{2 * lvl - 1}", t, training=training, slope=0.0)
        t = conv_block(sd, f"decode{2 * lvl}", t, training=training, slope=0.0)
    t = conv_block(sd, f"decode{2 * levels + 1}", t, training=training, slope=0.0)
    logits = F.conv2d(t, sd["final.0.weight"], sd["final.0.bias"])
    return torch.sigmoid(logits), logits


def attention_gate(sd: State, name: str, gate: torch.Tensor, skip: torch.Tensor, *, training: bool) -> torch.Tensor:
    """``AttentionBlock.forward`` (unet/attention_unet.py:159-181)."""
    g1 = conv_block(sd, f"{name}.W_gate", gate, training=training, slope=1.0)
    x1 = conv_block(sd, f"{name}.W_x", skip, training=training, slope=1.0)
    p = st(_relu(g1 + x1))
    psi = torch.sigmoid(conv_block(sd, f"{name}.psi", p, training=training, slope=1.0))
    return st(skip * psi)


def attention_unet_forward(sd: State, x: torch.Tensor, *, dilation: int = 1, training: bool = True):
    """``AttentionUnet.forward`` (unet/attention_unet.py:71-109): concat order is (attended skip, up-sampled)."""
    m4, skips = _encoder2d(sd, emu_input(x), training, dilation)
    t = conv_block(sd, "middle_conv1", m4, training=training, dilation=dilation)
    t = conv_block(sd, "middle_conv2", t, training=training, dilation=dilation)
    for lvl, skip in zip((1, 2, 3, 4), reversed(skips)):
        u = up_conv_t(sd, f"up{lvl}", t)
        a = attention_gate(sd, f"attention{lvl}", u, skip, training=training)
        t = checked_concat(a, u)
        t = conv_block(sd, f"decode{2 * lvl - 1}", t, training=training)
        t = conv_block(sd, f"decode{2 * lvl}", t, training=training)
    logits = F.conv2d(t, sd["final.0.weight"], sd["final.0.bias"])
    return torch.sigmoid(logits), logits


def _body3d(sd: State, x, *, training: bool, down: str, up: str):
    """Shared trunk of UNet3D and MultiOutputUnet3D; returns d6 (F//2 channels, full resolution).

    down: 'maxpool' | 'nearest' ; up: 'convT' | 'trilinear' | 'nearest_conv'.
    """
    def pool(t):
        if down == "maxpool":
            return st(_maxpool(t))
        return st(F.interpolate(t, scale_factor=0.5, mode="nearest"))

    def upsample(t, lvl):
        if up == "convT":
            return up_conv_t(sd, f"up{lvl}", t)
        if up == "trilinear":
            return st(F.interpolate(t, scale_factor=2, mode="trilinear", align_corners=False))
        t = st(F.interpolate(t, scale_factor=2, mode="nearest"))
        return conv_block(sd, f"up{lvl}_conv", t, training=training)

    skips = []
    t = emu_input(x)
    for lvl in range(3):
        t = conv_block(sd, f"encode{2 * lvl + 1}", t, training=training)
        t = conv_block(sd, f"encode{2 * lvl + 2}", t, training=training)
        skips.append(t)
        t = pool(t)
    t = conv_block(sd, "middle_conv1", t, training=training)
    t = conv_block(sd, "middle_conv2", t, training=training)
    for lvl, skip in zip((1, 2, 3), reversed(skips)):
        t = torch.cat((upsample(t, lvl), skip), 1)          # bare torch.cat in 3-D (unet3d.py:60-61)
        t = conv_block(sd, f"decode{2 * lvl - 1}", t, training=training)
        t = conv_block(sd, f"decode{2 * lvl}", t, training=training)
    return t


def unet3d_forward(sd: State, x: torch.Tensor, *, use_interpolation: bool = False, training: bool = True):
    """``UNet3D.forward`` -> (sigmoid(logits), logits); head key is ``final.weight``."""
    d6 = _body3d(sd, x, training=training, down="maxpool",
                 up="trilinear" if use_interpolation else "convT")
    logits = F.conv3d(d6, sd["final.weight"], sd["final.bias"])
    return torch.sigmoid(logits), logits


def mo3d_forward(sd: State, x: torch.Tensor, output_heads: Dict[str, dict], *,
                 use_interpolation: bool = True, training: bool = True) -> Dict[str, torch.Tensor]:
    """``MultiOutputUnet3D.forward`` -> dict of *activated* head outputs (no logits)."""
    d6 = _body3d(sd, x, training=training,
                 down="nearest" if use_interpolation else "maxpool",
                 up="nearest_conv" if use_interpolation else "convT")
    out = {}
    for name, cfg in output_heads.items():
        logits = F.conv3d(d6, sd[f"output_layers.{name}.weight"], sd[f"output_layers.{name}.bias"])
        out[name] = head_activation(logits, cfg.get("activation"))
    return out


# --------------------------------------------------------------------------------------------------
# losses (unet/losses.py)
# --------------------------------------------------------------------------------------------------
def soft_dice_loss(logits, targets, smooth: float = 1.0):
    p = torch.sigmoid(logits)
    n = targets.size(0)
    m1, m2 = p.reshape(n, -1), targets.reshape(n, -1)
    inter = (m1 * m2).sum(1)
    score = 2.0 * (inter + smooth) / (m1.sum(1) + m2.sum(1) + smooth)
    return 1 - score.mean()


def bce_dice_loss(logits, targets, alpha: float = 0.5, beta: float = 0.5):
    return alpha * F.binary_cross_entropy_with_logits(logits, targets) + beta * soft_dice_loss(logits, targets)


def siam_bce_dice_loss(logits, targets, alpha: float = 1.0, beta: float = 1.0):
    """siam_unet/losses.py:5-39 with its own BCELoss2d (:73-105): nn.BCELoss on sigmoid(logits), flattened."""
    bce = F.binary_cross_entropy(torch.sigmoid(logits).view(-1), targets.view(-1))
    return alpha * bce + beta * soft_dice_loss(logits, targets)


def tversky_loss(logits, targets, alpha: float = 0.5, beta: float = 0.5, smooth: float = 1.0):
    p = torch.sigmoid(logits).reshape(-1)
    t = targets.reshape(-1)
    tp = (p * t).sum()
    fp = ((1 - t) * p).sum()
    fn = (t * (1 - p)).sum()
    return 1 - (tp + smooth) / (tp + alpha * fp + beta * fn + smooth)


def logcosh_tversky_loss(logits, targets, alpha: float = 0.5, beta: float = 0.5, smooth: float = 1.0):
    return torch.log(torch.cosh(tversky_loss(logits, targets, alpha, beta, smooth)))


def trainer2d_loss(logits, y, out_channels: int, channel_weights=None, criterion=bce_dice_loss):
    """unet/train.py:133-134 -- NOTE the quirk: ``y_logits[ch]`` indexes the *batch* axis."""
    cw = torch.ones(out_channels) if channel_weights is None else torch.as_tensor(channel_weights)
    return sum(criterion(logits[ch], y[ch]) * cw[j] for j, ch in enumerate(range(out_channels))) / cw.sum()


def trainer3d_loss(logits, y, time_loss_weight: float = 0.1, criterion=bce_dice_loss):
    """unet3d/train.py:140-145 -- SmoothL1 between neighbouring *batch* entries."""
    # batch == 1 gives empty slices -> nan, exactly as the reference does
    return criterion(logits, y) + F.smooth_l1_loss(logits[1:], logits[:-1]) * time_loss_weight


def temporal_consistency_loss(pred):
    """multi_output_unet3d/losses.py:250-263 -- L1 between consecutive slices of axis 2 (Z) of a (B, C, Z, X, Y) prediction."""
    return F.l1_loss(pred[:, :, 1:], pred[:, :, :-1])


def bce_dice_temporal_loss(pred, targets, loss_params=(1.0, 0.1)):
    """multi_output_unet3d/losses.py:266-298 -- BCEDice(1, 1) + 0.1 * temporal consistency, both on ``pred`` as given."""
    return loss_params[0] * bce_dice_loss(pred, targets, 1, 1) + loss_params[1] * temporal_consistency_loss(pred)


MO3D_LOSS_MENU = {      # Trainer._get_loss_function, multi_output_unet3d/train.py:149-162
    "BCEDiceLoss": lambda p, t: bce_dice_loss(p, t, 1, 1),
    "DiceLoss": lambda p, t: bce_dice_loss(p, t, 0, 1),
    "TverskyLoss": tversky_loss,
    "logcoshTverskyLoss": logcosh_tversky_loss,
    "BCEDiceTemporalLoss": bce_dice_temporal_loss,
}


def trainer_mo3d_loss(pred: Dict[str, torch.Tensor], targets: Dict[str, torch.Tensor], output_heads: Dict[str, dict]):
    """multi_output_unet3d/train.py:183-196 -- sum over heads of weight * loss(pred, target); ``pred`` is the model's
    ALREADY ACTIVATED output and the criteria apply their own sigmoid on top (quirk 4 of SURVEY 8a)."""
    total = 0
    for name, cfg in output_heads.items():
        if cfg["loss"] not in MO3D_LOSS_MENU:
            raise ValueError(f'Loss "{cfg["loss"]}" not defined!')
        t = targets[name]
        if t.dim() == 4:
            t = t.unsqueeze(1)
        total = total + cfg.get("weight", 1.0) * MO3D_LOSS_MENU[cfg["loss"]](pred[name], t)
    return total


def adam_step(sd: State, grads: Dict[str, torch.Tensor], lr: float = 1e-3, clip: Optional[float] = None):
    """``optimizer.step()`` of a freshly built ``torch.optim.Adam(params, lr)`` (unet/train.py:102,139; betas (0.9, 0.999), eps 1e-8,
    first step), after ``clip_grad_norm_(params, clip)`` when ``clip`` is given (multi_output_unet3d/train.py:201).  Returns
    (updated parameters by key, total gradient norm before clipping or None)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_param(k)}
    for k, p in params.items():
        p.grad = grads[k].detach().clone()
    norm = torch.nn.utils.clip_grad_norm_(list(params.values()), max_norm=clip) if clip is not None else None
    torch.optim.Adam(list(params.values()), lr=lr).step()
    return {k: p.detach() for k, p in params.items()}, norm


# --------------------------------------------------------------------------------------------------
# parameter construction (reference key schema, PyTorch default init)
# --------------------------------------------------------------------------------------------------
def _conv_default_init(shape, fan_in, gen):
    bound = 1.0 / math.sqrt(fan_in)      # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +1/sqrt(fan_in))
    w = (torch.rand(shape, generator=gen) * 2 - 1) * bound
    return w


def _add_block(sd: State, name: str, cin: int, cout: int, nd: int, gen, kaiming_normal: bool):
    k = (3,) * nd
    fan_in = cin * 3 ** nd
    if kaiming_normal:   # utils/utils.py:76-78: kaiming_normal_(nonlinearity='leaky_relu') => std = sqrt(2/fan_in)
        sd[f"{name}.0.weight"] = torch.randn((cout, cin) + k, generator=gen) * math.sqrt(2.0 / fan_in)
    else:
        sd[f"{name}.0.weight"] = _conv_default_init((cout, cin) + k, fan_in, gen)
    sd[f"{name}.0.bias"] = _conv_default_init((cout,), fan_in, gen)
    sd[f"{name}.1.weight"] = torch.ones(cout)
    sd[f"{name}.1.bias"] = torch.zeros(cout)
    sd[f"{name}.1.running_mean"] = torch.zeros(cout)
    sd[f"{name}.1.running_var"] = torch.ones(cout)
    sd[f"{name}.1.num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _add_up(sd: State, name: str, cin: int, cout: int, nd: int, gen):
    # ConvTranspose weight (Cin, Cout, 2, 2[,2]); torch computes fan_in from dim 1 => cout * 2**nd
    fan_in = cout * 2 ** nd
    sd[f"{name}.weight"] = _conv_default_init((cin, cout) + (2,) * nd, fan_in, gen)
    sd[f"{name}.bias"] = _conv_default_init((cout,), fan_in, gen)


def _add_head(sd: State, name: str, cin: int, cout: int, nd: int, gen, kaiming_normal: bool = False):
    if kaiming_normal:
        sd[f"{name}.weight"] = torch.randn((cout, cin) + (1,) * nd, generator=gen) * math.sqrt(2.0 / cin)
    else:
        sd[f"{name}.weight"] = _conv_default_init((cout, cin) + (1,) * nd, cin, gen)
    sd[f"{name}.bias"] = _conv_default_init((cout,), cin, gen)


def init_unet2d(in_channels=1, out_channels=1, n_filter=32, *, seed=0, init_weights=True, siam_mode=None) -> State:
    """Keys/shapes of ``Unet`` (unet/unet.py:16-52) or ``Siam_UNet`` (siam_unet.py:18-57, in=out=1)."""
    g = torch.Generator().manual_seed(seed)
    f = n_filter
    sd: State = {}
    chans = [in_channels, f, f, 2 * f, 2 * f, 4 * f, 4 * f, 8 * f, 8 * f]
    for i in range(8):
        _add_block(sd, f"encode{i + 1}", chans[i], chans[i + 1], 2, g, init_weights)
    if siam_mode == "concat":
        _add_block(sd, "conv_concat", 16 * f, 8 * f, 2, g, init_weights)
    _add_block(sd, "middle_conv1", 8 * f, 16 * f, 2, g, init_weights)
    _add_block(sd, "middle_conv2", 16 * f, 16 * f, 2, g, init_weights)
    c = 16 * f
    for lvl in (1, 2, 3, 4):
        _add_up(sd, f"up{lvl}", c, c // 2, 2, g)
        _add_block(sd, f"decode{2 * lvl - 1}", c, c // 2, 2, g, init_weights)
        _add_block(sd, f"decode{2 * lvl}", c // 2, c // 2, 2, g, init_weights)
        c //= 2
    _add_head(sd, "final.0", f, out_channels, 2, g, init_weights)
    return sd


def _init_body3d(sd: State, in_channels, f, g, *, convT: bool, up_convs: bool):
    plan = [("encode1", in_channels, f // 2), ("encode2", f // 2, f), ("encode3", f, f), ("encode4", f, 2 * f),
            ("encode5", 2 * f, 2 * f), ("encode6", 2 * f, 4 * f), ("middle_conv1", 4 * f, 4 * f),
            ("middle_conv2", 4 * f, 8 * f)]
    for name, ci, co in plan:
        _add_block(sd, name, ci, co, 3, g, False)
    if convT:
        for lvl, c in ((1, 8 * f), (2, 4 * f), (3, 2 * f)):
            _add_up(sd, f"up{lvl}", c, c, 3, g)
    if up_convs:
        for lvl, c in ((1, 8 * f), (2, 4 * f), (3, 2 * f)):
            _add_block(sd, f"up{lvl}_conv", c, c, 3, g, False)
    for name, ci, co in [("decode1", 12 * f, 4 * f), ("decode2", 4 * f, 4 * f), ("decode3", 6 * f, 2 * f),
                         ("decode4", 2 * f, 2 * f), ("decode5", 3 * f, f), ("decode6", f, f // 2)]:
        _add_block(sd, name, ci, co, 3, g, False)


def init_unet3d(in_channels=1, out_channels=1, n_filter=16, use_interpolation=False, *, seed=0) -> State:
    """Keys/shapes of ``UNet3D`` (unet3d/unet3d.py:18-50); PyTorch default init (init_weights skips Conv3d)."""
    g = torch.Generator().manual_seed(seed)
    sd: State = {}
    _init_body3d(sd, in_channels, n_filter, g, convT=not use_interpolation, up_convs=False)
    _add_head(sd, "final", n_filter // 2, out_channels, 3, g)
    return sd


def init_mo3d(in_channels=1, output_heads=None, n_filter=16, use_interpolation=True, *, seed=0) -> State:
    """Keys/shapes of ``MultiOutputUnet3D`` (multi_output_unet3d.py:13-82)."""
    heads = output_heads or {"default": {"channels": 1, "activation": "sigmoid"}}
    g = torch.Generator().manual_seed(seed)
    sd: State = {}
    _init_body3d(sd, in_channels, n_filter, g, convT=not use_interpolation, up_convs=use_interpolation)
    for name, cfg in heads.items():
        _add_head(sd, f"output_layers.{name}", n_filter // 2, cfg["channels"], 3, g)
    return sd


# --------------------------------------------------------------------------------------------------
# helpers for tests / bench
# --------------------------------------------------------------------------------------------------
PARAM_SUFFIXES = (".weight", ".bias")


def is_param(key: str) -> bool:
    return key.endswith(PARAM_SUFFIXES)


def clone_state(sd: State, requires_grad: bool = False) -> State:
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and is_param(k) and t.is_floating_point():
            t.requires_grad_(True)
        out[k] = t
    return out


def grads_of(loss: torch.Tensor, sd: State) -> Dict[str, torch.Tensor]:
    keys = [k for k, v in sd.items() if v.requires_grad]
    gs = torch.autograd.grad(loss, [sd[k] for k in keys], allow_unused=True)
    return {k: (g if g is not None else torch.zeros_like(sd[k])) for k, g in zip(keys, gs)}
