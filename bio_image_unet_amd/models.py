"""The four hot-path networks with the reference's constructor signatures, attribute names and state_dict keys.

Every layer attribute (``encode1`` ... ``decode8``, ``up1`` ..., ``final``) is a plain ``torch.nn`` container that
*owns parameters and buffers only* -- so ``.to()``, ``.apply(init_weights)``, ``state_dict()`` /
``load_state_dict()`` of reference checkpoints behave exactly as in the reference -- while ``forward`` never calls
those containers: it runs the HIP graph built by ``_build`` (``engine.py``).

Reference classes mirrored here:
  Unet ................. unet/unet.py:5-104
  UNet3D ............... unet3d/unet3d.py:6-99
  Siam_UNet ............ siam_unet/siam_unet.py:7-148
  MultiOutputUnet3D .... multi_output_unet3d/multi_output_unet3d.py:7-170
"""
from __future__ import annotations

import logging
from collections import OrderedDict
from typing import Dict, Optional

import torch
from torch import nn

from . import engine as E


def _block(nd: int, cin: int, cout: int, dilation: int = 1, dropout: float = 0.0) -> nn.Sequential:
    """Parameter container of one conv block; same child indices (0 conv, 1 BN) as the reference's Sequential."""
    conv = nn.Conv3d if nd == 3 else nn.Conv2d
    bn = nn.BatchNorm3d if nd == 3 else nn.BatchNorm2d
    drop = nn.Dropout3d if nd == 3 else nn.Dropout2d
    return nn.Sequential(conv(cin, cout, 3, padding=dilation, dilation=dilation), bn(cout),
                         nn.LeakyReLU(negative_slope=0.1, inplace=True), drop(dropout))


class _HipNet(nn.Module):
    """Shared plumbing: engine cache keyed by input shape, compute dtype switch, loud CPU refusal."""
    nd = 2
    _max_cached = 2          # input shapes kept
    _max_live = 4            # engines per shape: forwards whose backward is still pending each hold one

    def __init__(self):
        super().__init__()
        self._engines: "OrderedDict[tuple, list]" = OrderedDict()
        self.compute_dtype = torch.float32

    # engines hold ctypes structs and device buffers: they are caches, never part of a copy or a pickle
    def __getstate__(self):
        st = self.__dict__.copy()
        st["_engines"] = OrderedDict()
        st.pop("_grad_hook", None)          # belongs to a process-local GradAverager
        return st

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_grad_hook":
                continue
            new.__dict__[k] = OrderedDict() if k == "_engines" else copy.deepcopy(v, memo)
        return new

    def register_grad_ready_hook(self, fn):
        """``fn(param, grad)`` is called from inside the backward pass the moment a parameter's gradient is complete (after the
        last of its contributions when weights are shared) -- what ``ddp.GradAverager`` overlaps its bucketed all-reduce with."""
        self._grad_hook = fn
        for engs in self._engines.values():
            for e in engs:
                e.grad_hook = fn

    def register_grad_alloc(self, fn):
        """``fn(param) -> tensor | None``: where the engine should compute that parameter's gradient (``ddp.GradAverager`` hands out views of
        its flat all-reduce buckets, so no gradient is copied on its way to the wire)."""
        self._grad_alloc = fn
        for engs in self._engines.values():
            for e in engs:
                e.grad_alloc = fn

    def invalidate_packed(self):
        """Call after writing parameters through ``.data`` (EMA, clipping, manual broadcast): such writes do not bump
        ``Tensor._version``, which is what the cached MFMA weight packings are keyed on.  In-place ops under ``no_grad``,
        optimizers and ``load_state_dict`` need nothing."""
        for engs in self._engines.values():
            for e in engs:
                e.invalidate_packed()

    def set_compute_dtype(self, dtype):
        """torch.float32 (reference numerics) or torch.bfloat16 (bf16 storage, fp32 accumulate)."""
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        self._engines.clear()
        return self

    def _engine_for(self, *xs: torch.Tensor) -> E.Engine:
        x = xs[0]
        if x.device.type != "cuda":
            raise RuntimeError(f"{type(self).__name__}: input is on '{x.device}'. This package has no CPU path -- "
                               "every layer is a HIP kernel for MI355X (gfx950).")
        p = next(self.parameters())
        if p.device != x.device:
            raise RuntimeError(f"parameters are on {p.device} but the input is on {x.device}")
        key = (tuple(x.shape), str(x.device), self.compute_dtype)
        pool = self._engines.get(key)
        eng = next((e for e in pool if not e.busy()), None) if pool else None
        if eng is None:
            if pool and len(pool) >= self._max_live:
                raise RuntimeError(f"{type(self).__name__}: {len(pool)} forwards of input shape {tuple(x.shape)} are waiting for "
                                   "their backward; each holds a full set of activation buffers (raise _max_live to allow more)")
            for name, t in list(self.named_parameters()) + list(self.named_buffers()):
                if t.is_floating_point() and (t.dtype != torch.float32 or not t.is_contiguous()):
                    raise RuntimeError(f"{type(self).__name__}.{name} is {t.dtype}, contiguous={t.is_contiguous()}: the kernels read "
                                       "parameters and BatchNorm buffers as dense fp32. Keep the module in fp32 and select the "
                                       "activation storage type with set_compute_dtype(torch.bfloat16).")
            eng = E.Engine(x.device, self.compute_dtype, self.nd)
            self._build(eng, *[tuple(t.shape) for t in xs])
            eng.finalize()
            eng.grad_hook = getattr(self, "_grad_hook", None)
            eng.grad_alloc = getattr(self, "_grad_alloc", None)
            names = {id(mod): name for name, mod in self.named_modules()}
            for nd_ in eng.nodes:       # labels for profiling: the reference attribute name of the layer
                mod = getattr(nd_, "conv", None) or getattr(nd_, "up", None)
                base = names.get(id(mod), type(nd_).__name__)
                if isinstance(nd_, E.ResampleNode):
                    base = f"{nd_.kind}@{'x'.join(str(v) for v in nd_.xin.space[1:])}"
                nd_.label = base[:-2] if base.endswith(".0") and not base.startswith("final") else base
            self._engines.setdefault(key, []).append(eng)
            while len(self._engines) > self._max_cached:
                self._engines.popitem(last=False)
        self._engines.move_to_end(key)
        return eng

    def _space(self, shape):
        if self.nd == 3:
            n, c, d, h, w = shape
            return (n, d, h, w), c
        n, c, h, w = shape
        return (n, 1, h, w), c

    def _apply(self, fn, *a, **k):      # parameters moved (.to / .cuda): graphs hold stale pointers
        self._engines.clear()
        return super()._apply(fn, *a, **k)


def _half(space):
    n, d, h, w = space
    return (n, d // 2 if d > 1 else 1, h // 2, w // 2)


# ------------------------------------------------------------------------------------------------------
# 2-D U-Net (+ Siamese variant)
# ------------------------------------------------------------------------------------------------------
class Unet(_HipNet):
    """``Unet(in_channels=1, out_channels=1, n_filter=32, dilation=1)`` -> ``forward(x) = (sigmoid(logits), logits)``."""
    nd = 2

    def __init__(self, in_channels=1, out_channels=1, n_filter=32, dilation=1):
        super().__init__()
        f = n_filter
        widths = [f, 2 * f, 4 * f, 8 * f]
        c = in_channels
        for lvl, wd in enumerate(widths):           # encode1..8 carry the ctor's dilation (reference :20-31)
            setattr(self, f"encode{2 * lvl + 1}", _block(2, c, wd, dilation))
            setattr(self, f"encode{2 * lvl + 2}", _block(2, wd, wd, dilation))
            setattr(self, f"maxpool{lvl + 1}", nn.MaxPool2d(kernel_size=2, stride=2))
            c = wd
        self._make_middle(f, dilation)
        c = 16 * f
        for lvl in (1, 2, 3, 4):                    # decoder always dilation 1 (reference :38-49)
            setattr(self, f"up{lvl}", nn.ConvTranspose2d(c, c // 2, kernel_size=2, stride=2))
            setattr(self, f"decode{2 * lvl - 1}", _block(2, c, c // 2))
            setattr(self, f"decode{2 * lvl}", _block(2, c // 2, c // 2))
            c //= 2
        self.final = nn.Sequential(nn.Conv2d(f, out_channels, kernel_size=1, padding=0))

    def _make_middle(self, f, dilation):
        self.middle_conv1 = _block(2, 8 * f, 16 * f, dilation)
        self.middle_conv2 = _block(2, 16 * f, 16 * f, dilation)

    # graph ---------------------------------------------------------------------------------------------
    def _check_divisible(self, space):
        n, d, h, w = space
        if h % 16 or w % 16:
            # the reference fails inside concat() when a pooled extent is odd (unet/unet.py:62-67)
            raise ValueError("concatenation failed: wrong dimensions")

    def _build_encoder(self, eng, x, spaces, cat_bufs, pool_out_slices=None):
        t = x
        skips = []
        for lvl in range(4):
            b1, b2 = getattr(self, f"encode{2 * lvl + 1}"), getattr(self, f"encode{2 * lvl + 2}")
            wd = b1[0].out_channels
            a = eng.new_act(spaces[lvl], wd, lazy=True)
            eng.add(E.ConvBlockNode(eng, b1, t, a))
            skip = cat_bufs[lvl].slice(wd, wd, lazy=True) if cat_bufs is not None else eng.new_act(spaces[lvl], wd, True)
            eng.add(E.ConvBlockNode(eng, b2, a, skip))
            skips.append(skip)
            if lvl == 3 and pool_out_slices is not None:
                pooled = pool_out_slices
            else:
                pooled = eng.new_act(spaces[lvl + 1], wd, lazy=False)
            eng.add(E.ResampleNode(eng, "maxpool", skip, pooled))
            t = pooled
        return t, skips

    def _build_decoder(self, eng, mid2, cat_bufs, spaces):
        t = mid2
        for lvl in (1, 2, 3, 4):
            buf = cat_bufs[4 - lvl]
            up = getattr(self, f"up{lvl}")
            u = buf.slice(0, up.out_channels, lazy=False)
            eng.add(E.ConvTNode(eng, up, t, u))
            cat = buf.full()
            b1, b2 = getattr(self, f"decode{2 * lvl - 1}"), getattr(self, f"decode{2 * lvl}")
            a = eng.new_act(spaces[4 - lvl], b1[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, b1, cat, a))
            t = eng.new_act(spaces[4 - lvl], b2[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, b2, a, t))
        head = E.HeadNode(eng, self.final[0], t, "sigmoid", want_logits=True, want_act=True)
        eng.add(head)
        eng.heads.append(head)

    def _spaces(self, space):
        sp = [space]
        for _ in range(4):
            sp.append(_half(sp[-1]))
        return sp

    def _build(self, eng, xshape):
        space, cin = self._space(xshape)
        assert cin == self.encode1[0].in_channels, f"expected {self.encode1[0].in_channels} input channels, got {cin}"
        self._check_divisible(space)
        spaces = self._spaces(space)
        # level l's decoder concat = (up-sampled | skip), each encode{2l+2}.out_channels wide; its reader is decode{7-2l}
        cat_bufs = [eng.new_cat(spaces[l], getattr(self, f"encode{2 * l + 2}")[0].out_channels,
                                getattr(self, f"encode{2 * l + 2}")[0].out_channels, getattr(self, f"decode{7 - 2 * l}")[0].out_channels, 1)
                    for l in range(4)]
        x = eng.new_input(space, cin)
        m4, _ = self._build_encoder(eng, x, spaces, cat_bufs)
        mid1 = eng.new_act(spaces[4], self.middle_conv1[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv1, m4, mid1))
        mid2 = eng.new_act(spaces[4], self.middle_conv2[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv2, mid1, mid2))
        self._build_decoder(eng, mid2, cat_bufs, spaces)

    def forward(self, x):
        eng = self._engine_for(x)
        logits, prob = E.run(eng, [x], [(0, "logits"), (0, "act")])
        return prob, logits


class Siam_UNet(Unet):
    """``Siam_UNet(n_filter=32, mode='concat')`` -> ``forward(x, prev_x)``; encoder weights shared by both frames."""

    def __init__(self, n_filter=32, mode="concat"):
        self.mode = mode
        super().__init__(in_channels=1, out_channels=1, n_filter=n_filter, dilation=1)

    def _make_middle(self, f, dilation):
        if self.mode == "concat":                       # reference siam_unet.py:38-39 (registered before middle_conv1)
            self.conv_concat = _block(2, 16 * f, 8 * f)
        super()._make_middle(f, dilation)

    def _build(self, eng, xshape, pshape):
        if self.mode not in ("concat", "max", "control", "corr"):
            raise NotImplementedError("Unknown mode: {}".format(self.mode))
        if tuple(xshape) != tuple(pshape):
            logging.critical(f"Shapes: {xshape}, {pshape}")
            raise ValueError("concatenation failed: wrong dimensions")
        space, cin = self._space(xshape)
        self._check_divisible(space)
        spaces = self._spaces(space)
        # level l's decoder concat = (up-sampled | skip), each encode{2l+2}.out_channels wide; its reader is decode{7-2l}
        cat_bufs = [eng.new_cat(spaces[l], getattr(self, f"encode{2 * l + 2}")[0].out_channels,
                                getattr(self, f"encode{2 * l + 2}")[0].out_channels, getattr(self, f"decode{7 - 2 * l}")[0].out_channels, 1)
                    for l in range(4)]
        x = eng.new_input(space, 1)
        px = eng.new_input(space, 1)
        c8 = self.encode8[0].out_channels
        if self.mode == "concat":
            jbuf = eng.new_cat(spaces[4], c8, c8, self.conv_concat[0].out_channels, 1)
            m4, _ = self._build_encoder(eng, x, spaces, cat_bufs, pool_out_slices=jbuf.slice(0, c8, lazy=False))
            mm4, _ = self._build_encoder(eng, px, spaces, None, pool_out_slices=jbuf.slice(c8, c8, lazy=False))
            join = eng.new_act(spaces[4], self.conv_concat[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, self.conv_concat, jbuf.full(), join))
        else:
            m4, _ = self._build_encoder(eng, x, spaces, cat_bufs)
            mm4, _ = self._build_encoder(eng, px, spaces, None)     # 'control' still runs it (BN running stats!)
            if self.mode == "max":
                join = eng.new_act(spaces[4], c8, lazy=False)
                eng.add(E.MaxJoinNode(eng, m4, mm4, join))
            elif self.mode == "corr":
                join = eng.new_act(spaces[4], c8, lazy=False)
                eng.add(E.XCorrNode(eng, m4, mm4, join))
            else:
                join = m4
        mid1 = eng.new_act(spaces[4], self.middle_conv1[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv1, join, mid1))
        mid2 = eng.new_act(spaces[4], self.middle_conv2[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv2, mid1, mid2))
        self._build_decoder(eng, mid2, cat_bufs, spaces)

    def forward(self, x, prev_x):
        eng = self._engine_for(x, prev_x)
        logits, prob = E.run(eng, [x, prev_x], [(0, "logits"), (0, "act")])
        return prob, logits


# ------------------------------------------------------------------------------------------------------
# 3-D U-Nets
# ------------------------------------------------------------------------------------------------------
class _Body3D(_HipNet):
    nd = 3

    def _make_body(self, in_channels, f, *, conv_t: bool, up_convs: bool, pools: bool):
        plan = [(in_channels, f // 2), (f // 2, f), (f, f), (f, 2 * f), (2 * f, 2 * f), (2 * f, 4 * f)]
        for i, (ci, co) in enumerate(plan):
            setattr(self, f"encode{i + 1}", _block(3, ci, co))
            if pools and i % 2 == 1:
                setattr(self, f"maxpool{i // 2 + 1}", nn.MaxPool3d(kernel_size=2, stride=2))
        self.middle_conv1 = _block(3, 4 * f, 4 * f)
        self.middle_conv2 = _block(3, 4 * f, 8 * f)
        if conv_t:
            for lvl, c in ((1, 8 * f), (2, 4 * f), (3, 2 * f)):
                setattr(self, f"up{lvl}", nn.ConvTranspose3d(c, c, kernel_size=2, stride=2))
        if up_convs:
            for lvl, c in ((1, 8 * f), (2, 4 * f), (3, 2 * f)):
                setattr(self, f"up{lvl}_conv", _block(3, c, c))
        for i, (ci, co) in enumerate([(12 * f, 4 * f), (4 * f, 4 * f), (6 * f, 2 * f), (2 * f, 2 * f), (3 * f, f),
                                      (f, f // 2)]):
            setattr(self, f"decode{i + 1}", _block(3, ci, co))

    def _build_body(self, eng, xshape, *, down: str, up: str):
        space, cin = self._space(xshape)
        n, d, h, w = space
        if d % 8 or h % 8 or w % 8:
            # reference: bare torch.cat raises RuntimeError on the first mismatching level (unet3d.py:60-61)
            raise RuntimeError("Sizes of tensors must match except in dimension 1 (input extents must be divisible by 8)")
        spaces = [space]
        for _ in range(3):
            spaces.append(_half(spaces[-1]))
        up_c = [self.middle_conv2[0].out_channels, self.decode2[0].out_channels, self.decode4[0].out_channels]
        skip_c = [self.encode6[0].out_channels, self.encode4[0].out_channels, self.encode2[0].out_channels]
        cat_bufs = [eng.new_cat(spaces[2 - i], up_c[i], skip_c[i], getattr(self, f"decode{2 * i + 1}")[0].out_channels, 3)
                    for i in range(3)]                                                    # level 2,1,0
        x = eng.new_input(space, cin)
        t = x
        for lvl in range(3):
            b1, b2 = getattr(self, f"encode{2 * lvl + 1}"), getattr(self, f"encode{2 * lvl + 2}")
            a = eng.new_act(spaces[lvl], b1[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, b1, t, a))
            buf = cat_bufs[2 - lvl]
            skip = buf.slice(up_c[2 - lvl], skip_c[2 - lvl], lazy=True)
            eng.add(E.ConvBlockNode(eng, b2, a, skip))
            pooled = eng.new_act(spaces[lvl + 1], skip.c, lazy=False)
            eng.add(E.ResampleNode(eng, "maxpool" if down == "maxpool" else "down", skip, pooled))
            t = pooled
        mid1 = eng.new_act(spaces[3], self.middle_conv1[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv1, t, mid1))
        t = eng.new_act(spaces[3], self.middle_conv2[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv2, mid1, t))
        for i, lvl in enumerate((1, 2, 3)):
            buf = cat_bufs[i]
            sp = spaces[3 - lvl]
            ctn = None
            if up == "convT":
                u = buf.slice(0, up_c[i], lazy=False)
                ctn = E.ConvTNode(eng, getattr(self, f"up{lvl}"), t, u)
                eng.add(ctn)
            elif up == "nearest_conv":
                r = eng.new_act(sp, t.c, lazy=False)
                rs = E.ResampleNode(eng, "up", t, r)
                eng.add(rs)
                u = buf.slice(0, up_c[i], lazy=True)
                blk = E.ConvBlockNode(eng, getattr(self, f"up{lvl}_conv"), r, u, fold_src=t)      # forward folded onto the coarse tensor
                rs.only_for_backward = blk.fold_src is not None
                rs.skip = blk.fold_all                                                            # (forward, data and weight gradient folded)
                if rs.skip:
                    r.buf.release()                                                               # ... so the up-sampled tensor is never materialised
                eng.add(blk)
            elif up == "trilinear":                       # F.interpolate(scale_factor=2, mode='trilinear') [unet3d/unet3d.py:82]
                u = buf.slice(0, up_c[i], lazy=False)
                assert u.c == t.c
                eng.add(E.ResampleNode(eng, "trilinear", t, u))
            else:
                raise NotImplementedError(f"unknown up-sampling '{up}'")
            cat = buf.full()
            b1, b2 = getattr(self, f"decode{2 * lvl - 1}"), getattr(self, f"decode{2 * lvl}")
            a = eng.new_act(sp, b1[0].out_channels, lazy=True)
            blk1 = E.ConvBlockNode(eng, b1, cat, a, convt=ctn)       # ConvT + concat + conv as one op when the folded kernels serve the level
            if ctn is not None and blk1.foldt is not None:
                ctn.folded_into = blk1
                if isinstance(buf, E.CatBuf):
                    u.buf.release()                              # the up-sampled tensor (and its gradient) is never materialised: no memory for it
            eng.add(blk1)
            t = eng.new_act(sp, b2[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, b2, a, t))
        return t


class UNet3D(_Body3D):
    """``UNet3D(in_channels=1, out_channels=1, n_filter=16, use_interpolation=False)`` -> (sigmoid(logits), logits)."""

    def __init__(self, in_channels=1, out_channels=1, n_filter=16, use_interpolation=False):
        super().__init__()
        self.use_interpolation = use_interpolation
        self._make_body(in_channels, n_filter, conv_t=not use_interpolation, up_convs=False, pools=True)
        self.final = nn.Conv3d(n_filter // 2, out_channels=out_channels, kernel_size=1, padding=0)

    def _build(self, eng, xshape):
        d6 = self._build_body(eng, xshape, down="maxpool", up="trilinear" if self.use_interpolation else "convT")
        head = E.HeadNode(eng, self.final, d6, "sigmoid", want_logits=True, want_act=True)
        eng.add(head)
        eng.heads.append(head)

    def forward(self, x):
        eng = self._engine_for(x)
        logits, prob = E.run(eng, [x], [(0, "logits"), (0, "act")])
        return prob, logits


class MultiOutputUnet3D(_Body3D):
    """``MultiOutputUnet3D(in_channels=1, output_heads=None, n_filter=16, use_interpolation=True)`` -> dict of
    activated head outputs (no logits), reference multi_output_unet3d.py:106-170."""

    def __init__(self, in_channels: int = 1, output_heads: Optional[Dict[str, dict]] = None, n_filter: int = 16,
                 use_interpolation: bool = True):
        super().__init__()
        self.output_heads = output_heads or {"default": {"channels": 1, "activation": "sigmoid"}}
        self.use_interpolation = use_interpolation
        self._make_body(in_channels, n_filter, conv_t=not use_interpolation, up_convs=use_interpolation,
                        pools=not use_interpolation)
        self.output_layers = nn.ModuleDict()
        for name, cfg in self.output_heads.items():
            self.output_layers[name] = nn.Conv3d(n_filter // 2, cfg["channels"], kernel_size=1)

    def _build(self, eng, xshape):
        d6 = self._build_body(eng, xshape, down="down" if self.use_interpolation else "maxpool",
                              up="nearest_conv" if self.use_interpolation else "convT")
        for name, cfg in self.output_heads.items():
            head = E.HeadNode(eng, self.output_layers[name], d6, cfg.get("activation"), want_logits=False, want_act=True)
            eng.add(head)
            eng.heads.append(head)

    def forward(self, x):
        eng = self._engine_for(x)
        outs = E.run(eng, [x], [(i, "act") for i in range(len(self.output_heads))])
        return {name: o for name, o in zip(self.output_heads, outs)}


# ------------------------------------------------------------------------------------------------------
# the remaining bio_image_unet.unet variants: legacy v0, three-level "baby", attention-gated decoder
# ------------------------------------------------------------------------------------------------------
def _block_relu(cin: int, cout: int, dropout: float = 0.0) -> nn.Sequential:
    """``Conv2d(k3, padding=1) -> BatchNorm2d -> ReLU -> Dropout2d(p)`` (unet/unet_v0.py:54-61, unet/baby_unet.py:51-58)."""
    return nn.Sequential(nn.Conv2d(kernel_size=3, in_channels=cin, out_channels=cout, padding=1), nn.BatchNorm2d(cout), nn.ReLU(),
                         nn.Dropout2d(dropout))


class _LegacyUnet(_HipNet):
    """Shared graph of Unet_v0 (4 pools) and BabyUnet (3 pools): ReLU blocks, ``Dropout2d(0.5)`` behind ``middle_conv2``, skips
    taken from the FIRST conv of each level (``concat(u1, e7)`` ..., unet/unet_v0.py:89-101), a last ``conv(F -> 1)`` block and a
    1 -> 1 channel 1x1 head."""
    nd = 2
    levels = 4

    def _make(self, f: int):
        c = 1
        for lvl in range(self.levels):
            wd = f << lvl
            setattr(self, f"encode{2 * lvl + 1}", _block_relu(c, wd))
            setattr(self, f"encode{2 * lvl + 2}", _block_relu(wd, wd))
            setattr(self, f"maxpool{lvl + 1}", nn.MaxPool2d(kernel_size=2, stride=2))
            c = wd
        top = f << self.levels
        self.middle_conv1 = _block_relu(c, top)
        self.middle_conv2 = _block_relu(top, top, dropout=0.5)
        c = top
        for lvl in range(1, self.levels + 1):
            setattr(self, f"up{lvl}", nn.ConvTranspose2d(c, c // 2, kernel_size=2, stride=2))
            setattr(self, f"decode{2 * lvl - 1}", _block_relu(c, c // 2))
            setattr(self, f"decode{2 * lvl}", _block_relu(c // 2, c // 2))
            c //= 2
        setattr(self, f"decode{2 * self.levels + 1}", _block_relu(f, 1))
        self.final = nn.Sequential(nn.Conv2d(1, 1, kernel_size=1, padding=0))

    def _build(self, eng, xshape):
        L = self.levels
        space, cin = self._space(xshape)
        assert cin == 1, f"{type(self).__name__} takes one input channel, got {cin}"
        n, d, h, w = space
        if h % (1 << L) or w % (1 << L):
            raise ValueError("concatenation failed: wrong dimensions")
        spaces = [space]
        for _ in range(L):
            spaces.append(_half(spaces[-1]))
        width = lambda l: getattr(self, f"encode{2 * l + 1}")[0].out_channels
        cat_bufs = [eng.new_cat(spaces[l], width(l), width(l), getattr(self, f"decode{2 * (L - l) - 1}")[0].out_channels, 1) for l in range(L)]
        t = eng.new_input(space, 1)
        for lvl in range(L):
            b1, b2 = getattr(self, f"encode{2 * lvl + 1}"), getattr(self, f"encode{2 * lvl + 2}")
            skip = cat_bufs[lvl].slice(width(lvl), width(lvl), lazy=True)           # e1 / e3 / e5 / e7: read by the next conv AND the decoder
            eng.add(E.ConvBlockNode(eng, b1, t, skip))
            a = eng.new_act(spaces[lvl], width(lvl), lazy=True)
            eng.add(E.ConvBlockNode(eng, b2, skip, a))
            t = eng.new_act(spaces[lvl + 1], width(lvl), lazy=False)
            eng.add(E.ResampleNode(eng, "maxpool", a, t))
        mid1 = eng.new_act(spaces[L], self.middle_conv1[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv1, t, mid1))
        mid2 = eng.new_act(spaces[L], self.middle_conv2[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv2, mid1, mid2, dropout_follows=True))
        t = eng.new_act(spaces[L], mid2.c, lazy=False)
        self._dropout_node = E.DropoutNode(eng, self.middle_conv2[3], mid2, t)
        eng.add(self._dropout_node)
        for lvl in range(1, L + 1):
            buf = cat_bufs[L - lvl]
            up = getattr(self, f"up{lvl}")
            u = buf.slice(0, up.out_channels, lazy=False)
            eng.add(E.ConvTNode(eng, up, t, u))
            b1, b2 = getattr(self, f"decode{2 * lvl - 1}"), getattr(self, f"decode{2 * lvl}")
            a = eng.new_act(spaces[L - lvl], b1[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, b1, buf.full(), a))
            t = eng.new_act(spaces[L - lvl], b2[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, b2, a, t))
        last = getattr(self, f"decode{2 * L + 1}")
        d9 = eng.new_act(space, 1, lazy=True)
        eng.add(E.ConvBlockNode(eng, last, t, d9))
        head = E.HeadNode(eng, self.final[0], d9, "sigmoid", want_logits=True, want_act=True)
        eng.add(head)
        eng.heads.append(head)

    def forward(self, x):
        eng = self._engine_for(x)
        logits, prob = E.run(eng, [x], [(0, "logits"), (0, "act")])
        return prob, logits


class Unet_v0(_LegacyUnet):
    """``Unet_v0(n_filter=32, **kwargs)`` (unet/unet_v0.py:5-106) -> ``forward(x) = (sigmoid(logits), logits)``."""
    levels = 4

    def __init__(self, n_filter=32, **kwargs):
        super().__init__()
        self._make(n_filter)


class BabyUnet(_LegacyUnet):
    """``BabyUnet(n_filter=4)`` (unet/baby_unet.py:5-93): three max-pools."""
    levels = 3

    def __init__(self, n_filter=4):
        super().__init__()
        self._make(n_filter)


class AttentionBlock(nn.Module):
    """Parameter container of the attention gate (unet/attention_unet.py:112-181): ``W_gate`` / ``W_x`` = 1x1 conv + BatchNorm,
    ``psi`` = 1x1 conv to one channel + BatchNorm + Sigmoid.  Same child names and indices as the reference (state_dict keys)."""

    def __init__(self, F_g, F_l, n_coefficients):
        super().__init__()
        self.W_gate = nn.Sequential(nn.Conv2d(F_g, n_coefficients, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(n_coefficients))
        self.W_x = nn.Sequential(nn.Conv2d(F_l, n_coefficients, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(n_coefficients))
        self.psi = nn.Sequential(nn.Conv2d(n_coefficients, 1, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(1), nn.Sigmoid())
        self.relu = nn.ReLU(inplace=True)


class AttentionUnet(Unet):
    """``AttentionUnet(in_channels=1, out_channels=1, n_filter=32, dilation=1)`` (unet/attention_unet.py:5-109): the 2-D U-Net
    with every skip multiplied by an attention coefficient computed from the up-sampled tensor; concat order is
    (attended skip, up-sampled) -- the reverse of ``Unet`` (``self.concat(a1, u1)``, :90)."""

    def __init__(self, in_channels=1, out_channels=1, n_filter=32, dilation=1):
        super().__init__(in_channels, out_channels, n_filter, dilation)
        f = n_filter
        # registration order of the reference (:38-52): up1, attention1, decode1, decode2, up2, ... -- state_dict order follows it
        mods = dict(self._modules)
        for k in [k for k in mods if k.startswith(("up", "decode")) or k == "final"]:
            del self._modules[k]
        for lvl, c in ((1, 8 * f), (2, 4 * f), (3, 2 * f), (4, f)):
            self._modules[f"up{lvl}"] = mods[f"up{lvl}"]
            setattr(self, f"attention{lvl}", AttentionBlock(c, c, n_coefficients=c // 2))
            self._modules[f"decode{2 * lvl - 1}"] = mods[f"decode{2 * lvl - 1}"]
            self._modules[f"decode{2 * lvl}"] = mods[f"decode{2 * lvl}"]
        self._modules["final"] = mods["final"]

    def _build(self, eng, xshape):
        space, cin = self._space(xshape)
        assert cin == self.encode1[0].in_channels, f"expected {self.encode1[0].in_channels} input channels, got {cin}"
        self._check_divisible(space)
        spaces = self._spaces(space)
        x = eng.new_input(space, cin)
        m4, skips = self._build_encoder(eng, x, spaces, None)          # skips are free-standing tensors here (gated, not concatenated)
        mid1 = eng.new_act(spaces[4], self.middle_conv1[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv1, m4, mid1))
        t = eng.new_act(spaces[4], self.middle_conv2[0].out_channels, lazy=True)
        eng.add(E.ConvBlockNode(eng, self.middle_conv2, mid1, t))
        for lvl in (1, 2, 3, 4):
            sp, e = spaces[4 - lvl], skips[4 - lvl]
            up, att = getattr(self, f"up{lvl}"), getattr(self, f"attention{lvl}")
            b1, b2 = getattr(self, f"decode{2 * lvl - 1}"), getattr(self, f"decode{2 * lvl}")
            buf = eng.new_cat(sp, e.c, up.out_channels, b1[0].out_channels, 1)      # (attended skip | up-sampled)
            u = buf.slice(e.c, up.out_channels, lazy=False)
            eng.add(E.ConvTNode(eng, up, t, u))
            nco = att.W_gate[0].out_channels
            g1 = eng.new_act(sp, nco, lazy=True)
            eng.add(E.ConvBlockNode(eng, att.W_gate, u, g1))
            x1 = eng.new_act(sp, nco, lazy=True)
            eng.add(E.ConvBlockNode(eng, att.W_x, e, x1))
            s = eng.new_act(sp, nco, lazy=False)
            eng.add(E.AddReluNode(eng, g1, x1, s))
            psi = eng.new_act(sp, 1, lazy=True)
            eng.add(E.ConvBlockNode(eng, att.psi, s, psi))
            a_att = buf.slice(0, e.c, lazy=False)
            eng.add(E.GateNode(eng, e, psi, a_att))
            a = eng.new_act(sp, b1[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, b1, buf.full(), a))
            t = eng.new_act(sp, b2[0].out_channels, lazy=True)
            eng.add(E.ConvBlockNode(eng, b2, a, t))
        head = E.HeadNode(eng, self.final[0], t, "sigmoid", want_logits=True, want_act=True)
        eng.add(head)
        eng.heads.append(head)
