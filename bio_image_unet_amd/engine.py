"""Host-side execution engine: a static op graph per (model, input shape) whose every node is a call into
``libbiu_hip.so``.  PyTorch is used for device memory (caching allocator), the current HIP stream, autograd
glue at the module boundary and nothing else -- no torch op touches an activation.

Data layout in HBM:
  * activations are channels-last ``[N, D, H, W, C]`` (2-D: ``D = 1``), bf16 or fp32;
  * a conv block stores only its *raw* convolution output ``y``; BatchNorm-affine + LeakyReLU is a per-channel
    transform ``T`` (scale, shift, slope) that every consumer applies while loading -- the activated tensor is
    never written to HBM;
  * skip-concat (reference ``unet/unet.py:62-67``) is zero-copy: the two producers write into channel slices
    of one buffer, and the per-channel transform vectors of the buffer are the concatenation of theirs.
Gradients mirror this: each buffer has a gradient twin; a conv block turns the incoming ``d a`` into ``d y``
in place (BatchNorm backward), then runs the weight- and data-gradient kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _lib
from ._lib import BIU_BF16, BIU_F32, BN_MAX_PARTIALS, biu_act, biu_xform, check, lib

LRELU_SLOPE = 0.1      # nn.LeakyReLU(negative_slope=0.1), reference unet/unet.py:58


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# Under stream capture the side stream forks off the capturing stream (wait_event on an event recorded there) and is joined again by the waits
# the eager path makes anyway (the decoder's first use of the composed weights; the hand-over of the deferred gradients), so the captured graph
# holds the same two branches.  BIU_DISABLE=capturefork keeps a captured step single-stream.
_NO_CAPTURE_FORK = os.environ.get("BIU_DISABLE", "").find("capturefork") >= 0
# voxels up to which a 3-D block's weight gradient runs on the side stream beside its data gradient (BIU_SIDE_WGRAD_VOX=0: never)
_SIDE_WGRAD_VOX = int(os.environ.get("BIU_SIDE_WGRAD_VOX", str(4 * 32 ** 3)))

class Buf:
    """One HBM buffer [N,D,H,W,C] + gradient twin + the consumer-side transform vectors of its channels."""

    def __init__(self, eng: "Engine", n, d, h, w, c):
        self.eng = eng
        self.shape = (n, d, h, w, c)
        self.t = torch.empty(self.shape, dtype=eng.tdtype, device=eng.device)
        self.g: Optional[torch.Tensor] = None
        self.scale = torch.ones(c, dtype=torch.float32, device=eng.device)
        self.shift = torch.zeros(c, dtype=torch.float32, device=eng.device)
        self.slope = torch.ones(c, dtype=torch.float32, device=eng.device)
        self.leaves: Dict[Tuple[int, int], bool] = {}       # (c0, c) -> gradient written this backward?
        self.lazy: Dict[Tuple[int, int], bool] = {}         # (c0, c) -> transform is not the identity
        self.ncons: Dict[Tuple[int, int], int] = {}         # (c0, c) -> number of nodes that read this slice
        self.nwr: Dict[Tuple[int, int], int] = {}           # (c0, c) -> gradient contributions received this backward
        self.producer: Dict[Tuple[int, int], "Node"] = {}   # (c0, c) -> ConvBlockNode that wrote it (if any)
        self.acts = []                                      # every Act handed out (release() clears their pointers)

    def slice(self, c0: int, c: int, lazy: bool) -> "Act":
        self.leaves[(c0, c)] = False
        self.lazy[(c0, c)] = lazy
        return Act(self, c0, c, [(c0, c)])

    def full(self) -> "Act":
        keys = sorted(self.leaves)
        cov = 0
        for c0, c in keys:
            assert c0 == cov, "concat buffer has a hole"
            cov += c
        assert cov == self.shape[4]
        return Act(self, 0, cov, keys)

    def grad(self) -> torch.Tensor:
        assert self.t is not None, "a released buffer has no gradient either"
        if self.g is None:
            self.g = torch.empty(self.shape, dtype=self.eng.tdtype, device=self.eng.device)
        return self.g

    def release(self):
        """The tensor is never materialised (the op that would write it runs folded into its consumer): free it.  Its Acts keep their
        channel counts and extents for the bookkeeping of the graph; their pointers become NULL, which every entry point refuses."""
        self.t = None
        self.g = None
        for a in self.acts:
            a._a.p = None


class Act:
    """A channel slice of a Buf as seen by a kernel (biu_act)."""

    def __init__(self, buf: Buf, c0: int, c: int, leaves):
        self.buf, self.c0, self.c, self.leaves = buf, c0, c, leaves
        n, d, h, w, ctot = buf.shape
        isz = buf.t.element_size()
        self.n, self.d, self.h, self.w = n, d, h, w
        self._a = biu_act(buf.t.data_ptr() + c0 * isz, n, d, h, w, c, ctot)
        self._g = None
        self._x = None
        self.is_input = False
        buf.acts.append(self)

    @property
    def nvox(self):
        return self.n * self.d * self.h * self.w

    @property
    def space(self):
        return (self.n, self.d, self.h, self.w)

    def a(self):
        return C.byref(self._a)

    def g(self):
        if self._g is None:
            gt = self.buf.grad()
            self._g = biu_act(gt.data_ptr() + self.c0 * gt.element_size(), self.n, self.d, self.h, self.w, self.c,
                              self.buf.shape[4])
        return C.byref(self._g)

    def is_lazy(self):
        return any(self.buf.lazy[k] for k in self.leaves)

    def xf(self):
        if not self.is_lazy():
            return None
        if self._x is None:
            o = 4 * self.c0
            self._x = biu_xform(self.buf.scale.data_ptr() + o, self.buf.shift.data_ptr() + o,
                                self.buf.slope.data_ptr() + o)
        return C.byref(self._x)

    def vec(self, name):   # view of a transform vector restricted to this slice
        return getattr(self.buf, name)[self.c0:self.c0 + self.c]

    # gradient bookkeeping -------------------------------------------------------------------------
    def g_written(self) -> bool:
        flags = [self.buf.leaves[k] for k in self.leaves]
        assert all(flags) or not any(flags), "partially written concat gradient"
        return flags[0]

    def mark_g(self):
        for k in self.leaves:
            self.buf.leaves[k] = True
            self.buf.nwr[k] = self.buf.nwr.get(k, 0) + 1

    def last_writer_producer(self):
        """ConvBlockNode that produced this slice if the caller is the last reader still to contribute its gradient."""
        if len(self.leaves) != 1:
            return None
        k = self.leaves[0]
        if self.buf.ncons.get(k, 0) - self.buf.nwr.get(k, 0) != 1:
            return None
        up = self.buf.producer.get(k)
        if up is None or not getattr(up, "batch_stats", False) or up.y.c != self.c or up.y.c0 != self.c0:
            return None
        return up

    def consumed(self):
        for k in self.leaves:
            self.buf.ncons[k] = self.buf.ncons.get(k, 0) + 1

    def sole_conv_block_producer(self):
        """The ConvBlockNode whose output this Act is, if this Act is one slice with exactly one reader."""
        if len(self.leaves) != 1 or self.buf.ncons.get(self.leaves[0], 0) != 1:
            return None
        return self.buf.producer.get(self.leaves[0])


class CatAct:
    """Channel concatenation (part0 | part1) of two dense tensors that stay in separate buffers: what the decoder's
    ``torch.cat`` becomes when the two-source kernels can serve its consumer (``biu_conv_*_cat``).  Only a ConvBlockNode reads it."""

    def __init__(self, p0: Act, p1: Act):
        assert p0.space == p1.space
        self.parts = (p0, p1)
        self.c = p0.c + p1.c
        self.n, self.d, self.h, self.w = p0.n, p0.d, p0.h, p0.w
        self.is_input = False

    @property
    def nvox(self):
        return self.parts[0].nvox

    @property
    def space(self):
        return self.parts[0].space

    def consumed(self):
        for p in self.parts:
            p.consumed()


class CatBuf:
    """Stand-in for the concat Buf of a decoder level: ``slice`` hands out the two dense parts, ``full`` their concatenation."""

    def __init__(self, eng: "Engine", space, c_first, c_second):
        n, d, h, w = space
        self.c_first, self.c_second = c_first, c_second
        self.bufs = (eng.new_buf(n, d, h, w, c_first), eng.new_buf(n, d, h, w, c_second))
        self.acts = [None, None]

    def slice(self, c0: int, c: int, lazy: bool) -> Act:
        k = 0 if (c0, c) == (0, self.c_first) else 1
        assert (c0, c) == ((0, self.c_first) if k == 0 else (self.c_first, self.c_second)), "a planar concat has exactly two slices"
        self.acts[k] = self.bufs[k].slice(0, c, lazy)
        return self.acts[k]

    def full(self) -> CatAct:
        assert self.acts[0] is not None and self.acts[1] is not None, "both parts must be produced before the concatenation is read"
        return CatAct(self.acts[0], self.acts[1])


# ======================================================================================================
# nodes
# ======================================================================================================
class Node:
    params: Sequence[nn.Parameter] = ()
    label = "node"

    def fwd(self, eng: "Engine"):
        raise NotImplementedError

    def bwd(self, eng: "Engine"):
        raise NotImplementedError


def _ksize(w: torch.Tensor):
    if w.dim() == 5:
        return tuple(w.shape[2:])
    return (1,) + tuple(w.shape[2:])


class ConvBlockNode(Node):
    """Conv(k=3, pad=dil, dilation=dil) -> BatchNorm -> LeakyReLU(0.1) -> Dropout(p=0)  [unet/unet.py:54-60].

    forward : y = conv(T_in(x)) + b ; batch statistics of y ; (scale, shift) of the consumer transform.
    backward: d a -> d y (BatchNorm + LeakyReLU backward, in place) ; dW, db ; d x.
    """

    def __init__(self, eng, seq: nn.Sequential, xin: Act, yout: Act, dropout_follows: bool = False, fold_src: Optional[Act] = None,
                 convt: Optional["ConvTNode"] = None):
        """``fold_src``: ``xin`` is the nearest-neighbour up-sampling (x2) of this coarse activation
        (multi_output_unet3d/multi_output_unet3d.py:138-139).  The forward then runs folded on the coarse tensor
        (``biu_upconv_fwd``: 8 parity classes x 2x2x2 taps instead of 27 taps, include/biu.h) when the kernel serves the shape, and
        the data gradient goes straight to ``fold_src`` (``biu_upconv_bwd_data``) and the weight gradient reads ``fold_src``
        (``biu_upconv_bwd_weight_bn``): with all three (``fold_all``) the up-sampled tensor ``xin`` is never needed and the up-sampling
        node is skipped."""
        conv, bn = seq[0], seq[1]
        self.conv, self.bn, self.xin, self.y = conv, bn, xin, yout
        self.fold_src, self.fold_slot, self.fold_dg_slot, self.fold_wg, self.fold_all = None, None, None, False, False
        # ``convt``: xin = concat(convt's output, skip) (unet3d/unet3d.py:84-90).  ConvTranspose + concat + this conv then run as ONE op with the up
        # half folded onto convt's coarse input (include/biu.h: biu_foldt_*), forward and backward; the ConvT node is skipped altogether and
        # this node delivers the gradients of its parameters too.
        self.foldt, self.foldt_blob, self.foldt_ver = None, None, None
        self.fold_ws, self.fold_ev = None, None      # own workspace + events of the folded weight gradient's side-stream chain rule
        self.wg_ws, self.wg_ev = None, None          # ... of a small layer's side-stream weight gradient
        # the block's activation as a leaky slope: LeakyReLU(s) -> s, ReLU -> 0 (Unet_v0 / BabyUnet), none or a later
        # non-piecewise-linear one (the attention gate's Sigmoid, applied by GateNode) -> 1
        act = seq[2] if len(seq) > 2 else None
        self.slope = float(act.negative_slope) if isinstance(act, nn.LeakyReLU) else (0.0 if isinstance(act, nn.ReLU) else 1.0)
        xin.consumed()
        if len(yout.leaves) == 1:
            yout.buf.producer[yout.leaves[0]] = self
        self.red_partial: Optional[torch.Tensor] = None      # BatchNorm-backward sums delivered by the consumer's dgrad
        self.red_nblk = 0
        self.kd, self.kh, self.kw = _ksize(conv.weight)
        self.dil = int(conv.dilation[0])
        drop = seq[3] if len(seq) > 3 else None
        if drop is not None and getattr(drop, "p", 0.0) != 0.0 and not dropout_follows:
            raise NotImplementedError("Dropout p != 0 needs a DropoutNode behind the block (models.Unet_v0 / BabyUnet build one)")
        assert tuple(conv.weight.shape[:2]) == (yout.c, xin.c), (conv.weight.shape, yout.c, xin.c)
        assert xin.space == yout.space
        self.params = [conv.weight, conv.bias, bn.weight, bn.bias]
        cout = yout.c
        dev = eng.device
        self.save_mean = torch.empty(cout, dtype=torch.float32, device=dev)
        self.save_invstd = torch.empty(cout, dtype=torch.float32, device=dev)
        yout.vec("slope").fill_(self.slope)
        eng.need_partial(cout)
        eng.need_partial_floats(lib.biu_conv_fwd_stats_floats(yout.a(), self.kd))
        if (convt is not None and isinstance(xin, CatAct) and len(xin.parts) == 2 and xin.parts[0] is convt.y and convt.kd == 2
                and (self.kd, self.kh, self.kw, self.dil) == (3, 3, 3, 1) and convt.up.bias is not None
                and lib.biu_foldt_ok(convt.xin.a(), xin.parts[1].a(), yout.a(), eng.dtype)
                and lib.biu_foldt_bwd_weight_workspace(convt.xin.c, xin.parts[1].c, cout, eng.dtype) > 0):
            self.foldt = convt
            lo, skip = convt.xin, xin.parts[1]
            self.foldt_blob = torch.empty(lib.biu_foldt_packed_bytes(lo.c, skip.c, cout, eng.dtype), dtype=torch.uint8, device=dev)
            eng.need_partial_floats(lib.biu_foldt_fwd_stats_floats(lo.a(), yout.a()))
            eng.need_ws(lib.biu_foldt_bwd_weight_workspace(lo.c, skip.c, cout, eng.dtype))
            self.params = [conv.weight, conv.bias, bn.weight, bn.bias]
        if (fold_src is not None and (self.kd, self.kh, self.kw, self.dil) == (3, 3, 3, 1) and not isinstance(xin, CatAct)
                and fold_src.c == xin.c and lib.biu_upconv_ok(fold_src.a(), yout.a(), eng.dtype)):
            self.fold_src = fold_src
            self.fold_slot = {"buf": torch.empty(lib.biu_upconv_packed_bytes(0, xin.c, cout, eng.dtype), dtype=torch.uint8, device=dev), "ver": None}
            nb1 = lib.biu_upconv_packed_bytes(1, xin.c, cout, eng.dtype)
            if nb1:
                self.fold_dg_slot = {"buf": torch.empty(nb1, dtype=torch.uint8, device=dev), "ver": None}
            wsf = lib.biu_upconv_bwd_weight_workspace(xin.c, cout, eng.dtype)
            if wsf:
                self.fold_wg = True
                eng.need_ws(wsf)
            self.fold_all = self.fold_dg_slot is not None and self.fold_wg
            eng.need_partial_floats(lib.biu_upconv_fwd_stats_floats(fold_src.a(), yout.a()))
        self.pk_f = eng.packed_slot(0, xin.c, cout, self.kd, self.kh, self.kw, self.dil)
        self.pk_b = eng.packed_slot(1, xin.c, cout, self.kd, self.kh, self.kw, self.dil)
        self.ws_bytes = lib.biu_conv_bwd_weight_workspace(xin.c, cout, self.kd, self.kh, self.kw, eng.dtype)
        eng.need_ws(self.ws_bytes)
        # scratch of the input-channel split (small fp32 grids): the forward writes y, the data gradient the input's gradient(s).
        # Sized here into the engine's one workspace -- the library owns no memory (include/biu.h, biu_conv_split_workspace)
        eng.need_ws(lib.biu_conv_split_workspace(xin.c, yout.a(), None, self.kd, self.kh, self.kw, self.dil, eng.dtype))
        if isinstance(xin, CatAct):
            eng.need_ws(lib.biu_conv_split_workspace(cout, xin.parts[0].a(), xin.parts[1].a(), self.kd, self.kh, self.kw, self.dil, eng.dtype))
        else:
            eng.need_ws(lib.biu_conv_split_workspace(cout, xin.a(), None, self.kd, self.kh, self.kw, self.dil, eng.dtype))

    def fwd(self, eng):
        st = _stream()
        w, b = self.conv.weight.data, self.conv.bias.data if self.conv.bias is not None else None
        packed = eng.pack(self.pk_f, 0, self.conv.weight, self.xin.c, self.y.c, self.kd, self.kh, self.kw)
        bn = self.bn
        scale, shift = self.y.vec("scale"), self.y.vec("shift")
        cat = self.xin.parts if isinstance(self.xin, CatAct) else None
        if self.foldt is not None:
            return self._fwd_foldt(eng, st)
        folded = None
        if self.fold_src is not None:                       # up-sampling + conv on the coarse tensor: weights folded once per version
            ver = (self.conv.weight.data_ptr(), self.conv.weight._version)
            if self.fold_slot["ver"] != ver:
                check(lib.biu_upconv_pack(0, _ptr(w), self.xin.c, self.y.c, eng.dtype, _ptr(self.fold_slot["buf"]), st), "upconv_pack")
                self.fold_slot["ver"] = ver
            folded = _ptr(self.fold_slot["buf"])
        if eng.bn_training(bn):
            # convolution + BatchNorm statistics in one call (the MFMA kernel reduces them in its epilogue)
            nblk = C.c_int(0)
            if folded:
                check(lib.biu_upconv_fwd(self.fold_src.a(), self.fold_src.xf(), folded, _ptr(b), self.y.a(), _ptr(eng.partial),
                                         eng.partial.numel(), C.byref(nblk), eng.dtype, st), "upconv_fwd")
            elif cat:
                check(lib.biu_conv_fwd_cat(cat[0].a(), cat[0].xf(), cat[1].a(), cat[1].xf(), _ptr(w), packed, _ptr(b), self.kd,
                                           self.kh, self.kw, self.dil, self.y.a(), _ptr(eng.partial), eng.partial.numel(),
                                           C.byref(nblk), _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "conv_fwd_cat")
            else:
                check(lib.biu_conv_fwd_stats(self.xin.a(), self.xin.xf(), _ptr(w), packed, _ptr(b), self.kd, self.kh, self.kw,
                                             self.dil, self.y.a(), _ptr(eng.partial), eng.partial.numel(), C.byref(nblk), _ptr(eng.ws), eng.ws_bytes,
                                             eng.dtype, st), "conv_fwd_stats")
            mom = bn.momentum if bn.momentum is not None else 0.1
            track = bn.track_running_stats and bn.running_mean is not None
            check(lib.biu_bn_finalize(_ptr(eng.partial), nblk.value, self.y.c, float(self.y.nvox), _ptr(bn.weight.data),
                                      _ptr(bn.bias.data), _ptr(bn.running_mean) if track else None,
                                      _ptr(bn.running_var) if track else None, mom, bn.eps, _ptr(scale), _ptr(shift),
                                      _ptr(self.save_mean), _ptr(self.save_invstd), st), "bn_finalize")
            if track and bn.num_batches_tracked is not None:
                eng.nbt_bump.append(bn.num_batches_tracked)
            self.batch_stats = True
        else:
            if folded:
                check(lib.biu_upconv_fwd(self.fold_src.a(), self.fold_src.xf(), folded, _ptr(b), self.y.a(), None, 0, None, eng.dtype, st),
                      "upconv_fwd")
            elif cat:
                check(lib.biu_conv_fwd_cat(cat[0].a(), cat[0].xf(), cat[1].a(), cat[1].xf(), _ptr(w), packed, _ptr(b), self.kd,
                                           self.kh, self.kw, self.dil, self.y.a(), None, 0, None, _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "conv_fwd_cat")
            else:
                check(lib.biu_conv_fwd(self.xin.a(), self.xin.xf(), _ptr(w), packed, _ptr(b), self.kd, self.kh, self.kw,
                                       self.dil, self.y.a(), _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "conv_fwd")
            check(lib.biu_bn_eval_affine(self.y.c, _ptr(bn.weight.data), _ptr(bn.bias.data), _ptr(bn.running_mean),
                                         _ptr(bn.running_var), bn.eps, _ptr(scale), _ptr(shift), st), "bn_eval_affine")
            self._eval_saved(eng)
            self.batch_stats = False

    def _eval_saved(self, eng):
        """Eval-mode forward under autograd (``model.eval(); loss.backward()``: frozen-BatchNorm fine-tuning, which the reference's plain
        nn.BatchNorm allows, unet/unet.py:54-60): the running statistics play the part of the saved batch statistics in the backward's
        (sum dz, sum dz * yhat).  Two C-sized vector ops, only when a backward can follow."""
        if eng.grad_mode:
            bn = self.bn
            self.save_mean.copy_(bn.running_mean)
            torch.rsqrt(bn.running_var.float() + bn.eps, out=self.save_invstd)

    def _foldt_version(self):
        ct = self.foldt
        ps = (self.conv.weight, self.conv.bias, ct.up.weight, ct.up.bias)
        return tuple((p.data_ptr(), p._version) for p in ps if p is not None)

    def _foldt_packed(self, st, eng=None):
        ct = self.foldt
        ver = self._foldt_version()
        if self.foldt_ver != ver:
            check(lib.biu_foldt_pack(_ptr(self.conv.weight.data), _ptr(self.conv.bias.data) if self.conv.bias is not None else None,
                                     _ptr(ct.up.weight.data), _ptr(ct.up.bias.data), ct.xin.c, ct.y.c, self.xin.parts[1].c, self.y.c, self._dtype,
                                     _ptr(self.foldt_blob), st), "foldt_pack")
            self.foldt_ver = ver
        elif eng is not None and eng._fold_ev is not None:
            # the image was packed ahead on the engine's side stream (Engine._prepack_folds): order this stream behind it
            torch.cuda.current_stream().wait_event(eng._fold_ev)
        return _ptr(self.foldt_blob)

    def _fwd_foldt(self, eng, st):
        self._dtype = eng.dtype
        ct, skip, bn = self.foldt, self.xin.parts[1], self.bn
        blob = self._foldt_packed(st, eng)
        scale, shift = self.y.vec("scale"), self.y.vec("shift")
        if eng.bn_training(bn):
            nblk = C.c_int(0)
            check(lib.biu_foldt_fwd(ct.xin.a(), ct.xin.xf(), skip.a(), skip.xf(), blob, self.y.a(), _ptr(eng.partial), eng.partial.numel(), C.byref(nblk),
                                    eng.dtype, st), "foldt_fwd")
            mom = bn.momentum if bn.momentum is not None else 0.1
            track = bn.track_running_stats and bn.running_mean is not None
            check(lib.biu_bn_finalize(_ptr(eng.partial), nblk.value, self.y.c, float(self.y.nvox), _ptr(bn.weight.data),
                                      _ptr(bn.bias.data), _ptr(bn.running_mean) if track else None,
                                      _ptr(bn.running_var) if track else None, mom, bn.eps, _ptr(scale), _ptr(shift),
                                      _ptr(self.save_mean), _ptr(self.save_invstd), st), "bn_finalize")
            if track and bn.num_batches_tracked is not None:
                eng.nbt_bump.append(bn.num_batches_tracked)
            self.batch_stats = True
        else:
            check(lib.biu_foldt_fwd(ct.xin.a(), ct.xin.xf(), skip.a(), skip.xf(), blob, self.y.a(), None, 0, None, eng.dtype, st), "foldt_fwd")
            check(lib.biu_bn_eval_affine(self.y.c, _ptr(bn.weight.data), _ptr(bn.bias.data), _ptr(bn.running_mean),
                                         _ptr(bn.running_var), bn.eps, _ptr(scale), _ptr(shift), st), "bn_eval_affine")
            self._eval_saved(eng)
            self.batch_stats = False

    def _bwd_foldt(self, eng, st, scale, shift, slope, A, B, Cc, dw, dy_sum=None):
        ct, skip, y = self.foldt, self.xin.parts[1], self.y
        lo = ct.xin
        dwt, dbt = eng.new_grad(ct.up.weight), eng.new_grad(ct.up.bias)
        side = eng.chain_stream()
        if side is None:
            check(lib.biu_foldt_bwd_weight_bn(lo.a(), lo.xf(), skip.a(), skip.xf(), y.g(), y.a(), scale, shift, slope, _ptr(A), _ptr(B), _ptr(Cc), _ptr(dy_sum),
                                              _ptr(self.conv.weight.data), _ptr(ct.up.weight.data), _ptr(ct.up.bias.data), ct.y.c, _ptr(dw), _ptr(dwt),
                                              _ptr(dbt), _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "foldt_bwd_weight_bn")
            eng.add_grad(ct.up.weight, dwt)
            eng.add_grad(ct.up.bias, dbt)
            deferred = False
        else:
            # the passes over the tensors here; the chain rule on their tables (seven small launches, 0.12-0.16 ms per level at cfg4, needed by
            # the optimizer only) on the side stream, beside the data gradient and the next layers: its gradients are handed over one node later
            if self.fold_ws is None:
                self.fold_ws = torch.empty(lib.biu_foldt_bwd_weight_workspace(lo.c, skip.c, y.c, eng.dtype), dtype=torch.uint8, device=eng.device)
                self.fold_ev = (torch.cuda.Event(), torch.cuda.Event())
            args = (lo.a(), lo.xf(), skip.a(), skip.xf(), y.g(), y.a(), scale, shift, slope, _ptr(A), _ptr(B), _ptr(Cc), _ptr(dy_sum),
                    _ptr(self.conv.weight.data), _ptr(ct.up.weight.data), _ptr(ct.up.bias.data), ct.y.c, _ptr(dw), _ptr(dwt), _ptr(dbt),
                    _ptr(self.fold_ws), self.fold_ws.numel(), eng.dtype)
            check(lib.biu_foldt_bwd_weight_bn_phase(*args, 1 | 4, st), "foldt_bwd_weight_bn_phase(1 | 4)")          # skip half (da -> dy) + G
            # (G on the side stream as well -- mask 4 | 2 there -- measured 11.38 -> 11.40 ms: two persistent MFMA kernels only share the chip)
            cur = torch.cuda.current_stream()
            ready, done = self.fold_ev
            ready.record(cur)
            side.wait_event(ready)
            label = lib.label
            with torch.cuda.stream(side):
                lib.label = label + "/chain"
                check(lib.biu_foldt_bwd_weight_bn_phase(*args, 2, _stream()), "foldt_bwd_weight_bn_phase(2)")              # border sums + chain rule
                done.record(side)
            lib.label = label
            eng.defer_grads(done, [(ct.up.weight, dwt), (ct.up.bias, dbt), (self.conv.weight, dw)], keep=(A, B, Cc, dy_sum))
            deferred = True
        blob = self._foldt_packed(st)
        want_lo = eng.wants_grad(lo)
        up = _fusable_producer(lo) if want_lo else None
        if not want_lo:
            raise NotImplementedError("foldt: the coarse input of a decoder level always wants its gradient")
        if up is not None:
            need = lib.biu_foldt_bwd_data_bnred_floats(lo.a())
            if up.red_partial is None or up.red_partial.numel() < need:
                up.red_partial = torch.empty(need, dtype=torch.float32, device=eng.device)
            n_up = C.c_int(0)
            check(lib.biu_foldt_bwd_data(y.g(), blob, lo.g(), 0, skip.g(), int(skip.g_written()), lo.a(), *up.red_coeffs(), _ptr(up.red_partial),
                                         up.red_partial.numel(), C.byref(n_up), _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "foldt_bwd_data")
            up.red_nblk = n_up.value
        else:
            check(lib.biu_foldt_bwd_data(y.g(), blob, lo.g(), int(lo.g_written()), skip.g(), int(skip.g_written()), None, None, None, None, None, None,
                                         None, 0, None, _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "foldt_bwd_data")
        lo.mark_g()
        skip.mark_g()
        return deferred

    def bwd(self, eng):
        if not self.y.g_written():
            return            # no gradient reaches this block (e.g. Siam 'control' branch)
        st = _stream()
        y, cout = self.y, self.y.c
        scale, shift, slope = _ptr(y.vec("scale")), _ptr(y.vec("shift")), _ptr(y.vec("slope"))
        nblk = C.c_int(0)
        if self.red_nblk:
            # the kernel that wrote d loss / d a (the reader's data gradient) already reduced (sum dz, sum dz*yhat)
            partial, nblk.value, self.red_nblk = self.red_partial, self.red_nblk, 0
        else:
            partial = eng.partial
            check(lib.biu_bn_bwd_reduce(y.g(), y.a(), scale, shift, slope, _ptr(self.save_mean), _ptr(self.save_invstd),
                                        _ptr(partial), C.byref(nblk), eng.dtype, st), "bn_bwd_reduce")
        dgamma, dbeta = eng.new_grad(self.bn.weight), eng.new_grad(self.bn.bias)
        A, B, Cc = eng.coef[0][:cout], eng.coef[1][:cout], eng.coef[2][:cout]
        dw = eng.new_grad(self.conv.weight)
        dy_sum = None
        if self.batch_stats:
            check(lib.biu_bn_bwd_finalize(_ptr(partial), nblk.value, cout, float(y.nvox), scale, _ptr(self.save_mean),
                                          _ptr(self.save_invstd), _ptr(dgamma), _ptr(dbeta), _ptr(A), _ptr(B), _ptr(Cc), st),
                  "bn_bwd_finalize")
            # d loss / d conv-bias: the bias is removed again by the batch mean, so sum_v dy == 0 identically
            # (A*S1 + B*M*mean + C*M cancels term by term); the reference's value is pure rounding noise (~1e-8).
            # Emit the exact zero instead of spending a pass over dy on it.
            db = eng.zero_like_bias(self.conv.bias) if self.conv.bias is not None else None
        else:
            # eval-mode BatchNorm: constants instead of statistics -- dy = scale * dz, and the conv bias gradient sum_v dy = scale * sum dz
            # is a real number again (it is also what the folded decoder's ConvT-bias chain rule needs: biu_foldt_bwd_weight_bn, dy_sum)
            dy_sum = torch.empty(cout, dtype=torch.float32, device=eng.device)
            check(lib.biu_bn_bwd_finalize_eval(_ptr(partial), nblk.value, cout, scale, _ptr(dgamma), _ptr(dbeta), _ptr(dy_sum), _ptr(A), _ptr(B),
                                               _ptr(Cc), st), "bn_bwd_finalize_eval")
            db = dy_sum if self.conv.bias is not None else None
        # BatchNorm+LeakyReLU backward (da -> dy, in place) rides inside the weight-gradient kernel's tile loader
        cat = self.xin.parts if isinstance(self.xin, CatAct) else None
        if self.foldt is not None:
            if not self._bwd_foldt(eng, st, scale, shift, slope, A, B, Cc, dw, dy_sum):
                eng.add_grad(self.conv.weight, dw)
            if db is not None:
                eng.add_grad(self.conv.bias, db)
            eng.add_grad(self.bn.weight, dgamma)
            eng.add_grad(self.bn.bias, dbeta)
            return
        if cat:
            check(lib.biu_conv_bwd_weight_cat(cat[0].a(), cat[0].xf(), cat[1].a(), cat[1].xf(), y.g(), y.a(), scale, shift, slope,
                                              _ptr(A), _ptr(B), _ptr(Cc), self.kd, self.kh, self.kw, self.dil, _ptr(dw), _ptr(eng.ws),
                                              eng.ws_bytes, eng.dtype, st), "conv_bwd_weight_cat")
        elif self.fold_wg:
            check(lib.biu_upconv_bwd_weight_bn(self.fold_src.a(), self.fold_src.xf(), y.g(), y.a(), scale, shift, slope, _ptr(A), _ptr(B),
                                               _ptr(Cc), _ptr(dw), _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "upconv_bwd_weight_bn")
        elif self.kd == 3 and eng.tdtype == torch.bfloat16 and y.nvox <= _SIDE_WGRAD_VOX and eng.chain_stream() is not None:
            # small volumes (the 32^3 / 16^3 levels of a 128^3 U-Net): weight and data gradient of a block each fill a fraction of the chip.
            # da -> dy here, then the weight gradient (needed by the optimizer only) on the side stream BESIDE the data gradient; its own
            # accumulator workspace, the parameter gradient handed over a node or two later (Engine._flush_deferred)
            side = eng.chain_stream()
            check(lib.biu_bn_bwd_apply(y.g(), y.a(), scale, shift, slope, _ptr(A), _ptr(B), _ptr(Cc), y.g(), eng.dtype, st), "bn_bwd_apply")
            if self.wg_ws is None:
                self.wg_ws = torch.empty(max(lib.biu_conv_bwd_weight_workspace(self.xin.c, cout, self.kd, self.kh, self.kw, eng.dtype), 16), dtype=torch.uint8,
                                         device=eng.device)
                self.wg_ev = (torch.cuda.Event(), torch.cuda.Event())
            ready, done = self.wg_ev
            ready.record(torch.cuda.current_stream())
            side.wait_event(ready)
            label = lib.label
            with torch.cuda.stream(side):
                lib.label = label + "/wgrad"
                check(lib.biu_conv_bwd_weight(self.xin.a(), self.xin.xf(), y.g(), self.kd, self.kh, self.kw, self.dil, _ptr(dw), None, _ptr(self.wg_ws),
                                              self.wg_ws.numel(), eng.dtype, _stream()), "conv_bwd_weight")
                done.record(side)
            lib.label = label
            eng.defer_grads(done, [(self.conv.weight, dw)])
            dw = None
        else:
            check(lib.biu_conv_bwd_weight_bn(self.xin.a(), self.xin.xf(), y.g(), y.a(), scale, shift, slope, _ptr(A), _ptr(B),
                                             _ptr(Cc), self.kd, self.kh, self.kw, self.dil, _ptr(dw), _ptr(eng.ws), eng.ws_bytes,
                                             eng.dtype, st), "conv_bwd_weight_bn")
        if dw is not None:
            eng.add_grad(self.conv.weight, dw)
        if db is not None:
            eng.add_grad(self.conv.bias, db)
        eng.add_grad(self.bn.weight, dgamma)
        eng.add_grad(self.bn.bias, dbeta)
        if cat:
            packed = eng.pack(self.pk_b, 1, self.conv.weight, self.xin.c, cout, self.kd, self.kh, self.kw)
            check(lib.biu_conv_bwd_data_cat(y.g(), _ptr(self.conv.weight.data), packed, self.kd, self.kh, self.kw, self.dil,
                                            cat[0].g(), int(cat[0].g_written()), cat[1].g(), int(cat[1].g_written()), _ptr(eng.ws), eng.ws_bytes, eng.dtype, st),
                  "conv_bwd_data_cat")
            cat[0].mark_g()
            cat[1].mark_g()
        elif self.fold_dg_slot is not None and eng.wants_grad(self.fold_src):
            # folded: d loss / d (coarse input) in one launch; xin's gradient is never written, so the up-sampling node's backward returns
            ver = (self.conv.weight.data_ptr(), self.conv.weight._version)
            if self.fold_dg_slot["ver"] != ver:
                check(lib.biu_upconv_pack(1, _ptr(self.conv.weight.data), self.xin.c, cout, eng.dtype, _ptr(self.fold_dg_slot["buf"]), st), "upconv_pack")
                self.fold_dg_slot["ver"] = ver
            check(lib.biu_upconv_bwd_data(y.g(), _ptr(self.fold_dg_slot["buf"]), self.fold_src.g(), int(self.fold_src.g_written()), eng.dtype, st),
                  "upconv_bwd_data")
            self.fold_src.mark_g()
        elif eng.wants_grad(self.xin):
            packed = eng.pack(self.pk_b, 1, self.conv.weight, self.xin.c, cout, self.kd, self.kh, self.kw)
            up = _fusable_producer(self.xin)
            if up is not None:
                part = up.red_buffer(eng, self.kd, 0)
                n_up = C.c_int(0)
                check(lib.biu_conv_bwd_data_bnred(y.g(), _ptr(self.conv.weight.data), packed, self.kd, self.kh, self.kw,
                                                  self.dil, self.xin.g(), self.xin.a(), *up.red_coeffs(), _ptr(part),
                                                  part.numel(), C.byref(n_up), _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "conv_bwd_data_bnred")
                up.red_nblk = n_up.value
            else:
                check(lib.biu_conv_bwd_data(y.g(), _ptr(self.conv.weight.data), packed, self.kd, self.kh, self.kw, self.dil,
                                            self.xin.g(), int(self.xin.g_written()), _ptr(eng.ws), eng.ws_bytes, eng.dtype, st), "conv_bwd_data")
            self.xin.mark_g()

    # ---- receiving side of the fused BatchNorm-backward reduction -----------------------------------
    def red_buffer(self, eng, kd, transposed) -> torch.Tensor:
        need = lib.biu_bwd_data_bnred_floats(self.y.a(), kd, transposed)
        if self.red_partial is None or self.red_partial.numel() < need:
            self.red_partial = torch.empty(need, dtype=torch.float32, device=eng.device)
        return self.red_partial

    def red_coeffs(self):
        y = self.y
        return (_ptr(y.vec("scale")), _ptr(y.vec("shift")), _ptr(y.vec("slope")), _ptr(self.save_mean),
                _ptr(self.save_invstd))


def _fusable_producer(xin: Act):
    """ConvBlockNode whose BatchNorm-backward sums the data-gradient kernel writing xin's gradient can also produce:
    xin is that block's output, this node is its only reader (so the gradient is complete after this one write) and the
    block normalised with batch statistics."""
    if xin.g_written():
        return None
    up = xin.sole_conv_block_producer()
    if up is None or not getattr(up, "batch_stats", False) or up.y.c != xin.c or up.y.c0 != xin.c0:
        return None
    return up


class ConvTNode(Node):
    """ConvTranspose(k=2, stride=2) [unet/unet.py:38, unet3d/unet3d.py:40]; output is final (no BN)."""

    def __init__(self, eng, up: nn.Module, xin: Act, yout: Act):
        self.up, self.xin, self.y = up, xin, yout
        xin.consumed()
        w = up.weight
        self.kd = 2 if w.dim() == 5 else 1
        assert tuple(w.shape[:2]) == (xin.c, yout.c)
        self.params = [up.weight, up.bias]
        self.folded_into = None              # the ConvBlockNode that computes ConvT + concat + conv as one op (biu_foldt_*): this node then does nothing
        self.pk_f = eng.packed_slot_convt(0, xin.c, yout.c, self.kd)
        self.pk_b = eng.packed_slot_convt(1, xin.c, yout.c, self.kd)
        eng.need_ws(lib.biu_convt_bwd_weight_workspace(xin.c, yout.c, self.kd, eng.dtype))

    def fwd(self, eng):
        if self.folded_into is not None:
            return
        packed = eng.pack_convt(self.pk_f, 0, self.up.weight, self.xin.c, self.y.c, self.kd)
        check(lib.biu_convt_fwd(self.xin.a(), self.xin.xf(), _ptr(self.up.weight.data), packed, _ptr(self.up.bias.data),
                                self.kd, self.y.a(), eng.dtype, _stream()), "convt_fwd")

    def bwd(self, eng):
        if self.folded_into is not None or not self.y.g_written():
            return
        st = _stream()
        dw, db = eng.new_grad(self.up.weight), eng.new_grad(self.up.bias)
        check(lib.biu_convt_bwd_weight(self.xin.a(), self.xin.xf(), self.y.g(), self.kd, _ptr(dw), _ptr(db), _ptr(eng.ws),
                                       eng.ws_bytes, eng.dtype, st), "convt_bwd_weight")
        eng.add_grad(self.up.weight, dw)
        eng.add_grad(self.up.bias, db)
        if eng.wants_grad(self.xin):
            packed = eng.pack_convt(self.pk_b, 1, self.up.weight, self.xin.c, self.y.c, self.kd)
            up = _fusable_producer(self.xin)
            if up is not None:
                part = up.red_buffer(eng, self.kd, 1)
                n_up = C.c_int(0)
                check(lib.biu_convt_bwd_data_bnred(self.y.g(), _ptr(self.up.weight.data), packed, self.kd, self.xin.g(),
                                                   self.xin.a(), *up.red_coeffs(), _ptr(part), part.numel(), C.byref(n_up),
                                                   eng.dtype, st), "convt_bwd_data_bnred")
                up.red_nblk = n_up.value
            else:
                check(lib.biu_convt_bwd_data(self.y.g(), _ptr(self.up.weight.data), packed, self.kd, self.xin.g(),
                                             int(self.xin.g_written()), eng.dtype, st), "convt_bwd_data")
            self.xin.mark_g()


class ResampleNode(Node):
    """MaxPool(2,2) / nearest x0.5 / nearest x2: output is materialised *activated* data (identity transform)."""

    def __init__(self, eng, kind: str, xin: Act, yout: Act):
        assert kind in ("maxpool", "down", "up", "trilinear")
        self.kind, self.xin, self.y = kind, xin, yout
        self.only_for_backward = False       # an up-sampling whose only reader folds it into its forward: needed by that reader's backward alone
        self.skip = False                    # ... and whose reader folds its backward too: the up-sampled tensor is never needed
        xin.consumed()

    def fwd(self, eng):
        if self.skip or (self.only_for_backward and not eng.grad_mode):
            return
        f = {"maxpool": lib.biu_maxpool_fwd, "down": lib.biu_nearest_down_fwd, "up": lib.biu_nearest_up_fwd,
             "trilinear": lib.biu_trilinear_up_fwd}[self.kind]
        check(f(self.xin.a(), self.xin.xf(), self.y.a(), eng.dtype, _stream()), self.kind + "_fwd")

    def bwd(self, eng):
        if not self.y.g_written() or not eng.wants_grad(self.xin):
            return
        acc, st = int(self.xin.g_written()), _stream()
        if self.kind == "maxpool":
            up = self.xin.last_writer_producer() if self.xin.xf() is not None else None
            if up is not None:
                # the pool is the last reader of its input: the finished gradient is reduced for the producer's BatchNorm here
                part = up.red_buffer(eng, up.kd, 0)
                n_up = C.c_int(0)
                check(lib.biu_maxpool_bwd_bnred(self.xin.a(), self.xin.xf(), self.y.g(), self.xin.g(), acc, _ptr(up.save_mean),
                                                _ptr(up.save_invstd), _ptr(part), part.numel(), C.byref(n_up), eng.dtype, st),
                      "maxpool_bwd_bnred")
                up.red_nblk = n_up.value
            else:
                check(lib.biu_maxpool_bwd(self.xin.a(), self.xin.xf(), self.y.g(), self.xin.g(), acc, eng.dtype, st), "maxpool_bwd")
        elif self.kind == "down":
            check(lib.biu_nearest_down_bwd(self.y.g(), self.xin.g(), acc, eng.dtype, st), "nearest_down_bwd")
        elif self.kind == "trilinear":
            check(lib.biu_trilinear_up_bwd(self.y.g(), self.xin.g(), acc, eng.dtype, st), "trilinear_up_bwd")
        else:
            check(lib.biu_nearest_up_bwd(self.y.g(), self.xin.g(), acc, eng.dtype, st), "nearest_up_bwd")
        self.xin.mark_g()


class MaxJoinNode(Node):
    """torch.maximum(m4, mm4) -- Siam 'max' join [siam_unet/siam_unet.py:117]."""

    def __init__(self, eng, a: Act, b: Act, out: Act):
        self.a_, self.b_, self.y = a, b, out
        a.consumed()
        b.consumed()

    def fwd(self, eng):
        check(lib.biu_max_join_fwd(self.a_.a(), self.a_.xf(), self.b_.a(), self.b_.xf(), self.y.a(), eng.dtype, _stream()),
              "max_join_fwd")

    def bwd(self, eng):
        if not self.y.g_written():
            return
        assert self.a_.g_written() == self.b_.g_written()
        check(lib.biu_max_join_bwd(self.a_.a(), self.a_.xf(), self.b_.a(), self.b_.xf(), self.y.g(), self.a_.g(),
                                   self.b_.g(), int(self.a_.g_written()), eng.dtype, _stream()), "max_join_bwd")
        self.a_.mark_g()
        self.b_.mark_g()


class XCorrNode(Node):
    """Depth-wise cross-correlation of the two pooled bottlenecks -- Siam 'corr' join [siam_unet/siam_unet.py:75-83,115]."""

    def __init__(self, eng, a: Act, b: Act, out: Act):
        self.a_, self.b_, self.y = a, b, out
        a.consumed()
        b.consumed()

    def fwd(self, eng):
        check(lib.biu_xcorr_fwd(self.a_.a(), self.a_.xf(), self.b_.a(), self.b_.xf(), self.y.a(), eng.dtype, _stream()), "xcorr_fwd")

    def bwd(self, eng):
        if not self.y.g_written():
            return
        assert self.a_.g_written() == self.b_.g_written()
        check(lib.biu_xcorr_bwd(self.a_.a(), self.a_.xf(), self.b_.a(), self.b_.xf(), self.y.g(), self.a_.g(), self.b_.g(),
                                int(self.a_.g_written()), eng.dtype, _stream()), "xcorr_bwd")
        self.a_.mark_g()
        self.b_.mark_g()


class CopyNode(Node):
    """out = T(x) into another slice (Siam concat of the two pooled bottlenecks, 'control' join)."""

    def __init__(self, eng, xin: Act, out: Act):
        self.xin, self.y = xin, out
        xin.consumed()

    def fwd(self, eng):
        check(lib.biu_xform_apply(self.xin.a(), self.xin.xf(), self.y.a(), eng.dtype, _stream()), "xform_apply")

    def bwd(self, eng):
        if not self.y.g_written() or not eng.wants_grad(self.xin):
            return
        assert not self.xin.is_lazy(), "CopyNode backward expects an identity transform"
        check(lib.biu_act_add(self.y.g(), self.xin.g(), int(self.xin.g_written()), eng.dtype, _stream()), "act_add")
        self.xin.mark_g()


class DropoutNode(Node):
    """nn.Dropout2d(p) behind a conv block (Unet_v0 / BabyUnet ``middle_conv2``, unet/unet_v0.py:31): whole channels of a
    sample are zeroed with probability p, the rest scaled by 1/(1-p).  ReLU / LeakyReLU commute with a non-negative factor,
    so sample n is materialised as T_n(y) with (scale, shift) * m[n, :] -- one ``biu_xform_apply`` per sample on the (small)
    bottleneck tensor; eval mode is the plain T.  ``mask_override`` ([N, C] of 0/1) pins the draw for parity tests."""

    def __init__(self, eng, drop: nn.Module, xin: Act, out: Act):
        self.drop, self.xin, self.y = drop, xin, out
        xin.consumed()
        self.mask_override: Optional[torch.Tensor] = None
        self.factor: Optional[torch.Tensor] = None           # [N, C] multiplier of the last forward
        self._keep = []

    def _sample(self, act: Act, tensor: torch.Tensor, n: int) -> biu_act:
        per = act.d * act.h * act.w * act.buf.shape[4] * tensor.element_size()
        return biu_act(tensor.data_ptr() + act.c0 * tensor.element_size() + n * per, 1, act.d, act.h, act.w, act.c, act.buf.shape[4])

    def fwd(self, eng):
        x, st = self.xin, _stream()
        p = float(self.drop.p)
        if self.drop.training and p > 0.0:
            if self.mask_override is not None:
                keep = self.mask_override.to(device=eng.device, dtype=torch.float32)
            else:
                keep = torch.bernoulli(torch.full((x.n, x.c), 1.0 - p, device=eng.device))
            self.factor = keep / (1.0 - p)
        else:
            self.factor = torch.ones((x.n, x.c), device=eng.device)
        sc = (x.vec("scale")[None, :] * self.factor).contiguous()
        sh = (x.vec("shift")[None, :] * self.factor).contiguous()
        sl = x.vec("slope").contiguous()
        self._keep = [sc, sh, sl]
        for n in range(x.n):
            xf = biu_xform(sc[n].data_ptr(), sh[n].data_ptr(), sl.data_ptr())
            a_in, a_out = self._sample(x, x.buf.t, n), self._sample(self.y, self.y.buf.t, n)
            check(lib.biu_xform_apply(C.byref(a_in), C.byref(xf), C.byref(a_out), eng.dtype, st), "dropout_fwd")

    def bwd(self, eng):
        if not self.y.g_written() or not eng.wants_grad(self.xin):
            return
        assert not self.xin.g_written(), "DropoutNode is the only reader of its input"
        st = _stream()
        f = self.factor.contiguous()
        zero, one = torch.zeros_like(f[0]), torch.ones_like(f[0])
        self._keep += [f, zero, one]
        for n in range(self.xin.n):
            xf = biu_xform(f[n].data_ptr(), zero.data_ptr(), one.data_ptr())           # T(v) = factor * v
            g_out, g_in = self._sample(self.y, self.y.buf.grad(), n), self._sample(self.xin, self.xin.buf.grad(), n)
            check(lib.biu_xform_apply(C.byref(g_out), C.byref(xf), C.byref(g_in), eng.dtype, st), "dropout_bwd")
        self.xin.mark_g()


class AddReluNode(Node):
    """relu(T(a) + T(b)) -- attention gate, ``psi = self.relu(g1 + x1)`` [unet/attention_unet.py:177]; materialised."""

    def __init__(self, eng, a: Act, b: Act, out: Act):
        self.a_, self.b_, self.y = a, b, out
        a.consumed()
        b.consumed()

    def fwd(self, eng):
        check(lib.biu_add_relu_fwd(self.a_.a(), self.a_.xf(), self.b_.a(), self.b_.xf(), self.y.a(), eng.dtype, _stream()), "add_relu_fwd")

    def bwd(self, eng):
        if not self.y.g_written():
            return
        assert self.a_.g_written() == self.b_.g_written()
        check(lib.biu_add_relu_bwd(self.y.a(), self.y.g(), self.a_.g(), self.b_.g(), int(self.a_.g_written()), eng.dtype, _stream()),
              "add_relu_bwd")
        self.a_.mark_g()
        self.b_.mark_g()


class GateNode(Node):
    """skip * sigmoid(T(psi)) -- attention gate output [unet/attention_unet.py:178-179]; psi has one channel; materialised."""

    def __init__(self, eng, e: Act, psi: Act, out: Act):
        assert psi.c == 1 and e.c == out.c
        self.e_, self.psi_, self.y = e, psi, out
        e.consumed()
        psi.consumed()

    def fwd(self, eng):
        check(lib.biu_gate_fwd(self.e_.a(), self.e_.xf(), self.psi_.a(), self.psi_.xf(), self.y.a(), eng.dtype, _stream()), "gate_fwd")

    def bwd(self, eng):
        if not self.y.g_written():
            return
        assert not self.psi_.g_written()
        check(lib.biu_gate_bwd(self.e_.a(), self.e_.xf(), self.psi_.a(), self.psi_.xf(), self.y.g(), self.e_.g(),
                               int(self.e_.g_written()), self.psi_.g(), eng.dtype, _stream()), "gate_bwd")
        self.e_.mark_g()
        self.psi_.mark_g()


_ACT_CODE = {None: 0, "none": 0, "sigmoid": 1, "tanh": 2, "relu": 3}


class HeadNode(Node):
    """1x1(x1) conv head + activation; fp32 NC[D]HW outputs [unet/unet.py:51,103-104; mo3d :164-168]."""

    def __init__(self, eng, conv: nn.Module, xin: Act, activation, want_logits: bool, want_act: bool):
        self.conv, self.xin = conv, xin
        xin.consumed()
        self.cout = conv.weight.shape[0]
        assert conv.weight.shape[1] == xin.c
        if activation not in _ACT_CODE:
            activation = None          # reference apply_activation: unknown strings fall through to identity
        self.activation = activation
        self.act = _ACT_CODE[activation]
        self.want_logits, self.want_act = want_logits, want_act
        self.params = [conv.weight, conv.bias]
        eng.need_ws(lib.biu_head_bwd_workspace(xin.c))
        self.logits = self.activated = None

    def out_shape(self, eng):
        x = self.xin
        return (x.n, self.cout, x.d, x.h, x.w) if eng.nd == 3 else (x.n, self.cout, x.h, x.w)

    def fwd(self, eng):
        shp = self.out_shape(eng)
        self.logits = torch.empty(shp, dtype=torch.float32, device=eng.device) if self.want_logits else None
        self.activated = torch.empty(shp, dtype=torch.float32, device=eng.device) if self.want_act else None
        w = self.conv.weight.data.reshape(self.cout, self.xin.c)
        check(lib.biu_head_fwd(self.xin.a(), self.xin.xf(), _ptr(w), _ptr(self.conv.bias.data), self.cout, self.act,
                               _ptr(self.logits), _ptr(self.activated), eng.dtype, _stream()), "head_fwd")

    def dlogits(self, eng, g_logits, g_act, a, dst=None, ctot=None, c0=0):
        """d loss / d logits from the caller's gradients w.r.t. (logits, activated output ``a``) in one kernel pass, written to
        ``dst`` (channels [c0, c0 + cout) of a [N, ctot, spatial] fp32 tensor); a gradient on the logits alone is used as is."""
        if g_act is None and dst is None and g_logits.dtype == torch.float32 and g_logits.is_contiguous():
            return g_logits
        shp = self.out_shape(eng)
        n, spatial = shp[0], 1
        for v in shp[2:]:
            spatial *= v
        if dst is None:
            dst, ctot, c0 = torch.empty(shp, dtype=torch.float32, device=eng.device), self.cout, 0
        f32 = lambda t: None if t is None else t.contiguous().float()
        gl, ga = f32(g_logits), f32(g_act)
        self._keep = (gl, ga, a)
        check(lib.biu_head_dlogits(_ptr(gl), _ptr(ga), _ptr(a) if ga is not None else None, self.act, n, self.cout, spatial,
                                   _ptr(dst), ctot, c0, _stream()), "head_dlogits")
        return dst

    def bwd_with(self, eng, dl: Optional[torch.Tensor]):
        if dl is None:
            return
        dl = dl.contiguous().float()
        st = _stream()
        dw, db = eng.new_grad(self.conv.weight), eng.new_grad(self.conv.bias)
        want_dx = eng.wants_grad(self.xin)
        assert not self.xin.g_written(), "heads that share a trunk go through Engine._backward_heads"
        up = self.xin.last_writer_producer() if (want_dx and self.xin.xf() is not None) else None
        if up is not None:
            # the head is the only reader of its input: its dx pass also reduces the producer's BatchNorm-backward sums
            part = up.red_buffer(eng, up.kd, 0)
            n_up = C.c_int(0)
            check(lib.biu_head_bwd_bnred(self.xin.a(), self.xin.xf(), _ptr(self.conv.weight.data), self.cout, _ptr(dl),
                                         self.xin.g(), _ptr(dw), _ptr(db), _ptr(eng.ws), eng.ws_bytes, _ptr(up.save_mean),
                                         _ptr(up.save_invstd), _ptr(part), part.numel(), C.byref(n_up), eng.dtype, st),
                  "head_bwd_bnred")
            up.red_nblk = n_up.value
        else:
            check(lib.biu_head_bwd(self.xin.a(), self.xin.xf(), _ptr(self.conv.weight.data), self.cout, _ptr(dl),
                                   self.xin.g() if want_dx else None, _ptr(dw), _ptr(db), _ptr(eng.ws), eng.ws_bytes,
                                   eng.dtype, st), "head_bwd")
        eng.add_grad(self.conv.weight, dw)
        eng.add_grad(self.conv.bias, db)
        if want_dx:
            self.xin.mark_g()


# ======================================================================================================
# engine
# ======================================================================================================
class Engine:
    """Static graph for one input shape.  ``nd`` = 2 or 3 spatial dims; ``tdtype`` = torch.float32 / bfloat16."""

    def __init__(self, device, tdtype, nd: int):
        if torch.device(device).type != "cuda":
            raise RuntimeError("bio_image_unet_amd executes only through its HIP kernels on an MI355X; "
                               f"got device '{device}'. There is no CPU fallback.")
        self.device, self.tdtype, self.nd = torch.device(device), tdtype, nd
        self.dtype = {torch.float32: BIU_F32, torch.bfloat16: BIU_BF16}[tdtype]
        self.nodes: List[Node] = []
        self.heads: List[HeadNode] = []
        self.bufs: List[Buf] = []
        self.inputs: List[Act] = []
        self._partial_c = 0
        self.ws_bytes = 0
        self.partial = self.ws = None
        self.coef = None
        self.grads: Dict[nn.Parameter, torch.Tensor] = {}
        self.nbt_bump: List[torch.Tensor] = []
        self.grad_mode = False
        self.module_training = True
        self.input_requires_grad = False
        self._packed: Dict[int, dict] = {}
        self._zero_flat, self._zero_used = None, 0
        self._slots: List[dict] = []
        self._job_key, self._job_tab = None, None
        self.grad_hook = None        # callable(param, grad) fired inside backward when a parameter's gradient is final (ddp.py)
        self.grad_alloc = None       # callable(param) -> fp32 tensor to compute the gradient INTO (a bucket view), or None (ddp.py)
        self.trace = None            # test hook: trace(phase, node, when) around every node ("fwd"/"bwd", node, "pre"/"post")
        self.generation = 0          # bumped by every forward: a backward must see the generation of ITS forward
        self._live = None            # weakref to the token of the autograd node that still needs this engine's buffers
        self._side = None            # side stream of the composed-weight packing of the folded decoder levels (_prepack_folds) and of their chain rule
        self._deferred = []
        self._no_side_chain = os.environ.get("BIU_DISABLE", "").find("sidechain") >= 0
        self._fold_ev = None

    def busy(self) -> bool:
        """A forward under autograd ran on this engine and its backward has neither run nor been dropped: the saved
        activations, statistics and partial sums are still owed to that node."""
        return self._live is not None and self._live() is not None

    def invalidate_packed(self):
        """Forget the cached MFMA weight packings (needed after a raw ``param.data`` write, which does not bump
        ``Tensor._version``)."""
        for s in self._slots:
            s["ver"] = None
        # the folded decoder levels keep their composed weights outside the slot list (biu_upconv_pack / biu_foldt_pack images)
        for nd_ in self.nodes:
            for name in ("fold_slot", "fold_dg_slot"):
                s = getattr(nd_, name, None)
                if s is not None:
                    s["ver"] = None
            if getattr(nd_, "foldt_ver", None) is not None:
                nd_.foldt_ver = None

    # ---- build helpers -------------------------------------------------------------------------------
    def new_buf(self, n, d, h, w, c) -> Buf:
        b = Buf(self, n, d, h, w, c)
        self.bufs.append(b)
        return b

    def new_cat(self, space, c_first, c_second, consumer_cout, kd):
        """Buffer(s) of a decoder level's concatenation (first | second).  Two dense buffers read through the two-source
        kernels when those can serve the consuming conv block, else one [.., c_first + c_second] buffer with channel slices."""
        n, d, h, w = space
        probe = lambda c: biu_act(256, n, d, h, w, c, c)              # dense rows, aligned dummy pointer
        a0, a1, ay = probe(c_first), probe(c_second), probe(consumer_cout)
        if lib.biu_conv_cat_ok(C.byref(a0), C.byref(a1), C.byref(ay), kd, 3, 3, 1, self.dtype) == 1:
            return CatBuf(self, space, c_first, c_second)
        return self.new_buf(n, d, h, w, c_first + c_second)

    def new_act(self, space, c, lazy: bool) -> Act:
        n, d, h, w = space
        return self.new_buf(n, d, h, w, c).slice(0, c, lazy)

    def add(self, node: Node):
        self.nodes.append(node)
        return node

    def need_partial(self, c):
        self._partial_c = max(self._partial_c, c)

    def need_partial_floats(self, n):
        self._partial_floats = max(getattr(self, "_partial_floats", 0), int(n))

    def need_ws(self, nbytes):
        self.ws_bytes = max(self.ws_bytes, int(nbytes))

    def finalize(self):
        dev = self.device
        c = max(self._partial_c, 1)
        self.partial = torch.empty(max(BN_MAX_PARTIALS * c * 2, getattr(self, "_partial_floats", 0)), dtype=torch.float32,
                                   device=dev)
        self.coef = torch.empty((3, c), dtype=torch.float32, device=dev)
        self.ws = torch.empty(max(self.ws_bytes, 16), dtype=torch.uint8, device=dev)
        self.params: List[nn.Parameter] = []
        seen = set()
        self._pcount: Dict[nn.Parameter, int] = {}      # contributions a parameter receives per backward (weight sharing: > 1)
        for nd_ in self.nodes:
            for p in nd_.params:
                if p is not None:
                    self._pcount[p] = self._pcount.get(p, 0) + 1
                    if id(p) not in seen:
                        seen.add(id(p))
                        self.params.append(p)

    # ---- MFMA weight packing cache ---------------------------------------------------------------------
    def packed_slot(self, kind, cin, cout, kd, kh, kw, dil):
        nbytes = lib.biu_conv_packed_bytes(kind, cin, cout, kd, kh, kw, dil, self.dtype)
        if nbytes == 0:
            return None
        slot = {"buf": torch.empty(nbytes, dtype=torch.uint8, device=self.device), "ver": None,
                "job": (0, kind, cin, cout, kd, kh, kw), "weight": None}
        self._slots.append(slot)
        return slot

    def pack(self, slot, kind, weight: nn.Parameter, cin, cout, kd, kh, kw):
        if slot is None:
            return None
        slot["weight"] = weight
        ver = (weight.data_ptr(), weight._version)
        if slot["ver"] != ver:
            check(lib.biu_conv_pack(kind, _ptr(weight.data), cin, cout, kd, kh, kw, self.dtype, _ptr(slot["buf"]), _stream()),
                  "conv_pack")
            slot["ver"] = ver
        return _ptr(slot["buf"])

    def packed_slot_convt(self, kind, cin, cout, kd):
        nbytes = lib.biu_convt_packed_bytes(kind, cin, cout, kd, self.dtype)
        if nbytes == 0:
            return None
        slot = {"buf": torch.empty(nbytes, dtype=torch.uint8, device=self.device), "ver": None,
                "job": (1, kind, cin, cout, kd, 2, 2), "weight": None}
        self._slots.append(slot)
        return slot

    def pack_convt(self, slot, kind, weight: nn.Parameter, cin, cout, kd):
        if slot is None:
            return None
        slot["weight"] = weight
        ver = (weight.data_ptr(), weight._version)
        if slot["ver"] != ver:
            check(lib.biu_convt_pack(kind, _ptr(weight.data), cin, cout, kd, self.dtype, _ptr(slot["buf"]), _stream()),
                  "convt_pack")
            slot["ver"] = ver
        return _ptr(slot["buf"])

    def pack_all(self):
        """Re-pack every stale weight tensor of the network in ONE launch (a step changes all of them; ~30 separate pack
        launches otherwise).  Slots learn their weight on first use, so the very first pass still packs one by one."""
        stale = [s for s in self._slots if s["weight"] is not None and s["ver"] != (s["weight"].data_ptr(), s["weight"]._version)]
        if len(stale) < 2:
            return
        key = tuple((id(s), s["weight"].data_ptr()) for s in stale)
        if self._job_key != key:                       # (re)build the device job table only when the set or a pointer changed
            from ._lib import biu_pack_job
            arr = (biu_pack_job * len(stale))()
            for j, s in zip(arr, stale):
                tr, kind, cin, cout, kd, kh, kw = s["job"]
                j.w, j.packed = s["weight"].data_ptr(), s["buf"].data_ptr()
                j.transposed, j.kind, j.cin, j.cout, j.kd, j.kh, j.kw, j.reserved = tr, kind, cin, cout, kd, kh, kw, 0
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._job_tab = raw.to(self.device)
            self._job_key = key
        check(lib.biu_pack_batch(_ptr(self._job_tab), len(stale), self.dtype, _stream()), "pack_batch")
        for s in stale:
            s["ver"] = (s["weight"].data_ptr(), s["weight"]._version)

    # ---- run-time helpers --------------------------------------------------------------------------------
    def bn_training(self, bn: nn.Module) -> bool:
        # nn.BatchNorm semantics: batch statistics when training, or when no running stats exist
        return bn.training or bn.running_mean is None

    def wants_grad(self, act: Act) -> bool:
        return self.input_requires_grad if act.is_input else True

    def new_input(self, space, c) -> Act:
        a = self.new_act(space, c, lazy=False)
        a.is_input = True
        self.inputs.append(a)
        return a

    def new_grad(self, p: nn.Parameter) -> torch.Tensor:
        if self.grad_alloc is not None and self._pcount.get(p, 1) == 1:      # (shared weights sum several contributions: own tensors)
            v = self.grad_alloc(p)
            if v is not None:
                return v
        return torch.empty(p.shape, dtype=torch.float32, device=self.device)

    def zero_like_bias(self, b: torch.Tensor) -> torch.Tensor:
        """Fresh zeros shaped like ``b``, carved from one flat buffer filled once per backward (one fill kernel per step
        instead of one per conv block)."""
        n = b.numel()
        if self._zero_flat is None or self._zero_used + n > self._zero_flat.numel() or self._zero_flat.dtype != b.dtype:
            self._zero_flat = torch.zeros(max(4096, 4 * n), dtype=b.dtype, device=b.device)
            self._zero_used = 0
        v = self._zero_flat[self._zero_used:self._zero_used + n].view_as(b)
        self._zero_used += n
        return v

    def add_grad(self, p: nn.Parameter, g: torch.Tensor):
        if p in self.grads:
            self.grads[p] = self.grads[p] + g
        else:
            self.grads[p] = g
        if self.grad_hook is not None:
            self._pseen[p] = self._pseen.get(p, 0) + 1
            if self._pseen[p] == self._pcount.get(p, 1):
                self.grad_hook(p, self.grads[p])

    # ---- execution ----------------------------------------------------------------------------------------
    def load_inputs(self, xs: Sequence[torch.Tensor]):
        st = _stream()
        for act, x in zip(self.inputs, xs):
            x = x.detach()
            if x.dtype == torch.uint8:
                # the reference's data contract: uint8 tiles scaled by 1/255 (unet/data.py:253-266, unet/predict.py:192-196);
                # the scaling rides in the layout kernel, the batch crossed PCIe as bytes
                x = x.contiguous()
                check(lib.biu_from_nchw_u8(_ptr(x), 1.0 / 255.0, act.a(), self.dtype, st), "from_nchw_u8")
                continue
            if x.dtype != torch.float32:
                x = x.float()
            x = x.contiguous()
            check(lib.biu_from_nchw(_ptr(x), act.a(), self.dtype, st), "from_nchw")

    def _prepack_folds(self):
        """The composed weights of the folded decoder levels (biu_foldt_pack: 3 x 216 small GEMMs per level, 0.07-0.13 ms at cfg4) depend on the
        parameters only: packed at the START of the forward on a side stream, they run beside the encoder instead of in front of the decoder
        (the decoder's first use waits on the event).  Not under stream capture (a captured step packs in line)."""
        self._fold_ev = None
        if os.environ.get("BIU_DISABLE", "").find("prepack") >= 0 or (torch.cuda.is_current_stream_capturing() and _NO_CAPTURE_FORK):
            return
        stale = [n for n in self.nodes if isinstance(n, ConvBlockNode) and n.foldt is not None and n.foldt_ver != n._foldt_version()]
        if not stale:
            return
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(cur)                      # the optimizer's update (and the last backward's reads of the old images) are on `cur`
        with torch.cuda.stream(self._side):
            st = _stream()
            for n in stale:
                n._dtype = self.dtype
                lib.label = n.label + ":pack"
                n._foldt_packed(st)
            self._fold_ev = torch.cuda.Event()
            self._fold_ev.record(self._side)

    def forward(self):
        self.generation += 1
        self.nbt_bump = []
        lib.label = "pack:fwd"
        self.pack_all()
        self._prepack_folds()
        for nd_ in self.nodes:
            lib.label = nd_.label + ":fwd"
            if self.trace:
                self.trace("fwd", nd_, "pre")
            nd_.fwd(self)
            if self.trace:
                self.trace("fwd", nd_, "post")
        if self.nbt_bump:       # a weight-shared block (Siam encoder) appears once per application
            counts: Dict[int, list] = {}
            for t in self.nbt_bump:
                counts.setdefault(id(t), [t, 0])[1] += 1
            by_k: Dict[int, list] = {}
            for t, k in counts.values():
                by_k.setdefault(k, []).append(t)
            for k, ts in by_k.items():
                torch._foreach_add_(ts, k)

    def chain_stream(self):
        """Side stream for work only the optimizer waits for (the chain rule of the folded decoder levels), or None: under stream capture, while a
        trace hook compares intermediate results, or with BIU_DISABLE=sidechain."""
        if self.trace or self._no_side_chain or (torch.cuda.is_current_stream_capturing() and _NO_CAPTURE_FORK):
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def defer_grads(self, done: "torch.cuda.Event", grads, keep=()):
        self._deferred.append((done, grads, keep))

    def _flush_deferred(self, keep_last: int):
        """Hand over the gradients whose side-stream work was enqueued at least `keep_last` nodes ago: the current stream waits for their event."""
        while len(self._deferred) > keep_last:
            done, grads, _ = self._deferred.pop(0)
            torch.cuda.current_stream().wait_event(done)
            for p, g in grads:
                self.add_grad(p, g)

    def backward(self, head_grads: Sequence[Optional[torch.Tensor]]):
        """head_grads[i] = None or (d loss / d logits, d loss / d activated output, activated output) of head i."""
        self.grads = {}
        self._deferred = []
        self._pseen: Dict[nn.Parameter, int] = {}
        self._zero_flat, self._zero_used = None, 0
        for b in self.bufs:
            for k in b.leaves:
                b.leaves[k] = False
                b.nwr[k] = 0
        lib.label = "head:bwd"
        if self.trace:
            self.trace("bwd", ("heads", head_grads), "pre")
        self._backward_heads(head_grads)
        if self.trace:
            self.trace("bwd", ("heads", head_grads), "post")
        for nd_ in reversed(self.nodes):
            if isinstance(nd_, HeadNode):
                continue
            lib.label = nd_.label + ":bwd"
            if self.trace:
                self.trace("bwd", nd_, "pre")
            self._flush_deferred(2)                      # (side-stream work is handed over two deferrals later, or at the end:
                                                         #  beside the persistent kernels of the main stream its small grids only get the CUs' spare slots)
            nd_.bwd(self)
            if self.trace:
                self.trace("bwd", nd_, "post")
        self._flush_deferred(0)
        return self.grads

    def _backward_heads(self, head_grads):
        """head_grads[i] = None or (g_logits, g_act, activated output) of head i."""
        live = [(h, g) for h, g in zip(self.heads, head_grads) if g is not None]
        if not live:
            return
        if len(live) == 1:
            h, (gl, ga, a) = live[0]
            h.bwd_with(self, h.dlogits(self, gl, ga, a))
            return
        # several heads read the same trunk output: one fused (cout-stacked) backward keeps d x single-pass; every head's
        # d logits goes straight into its channel slice of the stacked operand
        x = live[0][0].xin
        assert all(h.xin is x for h, _ in live)
        cout = sum(h.cout for h, _ in live)
        shp = live[0][0].out_shape(self)
        dl = torch.empty((shp[0], cout) + tuple(shp[2:]), dtype=torch.float32, device=self.device)
        key = tuple(id(h) for h, _ in live)
        if getattr(self, "_stack_key", None) != key:
            self._stack_key = key
            self._stack_w = torch.empty((cout, x.c), dtype=torch.float32, device=self.device)
        w, o = self._stack_w, 0
        for h, (gl, ga, a) in live:
            w[o:o + h.cout].copy_(h.conv.weight.data.reshape(h.cout, x.c))          # (cout x cin floats: tiny)
            h.dlogits(self, gl, ga, a, dst=dl, ctot=cout, c0=o)
            o += h.cout
        dw = torch.empty_like(w)
        db = torch.empty(cout, dtype=torch.float32, device=self.device)
        up = None
        if self.wants_grad(x) and x.xf() is not None and not x.g_written() and len(x.leaves) == 1:
            k = x.leaves[0]        # the heads are the trunk output's only readers: this one write completes its gradient
            cand = x.buf.producer.get(k)
            if (x.buf.ncons.get(k, 0) == len(live) and cand is not None and getattr(cand, "batch_stats", False)
                    and cand.y.c == x.c and cand.y.c0 == x.c0):
                up = cand
        if up is not None:
            part = up.red_buffer(self, up.kd, 0)
            n_up = C.c_int(0)
            check(lib.biu_head_bwd_bnred(x.a(), x.xf(), _ptr(w), cout, _ptr(dl), x.g(), _ptr(dw), _ptr(db), _ptr(self.ws), self.ws_bytes,
                                         _ptr(up.save_mean), _ptr(up.save_invstd), _ptr(part), part.numel(), C.byref(n_up), self.dtype,
                                         _stream()), "head_bwd_bnred(stacked)")
            up.red_nblk = n_up.value
        else:
            check(lib.biu_head_bwd(x.a(), x.xf(), _ptr(w), cout, _ptr(dl), x.g(), _ptr(dw), _ptr(db), _ptr(self.ws),
                                   self.ws_bytes, self.dtype, _stream()), "head_bwd(stacked)")
        x.mark_g()
        o = 0
        for h, _ in live:
            self.add_grad(h.conv.weight, dw[o:o + h.cout].reshape(h.conv.weight.shape).clone())
            self.add_grad(h.conv.bias, db[o:o + h.cout].clone())
            o += h.cout

    def input_grads(self) -> List[Optional[torch.Tensor]]:
        outs = []
        st = _stream()
        for act in self.inputs:
            if not self.input_requires_grad or not act.g_written():
                outs.append(None)
                continue
            shp = (act.n, act.c, act.d, act.h, act.w) if self.nd == 3 else (act.n, act.c, act.h, act.w)
            t = torch.empty(shp, dtype=torch.float32, device=self.device)
            ga = biu_act(act.buf.grad().data_ptr(), act.n, act.d, act.h, act.w, act.c, act.buf.shape[4])
            check(lib.biu_to_nchw(C.byref(ga), None, _ptr(t), self.dtype, st), "to_nchw")
            outs.append(t)
        return outs


# ======================================================================================================
# autograd glue at the module boundary
# ======================================================================================================
class _Token:
    """Lifetime marker of one autograd node: alive while that node can still ask the engine for its backward."""
    __slots__ = ("__weakref__",)


class _NetFn(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are the engine's kernel sequences."""

    @staticmethod
    def forward(ctx, eng: Engine, n_inputs: int, out_spec, *tensors):
        xs, ctx.n_inputs = tensors[:n_inputs], n_inputs
        if eng.busy():
            raise RuntimeError("engine re-entered while an earlier forward still awaits its backward (models take a second "
                               "engine for that case: this is a bug in the caller of engine.run)")
        eng.load_inputs(xs)
        eng.forward()
        ctx.eng, ctx.out_spec = eng, out_spec
        ctx.generation = eng.generation
        if eng.grad_mode:
            ctx.token = _Token()
            eng._live = weakref.ref(ctx.token)
        else:
            eng._live = None
        outs = []
        for hi, kind in out_spec:
            h = eng.heads[hi]
            outs.append(h.logits if kind == "logits" else h.activated)
        # the activated outputs are needed by the activations' backward; they are OUTPUTS of this node, so they go through
        # save_for_backward (a plain reference from the engine would keep the node -- and its claim on the engine -- alive)
        ctx.act_slot = {hi: i for i, (hi, kind) in enumerate(out_spec) if kind == "act"}
        ctx.save_for_backward(*[outs[i] for i in ctx.act_slot.values()])
        for h in eng.heads:
            h.logits = h.activated = None
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        eng: Engine = ctx.eng
        if ctx.generation != eng.generation:
            raise RuntimeError("backward of a forward whose activations were overwritten: the engine ran another forward of the "
                               "same input shape in between (generation %d, now %d)" % (ctx.generation, eng.generation))
        per_head: Dict[int, List[Optional[torch.Tensor]]] = {}
        for (hi, kind), g in zip(ctx.out_spec, gouts):
            slot = per_head.setdefault(hi, [None, None])
            slot[0 if kind == "logits" else 1] = g
        saved = dict(zip(ctx.act_slot.keys(), ctx.saved_tensors))
        head_grads = []
        for hi, h in enumerate(eng.heads):
            gl, ga = per_head.get(hi, [None, None])
            head_grads.append((gl, ga, saved.get(hi)) if (gl is not None or ga is not None) else None)
        grads = eng.backward(head_grads)
        dxs = eng.input_grads()
        pg = [grads.pop(p, None) for p in eng.params]
        eng.grads = {}              # sole owner is now autograd: AccumulateGrad can take the tensors without copying
        del grads
        ctx.token = None            # the engine's buffers are free again (a second backward of this node is not supported)
        eng._live = None
        eng.generation += 1
        return (None, None, None, *dxs, *pg)


def run(eng: Engine, xs: Sequence[torch.Tensor], out_spec):
    """Execute the graph under autograd.  Parameters are passed so that autograd routes their gradients."""
    eng.grad_mode = torch.is_grad_enabled()
    eng.input_requires_grad = eng.grad_mode and any(x.requires_grad for x in xs)
    return _NetFn.apply(eng, len(xs), tuple(out_spec), *xs, *eng.params)
