"""Per-kernel parity through the C ABI (libbiu_hip.so) against torch CPU fp32 functional ops -- the same ATen
kernels the reference's nn.Modules dispatch to.  Shapes include odd channel counts, channel-slice (pitch > C)
operands, 2-D (D = 1) and 3-D, dilation 2, and negative BatchNorm scales in the fused input transform."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.gpu_util import DT, XF, Dev, assert_close, check, lib, ptr, stream  # noqa: E402

DTYPES = ["f32", "bf16"]


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def conv_ref(x, w, b, dil):
    f = F.conv3d if w.dim() == 5 else F.conv2d
    return f(x, w, b, padding=dil, dilation=dil)


CONV_CASES = [
    # (nd, N, Cin, Cout, spatial, dil)
    (2, 2, 1, 4, (12, 20), 1),
    (2, 1, 5, 3, (9, 7), 2),
    (2, 2, 8, 16, (16, 16), 1),
    (3, 1, 3, 5, (4, 6, 10), 1),
    (3, 2, 4, 8, (8, 8, 8), 1),
    # first-layer special case (Cin = 1, Cout multiple of 8): vector-ALU kernels of biu_special.hip
    (3, 2, 1, 16, (6, 10, 12), 1),
    (3, 1, 1, 32, (4, 6, 8), 1),
    (3, 1, 1, 56, (3, 5, 7), 1),
    (2, 2, 1, 32, (20, 28), 1),
    (2, 1, 1, 64, (16, 16), 1),
    (2, 1, 1, 8, (9, 11), 1),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bwd(case, dtype):
    nd, n, cin, cout, sp, dil = case
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cout, cin, *([3] * nd), seed=2) * 0.3
    b = rnd(cout, seed=3)
    xf = XF(cin, seed=4)
    dx_ = Dev(x, dtype=dtype, pitch=cin + 3, c0=2)
    xr = dx_.ref().squeeze(2) if nd == 2 else dx_.ref()
    xa = xf.apply(xr).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yref = conv_ref(xa, wr, br, dil)
    dy = rnd(*yref.shape, seed=5)
    kd = 3 if nd == 3 else 1
    wd, bd = w.cuda(), b.cuda()
    yd = Dev(shape=(n, cout, 1 if nd == 2 else sp[0], sp[-2], sp[-1]), dtype=dtype, pitch=cout + 1, c0=1)
    check(lib.biu_conv_fwd(dx_.a(), xf.x(), ptr(wd), None, ptr(bd), kd, 3, 3, dil, yd.a(), None, 0, DT[dtype][1], stream()), "conv_fwd")
    got = yd.get(squeeze2d=(nd == 2))
    assert_close(got, yref.detach(), dtype, "conv_fwd")
    # backward: data gradient is wrt T(x) (the activated input), weight gradient sees T(x)
    dyd = Dev(dy, dtype=dtype)
    dyr = dyd.ref().squeeze(2) if nd == 2 else dyd.ref()
    yref.backward(dyr)
    dxd = Dev(shape=(n, cin, 1 if nd == 2 else sp[0], sp[-2], sp[-1]), dtype=dtype)
    check(lib.biu_conv_bwd_data(dyd.a(), ptr(wd), None, kd, 3, 3, dil, dxd.a(), 0, None, 0, DT[dtype][1], stream()), "conv_bwd_data")
    assert_close(dxd.get(squeeze2d=(nd == 2)), xa.grad, dtype, "conv_bwd_data")
    dw = torch.empty_like(wd)
    db = torch.empty_like(bd)
    ws = torch.empty(max(lib.biu_conv_bwd_weight_workspace(cin, cout, kd, 3, 3, DT[dtype][1]), 16), dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_bwd_weight(dx_.a(), xf.x(), dyd.a(), kd, 3, 3, dil, ptr(dw), ptr(db), ptr(ws), ws.numel(), DT[dtype][1],
                                  stream()), "conv_bwd_weight")
    assert_close(dw.cpu(), wr.grad, dtype, "conv_bwd_weight")
    assert_close(db.cpu(), br.grad, dtype, "conv dbias")
    # accumulate flag
    check(lib.biu_conv_bwd_data(dyd.a(), ptr(wd), None, kd, 3, 3, dil, dxd.a(), 1, None, 0, DT[dtype][1], stream()), "conv_bwd_data acc")
    assert_close(dxd.get(squeeze2d=(nd == 2)), 2 * xa.grad, dtype, "conv_bwd_data(accumulate)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 5, 1, 8, 12), (2, 16, 4, 6, 10), (1, 300, 1, 4, 4)])
def test_batchnorm_fwd_bwd(shape, dtype):
    n, c, d, h, w = shape
    y = rnd(*shape, seed=1) * 2 + 0.5
    yd = Dev(y, dtype=dtype, pitch=c + 8, c0=8)
    yr = yd.ref().requires_grad_(True)
    gamma = (rnd(c, seed=2) * 0.3 + 1).requires_grad_(True)
    beta = (rnd(c, seed=3) * 0.2).requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    out = F.leaky_relu(F.batch_norm(yr, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5), 0.1)
    # device
    partial = torch.empty(1024 * c * 2, device="cuda")
    nblk = C.c_int(0)
    check(lib.biu_bn_stats(yd.a(), ptr(partial), C.byref(nblk), DT[dtype][1], stream()), "bn_stats")
    g_d, b_d = gamma.detach().cuda(), beta.detach().cuda()
    rm_d, rv_d = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    scale, shift, mean, invstd = (torch.empty(c, device="cuda") for _ in range(4))
    check(lib.biu_bn_finalize(ptr(partial), nblk.value, c, float(n * d * h * w), ptr(g_d), ptr(b_d), ptr(rm_d), ptr(rv_d), 0.1, 1e-5,
                              ptr(scale), ptr(shift), ptr(mean), ptr(invstd), stream()), "bn_finalize")
    torch.testing.assert_close(rm_d.cpu(), rm, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv_d.cpu(), rv, rtol=1e-4, atol=1e-5)
    slope = torch.full((c,), 0.1, device="cuda")
    od = Dev(shape=shape, dtype=dtype)
    from tests.gpu_util import biu_xform
    xfs = biu_xform(scale.data_ptr(), shift.data_ptr(), slope.data_ptr())
    check(lib.biu_xform_apply(yd.a(), C.byref(xfs), od.a(), DT[dtype][1], stream()), "xform_apply")
    assert_close(od.get(), out.detach(), dtype, "bn+lrelu fwd")
    # backward
    da = rnd(*shape, seed=4)
    dad = Dev(da, dtype=dtype)
    out.backward(dad.ref())
    check(lib.biu_bn_bwd_reduce(dad.a(), yd.a(), ptr(scale), ptr(shift), ptr(slope), ptr(mean), ptr(invstd), ptr(partial),
                                C.byref(nblk), DT[dtype][1], stream()), "bn_bwd_reduce")
    dg, dbt, A, B, Cc = (torch.empty(c, device="cuda") for _ in range(5))
    check(lib.biu_bn_bwd_finalize(ptr(partial), nblk.value, c, float(n * d * h * w), ptr(scale), ptr(mean), ptr(invstd), ptr(dg),
                                  ptr(dbt), ptr(A), ptr(B), ptr(Cc), stream()), "bn_bwd_finalize")
    check(lib.biu_bn_bwd_apply(dad.a(), yd.a(), ptr(scale), ptr(shift), ptr(slope), ptr(A), ptr(B), ptr(Cc), dad.a(), DT[dtype][1],
                               stream()), "bn_bwd_apply")
    assert_close(dg.cpu(), gamma.grad, dtype, "dgamma")
    assert_close(dbt.cpu(), beta.grad, dtype, "dbeta")
    assert_close(dad.get(), yr.grad, dtype, "bn dy")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 5, 1, 8, 12), (1, 8, 4, 6, 10), (2, 16, 2, 4, 4)])
def test_maxpool_and_nearest(shape, dtype):
    n, c, d, h, w = shape
    nd = 2 if d == 1 else 3
    x = rnd(*shape, seed=1)
    x[0, 0, 0, 0, 0] = x[0, 0, 0, 0, 1] = 5.0            # a tie: gradient must go to the first maximum
    xf = XF(c, seed=2)
    xd = Dev(x, dtype=dtype, pitch=c + 8, c0=0)
    xa = xf.apply(xd.ref()).requires_grad_(True)
    xs = xa.squeeze(2) if nd == 2 else xa
    pool = F.max_pool2d if nd == 2 else F.max_pool3d
    ref = pool(xs, 2, 2)
    oshape = (n, c, max(d // 2, 1), h // 2, w // 2)
    od = Dev(shape=oshape, dtype=dtype)
    check(lib.biu_maxpool_fwd(xd.a(), xf.x(), od.a(), DT[dtype][1], stream()), "maxpool_fwd")
    assert_close(od.get(squeeze2d=nd == 2), ref.detach(), dtype, "maxpool_fwd")
    g = rnd(*ref.shape, seed=3)
    gd = Dev(g, dtype=dtype)
    gr = gd.ref().squeeze(2) if nd == 2 else gd.ref()
    ref.backward(gr)
    dxd = Dev(shape=shape, dtype=dtype, fill=0.0)
    check(lib.biu_maxpool_bwd(xd.a(), xf.x(), gd.a(), dxd.a(), 0, DT[dtype][1], stream()), "maxpool_bwd")
    assert_close(dxd.get(), xa.grad, dtype, "maxpool_bwd")
    check(lib.biu_maxpool_bwd(xd.a(), xf.x(), gd.a(), dxd.a(), 1, DT[dtype][1], stream()), "maxpool_bwd acc")
    assert_close(dxd.get(), 2 * xa.grad, dtype, "maxpool_bwd(accumulate)")
    # nearest x0.5 and x2 (3-D only in the reference; the kernels are dimension-agnostic)
    xa2 = xf.apply(xd.ref()).requires_grad_(True)
    xs2 = xa2.squeeze(2) if nd == 2 else xa2
    down = F.interpolate(xs2, scale_factor=0.5, mode="nearest")
    check(lib.biu_nearest_down_fwd(xd.a(), xf.x(), od.a(), DT[dtype][1], stream()), "nearest_down_fwd")
    assert_close(od.get(squeeze2d=nd == 2), down.detach(), dtype, "nearest_down_fwd")
    down.backward(gr)
    check(lib.biu_nearest_down_bwd(gd.a(), dxd.a(), 0, DT[dtype][1], stream()), "nearest_down_bwd")
    assert_close(dxd.get(), xa2.grad, dtype, "nearest_down_bwd")
    sm = rnd(*oshape, seed=5)
    sd_ = Dev(sm, dtype=dtype)
    sr = sd_.ref().requires_grad_(True)
    srs = sr.squeeze(2) if nd == 2 else sr
    up = F.interpolate(srs, scale_factor=2, mode="nearest")
    ud = Dev(shape=shape, dtype=dtype)
    check(lib.biu_nearest_up_fwd(sd_.a(), None, ud.a(), DT[dtype][1], stream()), "nearest_up_fwd")
    assert_close(ud.get(squeeze2d=nd == 2), up.detach(), dtype, "nearest_up_fwd")
    gu = Dev(rnd(*shape, seed=6), dtype=dtype)
    up.backward(gu.ref().squeeze(2) if nd == 2 else gu.ref())
    dsd = Dev(shape=oshape, dtype=dtype)
    check(lib.biu_nearest_up_bwd(gu.a(), dsd.a(), 0, DT[dtype][1], stream()), "nearest_up_bwd")
    assert_close(dsd.get(), sr.grad, dtype, "nearest_up_bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 2, 6, 3, (4, 6)), (3, 1, 4, 4, (2, 4, 6)), (3, 2, 16, 16, (2, 2, 2))])
def test_convtranspose(case, dtype):
    nd, n, cin, cout, sp = case
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cin, cout, *([2] * nd), seed=2) * 0.3
    b = rnd(cout, seed=3)
    xf = XF(cin, seed=4)
    xd = Dev(x, dtype=dtype)
    xr = xd.ref().squeeze(2) if nd == 2 else xd.ref()
    xa = xf.apply(xr).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    f = F.conv_transpose3d if nd == 3 else F.conv_transpose2d
    ref = f(xa, wr, br, stride=2)
    kd = 2 if nd == 3 else 1
    osp = tuple(2 * s for s in sp)
    oshape = (n, cout, 1 if nd == 2 else osp[0], osp[-2], osp[-1])
    yd = Dev(shape=oshape, dtype=dtype, pitch=cout + 5, c0=0)
    wd, bd = w.cuda(), b.cuda()
    check(lib.biu_convt_fwd(xd.a(), xf.x(), ptr(wd), None, ptr(bd), kd, yd.a(), DT[dtype][1], stream()), "convt_fwd")
    assert_close(yd.get(squeeze2d=nd == 2), ref.detach(), dtype, "convt_fwd")
    gd = Dev(rnd(*ref.shape, seed=5), dtype=dtype)
    ref.backward(gd.ref().squeeze(2) if nd == 2 else gd.ref())
    dxd = Dev(shape=(n, cin, 1 if nd == 2 else sp[0], sp[-2], sp[-1]), dtype=dtype)
    check(lib.biu_convt_bwd_data(gd.a(), ptr(wd), None, kd, dxd.a(), 0, DT[dtype][1], stream()), "convt_bwd_data")
    assert_close(dxd.get(squeeze2d=nd == 2), xa.grad, dtype, "convt_bwd_data")
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    ws = torch.empty(max(lib.biu_convt_bwd_weight_workspace(cin, cout, kd, DT[dtype][1]), 16), dtype=torch.uint8, device="cuda")
    check(lib.biu_convt_bwd_weight(xd.a(), xf.x(), gd.a(), kd, ptr(dw), ptr(db), ptr(ws), ws.numel(), DT[dtype][1], stream()),
          "convt_bwd_weight")
    assert_close(dw.cpu(), wr.grad, dtype, "convt dw")
    assert_close(db.cpu(), br.grad, dtype, "convt db")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 2, 7, 1, (6, 10), 1), (3, 1, 8, 2, (2, 4, 6), 0), (3, 2, 16, 3, (2, 4, 4), 2)])
def test_head(case, dtype):
    nd, n, cin, cout, sp, act = case
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cout, cin, seed=2) * 0.5
    b = rnd(cout, seed=3)
    xf = XF(cin, seed=4)
    xd = Dev(x, dtype=dtype)
    xr = xd.ref().squeeze(2) if nd == 2 else xd.ref()
    xa = xf.apply(xr).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    f = F.conv3d if nd == 3 else F.conv2d
    logits = f(xa, wr.view(cout, cin, *([1] * nd)), br)
    actf = {0: lambda t: t, 1: torch.sigmoid, 2: torch.tanh, 3: F.relu}[act]
    lo = torch.empty(logits.shape, device="cuda")
    ac = torch.empty(logits.shape, device="cuda")
    wd, bd = w.cuda(), b.cuda()
    check(lib.biu_head_fwd(xd.a(), xf.x(), ptr(wd), ptr(bd), cout, act, ptr(lo), ptr(ac), DT[dtype][1], stream()), "head_fwd")
    # outputs are fp32; the only bf16 effect is the stored input
    torch.testing.assert_close(lo.cpu(), logits.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(ac.cpu(), actf(logits).detach(), rtol=1e-4, atol=1e-4)
    dl = rnd(*logits.shape, seed=6)
    logits.backward(dl)
    dxd = Dev(shape=(n, cin, 1 if nd == 2 else sp[0], sp[-2], sp[-1]), dtype=dtype)
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    ws = torch.empty(lib.biu_head_bwd_workspace(cin), dtype=torch.uint8, device="cuda")
    dld = dl.cuda()
    check(lib.biu_head_bwd(xd.a(), xf.x(), ptr(wd), cout, ptr(dld), dxd.a(), ptr(dw), ptr(db), ptr(ws), ws.numel(), DT[dtype][1],
                           stream()), "head_bwd")
    assert_close(dxd.get(squeeze2d=nd == 2), xa.grad, dtype, "head dx")
    assert_close(dw.cpu(), wr.grad, "f32", "head dw")
    assert_close(db.cpu(), br.grad, "f32", "head db")


@pytest.mark.parametrize("dtype", DTYPES)
def test_elementwise_helpers(dtype):
    shape = (2, 8, 1, 4, 6)
    a, b = rnd(*shape, seed=1), rnd(*shape, seed=2)
    b[0, 0, 0, 0, 0] = a[0, 0, 0, 0, 0] = 1.0
    ad, bd = Dev(a, dtype=dtype), Dev(b, dtype=dtype)
    ar, br = ad.ref().requires_grad_(True), bd.ref().requires_grad_(True)
    ref = torch.maximum(ar, br)
    od = Dev(shape=shape, dtype=dtype)
    check(lib.biu_max_join_fwd(ad.a(), None, bd.a(), None, od.a(), DT[dtype][1], stream()), "max_join_fwd")
    assert_close(od.get(), ref.detach(), dtype, "max_join_fwd")
    gd = Dev(rnd(*shape, seed=3), dtype=dtype)
    ref.backward(gd.ref())
    dad, dbd = Dev(shape=shape, dtype=dtype), Dev(shape=shape, dtype=dtype)
    check(lib.biu_max_join_bwd(ad.a(), None, bd.a(), None, gd.a(), dad.a(), dbd.a(), 0, DT[dtype][1], stream()), "max_join_bwd")
    assert_close(dad.get(), ar.grad, dtype, "max_join da")
    assert_close(dbd.get(), br.grad, dtype, "max_join db")
    # nchw <-> channels-last
    x = rnd(2, 3, 4, 5, 6, seed=4)
    xd = Dev(shape=(2, 3, 4, 5, 6), dtype=dtype, pitch=7, c0=1)
    xs = x.cuda()
    check(lib.biu_from_nchw(ptr(xs), xd.a(), DT[dtype][1], stream()), "from_nchw")
    assert_close(xd.get(), x, dtype, "from_nchw")
    back = torch.empty_like(xs)
    check(lib.biu_to_nchw(xd.a(), None, ptr(back), DT[dtype][1], stream()), "to_nchw")
    torch.testing.assert_close(back.cpu(), xd.ref())


def test_adam_matches_torch():
    torch.manual_seed(0)
    ps = [torch.randn(s) for s in [(7,), (3, 5), (64, 3, 3, 3)]]
    gs = [torch.randn_like(p) for p in ps]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3)
    dp = [p.clone().cuda() for p in ps]
    m = [torch.zeros_like(p) for p in dp]
    v = [torch.zeros_like(p) for p in dp]
    for step in (1, 2, 3):
        for r, g in zip(ref, gs):
            r.grad = g.clone() * step
        opt.step()
        dg = [(g * step).cuda() for g in gs]
        mk = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.int64, device="cuda")
        tp, tg, tm, tv = mk(dp), mk(dg), mk(m), mk(v)          # keep the pointer tables alive across the launch
        numel = torch.tensor([p.numel() for p in dp], dtype=torch.int64, device="cuda")
        check(lib.biu_adam_step(len(dp), ptr(tp), ptr(tg), ptr(tm), ptr(tv), ptr(numel), 1e-3, 0.9, 0.999, 1e-8, step, 1.0,
                                stream()), "adam")
        torch.cuda.synchronize()
    for r, p in zip(ref, dp):
        torch.testing.assert_close(p.cpu(), r.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("max_norm", [1.0, 1e-2, 1e6])
def test_grad_clip_matches_torch(max_norm):
    """biu_grad_clip against torch.nn.utils.clip_grad_norm_ (multi_output_unet3d/train.py:201): the returned norm, the scaled gradients, and
    gradients left bit for bit alone when the norm is below max_norm.  Sizes off the 16-byte path and an unaligned view included."""
    torch.manual_seed(3)
    flat = torch.randn(1000, device="cuda")
    gs = [torch.randn(s, device="cuda") for s in [(7,), (3, 5), (64, 3, 3, 3), (1,), (4096, 33)]] + [flat[1:602]]     # (last: 4-byte aligned only)
    ref = [g.clone().cpu().requires_grad_(True) for g in gs]
    for r, g in zip(ref, gs):
        r.grad = g.clone().cpu()
    want_norm = torch.nn.utils.clip_grad_norm_(ref, max_norm=max_norm)
    before = [g.clone() for g in gs]
    n = len(gs)
    table = torch.tensor([g.data_ptr() for g in gs], dtype=torch.int64, device="cuda")
    numel = torch.tensor([g.numel() for g in gs], dtype=torch.int64, device="cuda")
    need = int(lib.biu_grad_clip_scratch_floats(n))
    scratch = torch.empty(need, device="cuda")
    total = torch.zeros(1, device="cuda")
    check(lib.biu_grad_clip(n, ptr(table), ptr(numel), max_norm, ptr(scratch), need, ptr(total), stream()), "grad_clip")
    torch.cuda.synchronize()
    assert abs(float(total) - float(want_norm)) <= 2e-6 * float(want_norm), (float(total), float(want_norm))
    for g, r, b in zip(gs, ref, before):
        if float(want_norm) + 1e-6 <= max_norm:
            assert torch.equal(g, b)                                           # coefficient clamped to 1: untouched
        torch.testing.assert_close(g.cpu(), r.grad, rtol=3e-6, atol=0.0)
    # too small a scratch is reported, not overrun
    assert lib.biu_grad_clip(n, ptr(table), ptr(numel), max_norm, ptr(scratch), need - 1, None, stream()) != 0


def test_shape_errors_are_reported_not_thrown():
    x = Dev(rnd(1, 4, 1, 8, 8), dtype="f32")
    y = Dev(shape=(1, 4, 1, 4, 8), dtype="f32")
    w = torch.zeros(4, 4, 3, 3, device="cuda")
    rc = lib.biu_conv_fwd(x.a(), None, ptr(w), None, None, 1, 3, 3, 1, y.a(), None, 0, 0, stream())
    assert rc == -1 and b"conv_fwd" in lib.biu_last_error()
    rc = lib.biu_maxpool_fwd(x.a(), None, x.a(), 0, stream())
    assert rc == -1


# ---------------------------------------------------------------------------------------------------------------
# MFMA implicit-GEMM path (packed weights): forward and data gradient
# ---------------------------------------------------------------------------------------------------------------
MFMA_CASES = [
    # (nd, N, Cin, Cout, spatial)   -- extents deliberately not multiples of the kernel's bricks
    (3, 1, 16, 32, (8, 16, 32)),
    (3, 2, 32, 16, (6, 10, 36)),
    (3, 1, 96, 32, (4, 8, 32)),
    (3, 1, 32, 64, (8, 8, 16)),
    (3, 1, 64, 128, (4, 4, 16)),
    (3, 1, 48, 48, (5, 7, 9)),
    # long enough columns for the rolling-window weight gradient (bf16: k_wgrad_roll): odd depth, H / W off the 8 x 16 window,
    # a 16-channel plain operand (half-empty tile), three tapped tiles
    (3, 2, 32, 48, (9, 20, 24)),
    (3, 1, 64, 16, (8, 24, 40)),
    (3, 1, 96, 32, (10, 24, 32)),
    # a batch of >= 8 images: the 2-D weight gradient takes the rolling-window kernel with the batch as its depth axis (odd batch, H / W off
    # the 8 x 16 window, 48 = one and a half channel tiles)
    (2, 8, 32, 32, (24, 40)),
    (2, 9, 64, 48, (20, 50)),
    (2, 2, 16, 32, (32, 32)),
    (2, 1, 64, 64, (24, 40)),
    (2, 1, 128, 128, (16, 16)),
    (2, 1, 32, 96, (20, 12)),
    # 16 output channels (forward) / 16 input channels (data gradient): the 16x16x32 MFMA kernel on bf16
    (3, 1, 64, 16, (5, 9, 20)),
    (2, 1, 64, 16, (20, 28)),
    (2, 2, 16, 64, (33, 17)),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", MFMA_CASES)
def test_conv_mfma_fwd_dgrad(case, dtype):
    nd, n, cin, cout, sp = case
    kd = 3 if nd == 3 else 1
    code = DT[dtype][1]
    nbytes = lib.biu_conv_packed_bytes(0, cin, cout, kd, 3, 3, 1, code)
    assert nbytes > 0, "this shape must be served by the MFMA kernels"
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cout, cin, *([3] * nd), seed=2) * (1.0 / (cin * 3 ** nd) ** 0.5)
    b = rnd(cout, seed=3)
    xf = XF(cin, seed=4)
    xd = Dev(x, dtype=dtype, pitch=cin + 16, c0=8)
    xr = xd.ref().squeeze(2) if nd == 2 else xd.ref()
    wq = w.bfloat16().float() if dtype == "bf16" else w          # the packed operand is stored in the compute dtype
    xa = xf.apply(xr)
    if dtype == "bf16":
        xa = xa.bfloat16().float()                               # T(x) is rounded to bf16 when staged into LDS
    xa.requires_grad_(True)
    yref = conv_ref(xa, wq, b, 1)
    wd, bd = w.cuda(), b.cuda()
    pk = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(0, ptr(wd), cin, cout, kd, 3, 3, code, ptr(pk), stream()), "conv_pack")
    oshape = (n, cout, 1 if nd == 2 else sp[0], sp[-2], sp[-1])
    yd = Dev(shape=oshape, dtype=dtype, pitch=cout + 8, c0=8)
    check(lib.biu_conv_fwd(xd.a(), xf.x(), ptr(wd), ptr(pk), ptr(bd), kd, 3, 3, 1, yd.a(), None, 0, code, stream()), "conv_fwd(mfma)")
    got = yd.get(squeeze2d=(nd == 2))
    t = dict(rtol=1e-4, atol=1e-4 * float(yref.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=1e-2 * float(yref.abs().max()))
    torch.testing.assert_close(got, yref.detach(), **t)
    # data gradient
    dyd = Dev(rnd(*yref.shape, seed=5), dtype=dtype)
    dyr = dyd.ref().squeeze(2) if nd == 2 else dyd.ref()
    yref.backward(dyr)
    nb2 = lib.biu_conv_packed_bytes(1, cin, cout, kd, 3, 3, 1, code)
    assert nb2 > 0
    pk2 = torch.empty(nb2, dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(1, ptr(wd), cin, cout, kd, 3, 3, code, ptr(pk2), stream()), "conv_pack(dgrad)")
    dxd = Dev(shape=(n, cin, 1 if nd == 2 else sp[0], sp[-2], sp[-1]), dtype=dtype, pitch=cin + 8, c0=0)
    check(lib.biu_conv_bwd_data(dyd.a(), ptr(wd), ptr(pk2), kd, 3, 3, 1, dxd.a(), 0, None, 0, code, stream()), "conv_bwd_data(mfma)")
    t2 = dict(rtol=1e-4, atol=1e-4 * float(xa.grad.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=1e-2 * float(xa.grad.abs().max()))
    torch.testing.assert_close(dxd.get(squeeze2d=(nd == 2)), xa.grad, **t2)
    check(lib.biu_conv_bwd_data(dyd.a(), ptr(wd), ptr(pk2), kd, 3, 3, 1, dxd.a(), 1, None, 0, code, stream()), "conv_bwd_data(mfma, acc)")
    torch.testing.assert_close(dxd.get(squeeze2d=(nd == 2)), 2 * xa.grad, rtol=t2["rtol"] * 2, atol=t2["atol"] * 2)
    # the pad region of the output buffer (channels outside the slice) must be untouched
    assert torch.isnan(yd.buf[..., :8].float()).all()
    # weight gradient (MFMA, split-K with fp32 atomics): reference sees the same rounded operands
    wq2 = wq.clone().requires_grad_(True)
    xa2 = xa.detach()
    conv_ref(xa2, wq2, None, 1).backward(dyr)
    wsz = lib.biu_conv_bwd_weight_workspace(cin, cout, kd, 3, 3, code)
    assert wsz > 0
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    dw = torch.full_like(wd, float("nan"))
    db = torch.empty_like(bd)
    check(lib.biu_conv_bwd_weight(xd.a(), xf.x(), dyd.a(), kd, 3, 3, 1, ptr(dw), ptr(db), ptr(ws), ws.numel(), code, stream()),
          "conv_bwd_weight(mfma)")
    t3 = dict(rtol=1e-3, atol=2e-4 * float(wq2.grad.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=1e-2 * float(wq2.grad.abs().max()))
    torch.testing.assert_close(dw.cpu(), wq2.grad, **t3)
    torch.testing.assert_close(db.cpu(), dyr.sum(dim=[0] + list(range(2, dyr.dim()))), rtol=1e-3, atol=1e-3 * float(dyr.abs().sum() ** 0.5))


# ---------------------------------------------------------------------------------------------------------------
# rolling-window 3x3x3 convolution with register-resident weights (biu_conv_roll.hip): narrow bf16 layers, forward + statistics and data
# gradient + the upstream block's BatchNorm-backward sums, through the ordinary entry points (tests/conftest.py: BIU_ROLL=always drops the
# size rule so that these small volumes take the kernel: windows of 8 x 32 in (H, W), depth segments, planes and columns off the window)
# ---------------------------------------------------------------------------------------------------------------
ROLL_CASES = [
    # (N, Cin, Cout, (D, H, W))
    (2, 16, 32, (7, 13, 37)),          # 32x32x16, two planes per step: odd depth, two windows each way with ragged edges
    (1, 32, 16, (9, 8, 32)),           # 16x16x32, one plane per step, one exact window
    (1, 32, 32, (20, 11, 40)),         # 32x32x16, 32-channel input: depth segments (20 planes over few columns)
    (3, 16, 32, (4, 5, 9)),            # minimum depth, a window mostly outside the volume
    (1, 32, 16, (24, 20, 70)),         # several depth segments and three windows along W
]


@pytest.mark.parametrize("case", ROLL_CASES)
def test_conv_roll_fwd_stats_and_dgrad_bnred(case):
    n, cin, cout, sp = case
    dtype, code = "bf16", DT["bf16"][1]
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cout, cin, 3, 3, 3, seed=2) * (1.0 / (cin * 27) ** 0.5)
    b = rnd(cout, seed=3)
    xf = XF(cin, seed=4)
    xd = Dev(x, dtype=dtype, pitch=cin + 16, c0=8)
    yd = Dev(shape=(n, cout, *sp), dtype=dtype, pitch=cout + 8, c0=8)
    xa = xf.apply(xd.ref()).bfloat16().float().requires_grad_(True)          # T(x) is rounded to bf16 where it is staged
    wq = w.bfloat16().float()
    yref = F.conv3d(xa, wq, b, padding=1)
    wd, bd = w.cuda(), b.cuda()
    pk = torch.empty(lib.biu_conv_packed_bytes(0, cin, cout, 3, 3, 3, 1, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(0, ptr(wd), cin, cout, 3, 3, 3, code, ptr(pk), stream()), "conv_pack")
    nfl = lib.biu_conv_fwd_stats_floats(yd.a(), 3)
    part = torch.full((nfl,), float("nan"), device="cuda")
    nblk = C.c_int(0)
    ws = torch.empty(16, dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_fwd_stats(xd.a(), xf.x(), ptr(wd), ptr(pk), ptr(bd), 3, 3, 3, 1, yd.a(), ptr(part), nfl, C.byref(nblk), ptr(ws), 0, code, stream()),
          "conv_fwd_stats(roll)")
    got = yd.get()
    torch.testing.assert_close(got, yref.detach(), rtol=1e-2, atol=1e-2 * float(yref.abs().max()))
    assert torch.isnan(yd.buf[..., :8].float()).all()                        # channels outside the slice untouched
    sums = part[:nblk.value * cout * 2].view(nblk.value, cout, 2).double().sum(0).cpu()
    gd = got.double()
    torch.testing.assert_close(sums[:, 0], gd.sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-4 * float(gd.abs().sum() / cout))
    torch.testing.assert_close(sums[:, 1], (gd * gd).sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-6)
    # the plain forward (no statistics, identity transform) through biu_conv_fwd
    yd2 = Dev(shape=(n, cout, *sp), dtype=dtype)
    check(lib.biu_conv_fwd(xd.a(), None, ptr(wd), ptr(pk), ptr(bd), 3, 3, 3, 1, yd2.a(), ptr(ws), 0, code, stream()), "conv_fwd(roll, identity)")
    yref2 = F.conv3d(xd.ref(), wq, b, padding=1)
    torch.testing.assert_close(yd2.get(), yref2, rtol=1e-2, atol=1e-2 * float(yref2.abs().max()))
    # data gradient + (sum dz, sum dz * yhat) of the upstream block whose raw output is y_up (the data gradient maps dy: cout -> cin channels)
    dyd = Dev(rnd(n, cout, *sp, seed=5), dtype=dtype)
    yref.backward(dyd.ref())
    pk2 = torch.empty(lib.biu_conv_packed_bytes(1, cin, cout, 3, 3, 3, 1, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(1, ptr(wd), cin, cout, 3, 3, 3, code, ptr(pk2), stream()), "conv_pack(dgrad)")
    dxd = Dev(shape=(n, cin, *sp), dtype=dtype, pitch=cin + 8, c0=0)
    yup = Dev(rnd(n, cin, *sp, seed=6), dtype=dtype, pitch=cin + 8, c0=8)
    uxf = XF(cin, seed=7)
    mean, invstd = rnd(cin, seed=8) * 0.2, rnd(cin, seed=9).abs() + 0.5
    md, isd = mean.cuda(), invstd.cuda()
    nfl2 = lib.biu_bwd_data_bnred_floats(dxd.a(), 3, 0)
    part2 = torch.full((nfl2,), float("nan"), device="cuda")
    nb2 = C.c_int(0)
    check(lib.biu_conv_bwd_data_bnred(dyd.a(), ptr(wd), ptr(pk2), 3, 3, 3, 1, dxd.a(), yup.a(), ptr(uxf.d[0]), ptr(uxf.d[1]), ptr(uxf.d[2]), ptr(md), ptr(isd),
                                      ptr(part2), nfl2, C.byref(nb2), ptr(ws), 0, code, stream()), "conv_bwd_data_bnred(roll)")
    gdx = dxd.get()
    torch.testing.assert_close(gdx, xa.grad, rtol=1e-2, atol=1e-2 * float(xa.grad.abs().max()))
    # reference sums on the STORED dx (as the separate reduce pass would see it)
    yu = yup.ref().double()
    shp = (1, -1, 1, 1, 1)
    t_ = yu * uxf.scale.double().view(shp) + uxf.shift.double().view(shp)
    dz = gdx.double() * torch.where(t_ > 0, torch.ones_like(t_), uxf.slope.double().view(shp).expand_as(t_))
    s1 = dz.sum(dim=(0, 2, 3, 4))
    s2 = (dz * (yu - mean.double().view(shp)) * invstd.double().view(shp)).sum(dim=(0, 2, 3, 4))
    got2 = part2[:nb2.value * cin * 2].view(nb2.value, cin, 2).double().sum(0).cpu()
    sc_ = float(dz.abs().sum() / cin)
    torch.testing.assert_close(got2[:, 0], s1, rtol=1e-3, atol=1e-4 * sc_)
    torch.testing.assert_close(got2[:, 1], s2, rtol=1e-3, atol=1e-4 * sc_ * float(invstd.max()) * 4)
    # plain data gradient, then accumulate (the accumulate form falls back to the brick kernels: both must agree)
    dxd2 = Dev(shape=(n, cin, *sp), dtype=dtype)
    check(lib.biu_conv_bwd_data(dyd.a(), ptr(wd), ptr(pk2), 3, 3, 3, 1, dxd2.a(), 0, ptr(ws), 0, code, stream()), "conv_bwd_data(roll)")
    # (32 -> 32 with the sums runs on the brick kernel -- no room beside 216 weight registers: two kernels, two summation orders)
    same = not (cin == 32 and cout == 32)
    torch.testing.assert_close(dxd2.get(), gdx, rtol=0 if same else 1e-2, atol=0 if same else 1e-2 * float(gdx.abs().max()))
    check(lib.biu_conv_bwd_data(dyd.a(), ptr(wd), ptr(pk2), 3, 3, 3, 1, dxd2.a(), 1, ptr(ws), 0, code, stream()), "conv_bwd_data(acc)")
    torch.testing.assert_close(dxd2.get(), 2 * xa.grad, rtol=2e-2, atol=2e-2 * float(xa.grad.abs().max()))


# ---------------------------------------------------------------------------------------------------------------
# nearest-neighbour up-sampling folded into the 3x3x3 convolution behind it (forward): 8 parity classes x 2x2x2 taps on the coarse tensor
# ---------------------------------------------------------------------------------------------------------------
UPCONV_CASES = [
    # (N, Cin, Cout, coarse extent)  -- extents off the 4 x 8 x 16 brick, one / two / three output tiles, 16- and 32-channel chunks
    (1, 32, 32, (4, 8, 16)),
    (2, 64, 64, (3, 5, 9)),
    (1, 48, 96, (5, 9, 17)),
    (1, 16, 24, (2, 3, 20)),
    (1, 128, 128, (4, 8, 8)),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", UPCONV_CASES)
def test_upconv_fwd_matches_upsample_then_conv(case, dtype):
    """biu_upconv_fwd == F.interpolate(scale_factor=2, mode='nearest') + Conv3d(k3, padding=1) on the lazily transformed input
    (multi_output_unet3d/multi_output_unet3d.py:138-139), with the BatchNorm statistics of the stored output from its epilogue."""
    n, cin, cout, sp = case
    code = DT[dtype][1]
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cout, cin, 3, 3, 3, seed=2) * (1.0 / (cin * 27) ** 0.5)
    b = rnd(cout, seed=3)
    xf = XF(cin, seed=4)
    xd = Dev(x, dtype=dtype, pitch=cin + 16, c0=8)
    hi = tuple(2 * v for v in sp)
    yd = Dev(shape=(n, cout, *hi), dtype=dtype, pitch=cout + 8, c0=8)
    assert lib.biu_upconv_ok(xd.a(), yd.a(), code) == 1
    xa = xf.apply(xd.ref())
    if dtype == "bf16":
        xa = xa.bfloat16().float()
    xa.requires_grad_(True)
    yref = F.conv3d(F.interpolate(xa, scale_factor=2, mode="nearest"), w, b, padding=1)
    wd, bd = w.cuda(), b.cuda()
    pk = torch.empty(lib.biu_upconv_packed_bytes(0, cin, cout, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_upconv_pack(0, ptr(wd), cin, cout, code, ptr(pk), stream()), "upconv_pack")
    nfl = lib.biu_upconv_fwd_stats_floats(xd.a(), yd.a())
    part = torch.full((nfl,), float("nan"), device="cuda")
    nblk = C.c_int(0)
    check(lib.biu_upconv_fwd(xd.a(), xf.x(), ptr(pk), ptr(bd), yd.a(), ptr(part), nfl, C.byref(nblk), code, stream()), "upconv_fwd")
    got = yd.get()
    t = dict(rtol=1e-4, atol=1e-4 * float(yref.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=1.5e-2 * float(yref.abs().max()))
    torch.testing.assert_close(got, yref.detach(), **t)
    assert torch.isnan(yd.buf[..., :8].float()).all()                       # channels outside the slice untouched
    sums = part[:nblk.value * cout * 2].view(nblk.value, cout, 2).double().sum(0).cpu()
    gd = got.double()
    torch.testing.assert_close(sums[:, 0], gd.sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-4 * float(gd.abs().sum() / cout))
    torch.testing.assert_close(sums[:, 1], (gd * gd).sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-6)
    # without statistics
    yd2 = Dev(shape=(n, cout, *hi), dtype=dtype)
    check(lib.biu_upconv_fwd(xd.a(), xf.x(), ptr(pk), ptr(bd), yd2.a(), None, 0, None, code, stream()), "upconv_fwd (no statistics)")
    assert torch.equal(yd2.get(), got)
    # data gradient straight onto the coarse tensor (= nearest_up_bwd(conv_bwd_data(dy))), plain and accumulated
    dyd = Dev(rnd(n, cout, *hi, seed=5), dtype=dtype, pitch=cout + 8, c0=0)
    yref.backward(dyd.ref())
    if lib.biu_upconv_packed_bytes(1, cin, cout, code) == 0:               # (bf16, Cout = 24: a reduction chunk would straddle two parity classes)
        assert (cin, cout, dtype) == (16, 24, "bf16")
        return
    pk1 = torch.empty(lib.biu_upconv_packed_bytes(1, cin, cout, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_upconv_pack(1, ptr(wd), cin, cout, code, ptr(pk1), stream()), "upconv_pack (data gradient)")
    dxd = Dev(shape=(n, cin, *sp), dtype=dtype, pitch=cin + 8, c0=8)
    check(lib.biu_upconv_bwd_data(dyd.a(), ptr(pk1), dxd.a(), 0, code, stream()), "upconv_bwd_data")
    t2 = dict(rtol=1e-4, atol=1e-4 * float(xa.grad.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=1.5e-2 * float(xa.grad.abs().max()))
    torch.testing.assert_close(dxd.get(), xa.grad, **t2)
    check(lib.biu_upconv_bwd_data(dyd.a(), ptr(pk1), dxd.a(), 1, code, stream()), "upconv_bwd_data (accumulate)")
    torch.testing.assert_close(dxd.get(), 2 * xa.grad, rtol=2 * t2["rtol"], atol=2 * t2["atol"])
    assert torch.isnan(dxd.buf[..., :8].float()).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(1, 32, 32, (4, 8, 16)), (2, 64, 32, (3, 5, 9)), (1, 96, 64, (5, 9, 17)), (1, 16, 24, (2, 3, 20))])
def test_upconv_bwd_weight_matches_the_unfolded_weight_gradient(case, dtype):
    """biu_upconv_bwd_weight_bn (coarse x, fine da / y) == biu_nearest_up_fwd + biu_conv_bwd_weight_bn on the up-sampled tensor: the same dy
    written back over da, the same dW up to the order of the fp32 sums; y = NULL: the plain weight gradient of a finished dy."""
    n, cin, cout, sp = case
    code = DT[dtype][1]
    hi = tuple(2 * v for v in sp)
    x = Dev(rnd(n, cin, *sp, seed=1), dtype=dtype, pitch=cin + 8, c0=0)
    xf = XF(cin, seed=2)
    y = Dev(rnd(n, cout, *hi, seed=3), dtype=dtype)
    da0 = rnd(n, cout, *hi, seed=4)
    yxf = XF(cout, seed=5)
    cA, cB, cC = (rnd(cout, seed=6) * 0.3 + 1.0), rnd(cout, seed=7) * 0.05, rnd(cout, seed=8) * 0.05
    coef = [t.cuda() for t in (cA, cB, cC)]
    # reference: materialise the up-sampled (transformed) tensor, then the unfolded BatchNorm-fused weight gradient
    up = Dev(shape=(n, cin, *hi), dtype=dtype)
    check(lib.biu_nearest_up_fwd(x.a(), xf.x(), up.a(), code, stream()), "nearest_up_fwd")
    ws0 = torch.empty(max(lib.biu_conv_bwd_weight_workspace(cin, cout, 3, 3, 3, code), 16), dtype=torch.uint8, device="cuda")
    da_ref = Dev(da0, dtype=dtype)
    dw_ref = torch.full((cout, cin, 3, 3, 3), float("nan"), device="cuda")
    check(lib.biu_conv_bwd_weight_bn(up.a(), None, da_ref.a(), y.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]), ptr(coef[1]),
                                     ptr(coef[2]), 3, 3, 3, 1, ptr(dw_ref), ptr(ws0), ws0.numel(), code, stream()), "conv_bwd_weight_bn")
    wsz = lib.biu_upconv_bwd_weight_workspace(cin, cout, code)
    assert wsz > 0
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    scale = float(dw_ref.abs().max())
    t2 = dict(rtol=1e-3, atol=2e-4 * scale) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2 * scale)
    for rep in range(2):
        da = Dev(da0, dtype=dtype)
        dw = torch.full((cout, cin, 3, 3, 3), float("nan"), device="cuda")
        check(lib.biu_upconv_bwd_weight_bn(x.a(), xf.x(), da.a(), y.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]), ptr(coef[1]),
                                           ptr(coef[2]), ptr(dw), ptr(ws), ws.numel(), code, stream()), "upconv_bwd_weight_bn")
        t = dict(rtol=1e-5, atol=1e-5) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(da.get(), da_ref.get(), **t)                 # dy written back over da, every parity class its own voxels
        torch.testing.assert_close(dw.cpu(), dw_ref.cpu(), **t2)
    # plain form on the finished dy
    dw2 = torch.full((cout, cin, 3, 3, 3), float("nan"), device="cuda")
    check(lib.biu_upconv_bwd_weight_bn(x.a(), xf.x(), da_ref.a(), None, None, None, None, None, None, None, ptr(dw2), ptr(ws), ws.numel(), code, stream()),
          "upconv_bwd_weight (plain)")
    torch.testing.assert_close(dw2.cpu(), dw_ref.cpu(), **t2)


# ---------------------------------------------------------------------------------------------------------------
# ConvTranspose(k2, s2) + concat + 3x3x3 conv of a decoder level, the up half folded onto the coarse tensor
# ---------------------------------------------------------------------------------------------------------------
FOLDT_CASES = [
    # (N, Cin_low, Cup, Cskip, Cout, coarse extent)
    (1, 64, 64, 32, 32, (4, 8, 16)),          # decode5 of UNet3D(n_filter = 32)
    (2, 64, 64, 32, 32, (5, 6, 19)),          # the same widths off the window: bf16 takes the rolling-window form (k_fold_roll + accumulating skip half)
    (1, 64, 64, 32, 32, (18, 9, 40)),         # ... with depth segments and two windows along W
    (2, 32, 32, 16, 32, (3, 5, 9)),
    (1, 128, 128, 64, 64, (2, 4, 8)),         # decode3
    (1, 48, 32, 32, 64, (5, 3, 7)),
    (1, 256, 256, 128, 128, (2, 3, 5)),       # decode1: 128-channel dy (four tiles in the one-launch folded weight gradient)
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", FOLDT_CASES)
def test_foldt_fwd_matches_convT_concat_conv(case, dtype):
    """biu_foldt_fwd == ConvTranspose3d(k2, s2) -> torch.cat([up, skip], 1) -> Conv3d(k3, padding=1) (unet3d/unet3d.py:84-90), both inputs
    lazily transformed, ConvT bias included (its border behaviour is the point of the 27-state table), BatchNorm statistics from the epilogue."""
    n, cl, cup, cs, cout, sp = case
    code = DT[dtype][1]
    hi = tuple(2 * v for v in sp)
    xl = Dev(rnd(n, cl, *sp, seed=1), dtype=dtype, pitch=cl + 8, c0=0)
    sk = Dev(rnd(n, cs, *hi, seed=2), dtype=dtype, pitch=cs + 16, c0=8)
    xfl, xfs = XF(cl, seed=3), XF(cs, seed=4)
    wt = rnd(cl, cup, 2, 2, 2, seed=5) * (1.0 / cl ** 0.5)
    bt = rnd(cup, seed=6)
    wc = rnd(cout, cup + cs, 3, 3, 3, seed=7) * (1.0 / ((cup + cs) * 27) ** 0.5)
    bc = rnd(cout, seed=8)
    yd = Dev(shape=(n, cout, *hi), dtype=dtype, pitch=cout + 8, c0=8)
    assert lib.biu_foldt_ok(xl.a(), sk.a(), yd.a(), code) == 1
    xa, sa = xfl.apply(xl.ref()), xfs.apply(sk.ref())
    if dtype == "bf16":
        xa, sa = xa.bfloat16().float(), sa.bfloat16().float()
    xa.requires_grad_(True); sa.requires_grad_(True)
    wt, bt, wc = wt.requires_grad_(True), bt.requires_grad_(True), wc.requires_grad_(True)
    up = F.conv_transpose3d(xa, wt, bt, stride=2)
    yref = F.conv3d(torch.cat([up, sa], 1), wc, bc, padding=1)
    dev = [t.detach().cuda() for t in (wc, bc, wt, bt)]
    pk = torch.empty(lib.biu_foldt_packed_bytes(cl, cs, cout, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_foldt_pack(ptr(dev[0]), ptr(dev[1]), ptr(dev[2]), ptr(dev[3]), cl, cup, cs, cout, code, ptr(pk), stream()), "foldt_pack")
    nfl = lib.biu_foldt_fwd_stats_floats(xl.a(), yd.a())
    part = torch.full((nfl,), float("nan"), device="cuda")
    nblk = C.c_int(0)
    check(lib.biu_foldt_fwd(xl.a(), xfl.x(), sk.a(), xfs.x(), ptr(pk), yd.a(), ptr(part), nfl, C.byref(nblk), code, stream()), "foldt_fwd")
    got = yd.get()
    t = dict(rtol=1e-4, atol=1e-4 * float(yref.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=2e-2 * float(yref.abs().max()))
    torch.testing.assert_close(got, yref.detach(), **t)
    assert torch.isnan(yd.buf[..., :8].float()).all()
    sums = part[:nblk.value * cout * 2].view(nblk.value, cout, 2).double().sum(0).cpu()
    gd = got.double()
    torch.testing.assert_close(sums[:, 0], gd.sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-4 * float(gd.abs().sum() / cout))
    torch.testing.assert_close(sums[:, 1], (gd * gd).sum(dim=(0, 2, 3, 4)), rtol=1e-4, atol=1e-6)
    # ---- backward: a GENERAL dy (per-channel sums far from zero): S_k = dy_sum - border sums; a train-mode BatchNorm's dy sums to zero and
    # passes dy_sum = NULL (the engine; the in-situ tests), any other dy must hand its channel sums in
    dy0 = rnd(n, cout, *hi, seed=9) + 0.3
    dyd = Dev(dy0, dtype=dtype, pitch=cout + 8, c0=0)
    dyr = dyd.ref()
    dy_sum = dyr.double().sum(dim=(0, 2, 3, 4)).float().cuda()          # of the values as stored
    yref.backward(dyr)
    dxl = Dev(shape=(n, cl, *sp), dtype=dtype, pitch=cl + 8, c0=8)
    dsk = Dev(shape=(n, cs, *hi), dtype=dtype)
    wsd = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    check(lib.biu_foldt_bwd_data(dyd.a(), ptr(pk), dxl.a(), 0, dsk.a(), 0, None, None, None, None, None, None, None, 0, None, ptr(wsd), wsd.numel(), code,
                                 stream()), "foldt_bwd_data")
    tg = lambda ref: dict(rtol=1e-4, atol=1e-4 * float(ref.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=2e-2 * float(ref.abs().max()))  # noqa: E731
    torch.testing.assert_close(dxl.get(), xa.grad, **tg(xa.grad))
    torch.testing.assert_close(dsk.get(), sa.grad, **tg(sa.grad))
    wsz = lib.biu_foldt_bwd_weight_workspace(cl, cs, cout, code)
    assert wsz > 0
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    dwc, dwt, dbt = (torch.full(t_.shape, float("nan"), device="cuda") for t_ in (wc, wt, bt))
    # a plain dy without its channel sums is refused (the zero-sum shortcut would silently give a wrong db_T / dW_conv)
    assert lib.biu_foldt_bwd_weight_bn(xl.a(), xfl.x(), sk.a(), xfs.x(), dyd.a(), None, None, None, None, None, None, None, None, ptr(dev[0]), ptr(dev[2]),
                                       ptr(dev[3]), cup, ptr(dwc), ptr(dwt), ptr(dbt), ptr(ws), ws.numel(), code, stream()) != 0
    assert b"dy_sum" in lib.biu_last_error()
    check(lib.biu_foldt_bwd_weight_bn(xl.a(), xfl.x(), sk.a(), xfs.x(), dyd.a(), None, None, None, None, None, None, None, ptr(dy_sum), ptr(dev[0]), ptr(dev[2]),
                                      ptr(dev[3]), cup, ptr(dwc), ptr(dwt), ptr(dbt), ptr(ws), ws.numel(), code, stream()), "foldt_bwd_weight")
    tw = lambda ref: dict(rtol=1e-3, atol=2e-4 * float(ref.abs().max())) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2 * float(ref.abs().max()))  # noqa: E731
    torch.testing.assert_close(dwc.cpu(), wc.grad, **tw(wc.grad))
    torch.testing.assert_close(dwt.cpu(), wt.grad, **tw(wt.grad))
    torch.testing.assert_close(dbt.cpu(), bt.grad, rtol=1e-3 if dtype == "f32" else 2e-2, atol=(2e-4 if dtype == "f32" else 2e-2) * float(bt.grad.abs().max()))
    # the engine's form: BatchNorm + LeakyReLU backward of the block in the loader (da -> dy written back), against bn_bwd_apply + the plain form
    yv = Dev(rnd(n, cout, *hi, seed=10), dtype=dtype)
    yxf = XF(cout, seed=11)
    cA, cB, cC = (rnd(cout, seed=12) * 0.3 + 1.0), rnd(cout, seed=13) * 0.05, rnd(cout, seed=14) * 0.05
    coef = [t_.cuda() for t_ in (cA, cB, cC)]
    da0 = rnd(n, cout, *hi, seed=15)
    da_ref = Dev(da0, dtype=dtype)
    check(lib.biu_bn_bwd_apply(da_ref.a(), yv.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), da_ref.a(), code,
                               stream()), "bn_bwd_apply")
    ref3 = [torch.full(t_.shape, float("nan"), device="cuda") for t_ in (wc, wt, bt)]
    da_sum = da_ref.get().double().sum(dim=(0, 2, 3, 4)).float().cuda()
    check(lib.biu_foldt_bwd_weight_bn(xl.a(), xfl.x(), sk.a(), xfs.x(), da_ref.a(), None, None, None, None, None, None, None, ptr(da_sum), ptr(dev[0]), ptr(dev[2]),
                                      ptr(dev[3]), cup, ptr(ref3[0]), ptr(ref3[1]), ptr(ref3[2]), ptr(ws), ws.numel(), code, stream()), "foldt_bwd_weight (plain, reference)")
    da = Dev(da0, dtype=dtype)
    got3 = [torch.full(t_.shape, float("nan"), device="cuda") for t_ in (wc, wt, bt)]
    check(lib.biu_foldt_bwd_weight_bn(xl.a(), xfl.x(), sk.a(), xfs.x(), da.a(), yv.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]), ptr(coef[1]),
                                      ptr(coef[2]), ptr(da_sum), ptr(dev[0]), ptr(dev[2]), ptr(dev[3]), cup, ptr(got3[0]), ptr(got3[1]), ptr(got3[2]), ptr(ws),
                                      ws.numel(), code, stream()), "foldt_bwd_weight_bn")
    tb = dict(rtol=1e-5, atol=1e-5) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(da.get(), da_ref.get(), **tb)                     # dy written back over da
    for g_, r_ in zip(got3, ref3):
        sc = float(r_.abs().max())
        torch.testing.assert_close(g_.cpu(), r_.cpu(), rtol=1e-3 if dtype == "f32" else 3e-2, atol=(2e-4 if dtype == "f32" else 3e-2) * sc)
    # the same call in two parts (biu_foldt_bwd_weight_bn_phase): the tensor passes on this stream, the chain rule on ANOTHER one behind an event
    da2 = Dev(da0, dtype=dtype)
    two = [torch.full(t_.shape, float("nan"), device="cuda") for t_ in (wc, wt, bt)]
    args = (xl.a(), xfl.x(), sk.a(), xfs.x(), da2.a(), yv.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(da_sum),
            ptr(dev[0]), ptr(dev[2]), ptr(dev[3]), cup, ptr(two[0]), ptr(two[1]), ptr(two[2]), ptr(ws), ws.numel(), code)
    check(lib.biu_foldt_bwd_weight_bn_phase(*args, 1, stream()), "foldt_bwd_weight_bn_phase(1)")          # skip half: da -> dy
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    side = torch.cuda.Stream()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        check(lib.biu_foldt_bwd_weight_bn_phase(*args, 4, C.c_void_p(side.cuda_stream)), "foldt_bwd_weight_bn_phase(4)")      # G
        check(lib.biu_foldt_bwd_weight_bn_phase(*args, 2, C.c_void_p(side.cuda_stream)), "foldt_bwd_weight_bn_phase(2)")      # border sums + chain rule
    torch.cuda.current_stream().wait_stream(side)
    assert lib.biu_foldt_bwd_weight_bn_phase(*args, 8, stream()) != 0                     # phases: a mask of 1 | 4 | 2
    assert torch.equal(da2.get(), da.get())
    for t2, g_ in zip(two, got3):
        # (the skip slice of dW_conv and G go through fp32 atomics: equal up to their summation order)
        torch.testing.assert_close(t2.cpu(), g_.cpu(), rtol=1e-4, atol=1e-5 * float(g_.abs().max()))


CONVT_MFMA_CASES = [
    # (nd, N, Cin, Cout, coarse spatial)
    (3, 1, 64, 64, (4, 8, 16)),
    (3, 2, 32, 32, (3, 5, 9)),
    (3, 1, 128, 128, (2, 4, 4)),
    (3, 1, 16, 48, (2, 2, 20)),
    (2, 2, 64, 32, (16, 16)),
    (2, 1, 256, 128, (8, 8)),
    (2, 1, 32, 16, (10, 36)),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONVT_MFMA_CASES)
def test_convtranspose_mfma(case, dtype):
    nd, n, cin, cout, sp = case
    kd = 2 if nd == 3 else 1
    code = DT[dtype][1]
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cin, cout, *([2] * nd), seed=2) * (1.0 / cin ** 0.5)
    b = rnd(cout, seed=3)
    xf = XF(cin, seed=4)
    xd = Dev(x, dtype=dtype, pitch=cin + 8, c0=8)
    xr = xd.ref().squeeze(2) if nd == 2 else xd.ref()
    xa = xf.apply(xr)
    wq = w
    if dtype == "bf16":
        xa, wq = xa.bfloat16().float(), w.bfloat16().float()
    xa.requires_grad_(True)
    wq = wq.clone().requires_grad_(True)
    f = F.conv_transpose3d if nd == 3 else F.conv_transpose2d
    ref = f(xa, wq, b, stride=2)
    osp = tuple(2 * s for s in sp)
    oshape = (n, cout, 1 if nd == 2 else osp[0], osp[-2], osp[-1])
    yd = Dev(shape=oshape, dtype=dtype, pitch=cout + 32, c0=0)
    wd, bd = w.cuda(), b.cuda()
    nb = lib.biu_convt_packed_bytes(0, cin, cout, kd, code)
    assert nb > 0
    pk = torch.empty(nb, dtype=torch.uint8, device="cuda")
    check(lib.biu_convt_pack(0, ptr(wd), cin, cout, kd, code, ptr(pk), stream()), "convt_pack")
    check(lib.biu_convt_fwd(xd.a(), xf.x(), ptr(wd), ptr(pk), ptr(bd), kd, yd.a(), code, stream()), "convt_fwd(mfma)")
    t = dict(rtol=1e-4, atol=1e-4 * float(ref.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=1e-2 * float(ref.abs().max()))
    torch.testing.assert_close(yd.get(squeeze2d=nd == 2), ref.detach(), **t)
    assert torch.isnan(yd.buf[..., cout:].float()).all()
    gd = Dev(rnd(*ref.shape, seed=5), dtype=dtype)
    gr = gd.ref().squeeze(2) if nd == 2 else gd.ref()
    ref.backward(gr)
    nb1 = lib.biu_convt_packed_bytes(1, cin, cout, kd, code)
    pk1 = torch.empty(nb1, dtype=torch.uint8, device="cuda")
    check(lib.biu_convt_pack(1, ptr(wd), cin, cout, kd, code, ptr(pk1), stream()), "convt_pack(dgrad)")
    dxd = Dev(shape=(n, cin, 1 if nd == 2 else sp[0], sp[-2], sp[-1]), dtype=dtype)
    check(lib.biu_convt_bwd_data(gd.a(), ptr(wd), ptr(pk1), kd, dxd.a(), 0, code, stream()), "convt_bwd_data(mfma)")
    t2 = dict(rtol=1e-4, atol=1e-4 * float(xa.grad.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=1e-2 * float(xa.grad.abs().max()))
    torch.testing.assert_close(dxd.get(squeeze2d=nd == 2), xa.grad, **t2)
    wsz = lib.biu_convt_bwd_weight_workspace(cin, cout, kd, code)
    assert wsz > 0
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    dw, db = torch.full_like(wd, float("nan")), torch.empty_like(bd)
    check(lib.biu_convt_bwd_weight(xd.a(), xf.x(), gd.a(), kd, ptr(dw), ptr(db), ptr(ws), ws.numel(), code, stream()), "convt_bwd_weight(mfma)")
    t3 = dict(rtol=1e-3, atol=2e-4 * float(wq.grad.abs().max())) if dtype == "f32" else dict(rtol=1e-2, atol=1e-2 * float(wq.grad.abs().max()))
    torch.testing.assert_close(dw.cpu(), wq.grad, **t3)
    # d bias = channel sums of dy, accumulated by the weight-gradient kernel while it stages dy (no second pass over the fine tensor)
    dbr = gr.sum(dim=[0] + list(range(2, gr.dim())))
    torch.testing.assert_close(db.cpu(), dbr, rtol=1e-3, atol=1e-3 * float(gr.abs().sum() ** 0.5))
    db.fill_(float("nan"))                           # a second call must not depend on what the buffer held
    check(lib.biu_convt_bwd_weight(xd.a(), xf.x(), gd.a(), kd, ptr(dw), ptr(db), ptr(ws), ws.numel(), code, stream()), "convt_bwd_weight(mfma) again")
    torch.testing.assert_close(db.cpu(), dbr, rtol=1e-3, atol=1e-3 * float(gr.abs().sum() ** 0.5))


# ---------------------------------------------------------------------------------------------------------------
# data gradient with the upstream block's BatchNorm-backward sums reduced in the epilogue
# ---------------------------------------------------------------------------------------------------------------
def _bn_bwd_sums_ref(da, y, scale, shift, slope, mean, invstd):
    """(sum dz, sum dz*yhat) per channel, dz = da * T'(scale*y+shift)  [torch BatchNorm backward, fp64]."""
    shp = (1, -1) + (1,) * (y.dim() - 2)
    t = y.double() * scale.double().view(shp) + shift.double().view(shp)
    dz = da.double() * torch.where(t > 0, torch.ones_like(t), slope.double().view(shp).expand_as(t))
    yhat = (y.double() - mean.double().view(shp)) * invstd.double().view(shp)
    dims = [0] + list(range(2, y.dim()))
    return dz.sum(dims), (dz * yhat).sum(dims)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(3, 1, 32, 32, (8, 16, 32)), (3, 2, 16, 64, (6, 10, 36)), (2, 2, 64, 32, (24, 40)),
                                  (3, 1, 6, 8, (4, 6, 10))])
def test_conv_bwd_data_bnred(case, dtype):
    nd, n, cin, cout, sp = case
    kd = 3 if nd == 3 else 1
    code = DT[dtype][1]
    w = rnd(cout, cin, *([3] * nd), seed=2) * (1.0 / (cin * 3 ** nd) ** 0.5)
    wd = w.cuda()
    nb2 = lib.biu_conv_packed_bytes(1, cin, cout, kd, 3, 3, 1, code)
    pk2 = torch.empty(max(nb2, 16), dtype=torch.uint8, device="cuda")
    if nb2:
        check(lib.biu_conv_pack(1, ptr(wd), cin, cout, kd, 3, 3, code, ptr(pk2), stream()), "conv_pack(dgrad)")
    dyd = Dev(rnd(n, cout, *sp, seed=5), dtype=dtype)
    yup = Dev(rnd(n, cin, *sp, seed=6), dtype=dtype, pitch=cin + 8, c0=0)
    xf = XF(cin, seed=7)
    mean, invstd = rnd(cin, seed=8) * 0.1, rnd(cin, seed=9).abs() + 0.5
    md, isd = mean.cuda(), invstd.cuda()
    dshape = (n, cin, 1 if nd == 2 else sp[0], sp[-2], sp[-1])
    dx_ref, dx = Dev(shape=dshape, dtype=dtype), Dev(shape=dshape, dtype=dtype)
    check(lib.biu_conv_bwd_data(dyd.a(), ptr(wd), ptr(pk2) if nb2 else None, kd, 3, 3, 1, dx_ref.a(), 0, None, 0, code, stream()), "dgrad")
    nfl = lib.biu_bwd_data_bnred_floats(dx.a(), kd, 0)
    part = torch.full((nfl,), float("nan"), device="cuda")
    nblk = C.c_int(0)
    check(lib.biu_conv_bwd_data_bnred(dyd.a(), ptr(wd), ptr(pk2) if nb2 else None, kd, 3, 3, 1, dx.a(), yup.a(), ptr(xf.d[0]),
                                      ptr(xf.d[1]), ptr(xf.d[2]), ptr(md), ptr(isd), ptr(part), nfl, C.byref(nblk), None, 0, code,
                                      stream()), "conv_bwd_data_bnred")
    if nd == 3 and cin == 32 and cout == 32 and dtype == "bf16":
        # the plain call takes the rolling-window kernel (BIU_ROLL=always in this suite), the reducing one has no 32 -> 32 rolling form and
        # stays on the brick kernel: two fp32 summation orders, each rounded once to bf16
        d = (dx.buf.float() - dx_ref.buf.float()).abs()
        assert float(d.max()) <= 2.0 ** -7 * float(dx_ref.buf.float().abs().max()) and float((d > 0).float().mean()) < 0.05
    else:
        assert torch.equal(dx.buf, dx_ref.buf), "the fused epilogue must not change the data gradient"
    sums = part[:nblk.value * cin * 2].view(nblk.value, cin, 2).double().sum(0).cpu()
    s1, s2 = _bn_bwd_sums_ref(dx.ref(), yup.ref(), xf.scale, xf.shift, xf.slope, mean, invstd)
    scale1 = float(dx.ref().abs().sum() / cin) + 1e-12
    torch.testing.assert_close(sums[:, 0], s1, rtol=1e-4, atol=1e-5 * scale1)
    torch.testing.assert_close(sums[:, 1], s2, rtol=1e-4, atol=1e-5 * scale1 * float(invstd.max()) * 4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(3, 1, 64, 64, (4, 8, 16)), (3, 2, 32, 32, (3, 5, 9)), (2, 2, 64, 32, (16, 16)),
                                  (2, 1, 6, 4, (5, 7))])
def test_convt_bwd_data_bnred(case, dtype):
    nd, n, cin, cout, sp = case
    kd = 2 if nd == 3 else 1
    code = DT[dtype][1]
    w = rnd(cin, cout, *([2] * nd), seed=2) * (1.0 / cin ** 0.5)
    wd = w.cuda()
    nb1 = lib.biu_convt_packed_bytes(1, cin, cout, kd, code)
    pk1 = torch.empty(max(nb1, 16), dtype=torch.uint8, device="cuda")
    if nb1:
        check(lib.biu_convt_pack(1, ptr(wd), cin, cout, kd, code, ptr(pk1), stream()), "convt_pack(dgrad)")
    osp = tuple(2 * s for s in sp)
    gd = Dev(rnd(n, cout, *osp, seed=5), dtype=dtype)
    yup = Dev(rnd(n, cin, *sp, seed=6), dtype=dtype, pitch=cin + 8, c0=0)
    xf = XF(cin, seed=7)
    mean, invstd = rnd(cin, seed=8) * 0.1, rnd(cin, seed=9).abs() + 0.5
    md, isd = mean.cuda(), invstd.cuda()
    dshape = (n, cin, 1 if nd == 2 else sp[0], sp[-2], sp[-1])
    dx_ref, dx = Dev(shape=dshape, dtype=dtype), Dev(shape=dshape, dtype=dtype)
    check(lib.biu_convt_bwd_data(gd.a(), ptr(wd), ptr(pk1) if nb1 else None, kd, dx_ref.a(), 0, code, stream()), "convt dgrad")
    nfl = lib.biu_bwd_data_bnred_floats(dx.a(), kd, 1)
    part = torch.full((nfl,), float("nan"), device="cuda")
    nblk = C.c_int(0)
    check(lib.biu_convt_bwd_data_bnred(gd.a(), ptr(wd), ptr(pk1) if nb1 else None, kd, dx.a(), yup.a(), ptr(xf.d[0]), ptr(xf.d[1]),
                                       ptr(xf.d[2]), ptr(md), ptr(isd), ptr(part), nfl, C.byref(nblk), code, stream()),
          "convt_bwd_data_bnred")
    assert torch.equal(dx.buf, dx_ref.buf)
    sums = part[:nblk.value * cin * 2].view(nblk.value, cin, 2).double().sum(0).cpu()
    s1, s2 = _bn_bwd_sums_ref(dx.ref(), yup.ref(), xf.scale, xf.shift, xf.slope, mean, invstd)
    scale1 = float(dx.ref().abs().sum() / cin) + 1e-12
    torch.testing.assert_close(sums[:, 0], s1, rtol=1e-4, atol=1e-5 * scale1)
    torch.testing.assert_close(sums[:, 1], s2, rtol=1e-4, atol=1e-5 * scale1 * float(invstd.max()) * 4)


# ---------------------------------------------------------------------------------------------------------------
# weight gradient with BatchNorm+LeakyReLU backward fused into its operand loader (da -> dy in place)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(3, 2, 96, 32, (8, 16, 32)), (3, 1, 32, 64, (6, 10, 20)), (2, 2, 128, 64, (24, 40)), (3, 2, 16, 32, (9, 10, 20)),
                                  (3, 1, 6, 8, (4, 6, 10)), (3, 2, 64, 48, (9, 20, 24)), (3, 1, 32, 16, (11, 16, 40)), (3, 2, 96, 16, (9, 12, 24)),
                                  (2, 8, 64, 32, (24, 40)), (2, 11, 32, 64, (16, 48))])
def test_conv_bwd_weight_bn(case, dtype):
    """Several 32-wide input-channel tiles (Cin = 96, 128) read the same da that one of them overwrites with dy."""
    nd, n, cin, cout, sp = case
    kd = 3 if nd == 3 else 1
    code = DT[dtype][1]
    x = Dev(rnd(n, cin, *sp, seed=1), dtype=dtype, pitch=cin + 8, c0=0)
    xf = XF(cin, seed=2)
    y = Dev(rnd(n, cout, *sp, seed=3), dtype=dtype)
    da0 = rnd(n, cout, *sp, seed=4)
    yxf = XF(cout, seed=5)
    cA, cB, cC = (rnd(cout, seed=6) * 0.3 + 1.0), rnd(cout, seed=7) * 0.05, rnd(cout, seed=8) * 0.05
    coef = [t.cuda() for t in (cA, cB, cC)]
    wsz = max(lib.biu_conv_bwd_weight_workspace(cin, cout, kd, 3, 3, code), 16)
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    wshape = (cout, cin) + (3,) * nd
    # reference path: separate BatchNorm-backward pass, then the plain weight gradient
    da_ref = Dev(da0, dtype=dtype)
    check(lib.biu_bn_bwd_apply(da_ref.a(), y.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]), ptr(coef[1]),
                               ptr(coef[2]), da_ref.a(), code, stream()), "bn_bwd_apply")
    dw_ref = torch.full(wshape, float("nan"), device="cuda")
    check(lib.biu_conv_bwd_weight(x.a(), xf.x(), da_ref.a(), kd, 3, 3, 1, ptr(dw_ref), None, ptr(ws), ws.numel(), code, stream()),
          "conv_bwd_weight")
    for rep in range(3):                              # repeated: a read/overwrite race would show as run-to-run differences
        da = Dev(da0, dtype=dtype)
        dw = torch.full(wshape, float("nan"), device="cuda")
        check(lib.biu_conv_bwd_weight_bn(x.a(), xf.x(), da.a(), y.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]),
                                         ptr(coef[1]), ptr(coef[2]), kd, 3, 3, 1, ptr(dw), ptr(ws), ws.numel(), code, stream()),
              "conv_bwd_weight_bn")
        # dy written back over da: the fused path parks dz in the storage type between its two sweeps (one more rounding)
        t = dict(rtol=1e-5, atol=1e-5) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(da.get(), da_ref.get(), **t)
        scale = float(dw_ref.abs().max())
        t2 = dict(rtol=1e-3, atol=2e-4 * scale) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2 * scale)
        torch.testing.assert_close(dw.cpu(), dw_ref.cpu(), **t2)


# ---------------------------------------------------------------------------------------------------------------
# trilinear x2 (UNet3D use_interpolation=True) and depth-wise cross-correlation (Siam 'corr')
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 5, 3, 4, 6), (1, 8, 1, 5, 7), (1, 16, 2, 2, 2)])
def test_trilinear_up(shape, dtype):
    n, c, d, h, w = shape
    code = DT[dtype][1]
    xf = XF(c, seed=3)
    xd = Dev(rnd(*shape, seed=1), dtype=dtype, pitch=c + 3, c0=2)
    xa = xf.apply(xd.ref()).requires_grad_(True)
    if d > 1:
        ref = F.interpolate(xa, scale_factor=2, mode="trilinear", align_corners=False)
    else:                                  # a D = 1 volume keeps its depth: bilinear in (h, w)
        ref = F.interpolate(xa.squeeze(2), scale_factor=2, mode="bilinear", align_corners=False).unsqueeze(2)
    od = Dev(shape=tuple(ref.shape), dtype=dtype, pitch=c + 1, c0=1)
    check(lib.biu_trilinear_up_fwd(xd.a(), xf.x(), od.a(), code, stream()), "trilinear_up_fwd")
    assert_close(od.get(), ref.detach(), dtype, "trilinear fwd")
    gd = Dev(rnd(*ref.shape, seed=5), dtype=dtype)
    ref.backward(gd.ref())
    dxd = Dev(shape=shape, dtype=dtype)
    check(lib.biu_trilinear_up_bwd(gd.a(), dxd.a(), 0, code, stream()), "trilinear_up_bwd")
    assert_close(dxd.get(), xa.grad, dtype, "trilinear bwd")
    check(lib.biu_trilinear_up_bwd(gd.a(), dxd.a(), 1, code, stream()), "trilinear_up_bwd(acc)")
    assert_close(dxd.get(), 2 * xa.grad, dtype, "trilinear bwd acc")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 6, 4, 4), (1, 8, 5, 7), (2, 3, 6, 3)])
def test_depthwise_xcorr(shape, dtype):
    n, c, h, w = shape
    code = DT[dtype][1]
    cur = Dev(rnd(*shape, seed=1), dtype=dtype, pitch=c + 2, c0=1)
    prev = Dev(rnd(*shape, seed=2), dtype=dtype)
    a = cur.ref().squeeze(2).requires_grad_(True)
    b = prev.ref().squeeze(2).requires_grad_(True)
    ref = F.conv2d(a.reshape(1, n * c, h, w), b.reshape(n * c, 1, h, w), groups=n * c, padding="same").view(n, c, h, w)
    od = Dev(shape=(n, c, 1, h, w), dtype=dtype)
    check(lib.biu_xcorr_fwd(cur.a(), None, prev.a(), None, od.a(), code, stream()), "xcorr_fwd")
    t = dict(rtol=1e-4, atol=1e-4 * float(ref.abs().max())) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2 * float(ref.abs().max()))
    torch.testing.assert_close(od.get(squeeze2d=True), ref.detach(), **t)
    gd = Dev(rnd(n, c, h, w, seed=5), dtype=dtype)
    ref.backward(gd.ref().squeeze(2))
    da, db = Dev(shape=(n, c, 1, h, w), dtype=dtype), Dev(shape=(n, c, 1, h, w), dtype=dtype)
    check(lib.biu_xcorr_bwd(cur.a(), None, prev.a(), None, gd.a(), da.a(), db.a(), 0, code, stream()), "xcorr_bwd")
    for got, want in ((da, a.grad), (db, b.grad)):
        t = dict(rtol=1e-4, atol=1e-4 * float(want.abs().max())) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2 * float(want.abs().max()))
        torch.testing.assert_close(got.get(squeeze2d=True), want, **t)


@pytest.mark.parametrize("shape", [(4, 1, 8, 16, 16), (2, 3, 33, 17), (1, 40, 40)])
def test_fused_bce_dice_loss(shape):
    """BCEDiceLoss through biu_bce_dice_fwd/bwd against the eager expression of unet/losses.py:78-112 (value and gradient)."""
    from bio_image_unet_amd.losses import BCEDiceLoss, BCELoss2d, SoftDiceLoss
    torch.manual_seed(0)
    lg = (torch.randn(shape) * 3).cuda().requires_grad_(True)
    tg = (torch.rand(shape) > 0.5).float().cuda()
    crit = BCEDiceLoss(0.3, 0.7)
    loss = crit(lg, tg) * 1.7
    loss.backward()
    lr = lg.detach().cpu().double().requires_grad_(True)
    tr = tg.cpu().double()
    ref = (0.3 * BCELoss2d()(lr, tr) + 0.7 * SoftDiceLoss()(lr, tr)) * 1.7
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    torch.testing.assert_close(lg.grad.cpu().double(), lr.grad, rtol=1e-4, atol=1e-6 * float(lr.grad.abs().max()) + 1e-12)


@pytest.mark.parametrize("which", ["bcedice+time", "tversky", "logcosh_tversky+time", "time_batch1"])
def test_fused_seg_losses_against_the_reference_expressions(which):
    """Tversky / logcoshTversky (unet/losses.py:145-239) and the 3-D trainer's SmoothL1 "time" term between neighbouring BATCH
    entries (unet3d/train.py:140-145) through the fused kernels, value and gradient against the eager float64 expressions."""
    from bio_image_unet_amd.losses import BCEDiceLoss, TverskyLoss, logcoshTverskyLoss
    from oracle import unet_oracle as O
    torch.manual_seed(1)
    shape = (1, 1, 4, 8, 8) if which == "time_batch1" else (3, 2, 4, 6, 10)
    lg = (torch.randn(shape) * 2).cuda().requires_grad_(True)
    tg = (torch.rand(shape) > 0.5).float().cuda()
    lr, tr = lg.detach().cpu().double().requires_grad_(True), tg.cpu().double()
    time = torch.nn.functional.smooth_l1_loss(lr[1:], lr[:-1])
    if which == "bcedice+time":
        loss, ref = BCEDiceLoss(0.5, 0.5)(lg, tg, time_weight=0.1), O.bce_dice_loss(lr, tr) + 0.1 * time
    elif which == "tversky":
        loss, ref = TverskyLoss(0.3, 0.7)(lg, tg), O.tversky_loss(lr, tr, 0.3, 0.7)
    elif which == "logcosh_tversky+time":
        loss, ref = logcoshTverskyLoss(0.6, 0.4)(lg, tg, time_weight=0.25), O.logcosh_tversky_loss(lr, tr, 0.6, 0.4) + 0.25 * time
    else:                    # a batch of one: SmoothL1 over empty slices is nan in the reference (and here)
        loss = BCEDiceLoss(0.5, 0.5)(lg, tg, time_weight=0.1)
        assert torch.isnan(loss) and torch.isnan(O.bce_dice_loss(lr, tr) + 0.1 * time)
        return
    (loss * 1.3).backward()
    (ref * 1.3).backward()
    assert abs(float(loss) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    torch.testing.assert_close(lg.grad.cpu().double(), lr.grad, rtol=2e-4, atol=2e-6 * float(lr.grad.abs().max()) + 1e-12)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 32, 4, 8, 12), (1, 16, 1, 10, 6), (2, 8, 2, 4, 4)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_maxpool_bwd_bnred(shape, dtype, accumulate):
    """Max-pool backward with the producer's BatchNorm-backward sums in the same pass == the two separate passes."""
    n, c, d, h, w = shape
    code = DT[dtype][1]
    xf = XF(c, seed=3)
    xd = Dev(rnd(*shape, seed=1), dtype=dtype)
    pd_ = 2 if d > 1 else 1
    gd = Dev(rnd(n, c, d // pd_, h // 2, w // 2, seed=2), dtype=dtype)
    mean, invstd = (rnd(c, seed=8) * 0.1).cuda(), (rnd(c, seed=9).abs() + 0.5).cuda()
    base = rnd(*shape, seed=4)
    dx_a, dx_b = Dev(base, dtype=dtype), Dev(base, dtype=dtype)
    nfl = 1024 * c * 2
    pa, pb = torch.zeros(nfl, device="cuda"), torch.zeros(nfl, device="cuda")
    na, nb = C.c_int(0), C.c_int(0)
    check(lib.biu_maxpool_bwd(xd.a(), xf.x(), gd.a(), dx_a.a(), accumulate, code, stream()), "maxpool_bwd")
    check(lib.biu_bn_bwd_reduce(dx_a.a(), xd.a(), ptr(xf.d[0]), ptr(xf.d[1]), ptr(xf.d[2]), ptr(mean), ptr(invstd), ptr(pa), C.byref(na),
                                code, stream()), "bn_bwd_reduce")
    check(lib.biu_maxpool_bwd_bnred(xd.a(), xf.x(), gd.a(), dx_b.a(), accumulate, ptr(mean), ptr(invstd), ptr(pb), nfl, C.byref(nb),
                                    code, stream()), "maxpool_bwd_bnred")
    assert torch.equal(dx_a.buf, dx_b.buf)
    sa = pa[:na.value * c * 2].view(na.value, c, 2).double().sum(0).cpu()
    sb = pb[:nb.value * c * 2].view(nb.value, c, 2).double().sum(0).cpu()
    torch.testing.assert_close(sb, sa, rtol=1e-4, atol=1e-4 * float(sa.abs().max()) + 1e-9)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 32, 4, 8, 12), (1, 16, 1, 10, 6), (2, 64, 2, 4, 8), (1, 128, 2, 4, 4)])
def test_maxpool_bwd_sends_ties_to_the_first_maximum(shape, dtype):
    """Inputs on a coarse grid of values: most windows hold their maximum several times, in every (d, h, w) position.  The gradient goes
    to the first one in window scan order (torch's rule) -- in the bf16 kernel the two w-neighbours of a window sit on two lanes and
    settle the argmax with one exchange, which is where an ordering mistake would show."""
    n, c, d, h, w = shape
    nd = 2 if d == 1 else 3
    x = (rnd(*shape, seed=21) * 1.5).round() / 2
    xf = XF(c, seed=2)
    xd = Dev(x, dtype=dtype)
    xa = xf.apply(xd.ref()).requires_grad_(True)
    xs = xa.squeeze(2) if nd == 2 else xa
    ref = (F.max_pool2d if nd == 2 else F.max_pool3d)(xs, 2, 2)
    g = rnd(*ref.shape, seed=3)
    gd = Dev(g, dtype=dtype)
    ref.backward(gd.ref().squeeze(2) if nd == 2 else gd.ref())
    dxd = Dev(shape=shape, dtype=dtype, fill=0.0)
    check(lib.biu_maxpool_bwd(xd.a(), xf.x(), gd.a(), dxd.a(), 0, DT[dtype][1], stream()), "maxpool_bwd")
    got = dxd.get()
    assert torch.equal(got != 0, xa.grad != 0), "gradient routed to a different window element"
    assert_close(got, xa.grad, dtype, "maxpool_bwd with ties")


def test_conv_mfma_sample_beyond_2gb():
    """Byte offsets inside a sample are 32-bit modular (buffer descriptors): a 2.6 GB sample must address correctly."""
    dtype, code = "bf16", DT["bf16"][1]
    n, cin, cout, sp = 1, 16, 16, (8, 400, 400)                  # 1.28 M voxels x pitch 1024 x 2 B = 2.6 GB
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cout, cin, 3, 3, 3, seed=2) * 0.05
    xd = Dev(x, dtype=dtype, pitch=1024, c0=1000)
    assert xd.buf.numel() * 2 > 2 ** 31
    wd = w.cuda()
    xr = xd.ref()
    wq = w.bfloat16().float().requires_grad_(True)
    xq = xr.clone().requires_grad_(True)
    ref = conv_ref(xq, wq, None, 1)
    pk = torch.empty(lib.biu_conv_packed_bytes(0, cin, cout, 3, 3, 3, 1, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(0, ptr(wd), cin, cout, 3, 3, 3, code, ptr(pk), stream()), "pack")
    yd = Dev(shape=(n, cout) + sp, dtype=dtype)
    check(lib.biu_conv_fwd(xd.a(), None, ptr(wd), ptr(pk), None, 3, 3, 3, 1, yd.a(), None, 0, code, stream()), "conv_fwd")
    torch.testing.assert_close(yd.get(), ref.detach(), rtol=1e-2, atol=1e-2 * float(ref.abs().max()))
    gd = Dev(rnd(n, cout, *sp, seed=3), dtype=dtype, pitch=1024, c0=8)
    ref.backward(gd.ref())
    pk1 = torch.empty(lib.biu_conv_packed_bytes(1, cin, cout, 3, 3, 3, 1, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(1, ptr(wd), cin, cout, 3, 3, 3, code, ptr(pk1), stream()), "pack1")
    dxd = Dev(shape=(n, cin) + sp, dtype=dtype)
    check(lib.biu_conv_bwd_data(gd.a(), ptr(wd), ptr(pk1), 3, 3, 3, 1, dxd.a(), 0, None, 0, code, stream()), "conv_bwd_data")
    torch.testing.assert_close(dxd.get(), xq.grad, rtol=1e-2, atol=1e-2 * float(xq.grad.abs().max()))
    ws = torch.empty(lib.biu_conv_bwd_weight_workspace(cin, cout, 3, 3, 3, code), dtype=torch.uint8, device="cuda")
    dw = torch.full_like(wd, float("nan"))
    check(lib.biu_conv_bwd_weight(xd.a(), None, gd.a(), 3, 3, 3, 1, ptr(dw), None, ptr(ws), ws.numel(), code, stream()), "conv_bwd_weight")
    torch.testing.assert_close(dw.cpu(), wq.grad, rtol=1e-2, atol=1e-2 * float(wq.grad.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(3, 2, 16, 1, (4, 6, 8)), (2, 2, 32, 2, (12, 10)), (3, 1, 8, 1, (2, 4, 4))])
def test_head_bwd_bnred(case, dtype):
    """Head backward with the producer's BatchNorm-backward sums in the same pass == head_bwd followed by bn_bwd_reduce."""
    nd, n, c, cout, sp = case
    code = DT[dtype][1]
    xf = XF(c, seed=3)
    xd = Dev(rnd(n, c, *sp, seed=1), dtype=dtype)
    w = (rnd(cout, c, seed=2) * 0.3).cuda()
    dl = rnd(n, cout, *sp, seed=4).cuda().contiguous()
    mean, invstd = (rnd(c, seed=8) * 0.1).cuda(), (rnd(c, seed=9).abs() + 0.5).cuda()
    shp = (n, c) + ((1,) + sp if nd == 2 else sp)
    dxa, dxb = Dev(shape=shp, dtype=dtype), Dev(shape=shp, dtype=dtype)
    wsz = lib.biu_head_bwd_workspace(c)
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    dwa, dba, dwb, dbb = (torch.empty_like(w), torch.empty(cout, device="cuda"), torch.empty_like(w), torch.empty(cout, device="cuda"))
    nfl = 1024 * c * 2
    pa, pb = torch.zeros(nfl, device="cuda"), torch.zeros(nfl, device="cuda")
    na, nb = C.c_int(0), C.c_int(0)
    check(lib.biu_head_bwd(xd.a(), xf.x(), ptr(w), cout, ptr(dl), dxa.a(), ptr(dwa), ptr(dba), ptr(ws), wsz, code, stream()), "head_bwd")
    check(lib.biu_bn_bwd_reduce(dxa.a(), xd.a(), ptr(xf.d[0]), ptr(xf.d[1]), ptr(xf.d[2]), ptr(mean), ptr(invstd), ptr(pa), C.byref(na),
                                code, stream()), "bn_bwd_reduce")
    check(lib.biu_head_bwd_bnred(xd.a(), xf.x(), ptr(w), cout, ptr(dl), dxb.a(), ptr(dwb), ptr(dbb), ptr(ws), wsz, ptr(mean), ptr(invstd),
                                 ptr(pb), nfl, C.byref(nb), code, stream()), "head_bwd_bnred")
    assert torch.equal(dxa.buf, dxb.buf)
    torch.testing.assert_close(dwb, dwa, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dbb, dba, rtol=1e-5, atol=1e-6)
    sa = pa[:na.value * c * 2].view(na.value, c, 2).double().sum(0).cpu()
    sb = pb[:nb.value * c * 2].view(nb.value, c, 2).double().sum(0).cpu()
    torch.testing.assert_close(sb, sa, rtol=1e-4, atol=1e-4 * float(sa.abs().max()) + 1e-9)


# ---------------------------------------------------------------------------------------------------------------
# conv block on a channel concatenation held in two dense buffers (no concat buffer)
# ---------------------------------------------------------------------------------------------------------------
CAT_CASES = [
    # (nd, N, c0, c1, Cout, spatial)
    (3, 1, 64, 32, 32, (8, 16, 32)),      # 3 input tiles -> block tile 32
    (3, 2, 64, 64, 64, (6, 10, 20)),      # 4 tiles -> block tile 64
    (2, 2, 128, 64, 64, (24, 40)),
    (3, 1, 64, 32, 48, (5, 7, 9)),
    (3, 2, 64, 32, 32, (10, 16, 40)),     # columns long enough for the rolling-window weight gradient
    (2, 8, 64, 32, 32, (24, 48)),         # ... and a batch of 8 images through its 2-D form
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CAT_CASES)
def test_conv_cat_forms_match_concat_buffer(case, dtype):
    nd, n, c0, c1, cout, sp = case
    kd = 3 if nd == 3 else 1
    code = DT[dtype][1]
    cin = c0 + c1
    x0, x1 = rnd(n, c0, *sp, seed=1), rnd(n, c1, *sp, seed=2)
    w = rnd(cout, cin, *([3] * nd), seed=3) * (1.0 / (cin * 3 ** nd) ** 0.5)
    b = rnd(cout, seed=4)
    xf0, xf1 = XF(c0, seed=5), XF(c1, seed=6, identity=True)                 # second source: no transform (ConvT output)
    d0, d1 = Dev(x0, dtype=dtype), Dev(x1, dtype=dtype)
    dc = Dev(torch.cat([x0, x1], 1), dtype=dtype)                            # the concat-buffer reference layout
    xfc_vec = [torch.cat([a, bb]).cuda() for a, bb in zip((xf0.scale, xf0.shift, xf0.slope), (xf1.scale, xf1.shift, xf1.slope))]
    from bio_image_unet_amd._lib import biu_xform
    xfc = biu_xform(*[t.data_ptr() for t in xfc_vec])
    yshape = (n, cout, 1 if nd == 2 else sp[0], sp[-2], sp[-1])
    wd, bd = w.cuda(), b.cuda()
    pk0 = torch.empty(lib.biu_conv_packed_bytes(0, cin, cout, kd, 3, 3, 1, code), dtype=torch.uint8, device="cuda")
    pk1 = torch.empty(lib.biu_conv_packed_bytes(1, cin, cout, kd, 3, 3, 1, code), dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_pack(0, ptr(wd), cin, cout, kd, 3, 3, code, ptr(pk0), stream()), "pack0")
    check(lib.biu_conv_pack(1, ptr(wd), cin, cout, kd, 3, 3, code, ptr(pk1), stream()), "pack1")
    ya = Dev(shape=yshape, dtype=dtype)
    assert lib.biu_conv_cat_ok(d0.a(), d1.a(), ya.a(), kd, 3, 3, 1, code) == 1
    # forward (+ statistics)
    yb = Dev(shape=yshape, dtype=dtype)
    nfl = lib.biu_conv_fwd_stats_floats(ya.a(), kd)
    pa, pb = torch.zeros(nfl, device="cuda"), torch.zeros(nfl, device="cuda")
    na, nb = C.c_int(0), C.c_int(0)
    check(lib.biu_conv_fwd_stats(dc.a(), C.byref(xfc), ptr(wd), ptr(pk0), ptr(bd), kd, 3, 3, 1, ya.a(), ptr(pa), nfl, C.byref(na), None, 0, code, stream()), "fwd")
    check(lib.biu_conv_fwd_cat(d0.a(), xf0.x(), d1.a(), None, ptr(wd), ptr(pk0), ptr(bd), kd, 3, 3, 1, yb.a(), ptr(pb), nfl, C.byref(nb), None, 0, code,
                               stream()), "fwd_cat")
    assert torch.equal(ya.buf, yb.buf)
    # ... and against torch: conv of the concatenation of the two transformed sources (operands as the MFMA kernel packs them)
    q = (lambda t: t.bfloat16().float()) if dtype == "bf16" else (lambda t: t)
    r0, r1 = (d.ref().squeeze(2) if nd == 2 else d.ref() for d in (d0, d1))
    xa = q(torch.cat([xf0.apply(r0), r1], 1)).requires_grad_(True)
    wq = q(w).requires_grad_(True)
    yref = conv_ref(xa, wq, b, 1)
    tl = (lambda ref, k=1.0: dict(rtol=1e-4 * k, atol=1e-4 * k * float(ref.abs().max()))) if dtype == "f32" else \
         (lambda ref, k=1.0: dict(rtol=1e-2 * k, atol=1e-2 * k * float(ref.abs().max())))
    torch.testing.assert_close(yb.get(squeeze2d=(nd == 2)), yref.detach(), **tl(yref))
    sa = pa[:na.value * cout * 2].view(na.value, cout, 2).double().sum(0)
    sb = pb[:nb.value * cout * 2].view(nb.value, cout, 2).double().sum(0)
    torch.testing.assert_close(sb, sa, rtol=1e-5, atol=1e-5 * float(sa.abs().max()))
    # data gradient into two tensors (second one accumulating)
    gd = Dev(rnd(*yshape, seed=7).squeeze(2) if nd == 2 else rnd(*yshape, seed=7), dtype=dtype)
    dxc = Dev(shape=dc.buf.permute(0, 4, 1, 2, 3).shape, dtype=dtype)
    check(lib.biu_conv_bwd_data(gd.a(), ptr(wd), ptr(pk1), kd, 3, 3, 1, dxc.a(), 0, None, 0, code, stream()), "dgrad")
    base1 = rnd(n, c1, *sp, seed=8)
    g0, g1 = Dev(shape=d0.buf.permute(0, 4, 1, 2, 3).shape, dtype=dtype), Dev(base1, dtype=dtype)
    check(lib.biu_conv_bwd_data_cat(gd.a(), ptr(wd), ptr(pk1), kd, 3, 3, 1, g0.a(), 0, g1.a(), 1, None, 0, code, stream()), "dgrad_cat")
    assert torch.equal(g0.buf, dxc.buf[..., :c0])
    gr = gd.ref().squeeze(2) if nd == 2 else gd.ref()
    (gx_ref,) = torch.autograd.grad(yref, xa, gr, retain_graph=True)                 # torch: data gradient of the concatenation
    torch.testing.assert_close(g0.get(squeeze2d=(nd == 2)), gx_ref[:, :c0], **tl(gx_ref))
    torch.testing.assert_close(g1.get(squeeze2d=(nd == 2)), gx_ref[:, c0:] + Dev(base1, dtype=dtype).get(squeeze2d=(nd == 2)), **tl(gx_ref, 2.0))
    want1 = (dxc.buf[..., c0:].float() + Dev(base1, dtype=dtype).buf.float())
    torch.testing.assert_close(g1.buf.float(), want1, rtol=2e-2 if dtype == "bf16" else 1e-5, atol=(2e-2 if dtype == "bf16" else 1e-5) * float(want1.abs().max()))
    # weight gradient, plain and with the fused BatchNorm backward
    wsz = lib.biu_conv_bwd_weight_workspace(cin, cout, kd, 3, 3, code)
    ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    dwa, dwb = torch.empty_like(wd), torch.empty_like(wd)
    check(lib.biu_conv_bwd_weight(dc.a(), C.byref(xfc), gd.a(), kd, 3, 3, 1, ptr(dwa), None, ptr(ws), wsz, code, stream()), "wgrad")
    check(lib.biu_conv_bwd_weight_cat(d0.a(), xf0.x(), d1.a(), None, gd.a(), None, None, None, None, None, None, None, kd, 3, 3, 1, ptr(dwb),
                                      ptr(ws), wsz, code, stream()), "wgrad_cat")
    torch.testing.assert_close(dwb, dwa, rtol=1e-4, atol=1e-5 * float(dwa.abs().max()))
    (gw_ref,) = torch.autograd.grad(yref, wq, gr)                                    # torch: weight gradient over both sources
    torch.testing.assert_close(dwb.cpu(), gw_ref, **tl(gw_ref, 2.0))
    yxf = XF(cout, seed=9)
    coef = [t.cuda() for t in ((rnd(cout, seed=10) * 0.3 + 1.0), rnd(cout, seed=11) * 0.05, rnd(cout, seed=12) * 0.05)]
    da0 = rnd(*yshape, seed=13).squeeze(2) if nd == 2 else rnd(*yshape, seed=13)
    daa, dab = Dev(da0, dtype=dtype), Dev(da0, dtype=dtype)
    check(lib.biu_conv_bwd_weight_bn(dc.a(), C.byref(xfc), daa.a(), ya.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]), ptr(coef[1]),
                                     ptr(coef[2]), kd, 3, 3, 1, ptr(dwa), ptr(ws), wsz, code, stream()), "wgrad_bn")
    check(lib.biu_conv_bwd_weight_cat(d0.a(), xf0.x(), d1.a(), None, dab.a(), ya.a(), ptr(yxf.d[0]), ptr(yxf.d[1]), ptr(yxf.d[2]), ptr(coef[0]),
                                      ptr(coef[1]), ptr(coef[2]), kd, 3, 3, 1, ptr(dwb), ptr(ws), wsz, code, stream()), "wgrad_bn_cat")
    assert torch.equal(daa.buf, dab.buf)
    torch.testing.assert_close(dwb, dwa, rtol=1e-4, atol=1e-5 * float(dwa.abs().max()))
    # torch: dy = cA * da * T'(scale*y+shift) + cB * y + cC on the stored operands, then the weight gradient w.r.t. it
    yr = ya.ref()
    dar = Dev(da0, dtype=dtype).ref()
    shp = (1, -1, 1, 1, 1)
    tt = yxf.scale.view(shp) * yr + yxf.shift.view(shp)
    dy_ref = coef[0].cpu().view(shp) * dar * torch.where(tt > 0, torch.ones_like(tt), yxf.slope.view(shp).expand_as(tt)) + \
        coef[1].cpu().view(shp) * yr + coef[2].cpu().view(shp)
    tb = dict(rtol=1e-5, atol=1e-5) if dtype == "f32" else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dab.get(), dy_ref, **tb)
    dyr2 = dab.get(squeeze2d=(nd == 2))                                              # the dy the kernel stored is what it multiplied
    (gw2,) = torch.autograd.grad(conv_ref(xa.detach(), wq, None, 1), wq, dyr2)
    torch.testing.assert_close(dwb.cpu(), gw2, **tl(gw2, 2.0))


@pytest.mark.parametrize("dtype", DTYPES)
def test_pack_batch_equals_the_single_packers(dtype):
    """biu_pack_batch (what a training step uses to refresh every packed weight in one launch) must write exactly the bytes of
    biu_conv_pack / biu_convt_pack -- including the 16-channel kernel's fragment image that follows the regular one."""
    from bio_image_unet_amd._lib import biu_pack_job
    code = DT[dtype][1]
    jobs = [(0, 0, 32, 16, 3), (0, 1, 16, 32, 3), (0, 0, 64, 16, 1), (0, 1, 16, 64, 1), (0, 0, 48, 32, 3), (0, 1, 32, 96, 1),
            (1, 0, 32, 16, 2), (1, 1, 64, 32, 1)]                       # (transposed, kind, cin, cout, kd)
    arr = (biu_pack_job * len(jobs))()
    keep, single = [], []
    for j, (tr, kind, cin, cout, kd) in zip(arr, jobs):
        if tr:
            w = rnd(cin, cout, *([2] * (3 if kd == 2 else 2)), seed=cin + cout).cuda()
            nb = lib.biu_convt_packed_bytes(kind, cin, cout, kd, code)
        else:
            w = rnd(cout, cin, *([3] * (3 if kd == 3 else 2)), seed=cin + cout).cuda()
            nb = lib.biu_conv_packed_bytes(kind, cin, cout, kd, 3, 3, 1, code)
        assert nb > 0
        a, b = torch.zeros(nb, dtype=torch.uint8, device="cuda"), torch.full((nb,), 0xAB, dtype=torch.uint8, device="cuda")
        if tr:
            check(lib.biu_convt_pack(kind, ptr(w), cin, cout, kd, code, ptr(a), stream()), "convt_pack")
        else:
            check(lib.biu_conv_pack(kind, ptr(w), cin, cout, kd, 3, 3, code, ptr(a), stream()), "conv_pack")
        j.w, j.packed = w.data_ptr(), b.data_ptr()
        j.transposed, j.kind, j.cin, j.cout, j.kd, j.kh, j.kw, j.reserved = tr, kind, cin, cout, kd, (2 if tr else 3), (2 if tr else 3), 0
        keep.append(w); single.append((a, b))
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    check(lib.biu_pack_batch(ptr(tab), len(jobs), code, stream()), "pack_batch")
    torch.cuda.synchronize()
    for (a, b), job in zip(single, jobs):
        assert torch.equal(a, b), f"batched packing differs from the single packer for job {job}"


@pytest.mark.parametrize("path", ["bn_stats", "conv_fwd_stats"])
def test_batchnorm_statistics_with_a_large_offset(path):
    """|mean| >> std: the variance comes from (sum y, sum y^2) partials (fp32 per block, merged in fp64), i.e. E[y^2] - mean^2 with
    cancellation.  At mean = 60, std = 1 (offset^2 / var = 3600) the batch variance must still be within 1 % and the mean within 1e-5
    relative -- through the separate statistics pass and through the conv epilogue's fused statistics (where a large conv bias is
    what produces such an offset)."""
    n, c, d, h, w = 2, 32, 8, 16, 32
    nvox = n * d * h * w
    rm_d, rv_d = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    g_d, b_d = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    scale, shift, mean, invstd = (torch.empty(c, device="cuda") for _ in range(4))
    nblk = C.c_int(0)
    if path == "bn_stats":
        y = rnd(n, c, d, h, w, seed=1) + 60.0
        yd = Dev(y, dtype="f32")
        partial = torch.empty(1024 * c * 2, device="cuda")
        check(lib.biu_bn_stats(yd.a(), ptr(partial), C.byref(nblk), 0, stream()), "bn_stats")
        yref = yd.ref()
    else:
        cin = 16
        x = rnd(n, cin, d, h, w, seed=2)
        wt = rnd(c, cin, 3, 3, 3, seed=3) * (1.0 / (cin * 27) ** 0.5)
        bias = torch.full((c,), 60.0)
        xd, yd = Dev(x, dtype="f32"), Dev(shape=(n, c, d, h, w), dtype="f32")
        wd, bd = wt.cuda(), bias.cuda()
        nbytes = lib.biu_conv_packed_bytes(0, cin, c, 3, 3, 3, 1, 0)
        pk = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        check(lib.biu_conv_pack(0, ptr(wd), cin, c, 3, 3, 3, 0, ptr(pk), stream()), "conv_pack")
        nfl = lib.biu_conv_fwd_stats_floats(yd.a(), 3)
        partial = torch.empty(nfl, device="cuda")
        check(lib.biu_conv_fwd_stats(xd.a(), None, ptr(wd), ptr(pk), ptr(bd), 3, 3, 3, 1, yd.a(), ptr(partial), nfl, C.byref(nblk), None, 0, 0, stream()),
              "conv_fwd_stats")
        yref = yd.get()
    check(lib.biu_bn_finalize(ptr(partial), nblk.value, c, float(nvox), ptr(g_d), ptr(b_d), ptr(rm_d), ptr(rv_d), 0.1, 1e-5,
                              ptr(scale), ptr(shift), ptr(mean), ptr(invstd), stream()), "bn_finalize")
    y64 = yref.double()
    m_ref = y64.mean(dim=(0, 2, 3, 4))
    v_ref = y64.var(dim=(0, 2, 3, 4), unbiased=False)
    torch.testing.assert_close(mean.cpu().double(), m_ref, rtol=1e-5, atol=0)
    var = 1.0 / invstd.cpu().double() ** 2 - 1e-5
    torch.testing.assert_close(var, v_ref, rtol=1e-2, atol=0)


# ------------------------------------------------------------------------------------------------------------------
# input-channel split of small fp32 launches with CALLER-OWNED scratch (include/biu.h: biu_conv_split_workspace)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nd,n,cin,cout,sp", [(2, 1, 256, 256, (16, 16)), (2, 2, 512, 256, (16, 32)), (3, 1, 128, 128, (8, 8, 16))])
def test_conv_split_workspace_is_the_callers(nd, n, cin, cout, sp):
    """The library keeps no scratch: the split runs only inside the workspace the caller hands in, gives the unsplit launch's result
    (and torch's), never writes past the queried size, and without a workspace the same call runs unsplit."""
    kd = 3 if nd == 3 else 1
    x = rnd(n, cin, *sp, seed=1)
    w = rnd(cout, cin, *([3] * nd), seed=2) * 0.05
    b = rnd(cout, seed=3)
    xd = Dev(x, dtype="f32")
    xa = xd.ref().squeeze(2) if nd == 2 else xd.ref()
    yref = conv_ref(xa, w, b, 1)
    wd, bd = w.cuda(), b.cuda()
    pk = [torch.empty(max(lib.biu_conv_packed_bytes(k, cin, cout, kd, 3, 3, 1, DT["f32"][1]), 16), dtype=torch.uint8, device="cuda") for k in (0, 1)]
    for k in (0, 1):
        check(lib.biu_conv_pack(k, ptr(wd), cin, cout, kd, 3, 3, DT["f32"][1], ptr(pk[k]), stream()), "conv_pack")
    shape_y = (n, cout, 1 if nd == 2 else sp[0], sp[-2], sp[-1])
    y0, y1 = Dev(shape=shape_y, dtype="f32"), Dev(shape=shape_y, dtype="f32")
    need = lib.biu_conv_split_workspace(cin, y0.a(), None, kd, 3, 3, 1, DT["f32"][1])
    assert need > 0, "this shape is meant to split (grid under half of the CUs)"
    assert lib.biu_conv_split_workspace(cin, y0.a(), None, kd, 3, 3, 1, DT["bf16"][1]) == 0          # fp32 launches only
    guard = 4096
    ws = torch.full((need + guard,), 0x5A, dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_fwd(xd.a(), None, ptr(wd), ptr(pk[0]), ptr(bd), kd, 3, 3, 1, y0.a(), ptr(ws), need, DT["f32"][1], stream()), "conv_fwd split")
    check(lib.biu_conv_fwd(xd.a(), None, ptr(wd), ptr(pk[0]), ptr(bd), kd, 3, 3, 1, y1.a(), None, 0, DT["f32"][1], stream()), "conv_fwd unsplit")
    assert bool((ws[need:] == 0x5A).all()), "the split wrote past the size its own query asked for"
    assert not bool((ws[:need] == 0x5A).all()), "the scratch was not used: the launch did not split"
    assert_close(y0.get(squeeze2d=(nd == 2)), yref, "f32", "conv_fwd (split)")
    torch.testing.assert_close(y0.get(), y1.get(), rtol=1e-5, atol=1e-5 * float(yref.abs().max()))
    # a workspace that is too small is not an error: the launch runs unsplit
    check(lib.biu_conv_fwd(xd.a(), None, ptr(wd), ptr(pk[0]), ptr(bd), kd, 3, 3, 1, y1.a(), ptr(ws), need - 1, DT["f32"][1], stream()), "conv_fwd small ws")
    torch.testing.assert_close(y0.get(), y1.get(), rtol=1e-5, atol=1e-5 * float(yref.abs().max()))
    # data gradient (accumulate = 1 on a pre-filled dx) and statistics through the split
    dy = rnd(*yref.shape, seed=5)
    dyd = Dev(dy, dtype="f32")
    xg = xa.clone().requires_grad_(True)
    conv_ref(xg, w, b, 1).backward(dy)
    dxd = Dev(torch.ones_like(x), dtype="f32")
    needb = lib.biu_conv_split_workspace(cout, dxd.a(), None, kd, 3, 3, 1, DT["f32"][1])
    assert needb > 0
    wsb = torch.empty(needb, dtype=torch.uint8, device="cuda")
    check(lib.biu_conv_bwd_data(dyd.a(), ptr(wd), ptr(pk[1]), kd, 3, 3, 1, dxd.a(), 1, ptr(wsb), needb, DT["f32"][1], stream()), "conv_bwd_data split")
    assert_close(dxd.get(squeeze2d=(nd == 2)), xg.grad + 1, "f32", "conv_bwd_data (split, accumulate)")
    part = torch.empty(lib.biu_conv_fwd_stats_floats(y0.a(), kd), device="cuda")
    nblk = C.c_int(0)
    check(lib.biu_conv_fwd_stats(xd.a(), None, ptr(wd), ptr(pk[0]), ptr(bd), kd, 3, 3, 1, y1.a(), ptr(part), part.numel(), C.byref(nblk),
                                 ptr(ws), need, DT["f32"][1], stream()), "conv_fwd_stats split")
    st_ = part[:nblk.value * cout * 2].view(nblk.value, cout, 2).double().sum(0).cpu()
    yy = yref.transpose(0, 1).flatten(1).double()
    torch.testing.assert_close(st_[:, 0], yy.sum(1), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(st_[:, 1], (yy * yy).sum(1), rtol=1e-4, atol=1e-3)
