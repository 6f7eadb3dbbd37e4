"""Module-API patterns the reference's plain nn.Module allows and a cached static engine must not silently break:
several forwards before a backward, a logging forward between forward and backward, raw ``.data`` writes next to the
packed-weight cache, optimizer state round trips, dtype casts and copies of a model that already ran."""
import copy
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import bio_image_unet_amd as B  # noqa: E402
from bio_image_unet_amd.optim import Adam  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402


def _model(nf=16, seed=3):
    sd = O.init_unet2d(1, 1, nf, seed=seed)
    m = B.Unet(1, 1, nf).cuda()
    m.load_state_dict(sd)
    m.train()
    return m, sd


def _grads(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters()}


def test_two_forwards_then_one_backward():
    """o1 = m(a); o2 = m(b); (f(o1) + f(o2)).backward() -- the second forward must not overwrite what the first one's
    backward needs (each pending forward holds its own engine)."""
    torch.manual_seed(0)
    a, b = torch.rand(2, 1, 64, 64).cuda(), torch.rand(2, 1, 64, 64).cuda()
    y = (torch.rand(2, 1, 64, 64) > 0.5).float().cuda()
    m, sd = _model()
    ref = {}
    for x in (a, b):                                     # separate forward/backward pairs from the same initial buffers
        m.load_state_dict(sd)
        m.zero_grad()
        O.bce_dice_loss(m(x)[1], y).backward()
        for k, g in _grads(m).items():
            ref[k] = ref.get(k, 0) + g
    m.load_state_dict(sd)
    m.zero_grad()
    _, l1 = m(a)
    _, l2 = m(b)
    (O.bce_dice_loss(l1, y) + O.bce_dice_loss(l2, y)).backward()
    for k, g in _grads(m).items():
        torch.testing.assert_close(g, ref[k], rtol=1e-3, atol=1e-5 * float(ref[k].abs().max()) + 1e-9, msg=lambda s: f"{k}: {s}")


def test_logging_forward_between_forward_and_backward():
    torch.manual_seed(1)
    a, b = torch.rand(2, 1, 64, 64).cuda(), torch.rand(2, 1, 64, 64).cuda()
    y = (torch.rand(2, 1, 64, 64) > 0.5).float().cuda()
    m, sd = _model()
    O.bce_dice_loss(m(a)[1], y).backward()
    ref = _grads(m)
    m.load_state_dict(sd)
    m.zero_grad()
    _, l1 = m(a)
    with torch.no_grad():
        m(b)                                             # same shape: would have reused the pending engine's buffers
    O.bce_dice_loss(l1, y).backward()
    for k, g in _grads(m).items():
        torch.testing.assert_close(g, ref[k], rtol=1e-3, atol=1e-5 * float(ref[k].abs().max()) + 1e-10, msg=lambda s: f"{k}: {s}")   # (wgrad sums with fp32 atomics: order varies)


def test_pending_forwards_are_bounded_and_released():
    m, _ = _model(nf=4)
    x = torch.rand(1, 1, 32, 32).cuda()
    outs = [m(x) for _ in range(m._max_live)]
    with pytest.raises(RuntimeError, match="waiting for their backward"):
        m(x)
    del outs                                             # dropping the outputs frees the autograd nodes and their engines
    m(x)


def test_data_write_needs_invalidate_and_inplace_op_does_not():
    """The packed MFMA weights are cached on (data_ptr, _version).  In-place ops under no_grad bump the version; a raw
    ``.data`` write does not and needs ``invalidate_packed()`` (what ddp.GradAverager does after its broadcast)."""
    torch.manual_seed(2)
    x = torch.rand(2, 1, 64, 64)
    m, sd = _model()
    m.eval()
    with torch.no_grad():
        m(x.cuda())                                      # packs
        for p in m.parameters():
            p.mul_(0.5)                                  # in place: version bump -> re-packed on the next forward
        _, got = m(x.cuda())
        sd2 = {k: (v * 0.5 if (k.endswith(".weight") or k.endswith(".bias")) else v) for k, v in sd.items()}
        _, want = O.unet2d_forward(sd2, x, training=False)
        assert float((got.cpu() - want).abs().max()) < 1e-3 * float(want.abs().max())
        for p in m.parameters():
            p.data.copy_(p.data * 2.0)                   # raw .data write: invisible to the cache
        m.invalidate_packed()
        _, got = m(x.cuda())
        _, want = O.unet2d_forward(sd, x, training=False)
        assert float((got.cpu() - want).abs().max()) < 1e-3 * float(want.abs().max())


@pytest.mark.parametrize("kind", ["foldt", "upconv"])
def test_data_write_and_invalidate_reach_the_folded_decoder_levels(kind):
    """The composed weights of the folded decoder levels (biu_foldt_pack / biu_upconv_pack images) are cached outside the engine's
    slot list: ``invalidate_packed()`` after a raw ``.data`` write must drop them too (ddp.GradAverager calls it after its broadcast).
    Compared with a FRESH model holding the written parameters -- forward, and the gradients of the folded level's parameters."""
    torch.manual_seed(4)
    heads = {"seg": {"channels": 1, "activation": None}}
    if kind == "foldt":
        mk = lambda: B.UNet3D(1, 1, 32).cuda()
        sd = O.init_unet3d(1, 1, 32, seed=21)
    else:
        mk = lambda: B.MultiOutputUnet3D(1, heads, 32, True).cuda()
        sd = O.init_mo3d(1, heads, 32, True, seed=22)
    x = torch.rand(2, 1, 16, 32, 32).cuda()

    def logits_of(o):
        return o["seg"] if isinstance(o, dict) else o[1]

    m = mk()
    m.load_state_dict(sd)
    m.train()
    logits_of(m(x)).square().mean().backward()          # packs every image, folded ones included, forward and backward
    m.zero_grad()
    g = torch.Generator().manual_seed(9)
    new = {k: (v + 0.05 * torch.randn(v.shape, generator=g) if v.is_floating_point() and v.dim() > 1 else v) for k, v in sd.items()}
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.data.copy_(new[k].to(p.device))            # raw .data write: no version bump
    m.invalidate_packed()
    out = logits_of(m(x))
    out.square().mean().backward()
    fresh = mk()
    fresh.load_state_dict(new)
    fresh.train()
    ref = logits_of(fresh(x))
    ref.square().mean().backward()
    assert float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), "folded levels kept stale composed weights"
    gm, gf = dict(m.named_parameters()), dict(fresh.named_parameters())
    for k in gm:
        if k.startswith(("decode5.0", "decode3.0", "up3", "up2", "up3_conv", "up2_conv")) and gm[k].grad is not None:
            torch.testing.assert_close(gm[k].grad, gf[k].grad, rtol=1e-3, atol=1e-5 * float(gf[k].grad.abs().max()) + 1e-10, msg=lambda s: f"{k}: {s}")


def test_adam_state_dict_round_trip_and_torch_checkpoint():
    torch.manual_seed(3)
    x = torch.rand(2, 1, 32, 32).cuda()
    y = (torch.rand(2, 1, 32, 32) > 0.5).float().cuda()

    def run(n_before, reload_, opt_cls_first=Adam):
        m, sd = _model(nf=4)
        opt = opt_cls_first(m.parameters(), lr=1e-3)
        for _ in range(n_before):
            opt.zero_grad()
            O.bce_dice_loss(m(x)[1], y).backward()
            opt.step()
        if reload_:
            state = copy.deepcopy(opt.state_dict())
            opt = Adam(m.parameters(), lr=1e-3)
            opt.load_state_dict(state)                   # new moment tensors; torch.optim.Adam stores `step` as a tensor
        for _ in range(2):
            opt.zero_grad()
            O.bce_dice_loss(m(x)[1], y).backward()
            opt.step()
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    straight = run(1, False)
    for first in (Adam, torch.optim.Adam):
        again = run(1, True, first)
        for k, v in straight.items():
            if v.is_floating_point() and not (k.endswith(".0.bias") and not k.startswith("final")):
                # (fp32 atomics order the weight-gradient sums differently from run to run; Adam's m / sqrt(v) turns that into up to a few
                # percent of one lr = 1e-3 step on an element whose gradient is near zero)
                torch.testing.assert_close(again[k], v, rtol=1e-4, atol=1e-4, msg=lambda s: f"{first.__name__} {k}: {s}")


def test_dtype_cast_is_refused_and_copies_drop_engines():
    m, _ = _model(nf=4)
    x = torch.rand(1, 1, 32, 32).cuda()
    m(x)
    m2 = copy.deepcopy(m)                                # after a forward: engines hold ctypes structs
    assert len(m2._engines) == 0
    with torch.no_grad():
        torch.testing.assert_close(m2(x)[1], m(x)[1])
    import io
    buf = io.BytesIO()
    torch.save(m, buf)                                   # whole-module pickle
    mh = copy.deepcopy(m).half()
    with pytest.raises(RuntimeError, match="set_compute_dtype"):
        mh(x)


_SIDE_CODE = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
import bio_image_unet_amd as B
torch.manual_seed(5)
m = B.UNet3D(1, 1, 32).cuda(); m.set_compute_dtype(torch.bfloat16); m.train()
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.rand(2, 1, 32, 64, 64, device="cuda", generator=g)
outs = []
for _ in range(2):                       # the second step reuses every event / workspace / deferred slot of the first
    m.zero_grad()
    p, l = m(x)
    (l.float().square().mean() + p.float().mean()).backward()
    torch.cuda.synchronize()
eng = list(m._engines.values())[-1][-1]
print("SIDE", eng._side is not None)
torch.save({k: v.grad.detach().float().cpu() for k, v in m.named_parameters()}, sys.argv[2])
"""


@pytest.mark.timeout(600)
def test_side_stream_gradients_match_the_single_stream_step(tmp_path):
    """Composed-weight packing, the chain rule of the folded weight gradients and the weight gradients of the small levels run on the
    engine's side stream: every parameter gradient must equal the single-stream step's up to the summation order of the
    fp32 atomics (and the one extra storage rounding where a small layer's BatchNorm backward becomes a pass of its own)."""
    import subprocess
    res = {}
    for tag, env in (("side", {}), ("single", {"BIU_SIDE_WGRAD_VOX": "0", "BIU_DISABLE": "sidechain,prepack"})):
        f = tmp_path / f"{tag}.pt"
        e = dict(os.environ)
        e.pop("BIU_DISABLE", None)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", _SIDE_CODE, ROOT, str(f)], env=e, capture_output=True, text=True, timeout=500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert ("SIDE True" in r.stdout) == (tag == "side"), r.stdout[-500:]
        res[tag] = torch.load(f)
    for k, gs in res["side"].items():
        g1 = res["single"][k]
        scale = float(g1.abs().max()) + 1e-20
        assert float((gs - g1).abs().max()) <= 2e-2 * scale, (k, float((gs - g1).abs().max()) / scale)


def test_tensors_a_fold_never_materialises_hold_no_memory():
    """ConvTranspose + concat + conv as one op (biu_foldt_*) never writes the ConvT output, nearest up-sampling + conv folded in all three
    directions never writes the up-sampled tensor: their buffers (and gradient twins) are released at engine build, their Acts keep extents
    and channel counts but a NULL pointer."""
    from bio_image_unet_amd import engine as E
    torch.manual_seed(0)
    for mk, shape in ((lambda: B.UNet3D(1, 1, 16), (1, 1, 16, 32, 32)),
                      (lambda: B.MultiOutputUnet3D(1, None, 16, use_interpolation=True), (1, 1, 16, 32, 32))):
        m = mk().cuda()
        m.set_compute_dtype(torch.bfloat16)
        m.train()
        x = torch.rand(*shape, device="cuda")
        out = m(x)
        t = out[1] if isinstance(out, tuple) else sum(v.float().mean() for v in out.values())
        t.float().square().mean().backward()
        torch.cuda.synchronize()
        eng = list(m._engines.values())[-1][-1]
        released = 0
        for nd in eng.nodes:
            if isinstance(nd, E.ConvTNode) and nd.folded_into is not None:
                assert nd.y.buf.t is None and nd.y.buf.g is None and not nd.y._a.p
                released += 1
            if isinstance(nd, E.ResampleNode) and getattr(nd, "skip", False):
                assert nd.y.buf.t is None and not nd.y._a.p
                released += 1
        assert released >= 2, released
        assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
