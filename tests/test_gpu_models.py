"""Whole-network parity on the GPU: the HIP engine against the CPU oracle (pinned to the reference by
tests/golden) on the golden inputs themselves and on larger seeded cases.

Tolerance (north_star): 1e-3 relative in fp32, stated per assertion below; thresholded masks are bit-exact on the golden
cases.  bf16: storage is 8-bit mantissa -> 5e-2 of the tensor's max, masks equal outside that band."""
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
    cls = {"Unet": B.Unet, "UNet3D": B.UNet3D, "Siam_UNet": B.Siam_UNet, "MultiOutputUnet3D": B.MultiOutputUnet3D,
           "AttentionUnet": B.AttentionUnet, "Unet_v0": B.Unet_v0, "BabyUnet": B.BabyUnet}[meta["model"]]
    return cls(**ctor)


def relerr(got, want):
    return float((got - want).abs().max()) / (float(want.abs().max()) + 1e-12)


def masks_agree(logits, ref_logits, band):
    safe = ref_logits.abs() > band
    return bool(((logits > 0) == (ref_logits > 0))[safe].all())


GOLDEN_GPU = ["unet2d_f4", "unet2d_f4_o2_dil2", "unet3d_f4", "unet3d_f4_interp", "siam_f4_concat", "siam_f4_max",
              "siam_f4_corr", "siam_f4_control", "mo3d_f4_interp", "mo3d_f4_convT", "attention_f4", "unet_v0_f4", "baby_f4",
              "mo3d_f4_trainer_convT", "mo3d_f4_trainer_interp"]


# Per-tensor exceptions to the 1e-3 gradient bound of the golden fixtures, as multiples of it: (case, parameter) -> factor.  An entry is
# allowed only where a measured LeakyReLU / max-pool decision flip against the reference's own CPU run moves that tensor;
# the measured worst ratio of every case is written to gpurun_out/golden_grad_ratios.txt by the test itself.
GOLDEN_GRAD_EXC = {}


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
    if "dropout_factor" in g["in"]:          # Unet_v0 / BabyUnet: the Dropout2d(0.5) draw of the reference run
        m._engine_for(*ins)
        m._dropout_node.mask_override = (g["in"]["dropout_factor"] > 0).float()
    outs = m(*ins)
    names = list(g["train"].keys())
    od = outs if isinstance(outs, dict) else dict(zip(("prob", "logits"), outs))
    for k in names:
        assert relerr(od[k].detach().cpu(), g["train"][k]) < REL, f"train.{k}"
    if "logits" in od:       # argmax masks bit-exact (north star): thresholded logits equal at EVERY voxel
        assert torch.equal(od["logits"].detach().cpu() > 0, g["train"]["logits"] > 0), "fp32 masks differ from the reference"
        assert torch.equal(od["prob"].detach().cpu() > 0.5, g["train"]["prob"] > 0.5)
    gi = {"meta": meta, "in": {k: v.cuda() for k, v in g["in"].items()}}
    loss = oracle_loss(gi, od)
    assert abs(float(loss) - float(g["loss"])) < REL * max(1.0, abs(float(g["loss"])))
    loss.backward()
    gscale = max(float(v.abs().max()) for v in g["grad"].values())
    worst = (0.0, "")
    for k, p in m.named_parameters():
        want = g["grad"][k]
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(want)
        # per-tensor 1e-3 of its own scale, with a floor at 1e-5 of the largest gradient in the net
        # (conv biases in front of a train-mode BN have true gradient 0: the reference value is rounding noise)
        tol = REL * float(want.abs().max()) + 1e-5 * gscale
        ratio = float((got - want).abs().max()) / tol
        worst = max(worst, (ratio, k))
        assert ratio <= GOLDEN_GRAD_EXC.get((case, k), 1.0), f"grad.{k}: {float((got - want).abs().max())} = {ratio:.2f} x {tol}"
    _record("golden_grad_ratios.txt", f"{case}: worst gradient error {worst[0]:.3f} x (1e-3 of the tensor's scale + 1e-5 of the net's) at {worst[1]}")
    sd_now = m.state_dict()
    for k, v in g["sd1"].items():
        torch.testing.assert_close(sd_now[k].cpu(), v, rtol=REL, atol=1e-5, msg=lambda s: f"sd1.{k}: {s}")
    m.eval()
    with torch.no_grad():
        outs_e = m(*ins)
    oe = outs_e if isinstance(outs_e, dict) else dict(zip(("prob", "logits"), outs_e))
    for k, v in g["eval"].items():
        assert relerr(oe[k].cpu(), v) < REL, f"eval.{k}"
    if "logits" in oe:
        assert torch.equal(oe["logits"].cpu() > 0, g["eval"]["logits"] > 0), "fp32 eval masks differ from the reference"
    # ---- the rest of the reference loop (unet/train.py:137-139; mo3d: clip_grad_norm_(1.0) first, train.py:201): the fused Adam
    # kernel on the engine's gradients against the parameters the reference's torch.optim.Adam produced ------------------------
    from bio_image_unet_amd.optim import Adam
    m.train()
    opt = Adam(m.parameters(), lr=1e-3)
    if g["gradnorm"] is not None:
        norm = opt.clip_grad_norm_(1.0)                  # biu_grad_clip against the norm the reference's clip_grad_norm_ returned
        assert abs(float(norm) - float(g["gradnorm"])) < REL * float(g["gradnorm"]), (float(norm), float(g["gradnorm"]))
    opt.step()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        want, g0 = g["adam1"][k], g["grad"][k]
        # first Adam step: lr * g / (|g| + 1e-8).  Entries whose reference gradient is resolved well above the gradient tolerance
        # must land on the reference's value; the rest (rounding-noise gradients: dead conv biases, ~0 entries) within 2 lr
        solid = g0.abs() > 20 * (2 * REL * float(g0.abs().max()) + 1e-5 * gscale)
        got = p.detach().cpu()
        assert float((got - want).abs().max()) <= 2.0e-3 + 1e-6, f"adam1.{k}: an entry moved by more than 2 lr"
        if solid.any():
            assert float((got - want)[solid].abs().max()) <= 2e-5, f"adam1.{k}: {float((got - want)[solid].abs().max())}"
    # ... and the next iteration's forward FROM THE REFERENCE'S updated parameters: loss and BN buffers after step 2
    sd_ref = {k: v.clone() for k, v in m.state_dict().items()}
    sd_ref.update(g["adam1"])
    m.load_state_dict(sd_ref)
    if "dropout_factor2" in g["in"]:
        m._dropout_node.mask_override = (g["in"]["dropout_factor2"] > 0).float()
    with torch.no_grad():
        outs2 = m(*ins)
    od2 = outs2 if isinstance(outs2, dict) else dict(zip(("prob", "logits"), outs2))
    loss2 = oracle_loss(gi, od2)
    assert abs(float(loss2) - float(g["loss2"])) < REL * max(1.0, abs(float(g["loss2"])))
    sd_now = m.state_dict()
    for k, v in g["sd2"].items():
        torch.testing.assert_close(sd_now[k].cpu(), v, rtol=REL, atol=1e-5, msg=lambda s: f"sd2.{k}: {s}")


# ------------------------------------------------------------------------------------------------------------------
# seeded mid-size networks at the MFMA widths of the BASELINE configs
# ------------------------------------------------------------------------------------------------------------------
HEADS = {"seg": {"channels": 1, "activation": "sigmoid"}, "flow": {"channels": 2, "activation": None},
         "dist": {"channels": 1, "activation": "tanh"}}

# cfg5 exactly as bench.py builds it (bench.HEADS5): base 64, three heads, activations (sigmoid, None, sigmoid)
HEADS5 = {"seg": {"channels": 1, "activation": "sigmoid"}, "flow": {"channels": 2, "activation": None},
          "dist": {"channels": 1, "activation": "sigmoid"}}

# kind -> (product ctor, oracle init, oracle forward(sd, xs, training) -> dict of outputs, input shape, number of inputs)
KINDS = {
    "unet2d_f16": (lambda: B.Unet(1, 1, 16), lambda s: O.init_unet2d(1, 1, 16, seed=s), "unet2d", (2, 1, 128, 128), 1),
    "cfg1_unet2d_f32": (lambda: B.Unet(1, 1, 32), lambda s: O.init_unet2d(1, 1, 32, seed=s), "unet2d", (2, 1, 256, 256), 1),       # cfg1 verbatim
    "cfg2_unet2d_f64_o2": (lambda: B.Unet(1, 2, 64), lambda s: O.init_unet2d(1, 2, 64, seed=s), "unet2d", (2, 1, 64, 64), 1),        # cfg2 widths
    "cfg3_siam_max_f32": (lambda: B.Siam_UNet(32, "max"), lambda s: O.init_unet2d(1, 1, 32, seed=s, init_weights=False), "siam_max", (2, 1, 64, 64), 2),
    "siam_concat_f16": (lambda: B.Siam_UNet(16, "concat"), lambda s: O.init_unet2d(1, 1, 16, seed=s, init_weights=False, siam_mode="concat"), "siam_concat", (2, 1, 64, 64), 2),
    "cfg4_unet3d_f32": (lambda: B.UNet3D(1, 1, 32), lambda s: O.init_unet3d(1, 1, 32, seed=s), "unet3d", (2, 1, 16, 32, 32), 1),
    "cfg5_mo3d_f32_interp": (lambda: B.MultiOutputUnet3D(1, HEADS, 32, True), lambda s: O.init_mo3d(1, HEADS, 32, True, seed=s), "mo3d_interp", (1, 1, 16, 32, 32), 1),
    "cfg5_mo3d_f32_convT": (lambda: B.MultiOutputUnet3D(1, HEADS, 32, False), lambda s: O.init_mo3d(1, HEADS, 32, False, seed=s), "mo3d_convT", (1, 1, 16, 32, 32), 1),
    # cfg5 at its STATED width (base 64: 3-D layers of 512 / 768 channels), both up-sampling modes, reduced extent
    "cfg5_mo3d_f64_interp": (lambda: B.MultiOutputUnet3D(1, HEADS5, 64, True), lambda s: O.init_mo3d(1, HEADS5, 64, True, seed=s), "mo3d5_interp", (1, 1, 16, 32, 32), 1),
    "cfg5_mo3d_f64_convT": (lambda: B.MultiOutputUnet3D(1, HEADS5, 64, False), lambda s: O.init_mo3d(1, HEADS5, 64, False, seed=s), "mo3d5_convT", (1, 1, 16, 32, 32), 1),
}


def _oracle_forward(fkind, sd, xs, training):
    if fkind == "unet2d":
        return dict(zip(("prob", "logits"), O.unet2d_forward(sd, xs[0], training=training)))
    if fkind == "unet3d":
        return dict(zip(("prob", "logits"), O.unet3d_forward(sd, xs[0], training=training)))
    if fkind.startswith("siam"):
        return dict(zip(("prob", "logits"), O.siam_forward(sd, xs[0], xs[1], mode=fkind.split("_")[1], training=training)))
    return O.mo3d_forward(sd, xs[0], HEADS5 if fkind.startswith("mo3d5") else HEADS, use_interpolation=fkind.endswith("interp"), training=training)


def _loss(outs, tg):
    if "logits" in outs:
        return O.bce_dice_loss(outs["logits"], tg["y"])
    return sum(((outs[k] - tg[k]) ** 2).mean() * w for k, w in (("seg", 1.0), ("flow", 0.5), ("dist", 0.25)))


def _problem(kind, seed):
    mk, init, fkind, shape, nin = KINDS[kind]
    g = torch.Generator().manual_seed(100 + seed)
    xs = [torch.rand(*shape, generator=g) for _ in range(nin)]
    if fkind.startswith("mo3d"):
        tg = {k: torch.rand((shape[0], v["channels"]) + tuple(shape[2:]), generator=g) for k, v in HEADS.items()}       # (HEADS5: same names / channels)
    else:
        oc = 2 if "o2" in kind else 1
        tg = {"y": (torch.rand((shape[0], oc) + tuple(shape[2:]), generator=g) > 0.5).float()}
    return mk, init(3 + seed), fkind, xs, tg


def _oracle_run(fkind, sd, xs, tg, dt, emu=False, training=True):
    osd = O.clone_state({k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}, requires_grad=True)
    with O.emulate_bf16(emu):
        outs = _oracle_forward(fkind, osd, [x.to(dt) for x in xs], training)
        loss = _loss(outs, {k: v.to(dt) for k, v in tg.items()})
        grads = O.grads_of(loss, osd)
    return {k: v.detach() for k, v in outs.items()}, loss.detach(), grads, osd


def _hip_run(mk, sd, xs, tg, dtype):
    m = mk().cuda()
    m.load_state_dict(sd)
    if dtype == "bf16":
        m.set_compute_dtype(torch.bfloat16)
    m.train()
    outs = m(*[x.cuda() for x in xs])
    od = outs if isinstance(outs, dict) else dict(zip(("prob", "logits"), outs))
    loss = _loss(od, {k: v.cuda() for k, v in tg.items()})
    loss.backward()
    torch.cuda.synchronize()
    return m, {k: v.detach().cpu() for k, v in od.items()}, float(loss), {k: p.grad.cpu() for k, p in m.named_parameters()}


def _dead(k):
    """Conv biases in front of a train-mode BatchNorm: true gradient exactly 0 (the reference value is rounding noise)."""
    return k.endswith(".0.bias") and not k.startswith("final")


def _grad_errors(grads, truth):
    """per parameter: (max-abs error / (max|truth| + 1e-2 of the largest gradient in the net), relative L2 error, cosine)"""
    gscale = max(float(v.abs().max()) for v in truth.values())
    out = {}
    for k, want in truth.items():
        got, want = grads[k].double(), want.double()
        e = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-2 * gscale)
        l2 = float((got - want).norm() / (want.norm() + 1e-3 * gscale * want.numel() ** 0.5))
        cos = float((got * want).sum() / (got.norm() * want.norm() + 1e-300))
        out[k] = (e, l2, cos)
    return out


def _record(name, text):
    """Measured errors go to gpurun_out/ (copied to profiles/ by hand when they are to be judged)."""
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "a") as f:
        f.write(text + "\n")


@pytest.mark.parametrize("kind", list(KINDS))
def test_midsize_fp32_vs_oracle(kind):
    """fp32 engine against the fp64 oracle: outputs, loss, masks, BN buffers, eval forward and EVERY parameter gradient within
    the north star's 1e-3 -- a flat bound, no conditioning terms.

    The gradient of this network is piecewise: a LeakyReLU or max-pool decision that sits within fp32 rounding of its boundary
    falls either way in ANY fp32 implementation (the reference's CPU path included: its own gradient is 1e-3 .. 2e-2 from the
    fp64 one on most of these seeds), and one flipped voxel moves a bottleneck weight gradient by that much -- an event, not
    an error level.  So the fp64 oracle is evaluated on the branch the engine took (``oracle.forced_decisions`` with the
    masks / argmax indices read back from the engine's stored activations); forward values do not depend on that choice and
    are compared with the free-running oracle."""
    from tests import insitu
    mk, sd, fkind, xs, tg = _problem(kind, 0)
    m, outs, loss, grads = _hip_run(mk, sd, xs, tg, "f32")
    q = insitu.extract_decisions(list(m._engines.values())[-1][-1])
    with O.record_decisions() as rq:
        f_outs, f_loss, _, osd = _oracle_run(fkind, sd, xs, tg, torch.float32)        # free-running fp32 oracle: the reference's arithmetic
    # how many of the engine's LeakyReLU / max-pool / max-join decisions differ from the ones the reference arithmetic took by
    # itself: only elements within fp32 rounding of a boundary may (a kernel that mis-decides systematically would be replayed
    # by forced_decisions below, not caught -- this is the check that catches it)
    flips = insitu.decision_mismatch(q, rq)
    _record("parity_fp32_decision_flips.txt", f"{kind}: " + "; ".join(f"{k} {d}/{t} = {d / t:.2e}" for k, (d, t) in flips.items()))
    for k, (d, t) in flips.items():
        assert d <= 1e-4 * t + 2, f"{k}: {d} of {t} decisions differ from the free-running fp32 oracle"
    with O.forced_decisions(q):
        t_outs, t_loss, t_grads, _ = _oracle_run(fkind, sd, xs, tg, torch.float64)
    with O.forced_decisions(q):
        _, _, r_grads, _ = _oracle_run(fkind, sd, xs, tg, torch.float32)
    for k, want in f_outs.items():
        assert relerr(outs[k], want) < REL, f"{k} rel err {relerr(outs[k], want)}"
        assert relerr(outs[k], t_outs[k].float()) < REL, f"{k} rel err vs fp64 {relerr(outs[k], t_outs[k].float())}"
    if "logits" in outs:
        assert masks_agree(outs["logits"], t_outs["logits"].float(), 1e-5 * float(t_outs["logits"].abs().max()))
    assert abs(loss - float(f_loss)) < REL * max(1.0, abs(float(f_loss)))
    mine, cpu32 = _grad_errors(grads, t_grads), _grad_errors(r_grads, t_grads)
    gmax = max(float(v.abs().max()) for v in t_grads.values())
    rows = sorted(((v[0], k, cpu32[k][0], v[2]) for k, v in mine.items() if not _dead(k)), reverse=True)
    _record("parity_fp32_grad_errors.txt", f"{kind}: worst " + "; ".join(f"{k} {e:.2e} (cpu32 {c:.2e})" for e, k, c, _ in rows[:6]))
    for e, k, c, cos in rows:
        assert e <= REL, f"grad {k}: err {e} > 1e-3 (CPU fp32 on the same branch: {c})"
        if float(t_grads[k].abs().max()) > 1e-3 * gmax:
            assert cos > 0.99999, f"grad {k}: cosine {cos}"
    for k in sd:
        if "running_" in k:
            torch.testing.assert_close(m.state_dict()[k].cpu(), osd[k].detach(), rtol=REL, atol=REL)
    m.eval()                # eval-mode forward from the updated buffers (no-statistics form of every kernel)
    with torch.no_grad():
        oe = m(*[x.cuda() for x in xs])
        oe = oe if isinstance(oe, dict) else dict(zip(("prob", "logits"), oe))
        re_ = _oracle_forward(fkind, {k: v.detach() for k, v in osd.items()}, xs, False)
    for k, want in re_.items():
        assert relerr(oe[k].cpu(), want) < REL, f"eval {k}: {relerr(oe[k].cpu(), want)}"


@pytest.mark.parametrize("kind", ["unet2d_f16", "cfg3_siam_max_f32", "cfg4_unet3d_f32", "cfg5_mo3d_f32_interp"])
def test_eval_mode_backward_fp32_vs_oracle(kind):
    """``model.eval(); loss.backward()`` -- frozen-BatchNorm fine-tuning, which the reference's plain nn.BatchNorm2d/3d blocks allow
    (unet/unet.py:54-60, unet3d/unet3d.py:52-58): running statistics are constants, dy = scale * dz, gamma / beta and -- unlike in train
    mode -- the conv biases receive real gradients.  Every parameter gradient against the fp64 oracle in eval mode on the engine's own
    LeakyReLU / max-pool branch (as test_midsize_fp32_vs_oracle); covers the folded decoder levels (cfg4: biu_foldt_bwd_weight_bn with
    dy_sum; cfg5: the folded up-convs) and the weight-shared Siam encoder."""
    from tests import insitu
    mk, sd, fkind, xs, tg = _problem(kind, 1)
    g = torch.Generator().manual_seed(77)
    sd = dict(sd)
    for k in sd:                                          # non-trivial running statistics
        if k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) + 0.5
    m = mk().cuda()
    m.load_state_dict(sd)
    m.eval()
    outs = m(*[x.cuda() for x in xs])
    od = outs if isinstance(outs, dict) else dict(zip(("prob", "logits"), outs))
    loss = _loss(od, {k: v.cuda() for k, v in tg.items()})
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.cpu() for k, p in m.named_parameters()}
    q = insitu.extract_decisions(list(m._engines.values())[-1][-1])
    with O.forced_decisions(q):
        t_outs, t_loss, t_grads, _ = _oracle_run(fkind, sd, xs, tg, torch.float64, training=False)
    for k, want in t_outs.items():
        assert relerr(od[k].detach().cpu(), want.float()) < REL, f"{k} rel err {relerr(od[k].detach().cpu(), want.float())}"
    assert abs(float(loss) - float(t_loss)) < REL * max(1.0, abs(float(t_loss)))
    errs = _grad_errors(grads, t_grads)
    rows = sorted(((v[0], k) for k, v in errs.items()), reverse=True)
    _record("parity_fp32_eval_backward.txt", f"{kind}: worst " + "; ".join(f"{k} {e:.2e}" for e, k in rows[:6]))
    for e, k in rows:
        assert e <= REL, f"grad {k}: err {e} > 1e-3"
    # the conv biases are live in eval mode: their gradients must not be the train-mode zeros
    live = [k for k in t_grads if _dead(k) and float(t_grads[k].abs().max()) > 0]
    assert live and all(float(grads[k].abs().max()) > 0 for k in live)
    # buffers untouched by an eval-mode step
    for k, v in sd.items():
        if "running_" in k:
            torch.testing.assert_close(m.state_dict()[k].cpu(), v)


@pytest.mark.parametrize("kind", list(KINDS))
def test_midsize_bf16_vs_oracle(kind):
    """bf16 engine (bf16 storage of activations, gradients and MFMA operands; fp32 accumulation).

    What bf16 storage costs is a property of the network, not of the kernels: the oracle's own bf16-storage emulation
    (pure torch CPU, rounding where the engine stores / packs bf16) is 15 - 40 % (relative L2) away from the fp64 gradient
    on these problems, all of it from rounding the forward values (rounding only the gradients costs 0.7 %), and fp64
    arithmetic with the conv weights perturbed by 2^-9 -- one bf16 rounding -- is already 14 % away
    (profiles/r02_bf16_error_budget.md).  Two correct bf16 implementations also differ from each other at that level
    (rounding is chaotic), so the sharp per-kernel statement for bf16 is tests/test_gpu_insitu.py; here:
      * outputs within 0.15 (worst single element) of the fp64 oracle, rms deviation <= 1.5 x the emulation's, masks equal outside a 5e-2 band;
      * the engine is no further from the fp64 gradient than bf16 storage itself puts the emulation:
        per parameter L2 error <= 1.6 x the emulation's + 0.03, over all parameters (rms) <= 1.25 x."""
    mk, sd, fkind, xs, tg = _problem(kind, 0)
    t_outs, t_loss, t_grads, _ = _oracle_run(fkind, sd, xs, tg, torch.float64)
    e_outs, e_loss, e_grads, _ = _oracle_run(fkind, sd, xs, tg, torch.float32, emu=True)
    m, outs, loss, grads = _hip_run(mk, sd, xs, tg, "bf16")
    for k, want in t_outs.items():
        assert relerr(outs[k], want.float()) < 0.15, f"{k} rel err {relerr(outs[k], want.float())}"     # worst single element, of the tensor's max
        d_h = float((outs[k].double() - want).pow(2).mean().sqrt())
        d_e = float((e_outs[k].double() - want).pow(2).mean().sqrt())
        assert d_h <= 1.5 * d_e + 1e-3 * float(want.abs().max()), f"{k}: rms deviation {d_h} vs the emulation's {d_e}"
    if "logits" in outs:
        assert masks_agree(outs["logits"], t_outs["logits"].float(), 5e-2 * float(t_outs["logits"].abs().max()))
    assert abs(loss - float(t_loss)) < 5e-2
    mine, emu = _grad_errors(grads, t_grads), _grad_errors(e_grads, t_grads)
    gmax = max(float(v.abs().max()) for v in t_grads.values())
    num = den = 0.0
    rows = []
    for k, (e, l2, cos) in mine.items():
        if _dead(k):
            continue
        rows.append((l2, k, emu[k][1], cos, emu[k][2]))
        assert l2 <= 1.6 * emu[k][1] + 0.03, f"grad {k}: L2 error {l2} vs the bf16 emulation's {emu[k][1]}"
        if float(t_grads[k].abs().max()) > 1e-3 * gmax:
            assert cos >= min(0.99, emu[k][2] - 0.05), f"grad {k}: cosine {cos} vs the emulation's {emu[k][2]}"
        num += l2 * l2
        den += emu[k][1] ** 2
    rows.sort(reverse=True)
    ratio = (num / max(den, 1e-30)) ** 0.5
    _record("parity_bf16_grad_errors.txt", f"{kind}: rms L2-error ratio engine/emulation {ratio:.3f}; worst " +
            "; ".join(f"{k} l2 {a:.3f} (emu {b:.3f}) cos {c:.4f} (emu {d:.4f})" for a, k, b, c, d in rows[:5]))
    assert ratio <= 1.25, f"engine gradients are {ratio:.2f}x as far from fp64 as bf16 storage itself explains"


def _blobs(shape, seed):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(shape, generator=g)
    k = torch.ones(1, 1, 9, 9) / 81
    for _ in range(2):
        z = torch.nn.functional.conv2d(z, k, padding=4)
    return z / z.std()


def test_training_curves_fp32_bf16_oracle():
    """25 Adam steps on a learnable problem (noisy blobs -> blob mask): the bf16 engine's loss curve follows the fp32
    engine's, which follows the CPU oracle's (unet/train.py:130-139 loop: forward, BCEDice, zero_grad, backward, step)."""
    from bio_image_unet_amd.optim import Adam
    shape, nf, steps = (4, 1, 64, 64), 16, 25
    b = _blobs(shape, 1)
    x = ((b - b.min()) / (b.max() - b.min()) + 0.1 * torch.randn(shape, generator=torch.Generator().manual_seed(2))).clamp(0, 1)
    y = (b > 0.3).float()
    sd = O.init_unet2d(1, 1, nf, seed=3)
    osd = O.clone_state(sd, requires_grad=True)
    opt = torch.optim.Adam([v for v in osd.values() if v.requires_grad], lr=1e-3)
    ref = []
    for _ in range(steps):
        _, lg = O.unet2d_forward(osd, x, training=True)
        loss = O.bce_dice_loss(lg, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref.append(float(loss))
    curves = {}
    for dtype in ("f32", "bf16"):
        m = B.Unet(1, 1, nf).cuda()
        m.load_state_dict(sd)
        if dtype == "bf16":
            m.set_compute_dtype(torch.bfloat16)
        m.train()
        o = Adam(m.parameters(), lr=1e-3)
        xc, yc = x.cuda(), y.cuda()
        cur = []
        for _ in range(steps):
            loss = O.bce_dice_loss(m(xc)[1], yc)
            o.zero_grad()
            loss.backward()
            o.step()
            cur.append(float(loss))
        curves[dtype] = cur
    _record("parity_training_curves.txt", "oracle " + " ".join(f"{v:.4f}" for v in ref) + "\nfp32   " + " ".join(f"{v:.4f}" for v in curves["f32"]) +
            "\nbf16   " + " ".join(f"{v:.4f}" for v in curves["bf16"]))
    assert ref[-1] < 0.7 * ref[0], "the problem must be learnable"
    for t in range(steps):
        assert abs(curves["f32"][t] - ref[t]) <= (0.005 if t < 10 else 0.03) * ref[0], f"fp32 step {t}: {curves['f32'][t]} vs oracle {ref[t]}"
        assert abs(curves["bf16"][t] - curves["f32"][t]) <= 0.04 * ref[0], f"bf16 step {t}: {curves['bf16'][t]} vs fp32 {curves['f32'][t]}"
    assert curves["bf16"][-1] < 0.7 * curves["bf16"][0]


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
