"""Pin the CPU oracle (oracle/unet_oracle.py) to vectors produced by the reference itself.

Chain of trust: reference (imported by path in the build container) -> tests/golden/*.npz -> oracle -> HIP engine.
Tolerances: the oracle calls the same ATen CPU kernels as the reference, so outputs agree to ~1e-6.
"""
import pytest
import torch

from oracle import unet_oracle as O
from tests.golden_util import ALL_CASES, load_case

RTOL, ATOL = 1e-5, 1e-6


def run_oracle(g, sd, training, step2=False):
    meta = g["meta"]
    ctor = meta["ctor"]
    x = g["in"]["x"]
    if meta["model"] == "Unet":
        return dict(zip(("prob", "logits"), O.unet2d_forward(sd, x, dilation=ctor["dilation"], training=training)))
    if meta["model"] == "AttentionUnet":
        return dict(zip(("prob", "logits"), O.attention_unet_forward(sd, x, dilation=ctor["dilation"], training=training)))
    if meta["model"] in ("Unet_v0", "BabyUnet"):
        f = g["in"]["dropout_factor2" if step2 else "dropout_factor"] if training else None
        return dict(zip(("prob", "logits"), O.legacy_unet_forward(sd, x, levels=4 if meta["model"] == "Unet_v0" else 3, training=training,
                                                                  dropout_factor=f)))
    if meta["model"] == "UNet3D":
        return dict(zip(("prob", "logits"),
                        O.unet3d_forward(sd, x, use_interpolation=ctor["use_interpolation"], training=training)))
    if meta["model"] == "Siam_UNet":
        return dict(zip(("prob", "logits"),
                        O.siam_forward(sd, x, g["in"]["prev_x"], mode=ctor["mode"], training=training)))
    if meta["model"] == "MultiOutputUnet3D":
        return O.mo3d_forward(sd, x, ctor["output_heads"], use_interpolation=ctor["use_interpolation"],
                              training=training)
    raise AssertionError(meta["model"])


def oracle_loss(g, outs):
    meta = g["meta"]
    if meta["model"] in ("Unet", "AttentionUnet"):
        crit = O.bce_dice_loss if "BCEDice" in meta["loss"] else O.tversky_loss
        return O.trainer2d_loss(outs["logits"], g["in"]["target"], meta["ctor"]["out_channels"], criterion=crit)
    if meta["model"] in ("Unet_v0", "BabyUnet"):
        return O.trainer2d_loss(outs["logits"], g["in"]["target"], 1)
    if meta["model"] == "UNet3D":
        return O.trainer3d_loss(outs["logits"], g["in"]["target"], 0.1)
    if meta["model"] == "Siam_UNet":       # the Siam package's own BCEDice (BCELoss on probabilities), loss_params (1, 1)
        return O.siam_bce_dice_loss(outs["logits"], g["in"]["target"], 1.0, 1.0)
    tg = {k.split(".", 1)[1]: v for k, v in g["in"].items() if k.startswith("target.")}
    if "train.py:183-201" in meta["loss"]:          # the mo3d trainer's per-head loss menu on the activated outputs
        return O.trainer_mo3d_loss(outs, tg, meta["ctor"]["output_heads"])
    return sum(((outs[k] - tg[k]) ** 2).mean() * w for k, w in (("seg", 1.0), ("flow", 0.5), ("dist", 0.25)))


@pytest.mark.parametrize("case", ALL_CASES)
def test_oracle_matches_reference_vectors(case):
    torch.set_num_threads(4)
    g = load_case(case)
    sd = O.clone_state(g["sd"], requires_grad=True)
    outs = run_oracle(g, sd, training=True)
    for k, v in g["train"].items():
        torch.testing.assert_close(outs[k].detach(), v, rtol=RTOL, atol=ATOL, msg=lambda m: f"train.{k}: {m}")
    loss = oracle_loss(g, outs)
    torch.testing.assert_close(loss.detach(), g["loss"], rtol=RTOL, atol=ATOL)
    grads = O.grads_of(loss, sd)
    assert set(grads) == set(g["grad"])
    gscale = max(float(v.abs().max()) for v in g["grad"].values())
    for k, v in g["grad"].items():
        # conv biases that feed a train-mode BN have an exactly-zero true gradient; the reference value is
        # rounding noise (~1e-8), so the absolute floor is tied to the global gradient scale
        scale = float(v.abs().max())
        torch.testing.assert_close(grads[k], v, rtol=1e-4, atol=1e-5 * scale + 1e-6 * gscale,
                                   msg=lambda m: f"grad.{k}: {m}")
    # BN buffers after the train-mode forward
    for k, v in g["sd1"].items():
        torch.testing.assert_close(sd[k].detach(), v, rtol=RTOL, atol=ATOL, msg=lambda m: f"sd1.{k}: {m}")
    # eval-mode forward from the updated buffers
    with torch.no_grad():
        outs_e = run_oracle(g, sd, training=False)
    for k, v in g["eval"].items():
        torch.testing.assert_close(outs_e[k], v, rtol=RTOL, atol=ATOL, msg=lambda m: f"eval.{k}: {m}")
    # the rest of the reference loop: (clip_grad_norm_,) Adam step, next iteration's forward (unet/train.py:137-139; mo3d :201)
    clip = 1.0 if g["gradnorm"] is not None else None
    new, norm = O.adam_step(sd, grads, lr=1e-3, clip=clip)
    if clip is not None:
        torch.testing.assert_close(norm, g["gradnorm"], rtol=1e-4, atol=0)
    assert set(new) == set(g["adam1"])
    for k, v in g["adam1"].items():
        # one Adam step moves an entry by lr * g / (|g| + 1e-8): entries whose gradient is rounding noise (dead conv biases,
        # |g| ~ 1e-8) move by an arbitrary fraction of lr in ANY implementation -- compare where the reference gradient is resolved
        solid = g["grad"][k].abs() > 1e-4 * gscale
        torch.testing.assert_close(new[k][solid], v[solid], rtol=1e-5, atol=2e-6, msg=lambda m: f"adam1.{k}: {m}")
        assert float((new[k] - v).abs().max()) <= 2.0e-3 + 1e-6, f"adam1.{k}: an entry moved by more than 2 lr"
    sd2 = O.clone_state({**{k: v for k, v in sd.items() if not O.is_param(k)}, **g["adam1"]}, requires_grad=False)
    with torch.no_grad():
        outs2 = run_oracle(g, sd2, training=True, step2=True)
        loss2 = oracle_loss(g, outs2)
    torch.testing.assert_close(loss2, g["loss2"], rtol=RTOL, atol=ATOL)
    for k, v in g["sd2"].items():
        torch.testing.assert_close(sd2[k], v, rtol=RTOL, atol=ATOL, msg=lambda m: f"sd2.{k}: {m}")


@pytest.mark.parametrize("kind", ["unet2d", "siam_concat", "siam_max", "unet3d", "unet3d_interp", "mo3d", "mo3d_convT"])
def test_oracle_init_key_schema_matches_reference(kind):
    """The oracle's own parameter constructors emit exactly the reference's state_dict keys and shapes."""
    case = {"unet2d": "unet2d_f4", "siam_concat": "siam_f4_concat", "siam_max": "siam_f4_max", "unet3d": "unet3d_f4",
            "unet3d_interp": "unet3d_f4_interp", "mo3d": "mo3d_f4_interp", "mo3d_convT": "mo3d_f4_convT"}[kind]
    g = load_case(case)
    ctor = g["meta"]["ctor"]
    if kind == "unet2d":
        sd = O.init_unet2d(ctor["in_channels"], ctor["out_channels"], ctor["n_filter"])
    elif kind.startswith("siam"):
        sd = O.init_unet2d(1, 1, ctor["n_filter"], siam_mode=ctor["mode"], init_weights=False)
    elif kind.startswith("unet3d"):
        sd = O.init_unet3d(ctor["in_channels"], ctor["out_channels"], ctor["n_filter"], ctor["use_interpolation"])
    else:
        sd = O.init_mo3d(ctor["in_channels"], ctor["output_heads"], ctor["n_filter"], ctor["use_interpolation"])
    assert set(sd) == set(g["sd"])
    for k in sd:
        assert tuple(sd[k].shape) == tuple(g["sd"][k].shape), k
