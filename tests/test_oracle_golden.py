"""Pin the CPU oracle (oracle/unet_oracle.py) to vectors produced by the reference itself.

Chain of trust: reference (imported by path in the build container) -> tests/golden/*.npz -> oracle -> HIP engine.
Tolerances: the oracle calls the same ATen CPU kernels as the reference, so outputs agree to ~1e-6.
"""
import pytest
import torch

from oracle import unet_oracle as O
from tests.golden_util import ALL_CASES, load_case

RTOL, ATOL = 1e-5, 1e-6


def run_oracle(g, sd, training):
    meta = g["meta"]
    ctor = meta["ctor"]
    x = g["in"]["x"]
    if meta["model"] == "Unet":
        return dict(zip(("prob", "logits"), O.unet2d_forward(sd, x, dilation=ctor["dilation"], training=training)))
    if meta["model"] == "AttentionUnet":
        return dict(zip(("prob", "logits"), O.attention_unet_forward(sd, x, dilation=ctor["dilation"], training=training)))
    if meta["model"] in ("Unet_v0", "BabyUnet"):
        f = g["in"]["dropout_factor"] if training else None
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
