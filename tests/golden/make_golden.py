"""Generate the golden fixtures in this directory from the REFERENCE implementation.

Run in the build container only (``/root/reference`` is absent on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's model files (and ``unet/losses.py``) import nothing but torch, so they are loaded *by file
path* (the package ``__init__`` files pull in tifffile/skimage, which are not installed -- SURVEY.md 8c).
Only tensors leave this script: inputs, state_dict, outputs, loss, gradients, BN buffers.  No reference
source or bytecode is written anywhere.

Each fixture ``<case>.npz`` holds:
  meta_json                         ctor kwargs, seeds, loss description
  in.<name>                         inputs (x, prev_x, target)
  sd.<key>                          initial state_dict (float32 / int64)
  train.<out>                       outputs of a train-mode forward (BN batch stats) from the initial state
  loss                              scalar loss driven through the reference's own loss code
  grad.<key>                        d loss / d parameter for every parameter
  sd1.<key>                         BN buffers after that one train-mode forward
  eval.<out>                        outputs of an eval-mode forward from the *post-step-1* buffers
  gradnorm                          (trainer cases with clipping) total L2 norm of the gradients before clip_grad_norm_(1.0)
  adam1.<key>                       every parameter after ``optimizer.step()`` of torch.optim.Adam(lr=1e-3) on those gradients
                                    (``zero_grad(); backward(); step()``, unet/train.py:137-139; mo3d: clipped first, :201)
  loss2, sd2.<key>                  loss and BN buffers of the SECOND train-mode forward (from the Adam-updated parameters)
  in.dropout_factor2                (Unet_v0 / BabyUnet) the Dropout2d draw of that second forward
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference/bio_image_unet"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


unet_mod = load("ref_unet", "unet/unet.py")
unet3d_mod = load("ref_unet3d", "unet3d/unet3d.py")
siam_mod = load("ref_siam", "siam_unet/siam_unet.py")
mo3d_mod = load("ref_mo3d", "multi_output_unet3d/multi_output_unet3d.py")
losses_mod = load("ref_losses", "unet/losses.py")
att_mod = load("ref_attention_unet", "unet/attention_unet.py")
v0_mod = load("ref_unet_v0", "unet/unet_v0.py")
baby_mod = load("ref_baby_unet", "unet/baby_unet.py")
mo3d_losses_mod = load("ref_mo3d_losses", "multi_output_unet3d/losses.py")   # the mo3d package's criteria (temporal term, Tversky defaults)
siam_losses_mod = load("ref_siam_losses", "siam_unet/losses.py")     # the Siam package's own criteria (BCELoss on probabilities)


def ref_init_weights(m):
    # restated from utils/utils.py:76-78 (that module imports tifffile at top level, so it cannot be imported here)
    if isinstance(m, torch.nn.Conv2d):
        torch.nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")


def dump(case, meta, model, inputs, target, loss_fn, out_names, call, clip=None, after_step2=None):
    """Drive one reference train-mode step (forward + loss + backward), an eval forward, the optimizer step of the reference
    loop (Adam, lr 1e-3; ``clip``: clip_grad_norm_ to that norm first, as the mo3d trainer does) and a second train-mode forward."""
    arrays = {"meta_json": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)}
    for k, v in inputs.items():
        arrays[f"in.{k}"] = v.numpy()
    if target is not None:
        if isinstance(target, dict):
            for k, v in target.items():
                arrays[f"in.target.{k}"] = v.numpy()
        else:
            arrays["in.target"] = target.numpy()
    for k, v in model.state_dict().items():
        arrays[f"sd.{k}"] = v.detach().clone().numpy()
    model.train()
    outs = call(model)
    if isinstance(outs, dict):
        outs_t = [outs[n] for n in out_names]
    else:
        outs_t = list(outs)
    for n, t in zip(out_names, outs_t):
        arrays[f"train.{n}"] = t.detach().numpy()
    loss = loss_fn(outs)
    arrays["loss"] = loss.detach().numpy()
    model.zero_grad()
    loss.backward()
    for k, p in model.named_parameters():
        arrays[f"grad.{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            arrays[f"sd1.{k}"] = v.detach().clone().numpy()
    model.eval()
    with torch.no_grad():
        outs = call(model)
    outs_t = [outs[n] for n in out_names] if isinstance(outs, dict) else list(outs)
    for n, t in zip(out_names, outs_t):
        arrays[f"eval.{n}"] = t.detach().numpy()
    # ---- the rest of the reference loop: (clip,) optimizer.step(), then the next iteration's forward -------------------
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    if clip is not None:
        arrays["gradnorm"] = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=clip).detach().numpy()
    opt.step()
    for k, p in model.named_parameters():
        arrays[f"adam1.{k}"] = p.detach().clone().numpy()
    model.train()
    outs = call(model)
    arrays["loss2"] = loss_fn(outs).detach().numpy()
    for k, v in model.state_dict().items():
        if "running_" in k or "num_batches" in k:
            arrays[f"sd2.{k}"] = v.detach().clone().numpy()
    if after_step2 is not None:
        for k, v in after_step2().items():
            arrays[k] = v.numpy()
    path = os.path.join(HERE, f"{case}.npz")
    np.savez_compressed(path, **arrays)
    print(f"{case}: {os.path.getsize(path) / 1e6:.2f} MB, loss={float(loss):.6f}")


def main():
    bce_dice = losses_mod.BCEDiceLoss(0.5, 0.5)
    tversky = losses_mod.TverskyLoss(0.5, 0.5)

    # ---- (i) Unet 2-D ------------------------------------------------------------------------
    for case, kw, hw, crit, crit_name in [
        ("unet2d_f4", dict(in_channels=1, out_channels=1, n_filter=4, dilation=1), (32, 48), bce_dice, "BCEDice(0.5,0.5)"),
        ("unet2d_f4_o2_dil2", dict(in_channels=2, out_channels=2, n_filter=4, dilation=2), (32, 32), tversky, "Tversky(0.5,0.5)"),
    ]:
        torch.manual_seed(0)
        m = unet_mod.Unet(**kw)
        m.apply(ref_init_weights)
        x = torch.rand(2, kw["in_channels"], *hw)
        y = (torch.rand(2, kw["out_channels"], *hw) > 0.5).float()
        oc = kw["out_channels"]
        # the reference Trainer's loss expression, verbatim semantics of unet/train.py:133-134
        loss_fn = lambda outs, y=y, oc=oc, crit=crit: sum(
            crit(outs[1][ch], y[ch]) * torch.ones(oc)[j] for j, ch in enumerate(range(oc))) / sum(torch.ones(oc))
        dump(case, dict(model="Unet", ctor=kw, seed=0, loss=f"unet/train.py:133-134 with {crit_name}", init="init_weights"),
             m, {"x": x}, y, loss_fn, ["prob", "logits"], lambda mod, x=x: mod(x))

    # ---- (i-b) the other bio_image_unet.unet networks: attention-gated, legacy v0, three-level "baby" --------------
    torch.manual_seed(4)
    kw = dict(in_channels=2, out_channels=2, n_filter=4, dilation=1)
    m = att_mod.AttentionUnet(**kw)
    m.apply(ref_init_weights)
    x = torch.rand(2, 2, 32, 48)
    y = (torch.rand(2, 2, 32, 48) > 0.5).float()
    loss_fn = lambda outs, y=y: sum(bce_dice(outs[1][ch], y[ch]) * torch.ones(2)[j] for j, ch in enumerate(range(2))) / sum(torch.ones(2))
    dump("attention_f4", dict(model="AttentionUnet", ctor=kw, seed=4, loss="unet/train.py:133-134 with BCEDice(0.5,0.5)", init="init_weights"),
         m, {"x": x}, y, loss_fn, ["prob", "logits"], lambda mod, x=x: mod(x))
    for case, cls, kw, hw in (("unet_v0_f4", v0_mod.Unet_v0, dict(n_filter=4), (32, 48)), ("baby_f4", baby_mod.BabyUnet, dict(n_filter=4), (24, 40))):
        torch.manual_seed(5)
        m = cls(**kw)
        m.apply(ref_init_weights)
        x = torch.rand(2, 1, *hw)
        y = (torch.rand(2, 1, *hw) > 0.5).float()
        # Dropout2d(0.5) behind middle_conv2 draws from torch's RNG: record the factor it applied in the train-mode forward
        seen = {}
        def hook(mod, inp, out, seen=seen):
            if mod.training:
                dead = (inp[0] == 0).flatten(2).all(2)
                zeroed = (out == 0).flatten(2).all(2)
                seen["f"] = torch.where(zeroed & ~dead, torch.zeros(()), torch.full((), 1.0 / (1.0 - mod.p)))
        h = m.middle_conv2[3].register_forward_hook(hook)
        loss_fn = lambda outs, y=y: bce_dice(outs[1][0], y[0])          # unet/train.py:133-134 with out_channels = 1
        class _In(dict):
            pass
        ins = _In(x=x)
        def call(mod, x=x):
            return mod(x)
        # the factor is only known after the forward: dump() stores inputs first, so run one throw-away forward with the same seed state
        st = torch.get_rng_state()
        m.train()
        with torch.no_grad():
            sd_keep = {k: v.clone() for k, v in m.state_dict().items()}
            m(x)
            m.load_state_dict(sd_keep)
        torch.set_rng_state(st)
        ins["dropout_factor"] = seen["f"].clone()
        first = {}
        def call(mod, x=x, first=first):                       # noqa: F811  (remembers the draw of the FIRST train-mode forward)
            out = mod(x)
            if mod.training and "f" not in first:
                first["f"] = seen["f"].clone()
            return out
        dump(case, dict(model=cls.__name__, ctor=kw, seed=5, loss="unet/train.py:133-134 with BCEDice(0.5,0.5), out_channels=1", init="init_weights"),
             m, ins, y, loss_fn, ["prob", "logits"], call, after_step2=lambda: {"in.dropout_factor2": seen["f"].clone()})
        assert torch.equal(first["f"], ins["dropout_factor"]), "the recorded Dropout2d draw must be the one of the dumped forward"
        h.remove()

    # ---- (ii) UNet3D ---------------------------------------------------------------------------
    smooth_l1 = torch.nn.SmoothL1Loss()
    for case, kw in [("unet3d_f4", dict(in_channels=1, out_channels=1, n_filter=4, use_interpolation=False)),
                     ("unet3d_f4_interp", dict(in_channels=1, out_channels=1, n_filter=4, use_interpolation=True))]:
        torch.manual_seed(1)
        m = unet3d_mod.UNet3D(**kw)
        x = torch.rand(2, 1, 8, 16, 24)
        y = (torch.rand(2, 1, 8, 16, 24) > 0.5).float()
        # unet3d/train.py:140-145
        loss_fn = lambda outs, y=y: bce_dice(outs[1], y) + smooth_l1(outs[1][1:, :, :], outs[1][:-1, :, :]) * 0.1
        dump(case, dict(model="UNet3D", ctor=kw, seed=1, loss="unet3d/train.py:140-145 BCEDice + 0.1*SmoothL1", init="default"),
             m, {"x": x}, y, loss_fn, ["prob", "logits"], lambda mod, x=x: mod(x))

    # ---- (iii) Siam_UNet -----------------------------------------------------------------------
    siam_bce_dice = siam_losses_mod.BCEDiceLoss(1, 1)          # siam_unet/train.py: loss_function='BCEDice', loss_params=(1, 1)
    for mode in ("concat", "max", "corr", "control"):
        torch.manual_seed(2)
        m = siam_mod.Siam_UNet(n_filter=4, mode=mode)
        x = torch.rand(2, 1, 32, 32)
        px = torch.rand(2, 1, 32, 32)
        y = (torch.rand(2, 1, 32, 32) > 0.5).float()
        loss_fn = lambda outs, y=y: siam_bce_dice(outs[1], y)     # criterion(y_logits, y_i), siam_unet/train.py:110
        dump(f"siam_f4_{mode}", dict(model="Siam_UNet", ctor=dict(n_filter=4, mode=mode), seed=2,
                                     loss="siam_unet/losses.py BCEDiceLoss(1,1): BCELoss(sigmoid(logits)) + SoftDice", init="default"),
             m, {"x": x, "prev_x": px}, y, loss_fn, ["prob", "logits"], lambda mod, x=x, px=px: mod(x, px))

    # ---- (iv) MultiOutputUnet3D ----------------------------------------------------------------
    heads = {"seg": {"channels": 1, "activation": "sigmoid"},
             "flow": {"channels": 2, "activation": None},
             "dist": {"channels": 1, "activation": "tanh"}}
    for case, interp in (("mo3d_f4_interp", True), ("mo3d_f4_convT", False)):
        torch.manual_seed(3)
        kw = dict(in_channels=1, output_heads=heads, n_filter=4, use_interpolation=interp)
        m = mo3d_mod.MultiOutputUnet3D(**kw)
        x = torch.rand(2, 1, 8, 16, 16)
        tgt = {"seg": (torch.rand(2, 1, 8, 16, 16) > 0.5).float(), "flow": torch.randn(2, 2, 8, 16, 16),
               "dist": torch.rand(2, 1, 8, 16, 16)}
        # simple weighted sum of MSE on the activated outputs (the mo3d trainer's loss menu is out of the hot path)
        loss_fn = lambda outs, tgt=tgt: sum(((outs[k] - tgt[k]) ** 2).mean() * w
                                            for k, w in (("seg", 1.0), ("flow", 0.5), ("dist", 0.25)))
        dump(case, dict(model="MultiOutputUnet3D", ctor=kw, seed=3, loss="sum_k w_k*MSE(out_k, tgt_k), w=(1,.5,.25)", init="default"),
             m, {"x": x}, tgt, loss_fn, ["seg", "flow", "dist"], lambda mod, x=x: mod(x))

    # ---- (v) the mo3d TRAINER's step: per-head criteria of multi_output_unet3d/losses.py on the activated outputs, weighted
    #      sum, clip_grad_norm_(1.0), Adam  (multi_output_unet3d/train.py:149-162 loss menu, :183-201 loop) -------------------
    menu = {"BCEDiceLoss": lambda: mo3d_losses_mod.BCEDiceLoss(1, 1), "DiceLoss": lambda: mo3d_losses_mod.BCEDiceLoss(0, 1),
            "TverskyLoss": mo3d_losses_mod.TverskyLoss, "logcoshTverskyLoss": mo3d_losses_mod.logcoshTverskyLoss,
            "BCEDiceTemporalLoss": mo3d_losses_mod.BCEDiceTemporalLoss}          # Trainer._get_loss_function, restated (:149-162)
    for case, interp, heads_t in (
        ("mo3d_f4_trainer_convT", False, {"mask": {"channels": 1, "activation": "sigmoid", "loss": "BCEDiceLoss", "weight": 1.0},
                                          "flow": {"channels": 2, "activation": "tanh", "loss": "DiceLoss", "weight": 0.5},
                                          "edge": {"channels": 1, "activation": None, "loss": "BCEDiceTemporalLoss", "weight": 0.25}}),
        ("mo3d_f4_trainer_interp", True, {"mask": {"channels": 1, "activation": "sigmoid", "loss": "TverskyLoss", "weight": 1.0},
                                          "dist": {"channels": 1, "activation": "relu", "loss": "logcoshTverskyLoss", "weight": 0.5},
                                          "seg2": {"channels": 2, "activation": "sigmoid", "loss": "BCEDiceLoss"}}),
    ):
        torch.manual_seed(6)
        kw = dict(in_channels=1, output_heads=heads_t, n_filter=4, use_interpolation=interp)
        m = mo3d_mod.MultiOutputUnet3D(**kw)
        x = torch.rand(2, 1, 8, 16, 16)
        tgt = {k: (torch.rand(2, v["channels"], 8, 16, 16) > 0.5).float() for k, v in heads_t.items()}
        crit = {k: menu[v["loss"]]() for k, v in heads_t.items()}
        wts = {k: v.get("weight", 1.0) for k, v in heads_t.items()}
        def loss_fn(outs, tgt=tgt, crit=crit, wts=wts, heads_t=heads_t):
            total = 0
            for name in heads_t:                                  # train.py:188-196
                total += wts[name] * crit[name](outs[name], tgt[name])
            return total
        dump(case, dict(model="MultiOutputUnet3D", ctor=kw, seed=6, init="default",
                        loss="multi_output_unet3d/train.py:183-201: sum_k weight_k * loss_k(activated out_k, tgt_k), clip_grad_norm_(1.0), Adam(1e-3)"),
             m, {"x": x}, tgt, loss_fn, list(heads_t), lambda mod, x=x: mod(x), clip=1.0)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
