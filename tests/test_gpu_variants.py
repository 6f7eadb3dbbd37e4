"""The kernel variants a launch can take must agree with the kernels they replace: the 16-row MFMA kernel (BIU_DISABLE=m16 falls back to
the 32-row one), the paired-tap weight gradient of a 16-channel input (BIU_DISABLE=rr16) and the input-channel split of small fp32 launches (BIU_DISABLE=ksplit).  The switches are read once per process, so
each side runs in its own subprocess (tests/variant_probe.py)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(which, disable, tmp_path):
    out = str(tmp_path / f"{which}_{disable or 'on'}.pt")
    env = dict(os.environ)
    env.pop("BIU_DISABLE", None)
    if disable:
        env["BIU_DISABLE"] = disable
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_probe.py"), which, out], check=True, env=env, timeout=300)
    return torch.load(out)


@pytest.mark.timeout(700)
@pytest.mark.parametrize("which,disable,tol_out,tol_grad", [
    # fp32: the split only re-associates the sum over input channels: outputs agree to 1e-5; a re-association flips a few LeakyReLU /
    # max-pool decisions, which moves single BatchNorm gradients by up to ~1e-2 even between two exact fp32 runs
    ("unet2d_f32", "ksplit", 1e-5, 2e-2),
    # bf16: two correct bf16 kernels differ by output rounding; discrete LeakyReLU / max-pool decisions then move gradients by ~1 %
    #  -- a wrong tap or tile would move them by tens of percent
    ("unet3d_bf16", "m16,rr16", 2e-2, 6e-2),
])
def test_variant_matches_the_kernel_it_replaces(which, disable, tol_out, tol_grad, tmp_path):
    on, off = _run(which, None, tmp_path), _run(which, disable, tmp_path)
    assert abs(on["loss"] - off["loss"]) <= tol_out * max(1.0, abs(off["loss"]))
    d = float((on["logits"] - off["logits"]).norm() / off["logits"].norm())
    assert d <= tol_out, f"logits differ by {d:.3e}"
    worst = ("", 0.0)
    for k in on:
        if k in ("loss", "logits"):
            continue
        r = float((on[k] - off[k]).norm() / (off[k].norm() + 1e-30))
        if r > worst[1]:
            worst = (k, r)
    assert worst[1] <= tol_grad, f"gradient of {worst[0]} differs by {worst[1]:.3e} between the variants"
