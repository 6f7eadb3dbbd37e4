"""The kernel variants a launch can take must agree with the kernels they replace: the 16-row MFMA kernel (BIU_DISABLE=m16 falls back to
the 32-row one), the paired-tap weight gradient of a 16-channel input (BIU_DISABLE=rr16), the rolling-window weight gradient (BIU_DISABLE=wroll), the one-launch folded weight gradient (BIU_DISABLE=foldall), the input-channel split of small fp32 launches (BIU_DISABLE=ksplit) and the opt-in bf16x3 products of the fp32 2-D kernels
(BIU_FP32_PRODUCTS=bf16x3) against the exact fp32 MFMA.  The switches are read once per process, so
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
    env.pop("BIU_FP32_PRODUCTS", None)
    if disable and disable.startswith("+"):          # an opt-in mode: the "off" side of the comparison switches it ON
        env["BIU_FP32_PRODUCTS"] = disable[1:]
    elif disable:
        env["BIU_DISABLE"] = disable
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "variant_probe.py"), which, out], check=True, env=env, timeout=300)
    return torch.load(out)


@pytest.mark.timeout(700)
@pytest.mark.parametrize("which,disable,tol_out,tol_grad", [
    # fp32: the split only re-associates the sum over input channels: outputs agree to 1e-5; a re-association flips a few LeakyReLU /
    # max-pool decisions, which moves single BatchNorm gradients by up to ~1e-2 even between two exact fp32 runs
    ("unet2d_f32", "ksplit", 1e-5, 2e-2),
    # opt-in: fp32 tensors multiplied as bf16x3 products (three bf16 MFMAs, <= 2^-15 per product) against the exact fp32 MFMA: 18 conv layers
    # deep the logits stay within 3e-4 (north-star tolerance for fp32: 1e-3); gradients as above
    ("unet2d_f32", "+bf16x3", 3e-4, 2e-2),
    # the default fp32 products (bf16x6: hi + mid + lo, six bf16 MFMA terms, <= 2^-23 per product) against the exact fp32 MFMA
    # (BIU_FP32_PRODUCTS=exact): 18 layers deep the logits agree like two fp32 summation orders do
    ("unet2d_f32", "+exact", 1e-5, 2e-2),
    # bf16: two correct bf16 kernels differ by output rounding; discrete LeakyReLU / max-pool decisions then move gradients by ~1 %
    #  -- a wrong tap or tile would move them by tens of percent
    ("unet3d_bf16", "m16,rr16", 2e-2, 6e-2),
    ("unet3d_bf16", "m16x2", 2e-2, 6e-2),                 # only the two-tile form of the 16-row kernel (32-channel tiles of single-chunk layers)
    # the rolling-window convolution with register-resident weights (biu_conv_roll.hip: the 16 <-> 32 and 32 -> 32 layers of the first level,
    # forward and data gradient) against the brick kernels it replaces (BIU_DISABLE=croll): same products, another summation order
    ("unet3d_bf16", "croll", 2e-2, 6e-2),
    # the rolling-window weight gradient (k_wgrad_roll) against the brick kernel it replaces (BIU_DISABLE=wroll): the forward is untouched
    # and the dy both write back over da is the same rounding sequence, so only the order of the fp32 sums inside dW differs
    ("unet3d_bf16", "wroll", 1e-6, 2.5e-4),
    ("unet3d_bf16", "wroll16", 1e-6, 2.5e-4),             # its form for a 16-channel plain operand (decode6) against the half-empty 32-row tile
    ("unet2d_bf16_n8", "wroll2d", 1e-6, 2.5e-4),            # its 2-D form (a batch of images as the depth axis)
    # nearest up-sampling + upN_conv forward folded onto the coarse tensor (biu_upconv_fwd) against up-sample + conv (BIU_DISABLE=upconv):
    # fp32: the same sums in another order; bf16: the folded weights are rounded after the fold, the unfolded ones tap by tap
    # ConvTranspose + concat + conv of the decoder levels as one folded op (biu_foldt_*) against the three separate ops (BIU_DISABLE=foldt)
    ("unet3d_f32", "foldt", 1e-5, 2e-2),
    ("unet3d_bf16", "foldt", 2e-2, 6e-2),
    # the folded weight gradient's G for all eight parity classes in one launch (k_wgrad_pipe<..., FALL>) against one launch per class
    # (BIU_DISABLE=foldall): the same operands, only the order of the fp32 sums over voxels differs
    ("unet3d_bf16", "foldall", 1e-6, 2.5e-4),
    # the weight-space products of the fold (composed weights, chain rule) on the fp32 matrix pipe (biu_fold_gemm.hip) against the scalar
    # kernels they replace (BIU_DISABLE=foldgemm): the same fp32 products in another summation order; the composed weights are then
    # rounded to bf16 operands, so an output can fall to the other side of a rounding boundary
    ("unet3d_f32", "foldgemm", 1e-5, 2e-2),
    ("unet3d_bf16", "foldgemm", 2e-2, 6e-2),
    # (the engine's side stream against the single-stream step: tests/test_gpu_engine_safety.py::test_side_stream_gradients_match_the_single_stream_step)
    # 64-channel chunks of the folded forward where the output is one 32-channel tile (decode5) against 32-channel chunks (BIU_DISABLE=foldck8):
    # the same products summed in the same order (chunk by chunk, tap by tap inside a chunk differs) -- outputs may round differently
    ("unet3d_bf16", "foldck8", 2e-2, 6e-2),
    ("mo3d_interp_f32", "upconv", 1e-5, 2e-2),
    # (this network is the sensitive one of DESIGN section 4 -- nearest down-sampling, 23 bf16 layers: swapping the 16-row kernels on the SAME
    # probe moves all gradients together by 0.206, the fold by 0.173; logits within 0.017-0.021 either way.  A wrong tap moves them by ~1)
    ("mo3d_interp_bf16", "upconv", 3e-2, 0.25),
])
def test_variant_matches_the_kernel_it_replaces(which, disable, tol_out, tol_grad, tmp_path):
    on, off = _run(which, None, tmp_path), _run(which, disable, tmp_path)
    assert abs(on["loss"] - off["loss"]) <= tol_out * max(1.0, abs(off["loss"]))
    d = float((on["logits"] - off["logits"]).norm() / off["logits"].norm())
    assert d <= tol_out, f"logits differ by {d:.3e}"
    # all gradients together (the large weight tensors dominate) within tol_grad; a single tensor within 4 x that: the small BatchNorm
    # vectors at the bottleneck (a few dozen voxels per channel) move by ~10 % when ONE LeakyReLU / max-pool decision upstream falls the
    # other way, which any two kernels that round a bf16 output differently produce (the sharp per-kernel statement is
    # tests/test_gpu_insitu.py)
    keys = [k for k in on if k not in ("loss", "logits")]
    num = sum(float((on[k] - off[k]).double().pow(2).sum()) for k in keys)
    den = sum(float(off[k].double().pow(2).sum()) for k in keys)
    assert (num / den) ** 0.5 <= tol_grad, f"all gradients together differ by {(num / den) ** 0.5:.3e}"
    worst = ("", 0.0)
    for k in keys:
        r = float((on[k] - off[k]).norm() / (off[k].norm() + 1e-30))
        if r > worst[1]:
            worst = (k, r)
    assert worst[1] <= 4 * tol_grad, f"gradient of {worst[0]} differs by {worst[1]:.3e} between the variants"


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["bf16x3", "exact"])
def test_fp32_op_tests_hold_in_the_other_product_modes(mode):
    """Every fp32 op test of tests/test_gpu_ops.py (MFMA convolutions, data / weight gradients, two-source and BatchNorm-fused forms,
    compared with torch fp32 at rtol 1e-4) runs with the default bf16x6 products in the main test process; here once more with the opt-in
    bf16x3 products and with the exact fp32 MFMA -- one child process each, the mode is process-wide."""
    env = dict(os.environ)
    env.pop("BIU_DISABLE", None)
    env["BIU_FP32_PRODUCTS"] = mode
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-q", "-x", "-k", "f32", "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=550)
    assert r.returncode == 0, r.stdout[-3000:]
