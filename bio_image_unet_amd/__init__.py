"""bio_image_unet_amd -- MI355X-native drop-in for the U-Net forward/backward hot path of danihae/bio-image-unet.

``import bio_image_unet_amd.unet as unet`` mirrors ``import bio_image_unet.unet as unet`` for the model classes
(and the Trainer / Predict counterparts that drive them).  All arithmetic runs in hand-written HIP kernels behind
the C ABI of ``include/biu.h``; importing the package fails if ``libbiu_hip.so`` has not been built.
"""
from . import _lib                      # noqa: F401  (raises ImportError when the HIP library is missing)
from .models import AttentionUnet, BabyUnet, MultiOutputUnet3D, Siam_UNet, UNet3D, Unet, Unet_v0  # noqa: F401

__version__ = "0.1.0"


def set_fp32_products(mode: str) -> None:
    """How the fp32 2-D 3x3 convolution kernels multiply: ``"exact"`` (default; fp32 MFMA) or ``"bf16x3"`` (operands split hi + lo in bf16,
    three bf16 MFMAs per product, fp32 accumulation: <= 2^-15 relative per product, 2.5-3x the throughput).  The counterpart of
    ``torch.backends.cudnn.allow_tf32`` for the reference's fp32 trainers.  Process-wide, to be called before the first forward
    (``include/biu.h: biu_set_fp32_products``)."""
    if mode not in ("exact", "bf16x3"):
        raise ValueError(f"set_fp32_products: {mode!r} (expected 'exact' or 'bf16x3')")
    _lib.check(_lib.lib.biu_set_fp32_products(1 if mode == "bf16x3" else 0), "set_fp32_products")
