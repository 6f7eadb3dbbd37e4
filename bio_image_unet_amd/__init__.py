"""bio_image_unet_amd -- MI355X-native drop-in for the U-Net forward/backward hot path of danihae/bio-image-unet.

``import bio_image_unet_amd.unet as unet`` mirrors ``import bio_image_unet.unet as unet`` for the model classes
(and the Trainer / Predict counterparts that drive them).  All arithmetic runs in hand-written HIP kernels behind
the C ABI of ``include/biu.h``; importing the package fails if ``libbiu_hip.so`` has not been built.
"""
from . import _lib                      # noqa: F401  (raises ImportError when the HIP library is missing)
from .models import AttentionUnet, BabyUnet, MultiOutputUnet3D, Siam_UNet, UNet3D, Unet, Unet_v0  # noqa: F401

__version__ = "0.1.0"


def set_fp32_products(mode: str) -> None:
    """How the fp32 2-D 3x3 convolution / ConvTranspose kernels multiply:

    * ``"bf16x6"`` (default): every fp32 operand is split hi + mid + lo in bf16 (24 significant bits) and a product is the six bf16 MFMA terms
      of order >= 2^-16, accumulated in fp32 -- <= 2^-23 relative per product, i.e. fp32-grade, at 2.7x the matrix-pipe rate of the fp32 MFMA;
    * ``"exact"``: ``v_mfma_f32_32x32x2_f32`` (fp32 FMA chains);
    * ``"bf16x3"`` (opt-in): hi + lo, three terms, <= 2^-15 relative per product, another 2x -- the counterpart of
      ``torch.backends.cudnn.allow_tf32`` for the reference's fp32 trainers.

    Process-wide, to be called before the first forward (``include/biu.h: biu_set_fp32_products``; environment: ``BIU_FP32_PRODUCTS``)."""
    modes = {"exact": 0, "bf16x3": 1, "bf16x6": 2}
    if mode not in modes:
        raise ValueError(f"set_fp32_products: {mode!r} (expected one of {sorted(modes)})")
    _lib.check(_lib.lib.biu_set_fp32_products(modes[mode]), "set_fp32_products")
