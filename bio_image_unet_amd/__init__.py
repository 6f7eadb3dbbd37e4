"""bio_image_unet_amd -- MI355X-native drop-in for the U-Net forward/backward hot path of danihae/bio-image-unet.

``import bio_image_unet_amd.unet as unet`` mirrors ``import bio_image_unet.unet as unet`` for the model classes
(and the Trainer / Predict counterparts that drive them).  All arithmetic runs in hand-written HIP kernels behind
the C ABI of ``include/biu.h``; importing the package fails if ``libbiu_hip.so`` has not been built.
"""
from . import _lib                      # noqa: F401  (raises ImportError when the HIP library is missing)
from .models import AttentionUnet, BabyUnet, MultiOutputUnet3D, Siam_UNet, UNet3D, Unet, Unet_v0  # noqa: F401

__version__ = "0.1.0"
