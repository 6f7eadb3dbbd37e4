"""``import bio_image_unet_amd.unet as unet`` mirrors ``import bio_image_unet.unet as unet`` for the hot path."""
from ..losses import *          # noqa: F401,F403
from ..models import AttentionUnet, BabyUnet, Unet, Unet_v0       # noqa: F401
from ..workflow import Predict2D as Predict, Trainer2D as Trainer   # noqa: F401
