"""``import bio_image_unet_amd.siam_unet as siam`` mirrors ``import bio_image_unet.siam_unet as siam`` for the hot path.
The Siam package has its own ``losses.py`` whose BCE works on probabilities (``siam_unet/losses.py:73-105``)."""
from ..losses import *          # noqa: F401,F403
from ..losses import BCEDiceLossSiam as BCEDiceLoss, BCELoss2dProb as BCELoss2d   # noqa: F401
from ..models import Siam_UNet  # noqa: F401
from ..workflow import TrainerSiam as Trainer   # noqa: F401
from ..workflow import PredictSiam as Predict   # noqa: F401
