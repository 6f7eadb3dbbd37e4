from ..losses import *          # noqa: F401,F403
from ..models import Siam_UNet  # noqa: F401
from ..workflow import TrainerSiam as Trainer   # noqa: F401
from ..workflow import PredictSiam as Predict   # noqa: F401
