from ..models import MultiOutputUnet3D  # noqa: F401
from ..workflow import PredictMo3d as Predict, TrainerMo3d as Trainer   # noqa: F401
