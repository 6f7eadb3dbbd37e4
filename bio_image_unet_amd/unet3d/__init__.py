from ..losses import *          # noqa: F401,F403
from ..models import UNet3D     # noqa: F401
from ..workflow import Trainer3D as Trainer   # noqa: F401
from ..workflow import Predict3D as Predict   # noqa: F401
