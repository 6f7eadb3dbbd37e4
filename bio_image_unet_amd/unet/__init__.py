from ..models import Unet  # noqa: F401
