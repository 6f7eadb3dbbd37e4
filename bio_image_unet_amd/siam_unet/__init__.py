from ..models import Siam_UNet  # noqa: F401
