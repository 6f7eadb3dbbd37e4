from ..models import UNet3D  # noqa: F401
