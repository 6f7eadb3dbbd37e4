from ..models import MultiOutputUnet3D  # noqa: F401
