"""Host utilities on the hot path: device selection and weight initialisation.

Mirrors ``bio_image_unet/utils/utils.py``: ``get_device`` (:56-73) and ``init_weights`` (:76-78).
"""
import torch
from torch import nn


def get_device(print_device: bool = False) -> torch.device:
    """Reference rule: ``cuda:0`` whenever torch was *built* with CUDA/ROCm (even with no GPU visible), else mps,
    else cpu (with a warning).  This package only executes on the GPU, so the cpu answer leads to a loud failure at
    the first forward."""
    if torch.backends.cuda.is_built():
        device = torch.device("cuda:0")
    elif torch.backends.mps.is_built():
        device = torch.device("mps")
    else:
        device = torch.device("cpu")
        print("Warning: No CUDA or MPS device found. Calculations will run on the CPU, which might be slower.")
    if print_device:
        print(f"Using device: {device}")
    return device


def init_weights(m: nn.Module) -> None:
    """Kaiming-normal (fan_in, gain sqrt(2)) on ``nn.Conv2d`` weights ONLY -- Conv3d, ConvTranspose and BatchNorm keep
    PyTorch's defaults, exactly as in the reference."""
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
