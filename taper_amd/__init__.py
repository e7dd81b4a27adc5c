"""taper_amd -- MI355X (gfx950) backend for taper's training hot path.

Layers (bottom-up):
  csrc/*.hip        hand-written HIP kernels + runtime  -> lib/libtaper_hip.so   (C ABI: include/taper_hip.h)
  csrc/host/*.cpp   C++ host: Tensor / Tape / nn / optim -> lib/libtaper_host.so  (C ABI: include/taper_host.h)
  api.py, hip.py    ctypes faces of the two ABIs (no arithmetic in Python)

Importing this package loads both libraries and fails loudly if they are
missing: there is no CPU or PyTorch fallback.
"""
from . import hip  # noqa: F401
from ._lib import TaperError, build_native  # noqa: F401
from . import dist  # noqa: F401
from .api import (  # noqa: F401
    SGD, Adam, AdamW, AdaptiveAvgPool2d, AvgPool2d, Communicator, Conv2d, Conv2dReLU, CosineAnnealingLR, DataLoader, Device,
    Dropout, ExponentialLR, Flatten, Linear, MaxPool2d, MNISTDataset, Module, ReduceLROnPlateau, ReLU, Sequential, Sigmoid,
    StepLR, Tape, Tensor, Trainer, accuracy, bce_loss, cross_entropy_loss, cross_entropy_loss_onehot, format_f32, log_softmax,
    mse_loss, one_hot, set_conv_chain, set_conv_chain_head, set_full_backward, softmax,
)

Layer = Module  # the north_star calls the trait nn::Layer
__all__ = [n for n in dir() if not n.startswith("_")]
