"""Layer library of the MI355X hot path: autograd Functions over the C ABI and nn.Module mirrors."""
from . import functional  # noqa: F401
from .modules import (Conv2d, BatchNorm2d, MaxPool2d, FusedReLU, Linear, LSTM, AdaptiveAvgPool2d,  # noqa: F401
                      Dropout2d)
