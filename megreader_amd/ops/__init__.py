"""Mirror of the reference's `ops` package (ops/__init__.py:1): `from ops import ctc_loss_2d`."""
from .ctc_loss_2d import CTCLoss2DFunction, ctc_loss_2d  # noqa: F401
