"""Mirror of the reference's `ops` package (ops/__init__.py:1): `from ops import ctc_loss_2d`."""
from .ctc_loss_2d import CTCLoss2DFunction, ctc_loss_2d  # noqa: F401
from . import ctc_2d  # noqa: F401,E402  (extension-level boundary: ops.ctc_2d.ctc_2d_csrc)
