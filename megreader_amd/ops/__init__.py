"""Mirror of the reference's `ops` package (ops/__init__.py:1): `from ops import ctc_loss_2d`."""
try:
    from .ctc_loss_2d import CTCLoss2DFunction, ctc_loss_2d  # noqa: F401
except ImportError:  # pragma: no cover - the 2D-CTC op lands after the CRNN path
    pass
