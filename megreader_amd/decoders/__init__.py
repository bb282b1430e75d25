"""Mirror of reference decoders/__init__.py:1-14 (the recognition heads on the hot path)."""
from .crnn import CRNNDecoder  # noqa: F401
from .ctc_decoder import CTCDecoder  # noqa: F401
from .ctc_decoder2d import CTCDecoder2D  # noqa: F401
from .attention_decoder import AttentionDecoder  # noqa: F401
from .seg_detector import SegDetector  # noqa: F401
from .seg_detector_loss import (BalanceCrossEntropyLoss, DiceLoss, L1BalanceCELoss, MaskL1Loss)  # noqa: F401
