"""GPU mirrors of the reference's evaluation-side structure classes (representers / measurers)."""
from .seg_detector_representer import SegDetectorRepresenter  # noqa: F401,E402
