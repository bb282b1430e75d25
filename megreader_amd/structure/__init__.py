"""GPU mirrors of the reference's evaluation-side structure classes (representers / measurers)."""
from .representers import CTCRepresenter, CTCRepresenter2D  # noqa: F401,E402
from .measurers import SequenceRecognitionMeasurer  # noqa: F401,E402
from .seg_detector_representer import SegDetectorRepresenter  # noqa: F401,E402


def member_from_config(obj, cmd=None):
    """The reference's Configurable instantiates nested `{'class': 'pkg.Name', ...}` dictionaries itself
    (concern/config.py:158-168 `create_member_from_config`); the mirrors are plain classes, so they resolve such an
    argument (e.g. `charset: ^charset` in the recognition YAMLs) the same way."""
    if isinstance(obj, dict) and 'class' in obj:
        import importlib
        package, name = obj['class'].rsplit('.', 1)
        return getattr(importlib.import_module(package), name)(**obj, cmd=cmd or {})
    return obj
