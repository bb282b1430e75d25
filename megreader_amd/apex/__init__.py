"""`apex` namespace shim: the reference imports apex at structure/model.py:6 and backbones/resnet.py:4 and uses
exactly two symbols, apex.parallel.DistributedDataParallel and apex.parallel.SyncBatchNorm."""
from . import parallel  # noqa: F401
