"""Mirror of reference backbones/__init__.py:1-4 (factories resolved by name from the YAML configs)."""
from .crnn import crnn_backbone  # noqa: F401
from .resnet import resnet18, resnet34, resnet50, resnet101, resnet152, deformable_resnet50  # noqa: F401
from .resnet_ppm import resnet50dilated_ppm  # noqa: F401
from .resnet_fpn import Resnet18FPN, Resnet34FPN, Resnet50FPN, Resnet101FPN, Resnet152FPN  # noqa: F401
