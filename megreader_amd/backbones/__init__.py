"""Mirror of reference backbones/__init__.py:1-4 (factories resolved by name from the YAML configs)."""
from .crnn import crnn_backbone  # noqa: F401
