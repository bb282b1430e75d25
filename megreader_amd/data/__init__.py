"""On-device input pipeline."""
from .device_pipeline import DevicePipeline, Prefetcher  # noqa: F401
