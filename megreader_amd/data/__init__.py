"""On-device input pipeline."""
from .device_pipeline import DevicePipeline, Prefetcher  # noqa: F401
from .msgpack_records import UnpackMsgpackData, records_to_batch  # noqa: F401
