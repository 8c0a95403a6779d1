from stdiff_amd.process.data import DeviceLoader, get_data_loader  # noqa: F401  (reference: process/data.py:64-75)
from stdiff_amd.process.preprocess import normalize_log1p_scale, preprocess_pair  # noqa: F401  (process/data.py:79-100 on arrays)

__all__ = ["DeviceLoader", "get_data_loader", "normalize_log1p_scale", "preprocess_pair"]
