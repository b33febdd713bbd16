"""livecc_amd -- MI355X-native streaming Qwen2-VL / LiveCC forward+generate hot path.

Only the pieces the hot path needs: HIP kernels + C-ABI (`csrc/`), the ctypes
binding (`_lib`), the host-side mirror of the reference's generate()/processor/plugin interface.
"""
from .config import LiveCCConfig, get_config  # noqa: F401

__version__ = "0.1.0"
