"""bxmi -- MI355X-native interval-intersection and basewise-bitset engine (host side)."""
from ._ffi import BxmiError, device_count, load  # noqa: F401

__all__ = ["BxmiError", "device_count", "load"]
