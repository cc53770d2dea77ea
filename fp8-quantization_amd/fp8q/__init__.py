"""fp8q -- MI355X-native FP8 fake-quantization engine (HIP kernels behind a C ABI)."""
from ._lib import Fp8qError, lib, so_path  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["Fp8qError", "lib", "so_path", "ops"]
