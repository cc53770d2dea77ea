"""Only the straight-through round is on the PTQ path."""
from ..fp8 import round_ste_func  # noqa: F401
