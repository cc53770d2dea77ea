from ..fp8 import QuantizerBase  # noqa: F401
