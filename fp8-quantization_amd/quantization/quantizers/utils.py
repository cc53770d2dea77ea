from ..fp8 import QuantizerNotInitializedError  # noqa: F401
