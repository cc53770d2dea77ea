from ..fp8 import QuantizerBase, FPQuantizer  # noqa: F401
from ..manager import AsymmetricUniformQuantizer, SymmetricUniformQuantizer  # noqa: F401
