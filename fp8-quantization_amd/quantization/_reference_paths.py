"""The reference's import paths, served from ONE table instead of a file per path.

`from quantization.range_estimators import FP_MSE_Estimator`, `import quantization.quantizers.fp8_quantizer`, ... keep working
for code written against the reference (its own scripts, tests and pickles name these modules): a meta-path finder resolves
each reference module name to the implementation module that holds the same classes -- lazily, so that e.g. the analysis
side-car (`quant_error_estimator` -> `quant_error`) is only imported when somebody asks for it.  The module object returned IS
the implementation module: classes keep their real `__module__`, so pickles and `torch.save(model)` are unaffected.
(Reference files: quantization/hijacker.py, autoquant_utils.py, quantization_manager.py, range_estimators.py,
quantized_folded_bn.py, base_quantized_classes.py, base_quantized_model.py, quant_error_estimator.py, quantizers/*.py.)
"""
import importlib
import importlib.abc
import importlib.util
import sys
import types

_PKG = __name__.rsplit(".", 1)[0]           # "quantization"

# reference module -> implementation module (relative to this package)
ALIASES = {
    "hijacker": "layers",                                  # QuantizationHijacker, activations_set
    "autoquant_utils": "layers",                           # Quant* / BNQ* layers, quantize_model, fold_bn ...
    "quantized_folded_bn": "layers",                       # BNFusedHijacker
    "base_quantized_classes": "layers",                    # QuantizedModule, QuantizedActivation, FP32Acts
    "base_quantized_model": "model",                       # QuantizedModel (+ range checkpointing, graphs, FP8 export)
    "quantization_manager": "manager",                     # QuantizationManager, Qstates, QMethods
    "range_estimators": "estimators",                      # the estimators, RangeEstimators, LineSearchEstimator
    "quant_error_estimator": "quant_error",                # compute_expected_quant_mse ...
    "quantizers.base_quantizers": "fp8",                   # QuantizerBase
    "quantizers.fp8_quantizer": "fp8",                     # FPQuantizer, quantize_to_fp8_ste_MM, grid enumerators
    "quantizers.rounding_utils": "fp8",                    # round_ste_func (the only rounding on the PTQ path)
    "quantizers.utils": "fp8",                             # QuantizerNotInitializedError
    "quantizers.uniform_quantizers": "uniform",            # the INT8 comparison quantizers
}
# the reference's `quantization.quantizers` package itself: the names its __init__ exports
_QUANTIZERS_EXPORTS = {"fp8": ("QuantizerBase", "FPQuantizer"), "uniform": ("AsymmetricUniformQuantizer", "SymmetricUniformQuantizer")}


class _ReferencePathFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PKG + "."):
            return None
        rel = fullname[len(_PKG) + 1:]
        if rel == "quantizers":
            return importlib.util.spec_from_loader(fullname, self, is_package=True)
        if rel in ALIASES:
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        rel = spec.name[len(_PKG) + 1:]
        if rel == "quantizers":
            m = types.ModuleType(spec.name, "The reference's quantizer package: names served from quantization.fp8 / .uniform.")
            m.__path__ = []
            for mod, names in _QUANTIZERS_EXPORTS.items():
                impl = importlib.import_module(f"{_PKG}.{mod}")
                for n in names:
                    setattr(m, n, getattr(impl, n))
            return m
        return importlib.import_module(f"{_PKG}.{ALIASES[rel]}")     # the implementation module itself

    def exec_module(self, module):
        pass


def install():
    if not any(isinstance(f, _ReferencePathFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _ReferencePathFinder())
