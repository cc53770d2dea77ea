"""Alias of quantization.estimators under the reference's module name."""
from .estimators import *  # noqa: F401,F403
from .estimators import (RangeEstimatorBase, CurrentMinMaxEstimator, AllMinMaxEstimator,  # noqa: F401
                         RunningMinMaxEstimator, FP_MSE_Estimator, RangeEstimators, NoDataPassedError,
                         estimate_range_line_search, LineSearchEstimator, OptMethod)
