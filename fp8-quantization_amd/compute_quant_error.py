#!/usr/bin/env python
"""SQNR of the FP8 formats (E5M2 ... E2M5) against INT8 on three input distributions -- the
reference's compute_quant_error.py (BASELINE config 1) on the MI355X engine.

For each distribution: draw `n_samples` values (numpy global RNG, seed 10 as in the reference), for
each format find the MSE-optimal clipping range by line search (one GPU pass per format), then
report the ANALYTIC expected quantization / dot-product error of the format's grid on that range.
"""
import argparse
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from quantization.distributions import ClippedGaussDistr, UniformDistr, ClippedStudentTDistr  # noqa: E402
from quantization.quant_error import compute_expected_quant_mse, compute_expected_dot_prod_mse  # noqa: E402
from quantization.quantizers.fp8_quantizer import FPQuantizer  # noqa: E402
from quantization.quantizers.uniform_quantizers import SymmetricUniformQuantizer  # noqa: E402
from quantization.range_estimators import estimate_range_line_search  # noqa: E402


FORMATS = ((5, "E5M2"), (4, "E4M3"), (3, "E3M4"), (2, "E2M5"), (0, "INT8"))   # exponent bits; 0 = uniform grid


def seed_all(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def _db(mse):
    return -10.0 * np.log10(mse)


def _make_quantizer(exp_bits, n_bits):
    if exp_bits == 0:
        return SymmetricUniformQuantizer(n_bits=n_bits)
    return FPQuantizer(n_bits=8, mantissa_bits=n_bits - 1 - exp_bits, set_maxval=True)


def format_row(row):
    """The two console lines the reference prints per format (compute_quant_error.py:47-58)."""
    e, m, _, q_mse, q_db, d_mse, d_db = row
    head = "FP8 {} E {} M Quantization: expected MSE {:.2e}".format(e, m, q_mse)
    return " ".join([head, " SQNR ", "{:.2e}\n".format(q_db), "Dot product:".rjust(23),
                     " expected MSE {:.2e}".format(d_mse), " SQNR ", "{:.2e}".format(d_db)])


def compute_quant_error(distr, n_bits=8, n_samples=5000000, seed=10, device="cuda", verbose=True):
    """One row per format: (exp_bits, mantissa_bits, range_max, quant_mse, quant_sqnr, dot_mse, dot_sqnr).
    The float64 samples live on the GPU; each format's clipping range comes from ONE pass of the float64 MSE-grid
    kernel over them (1000 candidates, the reference's float64 arithmetic: fp8q_mse_grid_f64), the expected errors
    are the analytic integrals over the format's grid."""
    seed_all(seed)
    samples = torch.as_tensor(distr.sample((n_samples,))).to(device=device)    # float64, as the reference (:19-20)
    table = []
    for exp_bits, _name in FORMATS:
        q = _make_quantizer(exp_bits, n_bits)
        lo, hi = estimate_range_line_search(samples, q)
        q_mse = compute_expected_quant_mse(distr, q, lo, hi, n_samples)
        d_mse = compute_expected_dot_prod_mse(distr, distr, q, q, lo, hi, lo, hi)
        table.append((exp_bits, n_bits - 1 - exp_bits, float(hi), q_mse, _db(q_mse), d_mse, _db(d_mse)))
        if verbose:
            print(format_row(table[-1]))
    return table


def default_distributions():
    spec = ((UniformDistr, {}, (-1.0, 1.0)), (ClippedGaussDistr, {"mu": 0.0, "sigma": 1.0}, (-10.0, 10.0)),
            (ClippedStudentTDistr, {"nu": 8.0}, (-100.0, 100.0)))
    return [cls(params_dict=params, range_min=lo, range_max=hi) for cls, params, (lo, hi) in spec]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--n-samples", type=int, default=5000000)
    ap.add_argument("--seed", type=int, default=10)
    ap.add_argument("--int8-float64", action="store_true",
                    help="integrate the INT8 grid in float64 throughout instead of the reference's mixed float32 / float64 "
                         "evaluation (its printed INT8 numbers carry up to 7.5 %% of float32 cancellation noise)")
    a = ap.parse_args(argv)
    if a.int8_float64:
        import quantization.quant_error as qe
        qe.INT_GRID_REFERENCE_PRECISION = False
    for d in default_distributions():
        print("*" * 80)
        d.print()
        compute_quant_error(d, n_samples=a.n_samples, seed=a.seed)


if __name__ == "__main__":
    main()
