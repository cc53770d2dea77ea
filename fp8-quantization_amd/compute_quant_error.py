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


def seed_all(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def compute_quant_error(distr, n_bits=8, n_samples=5000000, seed=10, device="cuda", verbose=True):
    """Returns [(exp_bits, mantissa_bits, range_max, quant_mse, quant_sqnr, dot_mse, dot_sqnr)]."""
    seed_all(seed)
    sample = torch.tensor(distr.sample((n_samples,))).to(device, torch.float32)
    rows = []
    for exp_bits in [5, 4, 3, 2, 0]:
        mantissa_bits = n_bits - 1 - exp_bits
        quant = FPQuantizer(n_bits=8, mantissa_bits=mantissa_bits, set_maxval=True) if exp_bits > 0 \
            else SymmetricUniformQuantizer(n_bits=n_bits)
        rmin, rmax = estimate_range_line_search(sample, quant)
        mse = compute_expected_quant_mse(distr, quant, rmin, rmax, n_samples)
        dp = compute_expected_dot_prod_mse(distr, distr, quant, quant, rmin, rmax, rmin, rmax)
        sqnr, dp_sqnr = -10.0 * np.log10(mse), -10.0 * np.log10(dp)
        rows.append((exp_bits, mantissa_bits, float(rmax), mse, sqnr, dp, dp_sqnr))
        if verbose:
            print("FP8 {} E {} M Quantization: expected MSE {:.2e}".format(exp_bits, mantissa_bits, mse),
                  " SQNR ", "{:.2e}\n".format(sqnr), "Dot product:".rjust(23),
                  " expected MSE {:.2e}".format(dp), " SQNR ", "{:.2e}".format(dp_sqnr))
    return rows


def default_distributions():
    return [UniformDistr(range_min=-1.0, range_max=1.0, params_dict={}),
            ClippedGaussDistr(params_dict={"mu": 0.0, "sigma": 1.0}, range_min=-10.0, range_max=10.0),
            ClippedStudentTDistr(params_dict={"nu": 8.0}, range_min=-100.0, range_max=100.0)]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-samples", type=int, default=5000000)
    ap.add_argument("--seed", type=int, default=10)
    a = ap.parse_args()
    for d in default_distributions():
        print("*" * 80)
        d.print()
        compute_quant_error(d, n_samples=a.n_samples, seed=a.seed)
