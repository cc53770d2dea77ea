"""The INT8 rows of BASELINE config 1 as the reference PRINTS them.

The reference hands its analytic integrator the INT grid as a float32 numpy array
(`quant.generate_grid().numpy()`, /root/reference/quantization/quant_error_estimator.py:141-143,
uniform_quantizers.py:328-331), so every grid point that enters its closed forms
(utils/distributions.py:97-160, 262-300, 362-383; utils/grid.py:46-93) is a numpy float32 scalar and, under numpy's
promotion rules, whole sub-expressions are evaluated in float32: on cells 0.008 wide the antiderivative differences cancel
catastrophically and the printed INT8 numbers carry +7.5 % (uniform), -0.14 % (Gauss), +1.2 % (Student-t) of rounding noise
against the float64 evaluation of the same integrals (`quantization/distributions.py`).

That noise is a function of the exact order of the floating-point operations, so "identical to the reference on the same
inputs" can only be had by evaluating the same expression trees with the same operand types.  This module does that and
nothing else: for each distribution the two interval integrals, written through one edge term per interval end where the
reference repeats it, with the association of every sum and product kept (constants made by numpy functions are float64,
literals and parameters are python floats -- both matter for the promotion).  It is used only when a grid of dtype float32
reaches the integrator (the INT quantizers: the comparison baseline of config 1, not a kernel target); float64 grids -- every
FP8 format -- take the float64 moments of `distributions.py`.  `compute_quant_error.py --int8-float64` switches it off.
"""
import numpy as np
from scipy import special

_RH = np.sqrt(0.5)              # numpy float64 scalars, as in the reference
_RHP = np.sqrt(0.5 * np.pi)


def _gauss_edge(z, u, mu, sigma):
    """the bracket that utils/distributions.py:104-125 evaluates once per interval end (z = a, then z = b)"""
    poly = -_RHP * mu**2 - _RHP * sigma**2 + 2.0 * _RHP * mu * u - _RHP * u**2
    return (np.exp((-0.5 * z**2 + 1.0 * z * mu - 0.5 * mu**2) / sigma**2) * sigma * (-1.0 * z - 1.0 * mu + 2.0 * u)
            + poly * special.erf((-_RH * z + _RH * mu) / sigma))


def gauss_p_sqr_r(d, a, b, u):
    mu, sigma = d.params_dict["mu"], d.params_dict["sigma"]
    t1 = -sigma * _gauss_edge(a, u, mu, sigma)
    t2 = sigma * _gauss_edge(b, u, mu, sigma)
    return (t1 + t2) * (1 / sigma / np.sqrt(2 * np.pi))


def _gauss_tail(z, mu, sigma):
    """utils/distributions.py:148-158: the x^2 part, once per interval end"""
    return (np.exp((-0.5 * z**2 + z * mu - 0.5 * mu**2) / sigma**2) * (-z * sigma - mu * sigma)
            + (-_RHP * mu**2 - _RHP * sigma**2) * special.erf((-_RH * z + _RH * mu) / sigma))


def gauss_x_p_signed_r(d, a, b, x0):
    mu, sigma = d.params_dict["mu"], d.params_dict["sigma"]
    first = (x0 * sigma
             * (np.exp(-((0.5 * mu**2) / sigma**2))
                * (np.exp((a * (-0.5 * a + mu)) / sigma**2) - np.exp((b * (-0.5 * b + mu)) / sigma**2)) * sigma
                - _RHP * mu * special.erf((_RH * a - _RH * mu) / sigma)
                + _RHP * mu * special.erf((_RH * b - _RH * mu) / sigma)))
    res = first + sigma * _gauss_tail(a, mu, sigma) - sigma * _gauss_tail(b, mu, sigma)
    return res * (1 / sigma / np.sqrt(2 * np.pi))


def _student_const(nu):
    return special.gamma(0.5 * (nu + 1.0)) / np.sqrt(np.pi * nu) / special.gamma(0.5 * nu)


def student_p_sqr_r(d, a, b, u):
    """utils/distributions.py:262-300: six terms, summed left to right"""
    nu = d.params_dict["nu"]
    terms = (
        (2.0 * nu * (-1.0 + ((a**2 + nu) / nu) ** (1.0 / 2.0 - nu / 2.0)) * u) / (1.0 - nu),
        -(2 * nu * (-1 + ((b**2 + nu) / nu) ** (1.0 / 2.0 - nu / 2)) * u) / (1.0 - nu),
        -a * u**2 * special.hyp2f1(1.0 / 2.0, (1.0 + nu) / 2.0, 3.0 / 2.0, -(a**2.0 / nu)),
        b * u**2.0 * special.hyp2f1(1.0 / 2.0, (1.0 + nu) / 2.0, 3.0 / 2.0, -(b**2 / nu)),
        -1.0 / 3.0 * a**3 * special.hyp2f1(3.0 / 2.0, (1.0 + nu) / 2.0, 5.0 / 2.0, -(a**2 / nu)),
        1.0 / 3.0 * b**3 * special.hyp2f1(3.0 / 2.0, (1.0 + nu) / 2.0, 5.0 / 2.0, -(b**2 / nu)),
    )
    res = terms[0]
    for t in terms[1:]:
        res = res + t
    return res * _student_const(nu)


def student_x_p_signed_r(d, a, b, x0):
    """utils/distributions.py:345-369"""
    df = d.params_dict["nu"]
    r1 = ((df ** ((1.0 + df) / 2.0) * (-((a**2 + df) ** (1.0 / 2.0 - df / 2.0)) + (b**2 + df) ** (1.0 / 2.0 - df / 2.0)) * x0)
          / (1.0 - df)) * _student_const(df)
    return r1 - student_p_sqr_r(d, a, b, 0.0)


def uniform_p_sqr_r(d, a, b, u):
    """utils/distributions.py:362-365"""
    res = -(a**3 / 3.0) + b**3 / 3.0 + a**2 * u - b**2 * u - a * u**2 + b * u**2
    return res * d.p


def uniform_x_p_signed_r(d, a, b, x0):
    """utils/distributions.py:380-383 (without the factor x of the other two distributions: a reproduced quirk)"""
    res = 0.5 * a**2 - 0.5 * b**2 + (b - a) * x0
    return res * d.p


FORMS = {
    "UniformDistr": {"integr_interv_p_sqr_r": uniform_p_sqr_r, "integr_interv_x_p_signed_r": uniform_x_p_signed_r},
    "ClippedGaussDistr": {"integr_interv_p_sqr_r": gauss_p_sqr_r, "integr_interv_x_p_signed_r": gauss_x_p_signed_r},
    "ClippedStudentTDistr": {"integr_interv_p_sqr_r": student_p_sqr_r, "integr_interv_x_p_signed_r": student_x_p_signed_r},
}


def integrate_float32_grid(distr, grid, func_name):
    """utils/grid.py:46-93 on a float32 grid: half-cells around every grid point, the point masses of a clipped
    distribution at its range ends; grid points, midpoints and everything derived from them stay numpy float32 scalars."""
    assert grid.dtype == np.float32
    grid = np.sort(grid)
    f = FORMS[type(distr).__name__][func_name]
    lo, hi = distr.range_min, distr.range_max
    res = 0.0
    if lo < grid[0]:
        res += f(distr, lo, grid[0], grid[0])
    for i in range(len(grid) - 1):
        mid = 0.5 * (grid[i] + grid[i + 1])
        a1, b1 = max(grid[i], lo), min(mid, hi)
        a2, b2 = max(mid, lo), min(grid[i + 1], hi)
        if a1 < b1:
            res += f(distr, a1, b1, grid[i])
        if a2 < b2:
            res += f(distr, a2, b2, grid[i + 1])
    if hi > grid[-1]:
        res += f(distr, grid[-1], hi, grid[-1])
    pm_lo, pm_hi = getattr(distr, "point_mass_range_min", 0.0), getattr(distr, "point_mass_range_max", 0.0)
    if type(distr).__name__ != "UniformDistr":
        # the nearest grid point of a range end: found in float32, as torch.Tensor([end]) - grid is (utils/grid.py:22-27)
        q_lo = grid[int(np.argmin(np.abs(np.float32(lo) - grid)))]
        q_hi = grid[int(np.argmin(np.abs(np.float32(hi) - grid)))]
        if func_name == "integr_interv_x_p_signed_r":
            res += lo * (q_lo - lo) * pm_lo + hi * (q_hi - hi) * pm_hi
        else:
            res += (q_lo - lo) ** 2 * pm_lo + (q_hi - hi) ** 2 * pm_hi
    return res
