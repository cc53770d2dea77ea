"""Analytic expected quantization error of a grid under a clipped distribution (BASELINE config 1).

Restates /root/reference/utils/grid.py:46-93 and quantization/quant_error_estimator.py:40-161:
every real number rounds to the nearest grid point, so the expected squared error is a sum over
half-cells of int p(x) (x - g)^2 dx, plus the point masses that clipping puts on the range ends.
CPU / float64; the quantizer itself is only used for the empirical cross-check (on the GPU, float64 lane).
"""
import numpy as np
import torch

from .fp8 import FPQuantizer, generate_all_float_values_scaled


def quant_scalar_nearest(x, grid):
    return grid[int(np.argmin(np.abs(x - grid)))]


# A float32 grid (what an INT quantizer's generate_grid() hands over) is integrated in the reference's own mixed precision,
# operation for operation (quantization/refprec.py): that is what the reference prints for INT8.  False: float64 throughout.
INT_GRID_REFERENCE_PRECISION = True


def integrate_pdf_grid_func_analyt(distr, grid, distr_attr_func_name):
    """sum over the Voronoi cells of the grid of distr.<func>(cell_lo, cell_hi, grid_point)."""
    if INT_GRID_REFERENCE_PRECISION and isinstance(grid, np.ndarray) and grid.dtype == np.float32:
        from .refprec import integrate_float32_grid
        return integrate_float32_grid(distr, grid, distr_attr_func_name)
    grid = np.sort(np.asarray(grid, dtype=np.float64))
    f = getattr(distr, distr_attr_func_name)
    lo, hi = distr.range_min, distr.range_max
    total = 0.0
    if lo < grid[0]:
        total += f(lo, grid[0], grid[0])
    for g0, g1 in zip(grid[:-1], grid[1:]):
        mid = 0.5 * (g0 + g1)
        a, b = max(g0, lo), min(mid, hi)        # left half-cell rounds down to g0
        if a < b:
            total += f(a, b, g0)
        a, b = max(mid, lo), min(g1, hi)        # right half-cell rounds up to g1
        if a < b:
            total += f(a, b, g1)
    if hi > grid[-1]:
        total += f(grid[-1], hi, grid[-1])
    # clipped distributions carry probability mass exactly at their range ends
    pm_lo, pm_hi = getattr(distr, "point_mass_range_min", 0.0), getattr(distr, "point_mass_range_max", 0.0)
    if pm_lo or pm_hi:
        q_lo, q_hi = quant_scalar_nearest(lo, grid), quant_scalar_nearest(hi, grid)
        if distr_attr_func_name == "integr_interv_p_sqr_r":
            total += (q_lo - lo) ** 2 * pm_lo + (q_hi - hi) ** 2 * pm_hi
        elif distr_attr_func_name == "integr_interv_x_p_signed_r":
            total += lo * (q_lo - lo) * pm_lo + hi * (q_hi - hi) * pm_hi
    return total


def _grid_of(quant, range_max):
    if isinstance(quant, FPQuantizer):
        return generate_all_float_values_scaled(quant.n_bits, quant.ebits, quant.default_bias, float(range_max))
    return quant.generate_grid().cpu().numpy()


def estimate_rounding_error_analyt(distr, grid):
    return integrate_pdf_grid_func_analyt(distr, grid, "integr_interv_p_sqr_r")


def estimate_dot_prod_error_analyt(distr_x, grid_x, distr_y, grid_y):
    """E[(xy - Q(x)Q(y))^2] for independent x, y, expanded in rounding-error moments (:40-64)."""
    ex = integrate_pdf_grid_func_analyt(distr_x, grid_x, "integr_interv_p_sqr_r")
    ey = integrate_pdf_grid_func_analyt(distr_y, grid_y, "integr_interv_p_sqr_r")
    sx = integrate_pdf_grid_func_analyt(distr_x, grid_x, "integr_interv_x_p_signed_r")
    sy = integrate_pdf_grid_func_analyt(distr_y, grid_y, "integr_interv_x_p_signed_r")
    mx, my = distr_x.eval_non_central_second_moment(), distr_y.eval_non_central_second_moment()
    return ex * my + ey * mx + 2.0 * sy * sx + ex * ey + 2.0 * ex * sy + 2.0 * ey * sx


def estimate_rounding_error_empirical(W, quantizer, range_min, range_max):
    quantizer.set_quant_range(range_min, range_max)
    return torch.mean(((quantizer.forward(W) - W) ** 2).flatten()).item()


def _device_sample(distr, n):
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    return torch.tensor(distr.sample((n,))).to(dev)     # float64, as the reference's empirical check (:146-151)


def compute_expected_quant_mse(distr, quant, quant_range_min, quant_range_max, num_samples):
    quant.set_quant_range(quant_range_min, quant_range_max)
    err_analyt = estimate_rounding_error_analyt(distr, _grid_of(quant, quant_range_max))
    err_emp = estimate_rounding_error_empirical(_device_sample(distr, num_samples), quant, quant_range_min,
                                                quant_range_max)
    if abs((err_emp - err_analyt) / err_analyt) > 0.1:
        print("Warning: the relative difference between the analytical and empirical error estimate is too high,\n"
              "please consider increasing the number of samples for the quantization range estimator.")
    return err_analyt


def compute_expected_dot_prod_mse(distr_x, distr_y, quant_x, quant_y, quant_x_range_min, quant_x_range_max,
                                  quant_y_range_min, quant_y_range_max, num_samples=2000000):
    quant_x.set_quant_range(quant_x_range_min, quant_x_range_max)
    quant_y.set_quant_range(quant_y_range_min, quant_y_range_max)
    grid_x = _grid_of(quant_x, quant_x_range_max)
    # the reference builds the second INT grid from quant_x as well (quant_error_estimator.py:117)
    grid_y = _grid_of(quant_y, quant_y_range_max) if isinstance(quant_y, FPQuantizer) else _grid_of(quant_x, quant_x_range_max)
    return estimate_dot_prod_error_analyt(distr_x, grid_x, distr_y, grid_y)
