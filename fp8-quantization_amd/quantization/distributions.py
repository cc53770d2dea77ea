"""Clipped input distributions for the analytic quantization-error study (BASELINE config 1).

Same classes, constructor arguments and method names as /root/reference/utils/distributions.py
(UniformDistr, ClippedGaussDistr, ClippedStudentTDistr); the closed forms are written through the
three partial moments of the density on an interval,

    M0 = int_a^b p,   M1 = int_a^b x p,   M2 = int_a^b x^2 p,

from which  int p (x-u)^2 = M2 - 2u M1 + u^2 M0   and   int p x (x0 - x) = x0 M1 - M2.
Clipping puts point masses at range_min / range_max.  float64 / numpy / scipy: CPU side-car.
"""
import numpy as np
import scipy.stats as stats
from scipy import special


class DistrBase:
    def __init__(self, params_dict, range_min, range_max, *args, **kwargs):
        assert range_max >= range_min
        self.params_dict = params_dict
        self.range_min = range_min
        self.range_max = range_max
        self.point_mass_range_min = 0.0
        self.point_mass_range_max = 0.0

    def moments(self, a, b):
        """(M0, M1, M2) of the un-clipped density on [a, b]."""
        raise NotImplementedError()

    def sample(self, shape):
        raise NotImplementedError()

    def print(self):
        raise NotImplementedError()

    def integr_interv_p_sqr_r(self, a, b, u):
        """int_a^b p(x) (x - u)^2 dx: squared rounding error when [a, b] rounds to u."""
        assert b >= a
        m0, m1, m2 = self.moments(a, b)
        return m2 - 2.0 * u * m1 + u * u * m0

    def integr_interv_x_p_signed_r(self, a, b, x0):
        """int_a^b p(x) x (x0 - x) dx: signed rounding error weighted by x."""
        assert b >= a
        _, m1, m2 = self.moments(a, b)
        return x0 * m1 - m2

    def integr_p_times_x(self, a, b):
        assert b >= a
        return self.moments(a, b)[1]

    def eval_non_central_second_moment(self):
        return (self.point_mass_range_min * self.range_min ** 2 + self.point_mass_range_max * self.range_max ** 2
                + self.moments(self.range_min, self.range_max)[2])


class UniformDistr(DistrBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.p = 1.0 / (self.range_max - self.range_min)

    def print(self):
        print("Uniform distribution on [", self.range_min, ",", self.range_max, "]")

    def pdf(self, x):
        return self.p

    def cdf(self, x):
        return (x - self.range_min) * self.p

    def sample(self, shape):
        return np.random.uniform(self.range_min, self.range_max, shape)

    def moments(self, a, b):
        return (b - a) * self.p, 0.5 * (b * b - a * a) * self.p, (b ** 3 - a ** 3) / 3.0 * self.p

    def integr_interv_x_p_signed_r(self, a, b, x0):
        # Reference quirk (utils/distributions.py:380-383): for the uniform distribution this term
        # is int p (x0 - x) dx -- without the factor x the Gaussian / Student-t versions carry.
        # Reproduced, because the published dot-product SQNR numbers depend on it.
        assert b >= a
        m0, m1, _ = self.moments(a, b)
        return x0 * m0 - m1


class ClippedGaussDistr(DistrBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.mu, self.sigma = self.params_dict["mu"], self.params_dict["sigma"]
        self.point_mass_range_min = stats.norm.cdf(self.range_min, loc=self.mu, scale=self.sigma)
        self.point_mass_range_max = 1.0 - stats.norm.cdf(self.range_max, loc=self.mu, scale=self.sigma)

    def print(self):
        print("Gaussian distr ", ", mu = ", self.mu, ", sigma = ", self.sigma, " clipped at [",
              self.range_min, ",", self.range_max, "]")

    def pdf(self, x):
        return stats.norm.pdf(np.asarray(x), self.mu, self.sigma)

    def cdf(self, x):
        return stats.norm.cdf(x, self.mu, self.sigma)

    def inverse_cdf(self, x):
        return stats.norm.ppf(x, loc=self.mu, scale=self.sigma)

    def sample(self, shape):
        return np.clip(np.random.normal(loc=self.mu, scale=self.sigma, size=shape), self.range_min, self.range_max)

    def moments(self, a, b):
        mu, s = self.mu, self.sigma
        za, zb = (a - mu) / s, (b - mu) / s
        # Phi(zb) - Phi(za) through erf keeps the difference accurate in the tails
        m0 = 0.5 * (special.erf(zb / np.sqrt(2.0)) - special.erf(za / np.sqrt(2.0)))
        fa = np.exp(-0.5 * za * za) / np.sqrt(2.0 * np.pi)
        fb = np.exp(-0.5 * zb * zb) / np.sqrt(2.0 * np.pi)
        m1 = mu * m0 - s * (fb - fa)
        m2 = (mu * mu + s * s) * m0 - s * ((b + mu) * fb - (a + mu) * fa)
        return m0, m1, m2


class ClippedStudentTDistr(DistrBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.nu = self.params_dict["nu"]
        self.point_mass_range_min = stats.t.cdf(self.range_min, self.nu)
        self.point_mass_range_max = 1.0 - stats.t.cdf(self.range_max, self.nu)

    def print(self):
        print("Student's-t distr", ", nu = ", self.nu, " clipped at [", self.range_min, ",", self.range_max, "]")

    def scale(self):
        nu = self.nu
        return special.gamma(0.5 * (nu + 1.0)) / np.sqrt(np.pi * nu) / special.gamma(0.5 * nu)

    def pdf(self, x):
        return stats.t.pdf(np.asarray(x), self.nu)

    def cdf(self, x):
        return stats.t.cdf(x, self.nu)

    def inverse_cdf(self, x):
        return stats.t.ppf(x, self.nu)

    def sample(self, shape):
        return np.clip(np.random.standard_t(self.nu, size=shape), self.range_min, self.range_max)

    def _antiderivatives(self, x):
        """F_m(x) = int_0^x t^m (1 + t^2/nu)^(-(nu+1)/2) dt for m = 0, 1, 2 (without the constant c).
        F_m = x^(m+1)/(m+1) * 2F1((m+1)/2, (nu+1)/2; (m+3)/2; -x^2/nu); m = 1 has an elementary form."""
        nu, k = self.nu, 0.5 * (self.nu + 1.0)
        z = -(x * x) / nu
        f0 = x * special.hyp2f1(0.5, k, 1.5, z)
        f1 = nu / (1.0 - nu) * ((1.0 + x * x / nu) ** (0.5 * (1.0 - nu)) - 1.0)
        f2 = x ** 3 / 3.0 * special.hyp2f1(1.5, k, 2.5, z)
        return f0, f1, f2

    def moments(self, a, b):
        c = self.scale()
        fa, fb = self._antiderivatives(a), self._antiderivatives(b)
        return tuple(c * (hi - lo) for lo, hi in zip(fa, fb))
