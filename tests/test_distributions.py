"""CPU: the device-order angle sampler (Best-Fisher wrapped-Cauchy rejection, Philox) draws from the
same distributions as the reference's Gaussian-envelope samplers ExpCosDistribution::draw
(distribution/expcosdistribution.hh:51-65) and ExpSin2Distribution::draw
(distribution/expsin2distribution.hh:45-58).  Pattern of the reference's own test_distribution.cc
(samples vs the analytic density), made quantitative with Kolmogorov-Smirnov tests."""
import numpy as np
import pytest
from scipy import integrate, stats


def vonmises_cdf(kappa, centre=0.0):
    """CDF on [-pi, pi) of p(x) ~ exp(kappa (cos(x - centre) - 1)) by numerical quadrature."""
    grid = np.linspace(-np.pi, np.pi, 4001)
    pdf = np.exp(kappa * (np.cos(grid - centre) - 1.0))
    cdf = integrate.cumulative_trapezoid(pdf, grid, initial=0.0)
    cdf /= cdf[-1]
    return lambda x: np.interp(x, grid, cdf)


@pytest.mark.parametrize("beta,x_p,x_m", [(1.0, 0.3, -0.4), (1.0, 2.9, -2.8), (0.05, 1.0, 2.0), (6.0, -1.0, 0.5),
                                          (1.0, 0.0, 3.14159)])
def test_expcos_device_order_equals_reference_distribution(orc, beta, x_p, x_m):
    L = orc.lib()
    n = 40000
    ref = np.zeros(n)
    L.orc_expcos_draws(12345, beta, x_p, x_m, n, ref)
    dev = np.array([L.orc_dev_expcos_draw(99, 0, 0, k, beta, x_p, x_m) for k in range(n)])
    assert ((dev >= -np.pi) & (dev < np.pi)).all()
    # analytic: expcosdistribution.cc:7-21 -- von Mises, tau = 2 beta |cos(dx/2)|, centred at the mean
    # staple angle (shifted by pi when |dx| > pi)
    dx = x_m - x_p
    tau = 2 * beta * abs(np.cos(0.5 * dx))
    centre = 0.5 * (x_p + x_m) + (np.pi if abs(dx) > np.pi else 0.0)
    centre = centre - 2 * np.pi * np.floor((centre + np.pi) / (2 * np.pi))
    cdf = vonmises_cdf(tau, centre)
    assert stats.kstest(dev, cdf).pvalue > 1e-3
    assert stats.kstest(ref, cdf).pvalue > 1e-3
    assert stats.ks_2samp(dev, ref).pvalue > 1e-3


@pytest.mark.parametrize("sigma", [0.02, 0.5, 4.0, 64.0, 1000.0])
def test_expsin2_device_order_equals_reference_distribution(orc, sigma):
    L = orc.lib()
    n = 40000
    ref = np.zeros(n)
    L.orc_expsin2_draws(4321, sigma, n, ref)
    dev = np.array([L.orc_dev_expsin2_draw(7, 1, 2, k, sigma) for k in range(n)])
    cdf = vonmises_cdf(0.5 * sigma)  # expsin2distribution.cc:20-24: exp(-sigma sin^2(x/2))
    assert stats.kstest(dev, cdf).pvalue > 1e-3
    assert stats.kstest(ref, cdf).pvalue > 1e-3
    assert stats.ks_2samp(dev, ref).pvalue > 1e-3


def test_vonmises_extreme_concentrations_terminate(orc):
    L = orc.lib()
    for sigma in (0.0, 1e-300, 1e-9, 1e9, float("nan")):
        v = [L.orc_dev_expsin2_draw(1, 0, 0, k, sigma) for k in range(200)]
        assert all(abs(x) <= np.pi for x in v)
    # beta = 0 or staples exactly opposite: flat conditional
    v = np.array([L.orc_dev_expcos_draw(1, 0, 0, k, 1.0, 0.0, np.pi) for k in range(4000)])
    assert stats.kstest(v, lambda x: (x + np.pi) / (2 * np.pi)).pvalue > 1e-3
