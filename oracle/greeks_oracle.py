"""CPU oracle for the Black-Scholes Greeks epilogue.  TEST INFRASTRUCTURE ONLY.

Restates reference ``src/interpolation/greeks.py:12-43`` line by line; ``scipy.stats.norm.cdf/pdf`` are
restated as 0.5*erfc(-x/sqrt(2)) (scipy's ndtr) and exp(-x^2/2)/sqrt(2*pi).  The reference's put rho carries NO
minus sign (greeks.py:35 uses +K*T*exp(-rT)*N(-d2)/100); parity means reproducing that.
Pinned by tests/golden/greeks.npz (outputs of the real reference)."""
import math

import numpy as np

_erfc = np.vectorize(math.erfc, otypes=[np.float64])


def norm_cdf(x):
    return 0.5 * _erfc(-np.asarray(x, np.float64) / math.sqrt(2.0))


def norm_pdf(x):
    x = np.asarray(x, np.float64)
    return np.exp(-x * x / 2.0) / math.sqrt(2.0 * math.pi)


def calculate_greeks(S, K, T, r, sigma, is_put):
    S, K, T, r, sigma = [np.asarray(a, np.float64) for a in (S, K, T, r, sigma)]
    is_put = np.broadcast_to(np.asarray(is_put, bool), S.shape)
    sq = np.sqrt(T)
    d1 = (np.log(S / K) + (r + 0.5 * sigma ** 2) * T) / (sigma * sq)         # greeks.py:21
    d2 = d1 - sigma * sq                                                       # :22
    disc = r * K * np.exp(-r * T)
    common = -S * norm_pdf(d1) * sigma / (2 * sq)
    delta = np.where(is_put, norm_cdf(d1) - 1, norm_cdf(d1))                   # :25,:29
    theta = np.where(is_put, (common + disc * norm_cdf(-d2)) / 365, (common - disc * norm_cdf(d2)) / 365)   # :26-27,:30-31
    gamma = norm_pdf(d1) / (S * sigma * sq)                                    # :33
    vega = S * norm_pdf(d1) * sq / 100                                         # :34
    rho = K * T * np.exp(-r * T) * np.where(is_put, norm_cdf(-d2), norm_cdf(d2)) / 100    # :35 (no sign flip for puts)
    return {"delta": delta, "gamma": gamma, "theta": theta, "vega": vega, "rho": rho}
