#!/usr/bin/env python3
"""Golden vectors for the Greeks epilogue from the REAL reference (src/interpolation/greeks.py), data only.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_greeks.py"""
import os, sys
sys.dont_write_bytecode = True
import numpy as np
sys.path.insert(0, "/root/reference/src")
from interpolation.greeks import BlackScholesGreeks   # the real reference

r = np.random.default_rng(20230320)
n = 4096
S = r.uniform(20000, 30000, n); K = S * r.uniform(0.7, 1.3, n); T = r.uniform(1 / 365, 1.5, n)
rate = r.uniform(0.0, 0.05, n); sig = r.uniform(0.05, 3.0, n)
out = {"S": S, "K": K, "T": T, "r": rate, "sigma": sig}
for typ in ("call", "put"):
    g = BlackScholesGreeks.calculate_greeks(S, K, T, rate, sig, typ)
    for k, v in g.items():
        out[f"{typ}/{k}"] = np.asarray(v)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "greeks.npz"), **out)
print("greeks golden:", n, "x 2 option types")
