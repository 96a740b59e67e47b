"""Greeks epilogue: oracle vs the real reference's golden outputs (CPU) and the HIP kernel vs both (GPU)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import greeks_oracle as G  # noqa: E402
from golden_io import GOLDEN  # noqa: E402

g = np.load(os.path.join(GOLDEN, "greeks.npz"))
RTOL = 1e-12   # erfc/exp/log implementations differ in the last bits (scipy ndtr vs libm vs ocml); measured <= 6e-14


def test_oracle_against_reference_golden():
    for typ in ("call", "put"):
        o = G.calculate_greeks(g["S"], g["K"], g["T"], g["r"], g["sigma"], typ == "put")
        for k, v in o.items():
            assert np.allclose(v, g[f"{typ}/{k}"], rtol=RTOL, atol=1e-300), (typ, k)


def test_known_answers_survey_8f():
    c = G.calculate_greeks(25000., 25500., .05, .01, .6, False)
    p = G.calculate_greeks(25000., 25500., .05, .01, .6, True)
    assert abs(c["delta"] - 0.4693948058617507) < 1e-14 and abs(c["rho"] - 5.3072242998553145) < 1e-12
    assert abs(p["delta"] + 0.5306051941382492) < 1e-14 and abs(p["rho"] - 7.436402293629095) < 1e-12   # + sign: reference quirk


@pytest.mark.gpu
def test_hip_greeks_against_reference_golden_and_oracle():
    import torch
    from iv_interpolation_amd import engine
    from iv_interpolation_amd.greeks import BlackScholesGreeks
    for typ in ("call", "put"):
        got = BlackScholesGreeks.calculate_greeks(g["S"], g["K"], g["T"], g["r"], g["sigma"], typ)
        for k, v in got.items():
            assert np.allclose(v, g[f"{typ}/{k}"], rtol=RTOL, atol=1e-300), (typ, k, np.max(np.abs(v - g[f"{typ}/{k}"])))
    r = BlackScholesGreeks.calculate_greeks(25000., 25500., .05, .01, .6, "put")
    assert abs(r["rho"] - 7.436402293629095) < 1e-12 and abs(r["theta"] + 36.14467931881973) < 1e-11
    # per-element option type + a large batch against the oracle on a sample
    n = 5_000_000
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    u = lambda lo, hi: torch.rand(n, generator=gen, dtype=torch.float64, device="cuda") * (hi - lo) + lo   # noqa: E731
    S = u(20000, 30000); K = S * u(.7, 1.3); T = u(1 / 365, 1.5); rr = u(0, .05); sg = u(.05, 3.0)
    put = (torch.rand(n, generator=gen, device="cuda") < 0.5).to(torch.uint8)
    out = engine.bs_greeks(S, K, T, rr, sg, is_put=put)
    idx = torch.arange(0, n, 997, device="cuda")
    ref = G.calculate_greeks(*[t[idx].cpu().numpy() for t in (S, K, T, rr, sg)], put[idx].cpu().numpy().astype(bool))
    for k in ref:
        assert np.allclose(out[k][idx].cpu().numpy(), ref[k], rtol=RTOL, atol=1e-300), k
