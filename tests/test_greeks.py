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


# ---- the epilogue wired to `preserve_greeks` (reference config.py:46; columns schema.py:36-40): frames through the
# interpolator with the flag set, against the real reference's interpolate_symbol + calculate_greeks on the same frames
gf = np.load(os.path.join(GOLDEN, "greeks_frames.npz"))
GREEKS = ["delta", "gamma", "theta", "vega", "rho"]


def _gf_input(name):
    import pandas as pd
    from golden_io import _dec
    cols = [str(c) for c in gf[f"{name}/in_columns"]]
    return pd.DataFrame({c: _dec(gf, f"{name}/in", c, str(gf[f"{name}/in_tag/{c}"])) for c in cols})


def _check_greeks_frame(got, name):
    assert got is not None and len(got) == int(gf[f"{name}/rows"]), name
    assert list(got.columns[-5:]) == GREEKS, list(got.columns)
    assert np.array_equal(got["date"].to_numpy().astype("datetime64[ns]").astype(np.int64), gf[f"{name}/date"])
    for c in ("iv", "underlying_price", "time_to_maturity"):
        assert np.allclose(got[c].to_numpy(np.float64), gf[f"{name}/out/{c}"], rtol=1e-12, atol=1e-13, equal_nan=True), (name, c)
    for c in GREEKS:
        g, e = got[c].to_numpy(np.float64), gf[f"{name}/greeks/{c}"]
        assert np.array_equal(np.isnan(g), np.isnan(e)), (name, c)
        # Greeks amplify the 1e-15 differences of the interpolated inputs (d1 divides by sigma sqrt(T)): 1e-10 relative
        assert np.allclose(g, e, rtol=1e-10, atol=1e-14, equal_nan=True), (name, c, np.nanmax(np.abs(g - e)))


@pytest.mark.parametrize("name", [str(n) for n in gf["names"]])
def test_preserve_greeks_host_logic_against_reference_golden(name):
    from oracle_backend import OracleBackend
    from iv_interpolation_amd import IVInterpolator
    method = str(gf["methods"][list(gf["names"]).index(name)])
    df = _gf_input(name)
    iv = IVInterpolator(method, 10, backend=OracleBackend(), preserve_greeks=True)
    _check_greeks_frame(iv.interpolate_symbol(df), name)
    _check_greeks_frame(iv.interpolate_frame(df), name)
    plain = IVInterpolator(method, 10, backend=OracleBackend()).interpolate_symbol(df)
    assert not set(GREEKS) & set(plain.columns)                      # flag off: the reference's 14 columns, nothing else


@pytest.mark.gpu
@pytest.mark.parametrize("name", [str(n) for n in gf["names"]])
def test_preserve_greeks_epilogue_on_gpu_against_reference_golden(name):
    from iv_interpolation_amd import IVInterpolator
    method = str(gf["methods"][list(gf["names"]).index(name)])
    df = _gf_input(name)
    iv = IVInterpolator(method, 10, preserve_greeks=True)
    _check_greeks_frame(iv.interpolate_symbol(df), name)
    _check_greeks_frame(iv.interpolate_frame(df), name)


def test_greeks_are_nan_when_a_channel_is_an_object_column():
    """Series.interpolate leaves an object-dtype channel alone (reference core.py:61): the epilogue has no interpolated
    value to work from and must report NaN, not numbers computed from placeholders."""
    from oracle_backend import OracleBackend
    from iv_interpolation_amd import IVInterpolator
    name = str(gf["names"][0])
    df = _gf_input(name)
    df["underlying_price"] = df["underlying_price"].astype(object)
    iv = IVInterpolator("linear", 10, backend=OracleBackend(), preserve_greeks=True)
    for got in (iv.interpolate_symbol(df), iv.interpolate_frame(df)):
        assert got is not None and len(got) > 0
        assert all(got[c].isna().all() for c in GREEKS), got[GREEKS].head()
