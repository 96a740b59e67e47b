"""GPU: randomized differential test of ivs_surface_batch_f64 against the oracle.  Every case draws a shape (strike
count uniform or ragged, 4..16 maturities, output grid), a layout (shared / per-surface strikes, maturities, query
grids), a method and a few awkward ingredients (NaN quotes, queries on knots and outside the hull, duplicated or
descending queries), so that every dispatch path (dense 64x16, variable-shape one- and two-wavefront kernels, generic
kernel, filtered redo pass) is hit with inputs nobody hand-picked.  Seeds are fixed: failures reproduce."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ivs_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

METHODS = {"linear": O.LINEAR, "cubic": O.CUBIC, "cubicspline": O.CUBICSPLINE, "slinear": O.SLINEAR,
           "nearest": O.NEAREST, "zero": O.ZERO, "pchip": O.PCHIP, "akima": O.AKIMA, "from_derivatives": O.FROM_DERIVATIVES,
           "quadratic": O.QUADRATIC}
EXACT = ("linear", "nearest", "zero", "from_derivatives")


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_case(seed):
    from iv_interpolation_amd import synth
    r = np.random.default_rng(seed)
    method = list(METHODS)[int(r.integers(len(METHODS)))]
    if r.random() < 0.55:                                     # favour the methods with fast kernels
        method = ["linear", "cubic", "cubicspline", "slinear", "pchip", "akima"][int(r.integers(6))]
    nT = int(r.choice([16, 16, 16, 8, 4, 5, 11, 15]))
    B = int(r.choice([1, 2, 37, 130, 400]))
    ragged = r.random() < 0.35
    mK = int(r.choice([1, 7, 64, 64, 65, 100, 130, 256]))
    mT = int(r.choice([1, 3, 16, 16, 17, 40, 64, 70]))
    case = dict(seed=seed, method=method, nT=nT, B=B, ragged=ragged, mK=mK, mT=mT)
    if ragged:
        lo, hi = [(8, 128), (4, 64), (65, 128), (2, 140), (4, 20)][int(r.integers(5))]
        d = synth.numpy_ragged_batch(B, nT, lo, hi, seed=seed)
        nk = np.diff(d["k_off"])
        K, sigma, k_off = d["K"], d["sigma"].copy(), d["k_off"]
        case.update(nK=f"{lo}..{hi}")
    else:
        nK = int(r.choice([64, 64, 64, 4, 5, 16, 17, 33, 48, 63, 65, 96, 128, 129, 150]))
        d = synth.numpy_batch(B, nK, nT, seed=seed)
        K, sigma, k_off = d["K"], d["sigma"].copy(), None
        nk = np.full(B, nK)
        if r.random() < 0.2:
            K = K[0].copy()                                   # strikes shared by the batch
        case.update(nK=nK)
    T = d["T"]
    if r.random() < 0.3:                                      # per-surface maturities
        T = T[None, :] * (1.0 + 0.4 * r.random((B, 1)))
    # NaN quotes: none, a few surfaces, or many
    mode = r.choice(["none", "none", "few", "many"])
    if mode != "none":
        flat = sigma.reshape(-1)
        p = 0.0005 if mode == "few" else 0.15
        flat[r.random(flat.size) < p] = np.nan
    # query grids
    kmin, kmax = float(np.min(K)), float(np.max(K))
    Kq = r.uniform(kmin - 0.1 * (kmax - kmin), kmax + 0.1 * (kmax - kmin), mK)
    if mK > 3:
        Kq[0] = np.ravel(K)[0]; Kq[1] = np.ravel(K)[min(3, np.ravel(K).size - 1)]; Kq[2] = Kq[3]      # knots, a duplicate
    if r.random() < 0.7:
        Kq.sort()
    tq_shape = (B, mT) if (r.random() < 0.3) else (mT,)
    Tlo, Thi = float(np.min(T)), float(np.max(T))
    Tq = np.exp(r.uniform(np.log(0.6 * Tlo), np.log(1.3 * Thi), tq_shape))
    Tq = np.sort(Tq, axis=-1)
    if mT > 2:
        Tq[..., 1] = np.ravel(T)[0]; Tq = np.sort(Tq, axis=-1)
    if r.random() < 0.1 and mT > 1:
        Tq = Tq[..., ::-1].copy()                             # descending queries: dense kernels hand over to the generic one
    if r.random() < 0.25:
        Kq = np.tile(Kq, (B, 1)) * (1.0 + 0.01 * r.random((B, 1)))
    # units: strikes in anything from milli-units to millions, vols from basis points to thousands
    ks, ss = 1.0, 1.0
    if r.random() < 0.3:
        ks = float(10.0 ** r.uniform(-3, 5)); ss = float(10.0 ** r.uniform(-4, 3))
        K = K * ks; Kq = Kq * ks; sigma = sigma * ss
    case.update(K=K, T=T, sigma=sigma, k_off=k_off, Kq=Kq, Tq=Tq, nk_max=int(nk.max()), nan=mode, scale=ss)
    return case


_lo, _hi = (int(x) for x in os.environ.get("IVS_FUZZ_SEEDS", "7000:7120").split(":"))      # widen for a one-off hunt


@pytest.mark.parametrize("seed", range(_lo, _hi))
def test_random_case_matches_oracle(seed):
    from iv_interpolation_amd import engine
    c = make_case(seed)
    kw = {}
    if c["k_off"] is not None:
        kw = dict(k_off=dev(c["k_off"]), nK_max=c["nk_max"], n_maturities=c["nT"])
    out, st = engine.surface_batch(dev(c["K"]), dev(c["T"]), dev(c["sigma"]), dev(c["Kq"]), dev(c["Tq"]), c["method"], **kw)
    kern = engine.last_kernel()
    Kfull = c["K"]
    if c["k_off"] is None and np.ndim(Kfull) == 1:
        Kfull = np.tile(Kfull, (c["B"], 1))
    ref, rst = O.surface_batch(Kfull, c["T"], c["sigma"], c["Kq"], c["Tq"], METHODS[c["method"]], k_off=c["k_off"])
    tagd = {k: c[k] for k in ("seed", "method", "nT", "B", "ragged", "nK", "mK", "mT", "nan")}
    tagd["kernel"] = kern
    got = out.cpu().numpy()
    assert np.array_equal(st.cpu().numpy(), rst), tagd
    assert np.array_equal(np.isnan(got), np.isnan(ref)), tagd
    if c["method"] in EXACT:
        assert np.array_equal(got, ref, equal_nan=True), (tagd, np.nanmax(np.abs(got - ref)))
    else:
        # Tolerance by conditioning (rounds 1-2: 1e-10 for everything).  Methods that return NaN outside the hull evaluate
        # inside it only: 1e-13 relative (measured on the 120 default seeds: <= 2.3e-15, profiles/r03/fuzz_errors.txt).
        # 'cubicspline' / 'pchip' EXTRAPOLATE to the right -- the queries reach 10 % (strikes) and 30 % (maturities) beyond
        # the last knot, where a cubic piece on a short last interval amplifies the rounding of its coefficients (scipy's
        # own two routes differ there as well): 1e-10 relative (measured: <= 1.2e-11, seed 7047, 5 maturities).
        path = os.environ.get("IVS_ERRLOG")
        if path:
            with np.errstate(all="ignore"):
                dd = np.abs(got - ref); okk = np.isfinite(dd)
                rel = float(np.max(dd[okk] / (0.1 * c["scale"] + np.abs(ref[okk])))) if okk.any() else 0.0
            with open(path, "a") as f:
                f.write(f"{rel:.3e} fuzz {tagd}\n")
        rt = 1e-10 if c["method"] in ("cubicspline", "pchip") else 1e-13
        assert np.allclose(got, ref, rtol=rt, atol=0.1 * rt * c["scale"], equal_nan=True), (tagd, np.nanmax(np.abs(got - ref)))
