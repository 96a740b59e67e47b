"""CPU: the host bookkeeping of the drop-in IVInterpolator (rules R1-R14) against the real
reference's golden outputs, with the two device calls answered by the oracle backend."""
import numpy as np
import pandas as pd
import pytest

from golden_io import SymbolCases, assert_long_frame, assert_symbol_frame, golden_frame_groups, method_tolerances
from oracle_backend import OracleBackend

from iv_interpolation_amd import EngineUnavailable, IVInterpolator

CASES = SymbolCases()
RTOL, ATOL = 1e-12, 1e-13


@pytest.mark.parametrize("name", CASES.names())
def test_symbol_cases(name):
    c = CASES.cases[name]
    df = CASES.input(name)
    before = df.copy(deep=True)
    got = IVInterpolator(c["method"], c["min_points"], backend=OracleBackend()).interpolate_symbol(df)
    assert_symbol_frame(got, CASES.expected(name), name=name, **method_tolerances(c["method"]))
    pd.testing.assert_frame_equal(df, before)          # caller's frame is not mutated (SURVEY 8b ownership)


def test_batch_equals_single():
    names = [n for n in CASES.names() if n.startswith(("g5_nan_linear", "g4_dup", "g6_too_few", "g1_linear", "fuzz0"))]
    names = [n for n in names if CASES.cases[n]["method"] == "linear"]
    frames = [CASES.input(n) for n in names]
    iv = IVInterpolator("linear", 2, backend=OracleBackend())
    batch = iv.interpolate_batch(frames)
    for n, f, b in zip(names, frames, batch):
        assert_symbol_frame(b, iv.interpolate_symbol(f), name=n)


def test_import_path_shim():
    from interpolation.core import IVInterpolator as Shim
    assert Shim is IVInterpolator
    iv = Shim()
    assert iv.method == "linear" and iv.min_points == 10     # reference core.py:12-14


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    df = CASES.input("g1_linear")
    with pytest.raises(EngineUnavailable):
        IVInterpolator().interpolate_symbol(df)


def _long_frame(names, rename=True):
    frames = []
    for i, n in enumerate(names):
        f = CASES.input(n).copy()
        if rename and "symbol" in f.columns:
            f["symbol"] = f["symbol"].where(f["symbol"].isna(), f"sym{i:03d}-" + n)
        frames.append(f)
    return frames


@pytest.mark.parametrize("method,min_points", [("linear", 2), ("linear", 10), ("cubic", 2), ("pchip", 2)])
def test_interpolate_frame_equals_concat_of_per_symbol_results(method, min_points):
    """Columnar path: one long frame with many symbols == concat of the per-symbol results (which are golden-checked)."""
    full = {"symbol", "date", "iv", "underlying_price", "time_to_maturity", "strike"}
    names = [n for n in CASES.names() if set(CASES.input(n).columns) == full or n.startswith(("g4_dup", "g3_", "g5_nan", "fuzz"))]
    # one long frame needs one column set: use the 13-column cases and the 6-column cases separately
    groups = {}
    for n in names:
        groups.setdefault(tuple(CASES.input(n).columns), []).append(n)
    iv = IVInterpolator(method, min_points, backend=OracleBackend())
    checked = 0
    for colset, ns in groups.items():
        if len(ns) < 3 or "symbol" not in colset:
            continue
        frames = _long_frame(ns)
        # symbols in arbitrary order; rows of one symbol keep their input order (ties between duplicate
        # timestamps are resolved by input order, as in the per-symbol path)
        long = pd.concat(frames[::-1], ignore_index=True)
        got = iv.interpolate_frame(long)
        per = [iv.interpolate_symbol(f) for f in frames]
        order = np.argsort([f["symbol"].dropna().iloc[0] for f in frames], kind="stable")
        exp = [per[i] for i in order if per[i] is not None]
        if not exp:
            assert len(got) == 0
            continue
        exp = pd.concat(exp, ignore_index=True)
        assert list(got.columns) == list(exp.columns)
        assert len(got) == len(exp), (len(got), len(exp))
        for c in exp.columns:
            g, e = got[c], exp[c]
            if e.dtype.kind == "f":
                assert np.array_equal(g.to_numpy(np.float64), e.to_numpy(np.float64), equal_nan=True), c
            elif e.dtype == object:
                assert ((g == e) | (g.isna() & e.isna())).all(), c
            else:
                assert (g.to_numpy() == e.to_numpy()).all(), c
        checked += 1
    assert checked >= 2


_EXACT = ("linear", "index", "values", "nearest", "zero", "from_derivatives", "piecewise_polynomial")


def test_interpolate_frame_against_reference_goldens():
    """(f)1 at frame level: the concatenated golden INPUTS through interpolate_frame against the concatenated golden
    OUTPUTS of the real reference (reference semantics core.py:16-85 per symbol; callers batch_processor.py:166-173,
    complete_pipeline.py:350-353 iterate the same rows).  Device calls are answered by the oracle backend here; the GPU
    twin of this test (tests/test_gpu_parity.py) runs the same frames through the HIP kernels."""
    n_groups = n_cases = 0
    for method, min_points, long_in, exp, names in golden_frame_groups(CASES):
        try:
            iv = IVInterpolator(method, min_points, backend=OracleBackend())
            got = iv.interpolate_frame(long_in)
        except ValueError:
            assert exp is None, (method, names)         # methods the reference itself rejects ('time', ...): every case is None
            continue
        assert_long_frame(got, exp, name=f"{method}/{min_points}/{len(names)} cases", **method_tolerances(method))
        n_groups += 1; n_cases += len(names)
    assert n_groups >= 15 and n_cases >= 100, (n_groups, n_cases)


@pytest.mark.parametrize("method", ["barycentric", "krogh"])
def test_polynomial_methods_give_up_beyond_32_knots(method):
    """Documented deviation: the reference returns the values of ONE polynomial through all knots;
    beyond IVS_POLY_MAX_KNOTS = 32 knots those values are numerical noise in the reference itself, the engine reports
    IVS_ST_ILL_CONDITIONED and the symbol becomes None.  Up to 32 knots the goldens y1/y5/y8/y9 pin the values."""
    from iv_interpolation_amd.frame_store import synthetic_symbol
    ok = synthetic_symbol("poly-ok", 30, seed=3)
    too_many = synthetic_symbol("poly-many", 40, seed=4)
    iv = IVInterpolator(method, 10, backend=OracleBackend())
    assert iv.interpolate_symbol(ok) is not None
    assert iv.interpolate_symbol(too_many) is None


def test_interpolate_batch_columnar_path_against_reference_goldens():
    """interpolate_batch on >= 4 frames with identical columns and dtypes takes the columnar path (one long frame grouped by
    frame number, results cut back into per-symbol frames): every golden case, batched with the cases that share its method,
    min_points and input schema, must come back exactly as the REAL reference returned it for the single symbol -- values,
    dtypes, column order AND the surviving merged-frame positions as index; frames the long frame cannot represent fall
    back to the per-symbol bookkeeping inside the call."""
    groups = {}
    for name in CASES.names():
        c = CASES.cases[name]
        groups.setdefault((c["method"], c["min_points"], tuple(map(tuple, c["in_columns"]))), []).append(name)
    n_fast = n_cases = 0
    for (method, min_points, _), names in groups.items():
        if len(names) < 4:
            continue
        iv = IVInterpolator(method, min_points, backend=OracleBackend())
        frames = [CASES.input(n) for n in names]
        before = [f.copy(deep=True) for f in frames]
        took_fast = iv._batch_via_frame(frames) is not None
        got = iv.interpolate_batch(frames)
        for n, g, f, b in zip(names, got, frames, before):
            assert_symbol_frame(g, CASES.expected(n), name=f"batch[{method}/{min_points}] {n}", **method_tolerances(method))
            pd.testing.assert_frame_equal(f, b)          # the caller's frames are not mutated
        n_fast += took_fast; n_cases += len(names)
        if took_fast and any(g is not None for g in got):      # the per-symbol frames accept the callers' new column silently
            import warnings
            g = next(x for x in got if x is not None)
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                g["batch_id"] = "b1"                             # batch_processor.py:105, complete_pipeline.py:323
    assert n_fast >= 8 and n_cases >= 120, (n_fast, n_cases)


def test_same_schema_concat_equals_pd_concat_or_declines():
    """interpolate_batch joins the callers' frames block by block (_concat_same_schema) when they share columns, dtypes and
    block layout: must be pd.concat(frames, ignore_index=True) exactly -- or None, and pd.concat then decides."""
    from iv_interpolation_amd.core import _concat_same_schema
    from iv_interpolation_amd.frame_store import synthetic_symbol
    frames = [synthetic_symbol(f"s{i}", 12 + i, seed=i) for i in range(9)]
    frames.append(frames[0].iloc[2:7])                                   # a slice: non-owning 2-D block views
    frames.append(frames[1].iloc[0:0])                                   # an empty frame
    f0 = frames[0]
    got = _concat_same_schema(frames, f0.columns, list(f0.dtypes))
    assert got is not None
    pd.testing.assert_frame_equal(got, pd.concat(frames, ignore_index=True))
    g = frames[5].copy(); g["volume"] = g["volume"].fillna(0).astype("int64")          # another dtype in one frame
    assert _concat_same_schema(frames[:5] + [g], f0.columns, list(f0.dtypes)) is None
    h = frames[3].copy(); h["extra"] = 1.0; h = h.drop(columns="extra")                # same columns, may differ in block layout
    r = _concat_same_schema(frames[:3] + [h], f0.columns, list(f0.dtypes))
    if r is not None:
        pd.testing.assert_frame_equal(r, pd.concat(frames[:3] + [h], ignore_index=True))
    tz = [f.assign(date=f["date"].dt.tz_localize("UTC")) for f in frames[:4]]          # extension block: declined
    assert _concat_same_schema(tz, tz[0].columns, list(tz[0].dtypes)) is None
    iv = IVInterpolator("linear", backend=OracleBackend())
    res = iv.interpolate_batch(tz + tz)                                  # ... and the batch call still answers through pd.concat
    assert all(r is not None and str(r["date"].dtype) == "datetime64[ns, UTC]" for r in res)


def test_batch_call_restores_the_collector_state():
    import gc
    from iv_interpolation_amd.frame_store import synthetic_symbol
    frames = [synthetic_symbol(f"s{i}", 12, seed=i) for i in range(5)]
    iv = IVInterpolator("linear", backend=OracleBackend())
    assert gc.isenabled()
    iv.interpolate_batch(frames)
    assert gc.isenabled()
    gc.disable()
    try:
        iv.interpolate_batch(frames)
        assert not gc.isenabled()                                        # a caller that had it off keeps it off
    finally:
        gc.enable()


def test_guards_speak_on_the_logger_like_the_reference(caplog):
    """core.py:27,38,50: every `None` comes with a line on the module's logger.  The columnar path decides silently, so the
    frames it answers with None are handed to the per-symbol bookkeeping, which stops at the same guard and says why --
    through interpolate_symbol (one frame) and through interpolate_batch (many)."""
    import logging
    from iv_interpolation_amd.frame_store import synthetic_symbol
    iv = IVInterpolator("linear", backend=OracleBackend())
    short = synthetic_symbol("s0", 5, seed=1)                            # 5 rows < min_points = 10
    good = [synthetic_symbol(f"s{i}", 12, seed=i) for i in range(1, 5)]
    with caplog.at_level(logging.WARNING, logger="interpolation.core"):
        assert iv.interpolate_symbol(short) is None
    assert any("Insufficient data points: 5 < 10" in r.getMessage() for r in caplog.records)
    caplog.clear()
    with caplog.at_level(logging.WARNING, logger="interpolation.core"):
        res = iv.interpolate_batch(good[:2] + [short] + good[2:])
    assert res[2] is None and all(r is not None for i, r in enumerate(res) if i != 2)
    assert sum("Insufficient data points" in r.getMessage() for r in caplog.records) == 1
