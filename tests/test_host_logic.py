"""CPU: the host bookkeeping of the drop-in IVInterpolator (rules R1-R14) against the real
reference's golden outputs, with the two device calls answered by the oracle backend."""
import numpy as np
import pandas as pd
import pytest

from golden_io import SymbolCases, assert_symbol_frame
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
    lin = c["method"] in ("linear", "index", "values", "nearest", "zero", "from_derivatives", "piecewise_polynomial")
    assert_symbol_frame(got, CASES.expected(name), rtol=0 if lin else RTOL, atol=0 if lin else ATOL, name=name)
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
