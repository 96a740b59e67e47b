"""CPU oracle for the DataFrame-level contract.  TEST INFRASTRUCTURE ONLY.

Restates ``IVInterpolator.interpolate_symbol`` (reference
``src/interpolation/core.py:16-85``) step by step: the guards (:26-28, :36-39,
:49-51, :76-78, :83-85), the minute timeline and exact-equality left join
(:32-33, :42-46, :54-55), the three interpolated channels (:58-61), the nine
forward-filled passthrough columns (:64-68), ``is_interpolated`` (:71) and the
final ``dropna`` (:74).  DataFrame bookkeeping uses pandas (the reference's own
container type); every number that the reference gets from
``Series.interpolate`` comes from ``ivs_oracle.interp1d`` instead.

Pinned by ``tests/golden/symbol_cases.*`` (real reference outputs).  Not imported
by the product path.
"""
from __future__ import annotations

from datetime import timedelta
from typing import Optional

import numpy as np
import pandas as pd

try:
    from . import ivs_oracle as O
except ImportError:  # imported as a top-level module with oracle/ on sys.path
    import ivs_oracle as O

NUMERIC_COLS = ["iv", "underlying_price", "time_to_maturity"]
FILL_COLS = ["symbol", "strike", "callput", "interest_rate", "mark_price", "index_price",
             "volume", "quote_volume", "record_time"]
VALID_PANDAS_METHODS = {"linear", "time", "index", "values", "nearest", "zero", "slinear", "quadratic",
                        "cubic", "barycentric", "krogh", "spline", "polynomial", "from_derivatives",
                        "piecewise_polynomial", "pchip", "akima", "cubicspline"}


def interpolate_symbol(symbol_data: pd.DataFrame, method: str = "linear",
                       min_points: int = 10) -> Optional[pd.DataFrame]:
    if len(symbol_data) < min_points:                                   # core.py:26-28
        return None
    try:
        df = symbol_data.sort_values("date").reset_index(drop=True)    # :32
        df["date"] = pd.to_datetime(df["date"])                         # :33
        if df["date"].max() - df["date"].min() > timedelta(days=30):    # :36-39
            return None
        timeline = pd.date_range(df["date"].min(), df["date"].max(), freq="1min")   # :42-46
        if len(timeline) > 100000:                                      # :49-51
            return None
        merged = pd.DataFrame({"date": timeline}).merge(df, on="date", how="left")  # :54-55
        code = O.method_code(method)
        if code is None:
            # methods the oracle does not restate: the reference either raises (-> None)
            # or runs another scipy routine that is out of this oracle's scope.
            if method not in VALID_PANDAS_METHODS or method in ("time", "spline", "polynomial"):
                return None
            raise NotImplementedError(method)
        fill = code in (O.PAD, O.BFILL)
        m = len(merged)
        pos = np.arange(m, dtype=np.float64)                            # RangeIndex positions (R8)
        for col in NUMERIC_COLS:                                        # :58-61
            if col in merged.columns:
                dt = merged[col].dtype
                if dt == object and not fill:
                    continue                                            # Series.interpolate on object dtype: (deprecated) no-op
                if dt == object:
                    # the fill methods DO fill object cells (pad_or_backfill), then the block is soft-converted
                    # (Block._maybe_downcast -> convert): restated on row numbers, which the fill rule moves like values
                    cells = merged[col].to_numpy()
                    rows = np.where(pd.isna(cells), np.nan, np.arange(m, dtype=np.float64))
                    if not np.isnan(rows).all():
                        val, _ = O.interp1d(pos, rows, pos, code)
                        filled = np.full(m, np.nan, dtype=object)
                        ok = ~np.isnan(val)
                        filled[ok] = cells[val[ok].astype(np.int64)]
                        merged[col] = pd.Series(filled, dtype=object).infer_objects().to_numpy()
                    continue
                y = merged[col].to_numpy(np.float64, na_value=np.nan)
                if np.isnan(y).all() or not np.isnan(y).any():
                    continue                                            # missing.py:468-472
                val, st = O.interp1d(pos, y, pos, code)
                if st != O.ST_OK:
                    return None                                         # scipy raises -> :83-85
                y = np.where(np.isnan(y), val, y)
                if str(dt) in ("Float64", "Float32"):                   # nullable floats keep their dtype
                    merged[col] = pd.array(y.astype(np.float32 if str(dt) == "Float32" else np.float64), dtype=str(dt))
                elif dt == np.float32:                                  # computed in float64, stored back as float32
                    merged[col] = y.astype(np.float32)
                else:
                    merged[col] = y
        for col in FILL_COLS:                                           # :64-68
            if col in merged.columns:
                merged[col] = merged[col].ffill()
        merged["is_interpolated"] = merged["symbol"].isna()             # :71
        merged = merged.dropna(subset=["symbol", "iv", "underlying_price", "time_to_maturity"])  # :74
        if merged.empty:                                                # :76-78
            return None
        return merged
    except NotImplementedError:
        raise
    except Exception:                                                   # :83-85
        return None
