"""CPU oracle for the IV -> OHLCV bridge.  TEST INFRASTRUCTURE ONLY.

Restates ``InterpolatedToOHLCVConverter._generate_ohlcv_from_interpolated`` and its four candle builders
(reference ``src/data_bridge/ohlcv_converter.py:138-187`` driver, ``:209-263`` spread_simulation, ``:265-290``
price_as_midpoint, ``:292-332`` trend_following, ``:334-357`` simple_spread, ``:359-369`` volume, ``:371-402``
quality check).  The reference draws from the process-global legacy NumPy generator (``np.random.uniform`` /
``normal`` / ``exponential``), unseeded; the numbers become reproducible once the caller seeds it, so this oracle
takes an explicit ``seed`` and consumes ``np.random.RandomState(seed)`` -- the same MT19937 stream, in the same
order: per valid row the strategy's draws, then one exponential draw iff the row has no positive volume.

Rounding: the reference rounds Python floats (``row[...]`` out of ``iterrows()`` on a mixed-type frame yields Python
floats), i.e. correctly rounded decimal rounding, ties on the exact binary value to even -- ``py_round`` below is that
(it simply calls Python's ``round`` on a Python float).

Pinned by ``tests/golden/bridge.npz`` (outputs of the real reference under ``np.random.seed``).  Not imported by the
product path.
"""
from __future__ import annotations

import math

import numpy as np

STRATEGIES = {"spread_simulation": 0, "price_as_midpoint": 1, "trend_following": 2, "simple_spread": 3}
PIPELINE_INLINE = 4      # CompleteOptimizedPipeline._generate_ohlcv_candles (reference complete_pipeline.py:473-510).
# PARITY UNPINNED for strategy 4: complete_pipeline.py cannot be imported here (python-dotenv / psycopg2 absent), so
# no golden vectors exist for it; it is restated from the source text only and tested GPU-vs-oracle.
OUT_COLS = ("open", "high", "low", "close", "volume", "source_price")


def py_round(x, nd):
    return round(float(x), nd)


def strategy_code(name: str) -> int:
    return STRATEGIES.get(name, STRATEGIES["simple_spread"])       # ohlcv_converter.py:167-169: anything else -> simple


def candles(price, volume, strategy: int, seed=None, rs=None, base_spread_pct=0.002, vol_factor=1.5):
    """One symbol.  price, volume: float arrays (volume may be None = column absent).  Returns (valid mask, dict of
    arrays over the VALID rows).  ``rs``: a RandomState to continue a stream across symbols (else seeded here)."""
    if rs is None:
        rs = np.random.RandomState(seed)
    price = np.asarray(price, np.float64)
    n = len(price)
    vol_in = None if volume is None else np.asarray(volume, np.float64)
    valid = np.zeros(n, bool)
    rows = []
    closes = []                                    # rounded closes of the candles emitted so far
    for i in range(n):
        base = float(price[i])
        if math.isnan(base) or base <= 0:          # :156-157
            continue
        valid[i] = True
        if strategy == 0:                          # :209-263
            vm = rs.uniform(0.5, vol_factor)
            spread = base * base_spread_pct * vm
            tb = 0
            if closes:
                tb = (base - closes[-1]) * 0.3
            o = base + (rs.uniform(-spread / 3, spread / 3) + tb * 0.2)
            c = base + (rs.uniform(-spread / 3, spread / 3) + tb * 0.5)
            mid = (o + c) / 2
            he = rs.uniform(0, spread / 2)
            lr = rs.uniform(0, spread / 2)
            h = max(o, c) + he
            lo = min(o, c) - lr
            if h - lo < base * 0.0005:
                h = mid + base * 0.00025
                lo = mid - base * 0.00025
        elif strategy == 1:                        # :265-290
            spread = base * 0.001
            o = base + rs.uniform(-spread / 4, spread / 4)
            c = base + rs.uniform(-spread / 4, spread / 4)
            h = base + spread / 2
            lo = base - spread / 2
        elif strategy == 2:                        # :292-332
            lookback = min(5, len(closes))
            trend = 0
            if lookback > 0:
                recent = closes[-lookback:]
                if len(recent) > 1:
                    trend = (recent[-1] - recent[0]) / len(recent)
            noise = rs.normal(0, base * 0.001)
            o = base + trend * 0.6 + noise
            c = base + trend * 0.6 * 1.2 + noise
            if trend > 0:
                h = max(o, c) + abs(trend) * 0.5
                lo = min(o, c) - abs(trend) * 0.2
            else:
                h = max(o, c) + abs(trend) * 0.2
                lo = min(o, c) - abs(trend) * 0.5
        elif strategy == PIPELINE_INLINE:          # complete_pipeline.py:484-494
            spread = base * 0.001
            o = base + rs.uniform(-spread / 3, spread / 3)
            c = base + rs.uniform(-spread / 3, spread / 3)
            h = max(o, c) + abs(rs.uniform(0, spread / 2))
            lo = min(o, c) - abs(rs.uniform(0, spread / 2))
        else:                                      # :334-357
            spread = base * 0.001
            o = base
            c = base + rs.uniform(-spread / 2, spread / 2)
            h = base + spread / 2
            lo = base - spread / 2
        v = 0.0 if vol_in is None else float(vol_in[i])            # :359-369
        if math.isnan(v) or v <= 0:
            v = rs.exponential(50)
        v = max(0, v)
        rc = py_round(c, 4)
        closes.append(rc)
        rows.append((py_round(o, 4), py_round(h, 4), py_round(lo, 4), rc, py_round(v, 6), base))
    arr = np.array(rows, np.float64).reshape(-1, 6)
    return valid, {k: arr[:, j] for j, k in enumerate(OUT_COLS)}


def candles_batch(price, volume, row_off, strategy: int, seed, **kw):
    """Many symbols (CSR rows), ONE stream consumed symbol after symbol like the reference's sequential loop
    (convert_batch, ohlcv_converter.py:463-...).  Returns (valid[total_rows], out[6][total_rows] with NaN on invalid rows)."""
    rs = np.random.RandomState(seed)
    price = np.asarray(price, np.float64)
    total = len(price)
    valid = np.zeros(total, bool)
    out = np.full((6, total), np.nan)
    for s in range(len(row_off) - 1):
        a, b = int(row_off[s]), int(row_off[s + 1])
        v, d = candles(price[a:b], None if volume is None else volume[a:b], strategy, rs=rs, **kw)
        valid[a:b] = v
        idx = a + np.flatnonzero(v)
        for j, k in enumerate(OUT_COLS):
            out[j, idx] = d[k]
    return valid, out


def pipeline_base_price(underlying, mark, index):
    """``row['underlying_price'] or row['mark_price'] or row['index_price']`` (complete_pipeline.py:479): the first
    TRUTHY value -- 0.0 falls through, NaN does not."""
    u, m, i = (np.asarray(a, np.float64) for a in (underlying, mark, index))
    return np.where(u != 0, u, np.where(m != 0, m, i))


def mt19937_words(seed: int, n: int) -> np.ndarray:
    """First n raw 32-bit outputs of np.random.seed(seed)'s generator."""
    return np.frombuffer(np.random.RandomState(seed).bytes(4 * n), dtype="<u4").copy()


def validate_quality(o, h, lo, c, src):
    """_validate_ohlcv_quality (ohlcv_converter.py:371-402) as (valid, reason)."""
    o, h, lo, c, src = (np.asarray(a, np.float64) for a in (o, h, lo, c, src))
    if (h < lo).any():
        return False, "High < Low found"
    if ((h < o) | (h < c)).any():
        return False, "High < Open/Close found"
    if ((lo > o) | (lo > c)).any():
        return False, "Low > Open/Close found"
    if (((h - lo) / src) > 0.1).any():
        return False, "Unrealistic spreads detected"
    if (np.stack([o, h, lo, c]) <= 0).any():
        return False, "Negative or zero prices found"
    return True, "All quality checks passed"
