"""CPU oracle for N-minute candle aggregation.  TEST INFRASTRUCTURE ONLY.

Restates reference ``src/candle_reconstruction/core.py:36-106`` (CandleReconstructor.reconstruct_symbol_candles):
sort by timestamp (:59), floor to the target frequency (:69), groupby-agg open=first / high=max / low=min /
close=last / volume=sum (:72-79; pandas' first/last/max/min/sum skip NaN, sum of nothing = 0), keep only groups
with at least N member rows (:86-88).  Pinned by tests/golden/candles.npz (real reference outputs)."""
import numpy as np

MINUTE_NS = 60_000_000_000


def _kahan_nansum(x):
    """pandas' groupby sum (pandas/_libs/groupby.pyx group_sum): Kahan summation in row order, NaN skipped."""
    s = 0.0; comp = 0.0
    for val in x:
        if val == val:
            y = val - comp
            t = s + y
            comp = t - s - y
            s = t
    return s


def aggregate(ts_ns, o, h, l, c, v, freq_minutes):
    """One symbol.  ts_ns int64 (any order), OHLCV float64.  Returns dict of arrays (bucket start ns, o, h, l, c, v)
    or None when there are fewer than freq_minutes rows (:63-66)."""
    ts_ns = np.asarray(ts_ns, np.int64)
    if ts_ns.size == 0 or ts_ns.size < freq_minutes:
        return None
    order = np.argsort(ts_ns, kind="stable")
    ts = ts_ns[order]
    cols = [np.asarray(a, np.float64)[order] for a in (o, h, l, c, v)]
    f = freq_minutes * MINUTE_NS
    bucket = (ts // f) * f                                     # dt.floor(f'{N}min')
    starts = np.flatnonzero(np.r_[True, bucket[1:] != bucket[:-1]])
    ends = np.r_[starts[1:], ts.size]
    out = {k: [] for k in ("timestamp", "open", "high", "low", "close", "volume")}
    for a, b in zip(starts, ends):
        if b - a < freq_minutes:                               # incomplete group (:86-88), counts rows incl. NaN cells
            continue
        oo, hh, ll, cc, vv = [x[a:b] for x in cols]
        first = oo[~np.isnan(oo)]; last = cc[~np.isnan(cc)]
        out["timestamp"].append(bucket[a])
        out["open"].append(first[0] if first.size else np.nan)
        out["high"].append(np.nanmax(hh) if (~np.isnan(hh)).any() else np.nan)
        out["low"].append(np.nanmin(ll) if (~np.isnan(ll)).any() else np.nan)
        out["close"].append(last[-1] if last.size else np.nan)
        out["volume"].append(_kahan_nansum(vv))
    return {k: np.asarray(x, np.int64 if k == "timestamp" else np.float64) for k, x in out.items()}
