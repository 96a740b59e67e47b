"""Test-only backend: lets the host bookkeeping of IVInterpolator (packing, join, gather,
assembly) run on CPU by answering the two device calls from the oracle.  Lives under tests/
on purpose: the product package has no CPU route."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ivs_oracle as O  # noqa: E402


class OracleBackend:
    def interp1d_batch(self, xk, yk, knot_off, q_off, total_q, code):
        return O.interp1d_batch(xk, yk, knot_off, None, q_off, code)

    def interp1d_greeks_batch(self, xk, yk, knot_off, q_off, total_q, code, src_pos, gvalid, strike_src, rate_src, put_src):
        """Oracle twin of the fused call: channels by ivs_oracle, rows that are knots of a channel keep the source cell,
        strike / rate / callput through the forward-fill index, Greeks by greeks_oracle."""
        import greeks_oracle as GO
        out, status = O.interp1d_batch(xk, yk, knot_off, None, q_off, code)
        fidx = self.ffill_index_batch(src_pos, knot_off, gvalid, q_off, total_q)
        vals = out.copy()
        for s in range(len(knot_off) - 1):
            a, b = int(knot_off[s]), int(knot_off[s + 1])
            rows = int(q_off[s]) + src_pos[a:b]
            for c in range(yk.shape[0]):
                src = yk[c, a:b]
                ok = ~np.isnan(src)
                vals[c, rows[ok]] = src[ok]
        nanv = np.nan
        K = np.where(fidx[0] >= 0, strike_src[np.clip(fidx[0], 0, None)], nanv)
        r = np.where(fidx[1] >= 0, rate_src[np.clip(fidx[1], 0, None)], nanv)
        put = np.where(fidx[2] >= 0, put_src[np.clip(fidx[2], 0, None)], 2)
        with np.errstate(all="ignore"):
            g = GO.calculate_greeks(vals[1], K, vals[2], r, vals[0], put == 1)
        gr = np.stack([g[k] for k in ("delta", "gamma", "theta", "vega", "rho")])
        gr[:, put == 2] = np.nan
        return out, status, gr

    def ffill_index_batch(self, src_pos, src_off, valid, q_off, total_q):
        n_cols = valid.shape[0]
        idx = np.full((n_cols, total_q), -1, np.int32)
        for s in range(len(src_off) - 1):
            a, b = int(src_off[s]), int(src_off[s + 1])
            m = int(q_off[s + 1] - q_off[s])
            last = np.searchsorted(src_pos[a:b], np.arange(m), side="right") - 1      # last source row <= i
            for c in range(n_cols):
                v = valid[c, a:b].astype(bool)
                prev_valid = np.where(v, np.arange(b - a), -1)
                prev_valid = np.maximum.accumulate(prev_valid)                          # last valid row <= j
                r = np.where(last >= 0, prev_valid[np.clip(last, 0, None)], -1)
                idx[c, q_off[s]:q_off[s + 1]] = np.where(r >= 0, r + a, -1)
        return idx


class OracleBridgeBackend:
    """Test-only stand-in for HipBridgeBackend: the bridge oracle on NumPy's own legacy generator, positioned at
    word `pos` of the stream of `seed` with the given cached normal deviate."""

    def bridge_candles(self, price, volume, row_off, strategy, seed, pos, tail, base_spread_pct, vol_factor):
        import bridge_oracle as BO
        rs = np.random.RandomState(seed)
        if pos:
            rs.bytes(4 * pos)
        st = rs.get_state()
        rs.set_state((st[0], st[1], st[2], int(tail[0]), float(np.int64(tail[1]).view(np.float64))))
        total = len(price)
        valid = np.zeros(total, bool)
        out = np.full((6, total), np.nan)
        for s in range(len(row_off) - 1):
            a, b = int(row_off[s]), int(row_off[s + 1])
            v, d = BO.candles(price[a:b], volume[a:b], strategy, rs=rs, base_spread_pct=base_spread_pct, vol_factor=vol_factor)
            valid[a:b] = v
            idx = a + np.flatnonzero(v)
            for j, k in enumerate(BO.OUT_COLS):
                out[j, idx] = d[k]
        st = rs.get_state()
        tail_out = (int(st[3]), int(np.float64(st[4]).view(np.int64)))
        # words consumed: locate the generator's next four words in the stream of `seed`
        probe = np.random.RandomState(); probe.set_state(st)
        nxt = np.frombuffer(probe.bytes(16), dtype="<u4")
        bound = pos + 16 * total + 64
        while True:
            w = BO.mt19937_words(seed, bound + 4)
            hit = np.flatnonzero((w[pos:-3] == nxt[0]) & (w[pos + 1:-2] == nxt[1]) & (w[pos + 2:-1] == nxt[2]) & (w[pos + 3:] == nxt[3]))
            if len(hit):
                return out, valid, pos + int(hit[0]), tail_out
            bound *= 2


class OracleCandleBackend:
    """Test-only stand-in for HipCandleBackend: the sparse per-row layout of ivs_candle_aggregate_f64 from NumPy
    (bucket heads carry the candle and the group's row count, all other rows count 0)."""

    def candle_aggregate(self, ts, cols, off, minutes):
        import candles_oracle as CO
        n = len(ts)
        f = minutes * 60_000_000_000
        out_ts = np.zeros(n, np.int64); outs = [np.full(n, np.nan) for _ in range(5)]; cnt = np.zeros(n, np.int32)
        for s in range(len(off) - 1):
            a, b = int(off[s]), int(off[s + 1])
            bucket = (ts[a:b] // f) * f
            starts = np.flatnonzero(np.r_[True, bucket[1:] != bucket[:-1]]) if b > a else np.zeros(0, np.int64)
            ends = np.r_[starts[1:], b - a]
            for i, j in zip(starts, ends):
                o, h, l, c, v = [x[a + i:a + j] for x in cols]
                first = o[~np.isnan(o)]; last = c[~np.isnan(c)]
                out_ts[a + i] = bucket[i]; cnt[a + i] = j - i
                outs[0][a + i] = first[0] if first.size else np.nan
                outs[1][a + i] = np.nanmax(h) if (~np.isnan(h)).any() else np.nan
                outs[2][a + i] = np.nanmin(l) if (~np.isnan(l)).any() else np.nan
                outs[3][a + i] = last[-1] if last.size else np.nan
                outs[4][a + i] = CO._kahan_nansum(v)
        return [out_ts, *outs, cnt]
