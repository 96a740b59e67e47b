"""Drop-in replacement for the reference's ``src/interpolation/core.py``.

Same class, same constructor, same ``interpolate_symbol(df) -> Optional[DataFrame]`` contract
(reference core.py:9-85; callers: complete_pipeline.py:318, batch_processor.py:97,
optimized_batch_processor.py:303,349), but the arithmetic -- the three interpolated channels
(core.py:58-61) and the forward-fill gather index (core.py:64-68) -- runs in HIP kernels on an
MI355X through the C ABI in include/ivs.h.  ``interpolate_batch`` is the columnar many-symbols
extension: every symbol of the batch goes to the device in ONE launch
per kernel instead of one DataFrame round trip per symbol.

Host code does what the reference's pandas calls do around the numbers (rules R1-R14 of
SURVEY.md section 8a): guards, sort, minute lattice, exact-timestamp join with duplicate expansion,
column order / dtypes / index of the result.  There is no CPU fallback for the numerics:
a missing libivs.so or GPU raises ``EngineUnavailable`` (it is NOT turned into ``None``).
"""
from __future__ import annotations

import gc
import logging
from datetime import timedelta
from typing import List, Optional, Sequence

import numpy as np
import pandas as pd

from ._lib import BFILL, METHOD_CODES, PAD, method_code, POLY_MAX_KNOTS, ST_ILL_CONDITIONED, ST_OK, EngineUnavailable

logger = logging.getLogger("interpolation.core")   # the reference's logger name (core.py:7)

NUMERIC_COLS = ["iv", "underlying_price", "time_to_maturity"]                      # core.py:58
FILL_COLS = ["symbol", "strike", "callput", "interest_rate", "mark_price",         # core.py:64-65
             "index_price", "volume", "quote_volume", "record_time"]
REQUIRED = ["symbol", "iv", "underlying_price", "time_to_maturity"]                 # core.py:74
MINUTE_NS = 60_000_000_000
GREEK_COLS = ["delta", "gamma", "theta", "vega", "rho"]                              # schema.py:36-40

# pandas accepts these names; the ones not in METHOD_CODES are not implemented by the engine yet
_PANDAS_METHODS = ["linear", "time", "index", "values", "nearest", "zero", "slinear", "quadratic", "cubic",
                   "barycentric", "krogh", "spline", "polynomial", "from_derivatives", "piecewise_polynomial",
                   "pchip", "akima", "cubicspline"]
# 'pad' / 'ffill' / 'bfill' / 'backfill': Series.interpolate still runs pandas' fill methods (pad_or_backfill, with a
# FutureWarning) -- and, unlike the interpolation methods, fills OBJECT columns too and then soft-converts them
_FILL_METHOD_CODES = (PAD, BFILL)


def _chan_kind(col: pd.Series, fill_method: bool = False) -> str:
    """How pandas' Series.interpolate treats a numeric channel of this dtype (reference core.py:61; verified against the
    real reference, golden cases t1/t2/t3/z*): 'obj' = object dtype: interpolate is a (deprecated) no-op, the column keeps
    its source cells -- except under the fill methods ('objfill'): pad_or_backfill fills object cells like any other and
    soft-converts the result (all floats -> float64); 'f32' = float32: computed in float64 from the upcast knots, stored
    as float32; 'ext' = nullable Float64 / Float32: result keeps the extension dtype; 'f64' = everything else (float64,
    ints: existing rules)."""
    dt = col.dtype
    if dt == object:
        return "objfill" if fill_method else "obj"
    if isinstance(dt, pd.api.extensions.ExtensionDtype):
        return "ext" if str(dt) in ("Float64", "Float32") else "f64"
    return "f32" if dt == np.float32 else "f64"


# interpolate_batch takes the columnar path from this many frames on (1: interpolate_symbol too -- the path is the same code
# for one frame, its vectorised bookkeeping replaces ~60 small pandas calls of the per-symbol bookkeeping)
_COLUMNAR_MIN_FRAMES = 1


def _concat_same_schema(frames, columns, dtypes) -> Optional[pd.DataFrame]:
    """``pd.concat(frames, ignore_index=True)`` for frames that share columns, dtypes AND block layout -- the callers' usual
    case: thousands of small frames cut from one query result -- or None when any frame differs (pd.concat decides then).
    pd.concat spends ~50 us per frame planning the join (column union, per-block join units); here a frame costs one
    look at its blocks, and every block position is one np.concatenate.  Reads the frames' block managers (the arrays
    pandas itself holds); builds the result through the public constructor."""
    try:
        b0 = frames[0]._mgr.blocks
        sig0 = [(b.dtype, b.mgr_locs.as_array.tobytes()) for b in b0]
        if not all(isinstance(d, np.dtype) for d, _ in sig0):           # extension blocks (tz-aware dates, nullable floats)
            return None
        nb = len(b0)
        parts = [[] for _ in range(nb)]
        for f in frames:
            bl = f._mgr.blocks
            if len(bl) != nb:
                return None
            for k in range(nb):
                b = bl[k]
                if b.dtype != sig0[k][0] or b.mgr_locs.as_array.tobytes() != sig0[k][1]:
                    return None
                parts[k].append(b.values)
        cols = [None] * len(columns)
        for k in range(nb):
            v0 = parts[k][0]
            arrs = parts[k] if isinstance(v0, np.ndarray) else [np.asarray(v) for v in parts[k]]   # DatetimeArray -> M8[ns]
            cat = np.concatenate(arrs, axis=1) if arrs[0].ndim == 2 else None
            if cat is None or cat.dtype != sig0[k][0]:
                return None
            for r, ci in enumerate(b0[k].mgr_locs.as_array):
                cols[int(ci)] = cat[r]
        if any(c is None for c in cols):
            return None
        data = pd.DataFrame(dict(zip(range(len(cols)), cols)), copy=False)
        data.columns = columns
        return data if list(data.dtypes) == list(dtypes) else None
    except (AttributeError, TypeError, ValueError):                       # another pandas: its own concat
        return None


def _objfill_source(cells: np.ndarray) -> np.ndarray:
    """Object channel under a fill method: the device fills ROW NUMBERS (exact in float64) instead of values -- the
    arithmetic-free fill rule is the same, and the host then takes the object cells through the filled numbers."""
    return np.where(np.asarray(pd.isna(cells)), np.nan, np.arange(len(cells), dtype=np.float64))


def _objfill_finish(filled_rows: np.ndarray, cells: np.ndarray) -> np.ndarray:
    """Filled row numbers -> object cells, then pandas' soft conversion of a filled object block (Block._maybe_downcast
    -> Block.convert -> lib.maybe_convert_objects: all floats become float64, strings stay object)."""
    miss = np.isnan(filled_rows)
    col = np.empty(len(filled_rows), dtype=object)
    col[:] = np.nan
    col[~miss] = cells[filled_rows[~miss].astype(np.int64)]
    return pd.Series(col, dtype=object).infer_objects().to_numpy()


def _chan_finish(values: np.ndarray, kind: str, dtype):
    """float64 result column -> the dtype the reference returns for this channel."""
    if kind == "f32":
        return values.astype(np.float32)
    if kind == "ext":
        return pd.array(values.astype(np.float32 if str(dtype) == "Float32" else np.float64), dtype=str(dtype))
    return values


class HipBackend:
    """Moves packed host columns to the current HIP device and calls the kernels.  Uploads of the (pageable, temporary)
    packed arrays are ordinary blocking copies: an asynchronous copy would outlive the temporary it reads from.  Results
    come back through pinned buffers with asynchronous copies behind the kernels (frame_columns)."""

    def interp1d_batch(self, xk, yk, knot_off, q_off, total_q, code):
        from . import engine
        torch = engine.require_device()
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        out, st = engine.interp1d_batch(d(xk), d(yk), d(knot_off), d(q_off), int(total_q), code)
        return out.cpu().numpy(), st.cpu().numpy()

    def ffill_index_batch(self, src_pos, src_off, valid, q_off, total_q):
        from . import engine
        torch = engine.require_device()
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        return engine.ffill_index_batch(d(src_pos), d(src_off), d(valid), d(q_off), int(total_q)).cpu().numpy()

    def interp1d_greeks_batch(self, xk, yk, knot_off, q_off, total_q, code, src_pos, gvalid, strike_src, rate_src, put_src):
        """Channels + Greeks epilogue in one pass: the forward-fill index of (strike, interest_rate, callput) is computed
        on the device and consumed by the eval kernel without leaving it."""
        from . import engine
        torch = engine.require_device()
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        ko, qo = d(knot_off), d(q_off)
        fidx = engine.ffill_index_batch(d(src_pos), ko, d(gvalid), qo, int(total_q))
        out, st, gr = engine.interp1d_greeks_batch(d(xk), d(yk), ko, qo, int(total_q), code, (0, 1, 2), fidx, (0, 1, 2),
                                                   d(strike_src), d(rate_src), d(put_src))
        return out.cpu().numpy(), st.cpu().numpy(), gr.cpu().numpy()


    fused = True       # one pass over the output rows (ivs_frame_columns_f64); False: the five separate calls of rounds 1-2 (A/B, tests)

    def frame_columns(self, pos, chan, src_off, q_off, total_q, code, valid, fsrc, f_rows, csrc, c_rows, host_rows, greek,
                      rows_info=None):
        """Everything interpolate_frame needs from the device in ONE round trip: the three merged channel columns, the
        forward-filled numeric columns (gathered on the device), the codes of the forward-filled object columns, the raw
        gather-index rows the host still wants (`host_rows`), optionally the Greeks.  Results land in pinned host memory
        through asynchronous copies behind the kernels (one synchronisation); the NumPy arrays returned are views of it."""
        from . import engine
        torch = engine.require_device()
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
        total_q = int(total_q)
        fidx = gr = dts = kp = None
        if self.fused:
            vall, sym_col = valid, -1
            host = {"ko": src_off, "qo": q_off, "pos": pos, "yk": np.stack(chan)}
            if greek is not None:
                gvalid, ksrc, rsrc, psrc = greek
                nv = valid.shape[0]
                vall = np.concatenate([valid, gvalid]) if nv else gvalid
                host.update(ksrc=ksrc, rsrc=rsrc, psrc=psrc)
            if vall.shape[0]:
                host["vall"] = vall
            if len(f_rows):
                host.update(fsrc=fsrc, f_rows=np.asarray(f_rows, np.int32))
            if len(c_rows):
                host.update(csrc=csrc, c_rows=np.asarray(c_rows, np.int32))
            if len(host_rows):
                host["host_rows"] = np.asarray(host_rows, np.int32)
            if rows_info is not None:
                fns, nds, sym_row = rows_info
                host.update(first_ns=fns, needs=np.ascontiguousarray(nds).astype(np.uint8))
                sym_col = -1 if sym_row is None else int(sym_row)
            dv = _upload(torch, host)
            g = ((nv, nv + 1, nv + 2), dv["ksrc"], dv["rsrc"], dv["psrc"]) if greek is not None else None
            ko, qo, pos_d, yk = dv["ko"], dv["qo"], dv["pos"], dv["yk"]
            r = engine.frame_columns(pos_d, ko, qo, total_q, yk, code, dv.get("vall"), dv.get("fsrc"), dv.get("f_rows"),
                                     dv.get("csrc"), dv.get("c_rows"), dv.get("host_rows"), dv.get("first_ns"), dv.get("needs"),
                                     sym_col, g)
            out, st, F, Cc, gr, dts, kp = r["chan"], r["status"], r["F"], r["C"], r["greeks"], r["date_ns"], r["keep"]
            idx_rows = r["idx"]
        else:
            ko, qo, pos_d, yk = d(src_off), d(q_off), d(pos), d(np.stack(chan))
            fidx = engine.ffill_index_batch(pos_d, ko, d(valid), qo, total_q) if valid.shape[0] else None
            xk = pos_d.to(torch.float64)
            if greek is not None:
                gvalid, ksrc, rsrc, psrc = greek
                gidx = engine.ffill_index_batch(pos_d, ko, d(gvalid), qo, total_q)
                out, st, gr = engine.interp1d_greeks_batch(xk, yk, ko, qo, total_q, code, (0, 1, 2), gidx, (0, 1, 2),
                                                           d(ksrc), d(rsrc), d(psrc))
            else:
                out, st = engine.interp1d_batch(xk, yk, ko, qo, total_q, code)
            F = engine.gather_rows(d(fsrc), fidx, d(np.asarray(f_rows, np.int32))) if len(f_rows) else None
            Cc = engine.gather_rows(d(csrc), fidx, d(np.asarray(c_rows, np.int32))) if len(c_rows) else None
            if rows_info is not None:                              # timestamps + keep flags formed on the device
                first_ns, needs, sym_row = rows_info
                sym_code = Cc[sym_row] if (Cc is not None and sym_row is not None) else None
                dts, kp = engine.frame_rows(qo, d(first_ns), out, sym_code, st, d(needs.astype(np.uint8)))
            idx_rows = None if not len(host_rows) else torch.stack([fidx[r] for r in host_rows])
        n64 = 3 + (len(f_rows) if F is not None else 0) + (5 if gr is not None else 0)
        h64 = torch.empty((n64, total_q), dtype=torch.float64, pin_memory=True)
        h64[:3].copy_(out, non_blocking=True)
        k = 3
        if F is not None:
            h64[k:k + len(f_rows)].copy_(F, non_blocking=True); k += len(f_rows)
        if gr is not None:
            h64[k:k + 5].copy_(gr, non_blocking=True)
        dates_h = keep_h = None
        if dts is not None:
            dates_h = torch.empty(total_q, dtype=torch.int64, pin_memory=True); dates_h.copy_(dts, non_blocking=True)
            keep_h = torch.empty(total_q, dtype=torch.uint8, pin_memory=True); keep_h.copy_(kp, non_blocking=True)
        n32 = (len(c_rows) if Cc is not None else 0) + len(host_rows)
        h32 = torch.empty((max(n32, 1), total_q), dtype=torch.int32, pin_memory=True)
        j = 0
        if Cc is not None:
            h32[:len(c_rows)].copy_(Cc, non_blocking=True); j = len(c_rows)
        if idx_rows is not None:
            h32[j:j + len(host_rows)].copy_(idx_rows, non_blocking=True)
        st_h = st.cpu()                                            # synchronises the stream: every copy above is complete
        a64, a32 = h64.numpy(), h32.numpy()
        res = {"chan": a64[:3], "status": st_h.numpy(), "F": a64[3:3 + len(f_rows)] if F is not None else None,
               "C": a32[:len(c_rows)] if Cc is not None else None,
               "rows": {r: a32[(len(c_rows) if Cc is not None else 0) + i] for i, r in enumerate(host_rows)},
               "greeks": a64[n64 - 5:] if gr is not None else None,
               "date_ns": None if dates_h is None else dates_h.numpy(),
               "keep": None if keep_h is None else keep_h.numpy().view(np.bool_)}
        return res


def _upload(torch, host):
    """{name: host array} -> {name: device tensor}.  A small call (interpolate_symbol: one symbol) sends everything in ONE
    blocking copy of a packed buffer and hands out views of it -- a dozen separate copies of a few hundred bytes cost ~15 us
    each, a sixth of such a call; large calls copy array by array (packing would add a pass over tens of megabytes)."""
    arrs = {k: np.ascontiguousarray(v) for k, v in host.items()}
    if sum(a.nbytes for a in arrs.values()) >= (1 << 20):
        return {k: torch.from_numpy(a).cuda() for k, a in arrs.items()}
    offs, n = {}, 0
    for k, a in arrs.items():
        offs[k] = n
        n += (a.nbytes + 15) & ~15
    buf = np.empty(max(n, 16), np.uint8)
    for k, a in arrs.items():
        buf[offs[k]:offs[k] + a.nbytes] = a.reshape(-1).view(np.uint8)
    dev = torch.from_numpy(buf).cuda()
    out = {}
    for k, a in arrs.items():
        t = dev[offs[k]:offs[k] + a.nbytes]
        out[k] = (t.view(getattr(torch, str(a.dtype))) if a.nbytes else torch.empty(0, dtype=getattr(torch, str(a.dtype)), device="cuda")).reshape(a.shape)
    return out


def _frame_columns_generic(be, pos, chan, src_off, q_off, total_q, code, valid, fsrc, f_rows, csrc, c_rows, host_rows, greek,
                           rows_info=None):
    """frame_columns for backends that only answer the two primitive calls (the oracle backend of the CPU tests): same
    results, formed with NumPy."""
    gr = None
    if greek is not None:
        gvalid, ksrc, rsrc, psrc = greek
        out, status, gr = be.interp1d_greeks_batch(pos.astype(np.float64), np.stack(chan), src_off, q_off, total_q, code,
                                                  pos.astype(np.int64), gvalid, ksrc, rsrc, psrc)
    else:
        out, status = be.interp1d_batch(pos.astype(np.float64), np.stack(chan), src_off, q_off, total_q, code)
    out = np.array(out, copy=True)
    ks = np.repeat(np.arange(len(src_off) - 1), np.diff(src_off))
    gpos = q_off[:-1][ks] + pos
    for ci in range(3):                                                # rows that are knots keep their source cell
        okk = ~np.isnan(chan[ci])
        out[ci, gpos[okk]] = chan[ci][okk]
    fidx = be.ffill_index_batch(pos.astype(np.int64), src_off, valid, q_off, total_q) if valid.shape[0] else None

    def take(src, rows, missing):
        o = np.empty((len(rows), total_q), src.dtype)
        for k, r in enumerate(rows):
            i = fidx[r]
            o[k] = np.where(i >= 0, src[k][np.clip(i, 0, None)], missing)
        return o
    return {"chan": out, "status": status, "F": take(fsrc, f_rows, np.nan) if len(f_rows) else None,
            "C": take(csrc, c_rows, -1) if len(c_rows) else None, "rows": {r: fidx[r] for r in host_rows}, "greeks": gr,
            "date_ns": None, "keep": None}


def _greek_sources(frames_src, n_rows):
    """Source-row arrays of the three option attributes the Greeks need, for symbols packed back to back.
    frames_src: per symbol a dict column -> ndarray (only the on-lattice source rows).  Returns (valid uint8 [3, n],
    strike f64, rate f64, put uint8 0 call / 1 put / 2 null).  A symbol WITHOUT the column gets the schema default
    (interest_rate 0.0: schema.py:34; callput: call) -- except strike, without which there are no Greeks (NaN)."""
    n = int(sum(n_rows))
    valid = np.zeros((3, n), np.uint8); strike = np.full(n, np.nan); rate = np.zeros(n); put = np.zeros(n, np.uint8)
    a = 0
    for src, m in zip(frames_src, n_rows):
        b = a + int(m)
        if "strike" in src:
            v = pd.to_numeric(pd.Series(src["strike"]), errors="coerce").to_numpy(np.float64, na_value=np.nan)
            strike[a:b] = v; valid[0, a:b] = ~np.isnan(v)
        if "interest_rate" in src:
            v = pd.to_numeric(pd.Series(src["interest_rate"]), errors="coerce").to_numpy(np.float64, na_value=np.nan)
            rate[a:b] = v; valid[1, a:b] = ~np.isnan(v)
        else:
            valid[1, a:b] = 1
        if "callput" in src:
            cp = pd.Series(src["callput"])
            null = cp.isna().to_numpy()
            is_call = cp.astype(str).str.lower().str.startswith("c").to_numpy()      # option_type == 'call' else put (greeks.py:24)
            put[a:b] = np.where(null, 2, np.where(is_call, 0, 1)); valid[2, a:b] = ~null
        else:
            valid[2, a:b] = 1
        a = b
    return valid, strike, rate, put


class _Prepared:
    __slots__ = ("df", "timeline", "pos", "rowlat", "M", "chan_src", "chan_needs", "chan_kind", "fill_cols",
                 "fill_valid", "src_np", "int_dtype")


class IVInterpolator:
    """Core interpolation engine for IV data (MI355X-native)."""

    def __init__(self, method: str = "linear", min_points: int = 10, backend=None, preserve_greeks: bool = False):
        self.method = method
        self.min_points = min_points
        self._backend = backend          # None -> HipBackend on first use
        # reference config.py:46 `preserve_greeks` ("Recalculate Greeks after interpolation"): the reference declares the
        # flag and the delta..rho columns (schema.py:36-40) but wires nothing up.  When set, every output row carries the
        # five Black-Scholes Greeks of greeks.py:12-43, formed in the interpolation kernel's epilogue.
        self.preserve_greeks = bool(preserve_greeks)

    # ------------------------------------------------------------------ public API
    def interpolate_symbol(self, symbol_data: pd.DataFrame) -> Optional[pd.DataFrame]:
        """Interpolate IV data for a single symbol (hourly rows -> 1-minute rows) or None."""
        return self.interpolate_batch([symbol_data])[0]

    def interpolate_batch(self, frames: Sequence[pd.DataFrame]) -> List[Optional[pd.DataFrame]]:
        """Many symbols, one device round trip.  Element i is what interpolate_symbol(frames[i]) returns."""
        # The columnar path creates a few small long-lived pandas objects per symbol and no cyclic garbage; with the cyclic
        # collector running, its passes over the caller's thousands of frames (and over the results made so far) are HALF
        # of the call (2048 symbols: 0.24 s vs 0.13 s, tools/profile_batch.py).  Paused for the duration, state restored.
        gc_was = gc.isenabled()
        if gc_was:
            gc.disable()
        try:
            fast = self._batch_via_frame(frames)
        finally:
            if gc_was:
                gc.enable()
        if fast is not None:
            return fast
        return self._batch_per_symbol(frames)

    def _batch_per_symbol(self, frames: Sequence[pd.DataFrame]) -> List[Optional[pd.DataFrame]]:
        """The reference's bookkeeping statement by statement per frame (guards and their log lines included), the device
        work of all frames in one round trip."""
        preps: List[Optional[_Prepared]] = []
        for f in frames:
            try:
                preps.append(self._prepare(f))
            except EngineUnavailable:
                raise
            except Exception as e:                                       # core.py:83-85
                logger.error(f"Interpolation failed: {e}")
                preps.append(None)
        live = [i for i, p in enumerate(preps) if p is not None]
        results: List[Optional[pd.DataFrame]] = [None] * len(frames)
        if not live:
            return results
        be = self._backend or HipBackend()
        code = method_code(self.method)
        # ---- pack (CSR over symbols)
        ps = [preps[i] for i in live]
        n_src = np.array([len(p.pos) for p in ps], np.int64)
        n_out = np.array([p.M for p in ps], np.int64)
        src_off = np.concatenate([[0], np.cumsum(n_src)]).astype(np.int64)
        q_off = np.concatenate([[0], np.cumsum(n_out)]).astype(np.int64)
        total_q = int(q_off[-1])
        xk = np.concatenate([p.pos for p in ps]).astype(np.float64)
        yk = np.stack([np.concatenate([p.chan_src[c] for p in ps]) for c in range(len(NUMERIC_COLS))])
        greeks = None
        if self.preserve_greeks:
            gvalid, ksrc, rsrc, psrc = _greek_sources([p.src_np for p in ps], n_src)
            out, status, greeks = be.interp1d_greeks_batch(xk, yk, src_off, q_off, total_q, code, xk.astype(np.int64),
                                                          gvalid, ksrc, rsrc, psrc)
        else:
            out, status = be.interp1d_batch(xk, yk, src_off, q_off, total_q, code)
        ncols = max((len(p.fill_cols) for p in ps), default=0)
        idx = None
        if ncols:
            valid = np.zeros((ncols, int(src_off[-1])), np.uint8)
            for k, p in enumerate(ps):
                for c in range(len(p.fill_cols)):
                    valid[c, src_off[k]:src_off[k + 1]] = p.fill_valid[c]
            src_pos = np.concatenate([p.pos for p in ps]).astype(np.int64)
            idx = be.ffill_index_batch(src_pos, src_off, valid, q_off, total_q)
        # ---- unpack
        for k, i in enumerate(live):
            p = ps[k]
            try:
                results[i] = self._assemble(p, out[:, q_off[k]:q_off[k + 1]], status[k],
                                            None if idx is None else idx[:, q_off[k]:q_off[k + 1]] - src_off[k],
                                            None if idx is None else idx[:, q_off[k]:q_off[k + 1]] < 0,
                                            None if greeks is None else greeks[:, q_off[k]:q_off[k + 1]])
            except EngineUnavailable:
                raise
            except Exception as e:                                       # core.py:83-85
                logger.error(f"Interpolation failed: {e}")
                results[i] = None
        return results

    def _batch_via_frame(self, frames) -> Optional[List[Optional[pd.DataFrame]]]:
        """interpolate_batch through the columnar path: the frames become ONE long frame grouped by frame
        number, one pass of vectorised bookkeeping and one device round trip, the result is cut back into per-symbol frames
        (views of the long result).  Only for the plain case -- frames with identical columns and dtypes, a datetime64 date
        column, an implemented method; anything else (and any frame the long frame cannot represent exactly, and every
        frame whose answer is None: its log line) goes through the per-symbol bookkeeping, which is the reference's
        contract statement by statement.  interpolate_symbol arrives here with one frame."""
        if len(frames) < _COLUMNAR_MIN_FRAMES:
            return None
        try:
            code = method_code(self.method)
        except KeyError:
            return None
        f0 = frames[0]
        if not isinstance(f0, pd.DataFrame) or "date" not in f0.columns or any(c not in f0.columns for c in REQUIRED):
            return None
        if not str(f0["date"].dtype).startswith("datetime64") or f0.columns.duplicated().any():
            return None
        c0, dt0 = f0.columns, list(f0.dtypes)
        for f in frames:                                                 # same columns in the same order (pd.concat would align by name)
            if not isinstance(f, pd.DataFrame) or not (f.columns is c0 or f.columns.equals(c0)):
                return None
        lens = np.array([len(f) for f in frames], np.int64)
        if int(lens.sum()) == 0:
            return None
        # one schema for all (dtype rules are per symbol); a single frame is its own long frame (nothing below writes to it)
        # (the bookkeeping below works on positions: the frame's own index is never looked at)
        data = f0 if len(frames) == 1 else _concat_same_schema(frames, c0, dt0)
        if data is None:
            if any(list(f.dtypes) != dt0 for f in frames):
                return None
            data = pd.concat(frames, ignore_index=True, copy=False)
            if list(data.dtypes) != dt0:
                return None
        grp = np.repeat(np.arange(len(frames), dtype=np.int64), lens)
        be = self._backend or HipBackend()
        try:
            out = self._frame_impl(data, grp, len(frames), lens, None, be, code, split=True)
        except EngineUnavailable:
            raise
        except Exception as e:                                           # anything unusual: the per-symbol bookkeeping decides
            logger.debug(f"columnar batch path declined: {e}")
            return None
        # NotImplemented: a frame the long frame cannot represent exactly.  None: a guard or a failed solve -- the per-symbol
        # bookkeeping says so on the logger like the reference does (core.py:27,38,50,77,84); those frames stop at its guards
        redo = [i for i, r in enumerate(out) if r is NotImplemented or r is None]
        if redo:
            again = self._batch_per_symbol([frames[i] for i in redo])
            for i, r in zip(redo, again):
                out[i] = r
        return out

    def interpolate_frame(self, data: pd.DataFrame) -> pd.DataFrame:
        """Columnar ingest/egress: ALL symbols of one long frame -- what the reference reads
        with ``SELECT ... FROM trading_tickers ORDER BY symbol, date`` -- in one pass of vectorised NumPy bookkeeping
        and one device round trip, instead of one DataFrame per symbol (batch_processor.py:67-142 loops over symbols,
        then ``iterrows()`` :166-173).  The result equals
        ``pd.concat([interpolate_symbol(g) for _, g in data.groupby("symbol", sort=True)], ignore_index=True)``
        (symbols whose result is ``None`` contribute nothing; ties between duplicate timestamps are ordered like the
        reference's ``sort_values('date')``).  Rows with a null symbol or an unparsable date are ignored."""
        for c in ["date"] + REQUIRED:
            if c not in data.columns:
                raise KeyError(c)
        try:
            code = method_code(self.method)
        except KeyError:
            raise ValueError(f"method '{self.method}' is not implemented by the MI355X engine") from None
        be = self._backend or HipBackend()
        if data["date"].dtype == object:
            # string dates: the reference sorts them LEXICOGRAPHICALLY before parsing (core.py:32-33); that order is a
            # per-symbol property of the strings, so these frames take the per-symbol bookkeeping (one device round trip)
            parts = [r for r in self.interpolate_batch([g for _, g in data.groupby("symbol", sort=True)]) if r is not None]
            if parts:
                return pd.concat(parts, ignore_index=True)
            cols_e = ["date"] + [c for c in data.columns if c != "date"]
            cols_e += [] if "is_interpolated" in cols_e else ["is_interpolated"]
            return pd.DataFrame({c: pd.Series(dtype=("datetime64[ns]" if c == "date" else bool if c == "is_interpolated"
                                                     else data[c].dtype)) for c in cols_e})
        sym_codes, sym_uniques = pd.factorize(data["symbol"], sort=True)
        return self._frame_impl(data, sym_codes, len(sym_uniques), None, (sym_codes, sym_uniques), be, code)

    def _frame_impl(self, data, grp, n_grp, grp_len, sym_fact, be, code, split=False):
        """The columnar path on EXPLICIT groups: `grp` [rows] = group number of every row of `data` (-1: ignored), groups
        in output order.  interpolate_frame groups by the symbol column (sym_fact = its factorisation, reused for the
        column's codes); interpolate_batch groups by frame number (grp_len = the frames' lengths for the min_points guard,
        the symbol column is then just another forward-filled column).  split: return the per-group results as a list."""
        cols_in = [c for c in data.columns if c != "date"]
        out_cols = ["date"] + cols_in + (["is_interpolated"] if "is_interpolated" not in cols_in else [])
        d_idx = pd.DatetimeIndex(pd.to_datetime(data["date"]))
        tz = d_idx.tz
        d_ns = d_idx.as_unit("ns").asi8
        sym_codes = grp
        ok_row = (sym_codes >= 0) & ~np.asarray(d_idx.isna())
        rows = np.flatnonzero(ok_row)
        rows = rows[np.lexsort((d_ns[rows], sym_codes[rows]))]          # by symbol, then date (stable)
        sc = sym_codes[rows]; dn = d_ns[rows]
        tie = (sc[1:] == sc[:-1]) & (dn[1:] == dn[:-1])
        if tie.any():
            # duplicate timestamps inside a symbol: the reference orders a symbol's rows with sort_values('date'), i.e.
            # numpy's quicksort on the datetime64 values (pandas nargsort; NOT the int64 view, whose vectorised sort
            # leaves other tie orders) -- not stable beyond 16 rows -- and the tie order decides which duplicate lands on
            # which merged-frame position (R7).  Same routine on the same values in the same input order.
            d64 = d_ns.view("datetime64[ns]")
            for s_id in np.unique(sc[1:][tie]):
                lo_ = int(np.searchsorted(sc, s_id, side="left")); hi_ = int(np.searchsorted(sc, s_id, side="right"))
                grp = np.sort(rows[lo_:hi_])                             # the symbol's rows in input order
                rows[lo_:hi_] = grp[np.argsort(d64[grp], kind="quicksort")]
            dn = d_ns[rows]
        S_all = int(n_grp)
        def empty():                                                     # (built on demand: 0.7 ms of pandas per call otherwise)
            return pd.DataFrame({c: pd.Series(dtype=(bool if c == "is_interpolated" else data[c].dtype if c in data.columns else "float64"))
                                 for c in out_cols})
        if len(rows) == 0:
            return [None] * S_all if split else empty()
        start = np.searchsorted(sc, np.arange(S_all), side="left")
        count = np.searchsorted(sc, np.arange(S_all), side="right") - start
        present = count > 0
        first_ns = np.where(present, dn[np.minimum(start, len(dn) - 1)], 0)
        last_ns = np.where(present, dn[np.minimum(start + count - 1, len(dn) - 1)], 0)
        span = last_ns - first_ns
        n_rows_grp = count if grp_len is None else np.asarray(grp_len, np.int64)                     # len(symbol_data), core.py:26
        keep_sym = present & (n_rows_grp >= self.min_points) & (span <= 30 * 24 * 3600 * 1_000_000_000)   # core.py:26-28, 36-39
        m0 = span // MINUTE_NS + 1
        keep_sym &= m0 <= 100000                                                                     # core.py:49-51
        rel = dn - first_ns[sc]
        on = keep_sym[sc] & (rel % MINUTE_NS == 0)                        # off-lattice rows vanish (R6)
        ridx = np.flatnonzero(on)
        if len(ridx) == 0:
            return [None] * S_all if split else empty()
        rsym = sc[ridx]; lat = rel[ridx] // MINUTE_NS
        # compact symbol numbering over the kept symbols
        kept = np.flatnonzero(keep_sym)
        renum = np.full(S_all, -1, np.int64); renum[kept] = np.arange(len(kept))
        ks = renum[rsym]                                                 # kept-symbol id of every on-lattice source row
        S = len(kept)
        q = np.bincount(ks, minlength=S).astype(np.int64)                # on-lattice rows per symbol (>= 1: the first row)
        src_off = np.concatenate([[0], np.cumsum(q)]).astype(np.int64)
        newsym = np.ones(len(ridx), bool); newsym[1:] = ks[1:] != ks[:-1]
        first = newsym.copy(); first[1:] |= lat[1:] != lat[:-1]          # first source row of its lattice point
        cum_first = np.cumsum(first)
        u_within = cum_first - (cum_first[src_off[:-1]] - 1)[ks]         # distinct lattice points so far, within the symbol
        i_within = np.arange(len(ridx)) - src_off[:-1][ks]
        pos = lat + i_within + 1 - u_within                              # merged-frame position (R7, R8)
        m0k = m0[kept]
        M = m0k + (q - u_within[src_off[1:] - 1])
        q_off = np.concatenate([[0], np.cumsum(M)]).astype(np.int64)
        total_q = int(q_off[-1])
        src_rows = rows[ridx]                                            # positions in `data` of the on-lattice rows
        # ---- channels
        kinds = [_chan_kind(data[c], code in _FILL_METHOD_CODES) for c in NUMERIC_COLS]
        chan = [np.zeros(len(src_rows)) if k == "obj" else
                _objfill_source(data[c].to_numpy()[src_rows]) if k == "objfill" else
                data[c].to_numpy(dtype=np.float64, na_value=np.nan)[src_rows]
                for c, k in zip(NUMERIC_COLS, kinds)]
        nan_cnt = np.stack([np.add.reduceat(np.isnan(v).astype(np.int64), src_off[:-1]) for v in chan], 1) + (M - q)[:, None]
        needs = (nan_cnt > 0) & (nan_cnt < M[:, None])                   # pandas leaves all-NaN / no-NaN columns alone
        for ci, k in enumerate(kinds):
            if k == "obj":
                needs[:, ci] = False                                     # object dtype: Series.interpolate is a no-op
        # ---- forward-filled columns: numeric ones are gathered on the device, object ones travel as integer codes
        fill_cols = [c for c in FILL_COLS if c in data.columns]
        src_np = {c: data[c].to_numpy()[src_rows] for c in cols_in}
        valid = (np.stack([(~pd.isna(src_np[c])).astype(np.uint8) for c in fill_cols]) if fill_cols
                 else np.zeros((0, len(src_rows)), np.uint8))
        f_names = [c for c in fill_cols if src_np[c].dtype.kind == "f"]
        o_names = [c for c in fill_cols if src_np[c].dtype == object]
        h_names = [c for c in fill_cols if c not in f_names and c not in o_names]      # ints, bools, datetimes: host gather
        fsrc = (np.stack([src_np[c].astype(np.float64, copy=False) for c in f_names]) if f_names
                else np.zeros((0, len(src_rows))))
        cats, codes = {}, []
        for c in o_names:
            if c == "symbol" and sym_fact is not None:
                cd, cat = sym_fact[0][src_rows].astype(np.int32), np.asarray(sym_fact[1], dtype=object)
            else:
                cd, cat = pd.factorize(src_np[c], use_na_sentinel=True)
                cd = cd.astype(np.int32); cat = np.asarray(cat, dtype=object)
            cats[c] = cat; codes.append(cd)
        csrc = np.stack(codes) if codes else np.zeros((0, len(src_rows)), np.int32)
        greek = None
        if self.preserve_greeks:
            srcg = {c: src_np[c] for c in ("strike", "interest_rate", "callput") if c in src_np}
            greek = _greek_sources([srcg], [len(src_rows)])
        dup = ~first
        rows_info = (first_ns[kept].astype(np.int64), needs, o_names.index("symbol") if "symbol" in o_names else None)
        args = (pos.astype(np.int64), chan, src_off, q_off, total_q, code, valid, fsrc, [fill_cols.index(c) for c in f_names],
                csrc, [fill_cols.index(c) for c in o_names], [fill_cols.index(c) for c in h_names], greek, rows_info)
        fc = be.frame_columns(*args) if hasattr(be, "frame_columns") else _frame_columns_generic(be, *args)
        out, status, greeks = fc["chan"], fc["status"], fc["greeks"]
        sym_ok = ~((status != ST_OK) & needs).any(1)                     # scipy would raise -> that symbol is None
        # ---- assemble the long output
        sym_of_row = None
        gpos = q_off[:-1][ks] + pos                                      # global output row of every source row
        nothing_missing = bool((M == q).all())
        lat_off = np.concatenate([[0], np.cumsum(m0k)])
        if fc["date_ns"] is not None and not dup.any():
            date_ns = fc["date_ns"]                                      # formed on the device (no duplicate timestamps)
        else:
            sym_of_row = np.repeat(np.arange(S), M)
            if dup.any():
                cnt = np.ones(int(lat_off[-1]), np.int64)
                np.add.at(cnt, lat_off[:-1][ks[dup]] + lat[dup], 1)      # duplicates multiply the timeline row (R7)
                glat = np.repeat(np.arange(int(lat_off[-1])), cnt)
                lat_in_sym = glat - lat_off[:-1][sym_of_row]
            else:
                lat_in_sym = np.arange(total_q) - q_off[:-1][sym_of_row]
            date_ns = first_ns[kept][sym_of_row] + lat_in_sym * MINUTE_NS
        dates = pd.DatetimeIndex(date_ns.view("datetime64[ns]"))
        if tz is not None:
            dates = dates.tz_localize("UTC").tz_convert(tz)
        cols = {"date": dates}
        raw_idx = None
        for name in cols_in:
            v = src_np[name]
            int_dtype = v.dtype if v.dtype.kind in "iub" else None
            if name in NUMERIC_COLS:
                ci = NUMERIC_COLS.index(name)
                if kinds[ci] == "obj":
                    merged = np.full(total_q, np.nan, dtype=object)
                    merged[gpos] = v
                    cols[name] = merged
                    continue
                if kinds[ci] == "objfill":                               # the channel carried row numbers of `v`
                    cols[name] = _objfill_finish(out[ci], v)
                    continue
                merged = out[ci]                                         # knots keep their cells, the rest is interpolated
                # (a column pandas leaves alone -- no NaN at all, or nothing but NaN -- comes back as its own cells)
                if nothing_missing and int_dtype is not None:
                    merged = merged.astype(int_dtype)
                else:
                    merged = _chan_finish(merged, kinds[ci], data[name].dtype)
                cols[name] = merged
            elif name in f_names:
                col = fc["F"][f_names.index(name)]
                cols[name] = col if v.dtype == np.float64 else col.astype(v.dtype)
            elif name in o_names:
                cat = cats[name]
                cols[name] = np.append(cat, np.nan)[fc["C"][o_names.index(name)]]      # code -1 -> the appended NaN
            elif name in h_names:
                cols[name] = _gather(v, fc["rows"][fill_cols.index(name)].astype(np.int64), int_dtype, nothing_missing)
            else:
                if raw_idx is None:
                    raw_idx = np.full(total_q, -1, np.int64)
                    raw_idx[gpos] = np.arange(len(gpos))
                cols[name] = _gather(v, raw_idx, int_dtype, nothing_missing)
        # the forward-fill index only ever points at VALID source cells, so "symbol is null" == "no source row yet"
        # (an integer compare instead of pd.isna over millions of Python strings)
        if "symbol" in o_names:
            sym_na = fc["C"][o_names.index("symbol")] < 0
        else:
            sym_na = np.asarray(pd.isna(cols["symbol"]))
        if fc["keep"] is not None and all(k != "obj" for k in kinds):
            keep = fc["keep"]                                            # formed on the device
        else:
            if sym_of_row is None:
                sym_of_row = np.repeat(np.arange(S), M)
            keep = ~sym_na & sym_ok[sym_of_row]
            for c in REQUIRED[1:]:
                keep &= ~np.asarray(pd.isna(cols[c]))
        cols["is_interpolated"] = sym_na
        if greeks is not None:
            # a channel pandas does not compute on (object dtype) gives the epilogue nothing to work from: undefined -> NaN
            undefined = any(k in ("obj", "objfill") for k in kinds)
            for gi, gname in enumerate(GREEK_COLS):
                cols[gname] = np.full(total_q, np.nan) if undefined else greeks[gi]
                if gname not in out_cols:
                    out_cols.append(gname)
        res = pd.DataFrame({c: cols[c] for c in out_cols}, copy=False)
        if not split:
            if not keep.all():
                res = res[keep].reset_index(drop=True)
            return res
        # per-group results (interpolate_batch): slices of the long frame, indexed by merged-frame position like the
        # per-symbol path (core.py:74 leaves the surviving positions as the index).  A symbol whose int / bool columns
        # could stay integral (no row missing in ITS merged frame) while the long frame's did not is left to the caller.
        results = [None] * S_all
        int_cols = any(src_np[c].dtype.kind in "iub" for c in cols_in)
        kall = bool(keep.all())
        for k, gid in enumerate(kept):
            a, b = int(q_off[k]), int(q_off[k + 1])
            if not sym_ok[k]:
                continue                                              # scipy would have raised: None (core.py:83-85)
            if int_cols and M[k] == q[k] and not nothing_missing:
                results[gid] = NotImplemented                         # dtype differs from the long frame's: per-symbol path
                continue
            if kall:
                sub = res.iloc[a:b]; sub.index = pd.RangeIndex(b - a)
            else:
                kk = keep[a:b]
                nk = int(kk.sum())
                if nk == 0:
                    continue                                          # "No valid data after interpolation" (core.py:76-78)
                lo_ = int(kk.argmax()); hi_ = b - a - int(kk[::-1].argmax())
                if hi_ - lo_ == nk:                                   # the usual case: leading / trailing rows dropped
                    sub = res.iloc[a + lo_:a + hi_]; sub.index = pd.RangeIndex(lo_, hi_)
                else:
                    sub = res.iloc[a:b]; sub.index = pd.RangeIndex(b - a); sub = sub[kk]
            sub._is_copy = None       # a slice of a frame nobody else holds: callers add columns (batch_id) without pandas' view warning
            results[gid] = sub
        return results

    # ------------------------------------------------------------------ host bookkeeping
    def _prepare(self, symbol_data: pd.DataFrame) -> Optional[_Prepared]:
        if len(symbol_data) < self.min_points:                           # core.py:26-28
            logger.warning(f"Insufficient data points: {len(symbol_data)} < {self.min_points}")
            return None
        df = symbol_data.sort_values("date").reset_index(drop=True)     # core.py:32 (same pandas call, same order)
        df["date"] = pd.to_datetime(df["date"])                          # core.py:33
        dmin, dmax = df["date"].min(), df["date"].max()
        time_range = dmax - dmin
        if time_range > timedelta(days=30):                              # core.py:36-39
            logger.warning(f"Time range too large: {time_range}")
            return None
        timeline = pd.date_range(start=dmin, end=dmax, freq="1min")      # core.py:42-46
        if len(timeline) > 100000:                                       # core.py:49-51
            logger.warning(f"Timeline too long: {len(timeline)} minutes")
            return None
        try:
            code = method_code(self.method)
        except KeyError:
            code = None
        if code is None:
            if self.method not in _PANDAS_METHODS:
                raise ValueError(f"method must be one of {_PANDAS_METHODS}. Got '{self.method}' instead.")
            if self.method == "time":
                raise ValueError("time-weighted interpolation only works on Series or DataFrames with a DatetimeIndex")
            if self.method in ("spline", "polynomial"):
                raise ValueError("You must specify the order of the spline or polynomial.")
            raise ValueError(f"method '{self.method}' is valid for pandas but not implemented by the MI355X engine "
                             f"(implemented: {sorted(METHOD_CODES)})")
        for c in REQUIRED:                                               # dropna(subset=...) KeyError, core.py:74
            if c not in df.columns:
                raise KeyError(c)
        # exact-timestamp left join (core.py:54-55) as integer lattice arithmetic
        t_ns = pd.DatetimeIndex(timeline).as_unit("ns").asi8
        d_idx = pd.DatetimeIndex(df["date"]).as_unit("ns")
        d_ns = d_idx.asi8
        rel = d_ns - t_ns[0]
        on = (~np.asarray(d_idx.isna())) & (rel >= 0) & (rel % MINUTE_NS == 0)
        rows_on = np.flatnonzero(on)                                     # off-lattice rows vanish (R6)
        lat = rel[rows_on] // MINUTE_NS
        q = len(rows_on)
        first = np.ones(q, bool); first[1:] = lat[1:] != lat[:-1]
        u = np.cumsum(first)
        pos = lat + np.arange(q) + 1 - u                                 # duplicates take consecutive rows (R7, R8)
        m0 = len(timeline)
        M = m0 + (q - int(u[-1]))
        if M == m0:
            rowlat = None
        else:
            cnt = np.maximum(np.bincount(lat, minlength=m0), 1)
            rowlat = np.repeat(np.arange(m0), cnt)
        p = _Prepared()
        p.df, p.timeline, p.pos, p.rowlat, p.M = df, timeline, pos, rowlat, M
        # source columns as NumPy arrays restricted to the rows that landed on the lattice
        p.src_np, p.int_dtype = {}, {}
        for c in df.columns:
            if c == "date":
                continue
            v = df[c].to_numpy()
            p.int_dtype[c] = v.dtype if v.dtype.kind in "iub" else None
            p.src_np[c] = v[rows_on]
        p.chan_src, p.chan_needs, p.chan_kind = [], [], []
        for c in NUMERIC_COLS:                                           # core.py:58-61
            kind = _chan_kind(df[c], code in _FILL_METHOD_CODES)
            p.chan_kind.append(kind)
            if kind == "obj":                                            # object dtype: Series.interpolate leaves it alone
                p.chan_src.append(np.zeros(q)); p.chan_needs.append(False)
                continue
            v = (_objfill_source(df[c].to_numpy()[rows_on]) if kind == "objfill"
                 else df[c].to_numpy(dtype=np.float64, na_value=np.nan)[rows_on])
            p.chan_src.append(v)
            nn = int(np.isnan(v).sum()) + (M - q)                        # NaNs of the merged column
            p.chan_needs.append(0 < nn < M)                              # pandas: all-NaN / no-NaN columns are left alone
        p.fill_cols = [c for c in FILL_COLS if c in df.columns]          # core.py:64-68
        p.fill_valid = [(~pd.isna(p.src_np[c])).astype(np.uint8) for c in p.fill_cols]
        return p

    def _assemble(self, p: _Prepared, out, status, fill_idx, fill_missing, greeks=None) -> Optional[pd.DataFrame]:
        df, M = p.df, p.M
        q = len(p.pos)
        for c in range(len(NUMERIC_COLS)):
            if p.chan_needs[c] and status[c] == ST_ILL_CONDITIONED:
                # documented deviation: the reference returns the (numerically meaningless) values of a polynomial of
                # degree > POLY_MAX_KNOTS - 1; the engine does not reproduce noise and gives the symbol up
                raise ValueError(f"method '{self.method}': one polynomial through more than {POLY_MAX_KNOTS} knots of "
                                 f"'{NUMERIC_COLS[c]}' is ill-conditioned; not computed")
            if p.chan_needs[c] and status[c] != ST_OK:
                # scipy raises inside Series.interpolate (too few knots) -> core.py:83-85
                raise ValueError("The number of derivatives at boundaries does not match: "
                                 f"too few valid knots in '{NUMERIC_COLS[c]}' for method '{self.method}'")
        nothing_missing = M == q
        raw_idx = None
        dates = p.timeline if p.rowlat is None else p.timeline[p.rowlat]
        cols = {"date": dates}
        for name in df.columns:
            if name == "date":
                continue
            if name in NUMERIC_COLS:
                ci = NUMERIC_COLS.index(name)
                if p.chan_kind[ci] == "obj":                             # untouched by interpolate: source cells only
                    merged = np.full(M, np.nan, dtype=object)
                    merged[p.pos] = p.src_np[name]
                    cols[name] = merged
                    continue
                merged = np.full(M, np.nan)
                merged[p.pos] = p.chan_src[ci]
                if p.chan_needs[ci]:
                    merged = np.where(np.isnan(merged), out[ci], merged)
                if p.chan_kind[ci] == "objfill":                         # row numbers of the object cells
                    cols[name] = _objfill_finish(merged, p.src_np[name])
                    continue
                if nothing_missing and p.int_dtype[name] is not None:    # nothing missing: int column stays int
                    merged = merged.astype(p.int_dtype[name])
                else:
                    merged = _chan_finish(merged, p.chan_kind[ci], df[name].dtype)
                cols[name] = merged
            elif name in p.fill_cols:
                fi = p.fill_cols.index(name)
                cols[name] = _gather(p.src_np[name], np.where(fill_missing[fi], -1, fill_idx[fi]), p.int_dtype[name],
                                     nothing_missing)
            else:
                if raw_idx is None:
                    raw_idx = np.full(M, -1, np.int64)
                    raw_idx[p.pos] = np.arange(q)
                cols[name] = _gather(p.src_np[name], raw_idx, p.int_dtype[name], nothing_missing)
        sym = cols["symbol"]
        sym_na = pd.isna(sym)
        keep = ~sym_na                                                    # core.py:74
        for c in REQUIRED[1:]:
            keep &= ~pd.isna(cols[c])
        cols["is_interpolated"] = sym_na                                  # core.py:71 (after ffill -> always False)
        order = ["date"] + [c for c in df.columns if c != "date"]
        if "is_interpolated" not in order:
            order.append("is_interpolated")
        if greeks is not None:
            # a channel pandas does not compute on (object dtype) gives the epilogue nothing to work from: undefined -> NaN
            undefined = any(k in ("obj", "objfill") for k in p.chan_kind)
            for gi, gname in enumerate(GREEK_COLS):
                cols[gname] = np.full(M, np.nan) if undefined else greeks[gi]
                if gname not in order:
                    order.append(gname)
        merged = pd.DataFrame({c: cols[c] for c in order}, copy=False)
        if not keep.all():
            merged = merged[keep]
        if merged.empty:                                                 # core.py:76-78
            logger.warning("No valid data after interpolation")
            return None
        logger.debug(f"Interpolated {len(df)} → {len(merged)} rows")
        return merged


def _gather(values: np.ndarray, idx: np.ndarray, int_dtype, nothing_missing: bool) -> np.ndarray:
    """values[idx] with -1 -> missing, dtype-promoting exactly like the reference's left merge + ffill
    (int/bool -> float64 as soon as the merge introduced a missing row, object -> NaN, datetime -> NaT)."""
    miss = idx < 0
    any_miss = bool(miss.any())
    safe = np.where(miss, 0, idx) if any_miss else idx
    if int_dtype is not None:
        if nothing_missing and not any_miss:
            return values[safe]
        if int_dtype.kind == "b":                                        # bool + NaN -> object in pandas
            r = values[safe].astype(object)
            if any_miss:
                r[miss] = np.nan
            return r
        r = values[safe].astype(np.float64)
        if any_miss:
            r[miss] = np.nan
        return r
    r = values[safe]
    if any_miss:
        if r.dtype.kind == "f":
            r = r.copy(); r[miss] = np.nan
        elif r.dtype.kind in "mM":
            r = r.copy(); r[miss] = np.datetime64("NaT") if r.dtype.kind == "M" else np.timedelta64("NaT")
        else:
            r = r.astype(object); r[miss] = np.nan
    return r
