"""Thin device-side wrappers: torch-ROCm tensors are only device buffers and streams here;
all arithmetic happens in the HIP kernels behind the C ABI (include/ivs.h)."""
from __future__ import annotations

from typing import Optional, Tuple

from . import _lib
from ._lib import EngineUnavailable


def _torch():
    import torch
    return torch


def require_device():
    """Return torch, after checking that the HIP library loads and a GPU is visible."""
    lib = _lib.load()
    torch = _torch()
    if not torch.cuda.is_available() or lib.ivs_device_count() < 1:
        raise EngineUnavailable("no MI355X / HIP device visible: the interpolation engine has no CPU fallback")
    return torch


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _stream(torch, stream):
    s = torch.cuda.current_stream() if stream is None else stream
    return s.cuda_stream


def _f64(torch, t, name):
    if t.dtype != torch.float64 or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA float64 tensor")
    return t.contiguous()


def _hold_for_stream(torch, stream, *tensors):
    """Tensors that were allocated on torch's CURRENT stream (per-call workspaces, contiguous temporaries, outputs) but are
    read or written by kernels launched on an explicit `stream`: tell the caching allocator, so that the block is not
    handed to the next allocation on the current stream while those kernels still run."""
    if stream is None or stream == torch.cuda.current_stream():
        return
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(stream)


def validate_ragged(K, sigma, k_off, nK_max: int, n_maturities: int):
    """Optional host-side check of CSR offsets with a readable exception (one small device reduction + one D2H read, i.e. it
    SYNCHRONISES: not for capture).  The kernels do not depend on it: they check every span against nK_max and against the
    total strike count passed through the C ABI and flag offending surfaces with ST_BAD_SHAPE."""
    torch = _torch()
    if k_off.numel() < 1:
        raise ValueError("k_off must hold B+1 offsets")
    span = k_off[1:] - k_off[:-1]
    stats = torch.stack([k_off[0], k_off[-1], span.min() if span.numel() else k_off[0] * 0,
                         span.max() if span.numel() else k_off[0] * 0]).tolist()
    first, last, smin, smax = (int(v) for v in stats)
    if first != 0 or smin < 0:
        raise ValueError("k_off must start at 0 and be non-decreasing")
    if smax > nK_max:
        raise ValueError(f"k_off: a surface has {smax} strikes but nK_max={nK_max}")
    if K.numel() != last or sigma.numel() != n_maturities * last:
        raise ValueError(f"ragged batch: K has {K.numel()} and sigma {sigma.numel()} entries, k_off[-1]={last}, nT={n_maturities}")


def surface_workspace(B: int, ragged: bool, device=None):
    """Device scratch for one surface_batch call (ivs_surface_workspace_bytes): uint8 CUDA tensor, 256-byte aligned."""
    torch = require_device()
    n = _lib.load().ivs_surface_workspace_bytes(int(B), 1 if ragged else 0)
    return torch.empty(n, dtype=torch.uint8, device=device or "cuda")


def surface_batch(K, T, sigma, Kq, Tq, method="linear", *, k_off=None, nK_max: Optional[int] = None,
                  n_maturities: Optional[int] = None, out=None, status=None, stream=None,
                  force_generic: bool = False, workspace=None, map_groups: int = 0, one_pass: bool = False,
                  validate: bool = False):
    """Interpolate a batch of (strike x maturity) surfaces on the current device.

    Uniform: K [B,nK] or [nK] (shared), sigma [B,nT,nK].  Ragged: K flat [total], sigma flat
    [nT*total] (surface b row-major [nT][nK_b]), k_off int64 [B+1], nK_max, n_maturities.
    T [nT] or [B,nT]; Kq [mK] or [B,mK]; Tq [mT] or [B,mT].  Returns (out [B,mT,mK], status [B]).
    `workspace`: uint8 CUDA tensor from surface_workspace() (allocated per call when omitted; pass one to keep the
    call allocation-free, e.g. under hipGraph capture).  Ragged offsets are checked ON THE DEVICE (a surface whose span is
    negative, exceeds nK_max or leaves K gets ST_BAD_SHAPE and is skipped), so the call never synchronises;
    `validate=True` adds validate_ragged()'s host-side check (one D2H read) with a readable exception.
    """
    torch = require_device()
    lib = _lib.load()
    code = _lib.METHOD_CODES[method] if isinstance(method, str) else int(method)
    K = _f64(torch, K, "K"); T = _f64(torch, T, "T"); sigma = _f64(torch, sigma, "sigma")
    Kq = _f64(torch, Kq, "Kq"); Tq = _f64(torch, Tq, "Tq")
    if k_off is None:
        B, nT, nK = sigma.shape
        k_stride = 0 if K.dim() == 1 else nK
        if K.shape[-1] != nK or (K.dim() == 2 and K.shape[0] != B):
            raise ValueError("K shape does not match sigma")
    else:
        if k_off.dtype != torch.int64 or not k_off.is_cuda:
            raise TypeError("k_off must be a CUDA int64 tensor")
        if nK_max is None or n_maturities is None:
            raise ValueError("ragged batches need nK_max and n_maturities")
        k_off = k_off.contiguous()
        B = k_off.numel() - 1; nT = int(n_maturities); nK = int(nK_max)
        if B < 0:
            raise ValueError("k_off must hold B+1 offsets")
        if sigma.numel() != nT * K.numel():
            raise ValueError(f"ragged batch: K has {K.numel()} strikes, sigma {sigma.numel()} quotes, nT={nT}")
        if validate:
            validate_ragged(K, sigma, k_off, nK, nT)
        k_stride = K.numel()       # ragged: the C ABI takes the total strike count here and bounds every span by it
    if T.shape[-1] != nT or (T.dim() == 2 and T.shape[0] != B):
        raise ValueError("T shape does not match sigma")
    t_stride = 0 if T.dim() == 1 else nT
    mK, mT = Kq.shape[-1], Tq.shape[-1]
    if (Kq.dim() == 2 and Kq.shape[0] != B) or (Tq.dim() == 2 and Tq.shape[0] != B):
        raise ValueError("per-surface query grids must have B rows")
    kq_stride = 0 if Kq.dim() == 1 else mK
    tq_stride = 0 if Tq.dim() == 1 else mT
    if out is None:
        out = torch.empty((B, mT, mK), dtype=torch.float64, device=sigma.device)
    if status is None:
        status = torch.empty((B,), dtype=torch.int32, device=sigma.device)
    need = lib.ivs_surface_workspace_bytes(B, 0 if k_off is None else 1)
    if workspace is None:
        workspace = torch.empty(need, dtype=torch.uint8, device=sigma.device)
    elif workspace.numel() * workspace.element_size() < need or not workspace.is_cuda:
        raise ValueError(f"workspace too small: {workspace.numel() * workspace.element_size()} < {need} bytes")
    flags = ((_lib.FLAG_FORCE_GENERIC if force_generic else 0) | (_lib.FLAG_ONE_PASS if one_pass else 0)
             | _lib.flag_map_groups(map_groups))
    rc = lib.ivs_surface_batch_f64(_ptr(K), _ptr(k_off), k_stride, nK, _ptr(T), t_stride, nT, _ptr(sigma), B,
                                   _ptr(Kq), kq_stride, mK, _ptr(Tq), tq_stride, mT, _ptr(out), _ptr(status),
                                   code, flags, _ptr(workspace), workspace.numel() * workspace.element_size(),
                                   _stream(torch, stream))
    _hold_for_stream(torch, stream, K, T, sigma, Kq, Tq, k_off, out, status, workspace)
    _lib.check(rc, "ivs_surface_batch_f64")
    return out, status


def place_output(run, shape, tries: int = 8, dtype=None, warm: int = 8, timed: int = 3):
    """Pick the output buffer a persistent caller should keep.  On MI355X the same surface kernel on the same inputs runs up
    to 8 % faster or slower depending on WHICH allocation it writes to (stable per buffer, independent of offsets inside
    it).  `run(out)` must launch the call on the current stream with `out` as its output tensor.
    Allocates `tries` candidates (all alive at once: distinct allocations), times `run` on each with HIP events (median of
    `timed` after `warm` untimed launches), returns (best_tensor, [median ms per candidate]); the others are freed."""
    torch = require_device()
    dtype = dtype or torch.float64
    cands = [torch.empty(shape, dtype=dtype, device="cuda") for _ in range(max(1, tries))]
    ms = []
    for o in cands:
        for _ in range(warm):
            run(o)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(timed)]
        for s_, e_ in evs:
            s_.record(); run(o); e_.record()
        torch.cuda.synchronize()
        ms.append(sorted(s_.elapsed_time(e_) for s_, e_ in evs)[timed // 2])
    best = cands[ms.index(min(ms))]
    del cands
    torch.cuda.empty_cache()
    return best, ms


def last_kernel() -> str:
    return _lib.load().ivs_last_kernel().decode()


def interp1d_batch(xk, yk, knot_off, q_off, total_q: int, method, xq=None, stream=None) -> Tuple[object, object]:
    """CSR batch of 1-D series.  xk [TK], yk [C,TK], knot_off/q_off int64 [S+1] (device).
    Returns (out [C,total_q], status [S,C])."""
    torch = require_device()
    lib = _lib.load()
    code = _lib.METHOD_CODES[method] if isinstance(method, str) else int(method)
    xk = _f64(torch, xk, "xk"); yk = _f64(torch, yk, "yk")
    Cn, TK = yk.shape
    S = knot_off.numel() - 1
    dev = yk.device
    out = torch.empty((Cn, total_q), dtype=torch.float64, device=dev)
    status = torch.zeros((S, Cn), dtype=torch.int32, device=dev)
    wsb = lib.ivs_interp1d_workspace_bytes(TK, S, Cn)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
    rc = lib.ivs_interp1d_batch_f64(_ptr(xk), _ptr(yk), TK, _ptr(knot_off), S, Cn, TK,
                                    _ptr(xq), _ptr(q_off), total_q, _ptr(out), total_q, _ptr(status), code,
                                    _ptr(ws), ws.numel() * 8, _stream(torch, stream))
    _hold_for_stream(torch, stream, xk, yk, out, status, ws)
    _lib.check(rc, "ivs_interp1d_batch_f64")
    return out, status


def interp1d_greeks_batch(xk, yk, knot_off, q_off, total_q: int, method, channels, fidx, rows, strike_src, rate_src,
                          put_src, stream=None):
    """interp1d_batch with the Greeks epilogue (ivs_interp1d_greeks_batch_f64).  channels = (ch_iv, ch_S, ch_T); fidx int32
    [n_cols, total_q] from ffill_index_batch; rows = (row_strike, row_rate, row_callput) inside fidx or -1; *_src = source
    columns (float64, float64, uint8 0/1/2) or None.  Returns (out [C,total_q], status [S,C], greeks [5,total_q])."""
    torch = require_device()
    lib = _lib.load()
    code = _lib.METHOD_CODES[method] if isinstance(method, str) else int(method)
    xk = _f64(torch, xk, "xk"); yk = _f64(torch, yk, "yk")
    Cn, TK = yk.shape
    S = knot_off.numel() - 1
    dev = yk.device
    out = torch.empty((Cn, total_q), dtype=torch.float64, device=dev)
    greeks = torch.empty((5, total_q), dtype=torch.float64, device=dev)
    status = torch.zeros((S, Cn), dtype=torch.int32, device=dev)
    wsb = lib.ivs_interp1d_workspace_bytes(TK, S, Cn)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
    fidx = None if fidx is None else fidx.contiguous()
    rc = lib.ivs_interp1d_greeks_batch_f64(_ptr(xk), _ptr(yk), TK, _ptr(knot_off), S, Cn, TK, None, _ptr(q_off), total_q,
                                           _ptr(out), total_q, _ptr(status), code, int(channels[0]), int(channels[1]),
                                           int(channels[2]), _ptr(fidx), 0 if fidx is None else fidx.shape[1],
                                           int(rows[0]), int(rows[1]), int(rows[2]), _ptr(strike_src), _ptr(rate_src),
                                           _ptr(put_src), _ptr(greeks), total_q, _ptr(ws), ws.numel() * 8,
                                           _stream(torch, stream))
    _hold_for_stream(torch, stream, xk, yk, out, status, greeks, ws, fidx)
    _lib.check(rc, "ivs_interp1d_greeks_batch_f64")
    return out, status, greeks


def ffill_index_batch(src_pos, src_off, valid, q_off, total_q: int, stream=None):
    """valid uint8 [n_cols, total_src] -> int32 [n_cols, total_q] flat source-row index or -1."""
    torch = require_device()
    lib = _lib.load()
    n_cols, TS = valid.shape
    S = src_off.numel() - 1
    idx = torch.empty((n_cols, total_q), dtype=torch.int32, device=valid.device)
    valid = valid.contiguous()
    rc = lib.ivs_ffill_index_batch(_ptr(src_pos), _ptr(src_off), _ptr(valid), TS, n_cols, _ptr(q_off),
                                   S, total_q, _ptr(idx), total_q, _stream(torch, stream))
    _hold_for_stream(torch, stream, valid, idx)
    _lib.check(rc, "ivs_ffill_index_batch")
    return idx


def gather_rows(src, idx, idx_row, stream=None):
    """Columnar egress: out[c][g] = src[c][idx[idx_row[c]][g]] (NaN / -1 where the index is negative).  src float64 or int32
    [n_cols, n_src]; idx int32 [rows, n]; idx_row int32 [n_cols] (device)."""
    torch = require_device()
    lib = _lib.load()
    n_cols, n_src = src.shape
    n = idx.shape[1]
    out = torch.empty((n_cols, n), dtype=src.dtype, device=src.device)
    fn = lib.ivs_gather_rows_f64 if src.dtype == torch.float64 else lib.ivs_gather_rows_i32
    src = src.contiguous()
    rc = fn(_ptr(src), n_src, _ptr(idx), idx.shape[1], _ptr(idx_row), n_cols, n, _ptr(out), n, _stream(torch, stream))
    _hold_for_stream(torch, stream, src, out)
    _lib.check(rc, "ivs_gather_rows")
    return out


def frame_rows(q_off, first_ns, chan, sym_code, status, needs, stream=None):
    """Per output row: timestamp (int64 ns) and the dropna / failed-symbol keep flag (ivs_frame_rows)."""
    torch = require_device()
    lib = _lib.load()
    S = q_off.numel() - 1
    Cn, total_q = chan.shape
    dates = torch.empty(total_q, dtype=torch.int64, device=chan.device)
    keep = torch.empty(total_q, dtype=torch.uint8, device=chan.device)
    status = status.contiguous(); needs = needs.contiguous()
    rc = lib.ivs_frame_rows(_ptr(q_off), S, total_q, _ptr(first_ns), _ptr(chan), total_q, Cn, _ptr(sym_code), _ptr(status),
                            _ptr(needs), _ptr(dates), _ptr(keep), _stream(torch, stream))
    _hold_for_stream(torch, stream, status, needs, dates, keep)
    _lib.check(rc, "ivs_frame_rows")
    return dates, keep


def frame_columns(src_pos, src_off, q_off, total_q: int, yk, method, valid, fsrc, f_rows, csrc, c_rows, idx_rows,
                  first_ns=None, needs=None, sym_col: int = -1, greek=None, stream=None):
    """The long output frame in ONE pass over its rows (ivs_frame_columns_f64): channels on the integer lattice, the
    forward-filled f64 / code columns gathered straight from their source columns, raw gather-index rows for the columns the
    host gathers itself, the date column and the dropna keep flag, optionally the Greeks -- what interp1d_batch +
    ffill_index_batch + gather_rows (x2) + frame_rows do together, without the index arrays in between.

    src_pos int64 [n_src], src_off / q_off int64 [S+1], yk float64 [C, n_src], valid uint8 [V, n_src], fsrc float64
    [nF, n_src], csrc int32 [nC, n_src]; f_rows / c_rows / idx_rows: int32 device tensors naming the validity row of each
    column (or None); first_ns int64 [S] + needs uint8 [S, C] switch the date / keep outputs on; greek = (g_rows (3 ints, -1
    = column absent), strike_src, rate_src, put_src) or None.
    Returns dict(chan [C,total_q], status [S,C], F, C, idx, date_ns, keep, greeks) of device tensors (None where not asked)."""
    torch = require_device()
    lib = _lib.load()
    code = _lib.method_code(method) if isinstance(method, str) else int(method)
    dev = yk.device
    yk = _f64(torch, yk, "yk")
    Cn, n_src = yk.shape
    S = src_off.numel() - 1
    total_q = int(total_q)
    a = _lib.FrameArgs()
    keepalive = []

    def cont(t):
        t = t.contiguous(); keepalive.append(t); return t
    src_pos, src_off, q_off = cont(src_pos), cont(src_off), cont(q_off)
    a.src_pos, a.src_off, a.q_off = _ptr(src_pos), _ptr(src_off), _ptr(q_off)
    a.n_series, a.total_src, a.total_queries = S, n_src, total_q
    a.yk, a.yk_stride, a.n_channels, a.method = _ptr(yk), n_src, Cn, code
    chan = torch.empty((Cn, total_q), dtype=torch.float64, device=dev)
    status = torch.zeros((S, Cn), dtype=torch.int32, device=dev)
    a.chan_out, a.chan_stride, a.status = _ptr(chan), total_q, _ptr(status)
    nV = 0 if valid is None else valid.shape[0]
    if nV:
        valid = cont(valid)
    a.valid, a.valid_stride, a.n_valid = (_ptr(valid) if nV else 0), n_src, nV
    F = Cc = idx = None
    nF = 0 if fsrc is None else fsrc.shape[0]
    if nF:
        fsrc, f_rows = cont(fsrc), cont(f_rows)
        F = torch.empty((nF, total_q), dtype=torch.float64, device=dev)
        a.fsrc, a.fsrc_stride, a.f_rows, a.n_f, a.f_out, a.f_stride = _ptr(fsrc), n_src, _ptr(f_rows), nF, _ptr(F), total_q
    nC = 0 if csrc is None else csrc.shape[0]
    if nC:
        csrc, c_rows = cont(csrc), cont(c_rows)
        Cc = torch.empty((nC, total_q), dtype=torch.int32, device=dev)
        a.csrc, a.csrc_stride, a.c_rows, a.n_c, a.c_out, a.c_stride = _ptr(csrc), n_src, _ptr(c_rows), nC, _ptr(Cc), total_q
    nI = 0 if idx_rows is None else idx_rows.numel()
    if nI:
        idx_rows = cont(idx_rows)
        idx = torch.empty((nI, total_q), dtype=torch.int32, device=dev)
        a.idx_rows, a.n_idx, a.idx_out, a.idx_stride = _ptr(idx_rows), nI, _ptr(idx), total_q
    dates = keep = None
    a.sym_col = int(sym_col)
    if first_ns is not None:
        first_ns, needs = cont(first_ns), cont(needs)
        dates = torch.empty(total_q, dtype=torch.int64, device=dev)
        keep = torch.empty(total_q, dtype=torch.uint8, device=dev)
        a.first_ns, a.needs, a.date_ns, a.keep = _ptr(first_ns), _ptr(needs), _ptr(dates), _ptr(keep)
    greeks = None
    a.g_strike = a.g_rate = a.g_put = -1
    if greek is not None:
        g_rows, ksrc, rsrc, psrc = greek
        ksrc, rsrc, psrc = cont(ksrc), cont(rsrc), cont(psrc)
        greeks = torch.empty((5, total_q), dtype=torch.float64, device=dev)
        a.g_strike, a.g_rate, a.g_put = (int(x) for x in g_rows)
        a.strike_src, a.rate_src, a.put_src = _ptr(ksrc), _ptr(rsrc), _ptr(psrc)
        a.ch_iv, a.ch_underlying, a.ch_ttm = 0, 1, 2
        a.greeks, a.greeks_stride = _ptr(greeks), total_q
    wsb = lib.ivs_frame_workspace_bytes(n_src, S, Cn)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
    rc = lib.ivs_frame_columns_f64(a, _ptr(ws), ws.numel() * 8, _stream(torch, stream))
    _hold_for_stream(torch, stream, ws, chan, status, F, Cc, idx, dates, keep, greeks, *keepalive)
    _lib.check(rc, "ivs_frame_columns_f64")
    return {"chan": chan, "status": status, "F": F, "C": Cc, "idx": idx, "date_ns": dates, "keep": keep, "greeks": greeks}


def bs_greeks(S, K, T, r, sigma, is_put=None, default_is_put: bool = False, stream=None):
    """Black-Scholes Greeks on the device.  All inputs CUDA float64 tensors of one shape (is_put: uint8 or None).
    Returns dict(delta, gamma, theta, vega, rho) of tensors with that shape."""
    torch = require_device()
    lib = _lib.load()
    ins = [_f64(torch, t, n) for t, n in zip((S, K, T, r, sigma), ("S", "K", "T", "r", "sigma"))]
    n = ins[0].numel()
    if any(t.numel() != n for t in ins):
        raise ValueError("all inputs must have the same number of elements")
    if is_put is not None:
        is_put = is_put.to(torch.uint8).contiguous()
    outs = [torch.empty_like(ins[0]) for _ in range(5)]
    rc = lib.ivs_bs_greeks_f64(*[_ptr(t) for t in ins], _ptr(is_put), int(bool(default_is_put)), n,
                               *[_ptr(t) for t in outs], _stream(torch, stream))
    _lib.check(rc, "ivs_bs_greeks_f64")
    return dict(zip(("delta", "gamma", "theta", "vega", "rho"), outs))


def candle_aggregate(ts_ns, o, h, l, c, v, series_off, freq_minutes: int, stream=None):
    """Sparse N-minute aggregation on the device (see ivs_candle_aggregate_f64).  ts_ns int64, OHLCV float64, series_off
    int64 [S+1]; all CUDA tensors.  Returns (out_ts, open, high, low, close, volume, count) of length n."""
    torch = require_device()
    lib = _lib.load()
    n = ts_ns.numel()
    S = series_off.numel() - 1
    cols = [_f64(torch, t, nm) for t, nm in zip((o, h, l, c, v), "ohlcv")]
    out_ts = torch.empty(n, dtype=torch.int64, device=ts_ns.device)
    outs = [torch.empty(n, dtype=torch.float64, device=ts_ns.device) for _ in range(5)]
    cnt = torch.empty(n, dtype=torch.int32, device=ts_ns.device)
    rc = lib.ivs_candle_aggregate_f64(_ptr(ts_ns.contiguous()), *[_ptr(t) for t in cols], _ptr(series_off), S, n,
                                      int(freq_minutes) * 60_000_000_000, _ptr(out_ts), *[_ptr(t) for t in outs], _ptr(cnt),
                                      _stream(torch, stream))
    _lib.check(rc, "ivs_candle_aggregate_f64")
    return (out_ts, *outs, cnt)


BRIDGE_STRATEGIES = {"spread_simulation": 0, "price_as_midpoint": 1, "trend_following": 2, "simple_spread": 3,
                     "pipeline_inline": 4}


def mt19937_words(seed: int, n_words: int, device=None, stream=None):
    """The first n_words raw 32-bit outputs of ``np.random.seed(seed)`` as an int32 CUDA tensor (bit pattern of uint32)."""
    torch = require_device()
    lib = _lib.load()
    if not 0 <= int(seed) <= 0xFFFFFFFF:
        raise ValueError("Seed must be between 0 and 2**32 - 1")          # numpy's own message
    words = torch.empty(int(n_words), dtype=torch.int32, device=device or "cuda")
    rc = lib.ivs_mt19937_words_u32(int(seed), _ptr(words), int(n_words), _stream(torch, stream))
    _lib.check(rc, "ivs_mt19937_words_u32")
    return words


def bridge_words_bound(total_rows: int, strategy: int) -> int:
    """Words that certainly cover one bridge call: exact upper bound for the uniform strategies; for trend_following
    (rejection sampling) a generous estimate -- the call reports when it was not enough."""
    per_row = {0: 12, 1: 6, 2: 10, 3: 4, 4: 10}[int(strategy)]
    return int(total_rows) * per_row + 4096


def bridge_candles(price, volume, row_off, strategy: int, words, rng_tail=None, base_spread_pct: float = 0.002,
                   vol_factor: float = 1.5, stream=None):
    """IV -> OHLCV candles on the device (see ivs_bridge_candles_f64).  price float64 [n], volume float64 [n] or None,
    row_off int64 [S+1], words int32 [n_words] (from mt19937_words, positioned at this call's first draw), rng_tail
    int64 [4] or None (fresh generator).  Returns (out [6, n], valid uint8 [n], rng_tail)."""
    torch = require_device()
    lib = _lib.load()
    price = _f64(torch, price, "price")
    n = price.numel()
    S = row_off.numel() - 1
    if volume is not None:
        volume = _f64(torch, volume, "volume")
        if volume.numel() != n:
            raise ValueError("price and volume must have the same length")
    if rng_tail is None:
        rng_tail = torch.zeros(4, dtype=torch.int64, device=price.device)
    out = torch.empty((6, n), dtype=torch.float64, device=price.device)
    valid = torch.empty(n, dtype=torch.uint8, device=price.device)
    wsb = lib.ivs_bridge_workspace_bytes(n)
    ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=price.device)
    rc = lib.ivs_bridge_candles_f64(_ptr(price), _ptr(volume), _ptr(row_off), S, n, int(strategy), float(base_spread_pct),
                                    float(vol_factor), _ptr(words), words.numel(), _ptr(out), _ptr(valid), _ptr(rng_tail),
                                    _ptr(ws), ws.numel() * 8, _stream(torch, stream))
    _lib.check(rc, "ivs_bridge_candles_f64")
    return out, valid, rng_tail
