"""ctypes binding of libivs.so (the C ABI in include/ivs.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C iv_interpolation_amd/csrc``.
There is NO CPU fallback: if the shared object is missing or no HIP device is visible the
product path raises ``EngineUnavailable``.
"""
from __future__ import annotations

import ctypes as C
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libivs.so")

LINEAR, CUBIC, CUBICSPLINE, SLINEAR = 0, 1, 2, 3
ST_OK, ST_TOO_FEW_KNOTS, ST_BAD_SHAPE, ST_ILL_CONDITIONED = 0, 1, 2, 4
FLAG_FORCE_GENERIC = 1
FLAG_ONE_PASS = 2          # IVS_FLAG_ONE_PASS: skip the row-pass kernels (testing / A-B timing)


def flag_map_groups(n: int) -> int:
    """IVS_FLAG_MAP_GROUPS(n): tuning override of the 64x16 kernel's surface -> workgroup mapping (0 = default)."""
    return (int(n) & 0xff) << 8

ABI_VERSION = 3

# pandas method names (reference core.py:61 forwards self.method) -> engine codes
NEAREST, ZERO, PCHIP, AKIMA, FROM_DERIVATIVES = 4, 5, 6, 7, 8
QUADRATIC = 9
BARYCENTRIC, KROGH = 10, 11
PAD, BFILL = 12, 13        # pandas' fill methods, which Series.interpolate still executes (pad_or_backfill)
POLY_MAX_KNOTS = 32
METHOD_CODES = {"linear": LINEAR, "index": LINEAR, "values": LINEAR,
                "cubic": CUBIC, "cubicspline": CUBICSPLINE, "slinear": SLINEAR,
                "nearest": NEAREST, "zero": ZERO, "pchip": PCHIP, "akima": AKIMA,
                "from_derivatives": FROM_DERIVATIVES, "piecewise_polynomial": FROM_DERIVATIVES, "quadratic": QUADRATIC,
                "barycentric": BARYCENTRIC, "krogh": KROGH,
                "pad": PAD, "ffill": PAD, "bfill": BFILL, "backfill": BFILL}


def method_code(name) -> int:
    """Engine code of a pandas method name, or KeyError.  pandas matches the FILL methods case-insensitively
    (`method.lower() in fillna_methods`, verified against the reference: 'PAD', 'Backfill' work) and every other name
    exactly ('Linear' raises -> None)."""
    if name in METHOD_CODES:
        return METHOD_CODES[name]
    if isinstance(name, str) and name.lower() in ("pad", "ffill", "bfill", "backfill"):
        return METHOD_CODES[name.lower()]
    raise KeyError(name)


class EngineUnavailable(RuntimeError):
    """libivs.so missing / not loadable / no MI355X visible.  Never swallowed by the host code."""


class EngineError(RuntimeError):
    """A C-ABI call returned a negative code."""


_p = C.c_void_p
_i64, _i32, _sz = C.c_int64, C.c_int32, C.c_size_t



class FrameArgs(C.Structure):
    """ivs_frame_args of include/ivs.h (field for field)."""
    _fields_ = [("src_pos", _p), ("src_off", _p), ("q_off", _p),
                ("n_series", _i64), ("total_src", _i64), ("total_queries", _i64),
                ("yk", _p), ("yk_stride", _i64), ("n_channels", _i32), ("method", _i32),
                ("chan_out", _p), ("chan_stride", _i64), ("status", _p),
                ("valid", _p), ("valid_stride", _i64), ("n_valid", _i32),
                ("fsrc", _p), ("fsrc_stride", _i64), ("f_rows", _p), ("n_f", _i32), ("f_out", _p), ("f_stride", _i64),
                ("csrc", _p), ("csrc_stride", _i64), ("c_rows", _p), ("n_c", _i32), ("c_out", _p), ("c_stride", _i64),
                ("idx_rows", _p), ("n_idx", _i32), ("idx_out", _p), ("idx_stride", _i64),
                ("first_ns", _p), ("needs", _p), ("sym_col", _i32), ("date_ns", _p), ("keep", _p),
                ("g_strike", _i32), ("g_rate", _i32), ("g_put", _i32), ("strike_src", _p), ("rate_src", _p), ("put_src", _p),
                ("ch_iv", _i32), ("ch_underlying", _i32), ("ch_ttm", _i32), ("greeks", _p), ("greeks_stride", _i64)]


# every symbol include/ivs.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "ivs_version": (C.c_int, []),
    "ivs_last_error": (C.c_char_p, []),
    "ivs_last_kernel": (C.c_char_p, []),
    "ivs_device_count": (C.c_int, []),
    "ivs_interp1d_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "ivs_interp1d_batch_f64": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _i64, _p, _p, _i64, _p, _i64, _p, _i32,
                                         _p, _sz, _p]),
    "ivs_interp1d_greeks_batch_f64": (C.c_int, [_p, _p, _i64, _p, _i64, _i32, _i64, _p, _p, _i64, _p, _i64, _p, _i32,
                                                _i32, _i32, _i32, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _i64,
                                                _p, _sz, _p]),
    "ivs_ffill_index_batch": (C.c_int, [_p, _p, _p, _i64, _i32, _p, _i64, _i64, _p, _i64, _p]),
    "ivs_gather_rows_f64": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _i64, _p, _i64, _p]),
    "ivs_gather_rows_i32": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _i64, _p, _i64, _p]),
    "ivs_frame_rows": (C.c_int, [_p, _i64, _i64, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p]),
    "ivs_frame_workspace_bytes": (_sz, [_i64, _i64, _i32]),
    "ivs_frame_columns_f64": (C.c_int, [C.POINTER(FrameArgs), _p, _sz, _p]),
    "ivs_bs_greeks_f64": (C.c_int, [_p, _p, _p, _p, _p, _p, _i32, _i64, _p, _p, _p, _p, _p, _p]),
    "ivs_candle_aggregate_f64": (C.c_int, [_p] * 7 + [_i64, _i64, _i64] + [_p] * 8),
    "ivs_bridge_workspace_bytes": (C.c_size_t, [_i64]),
    "ivs_mt19937_words_u32": (C.c_int, [C.c_uint32, _p, _i64, _p]),
    "ivs_bridge_candles_f64": (C.c_int, [_p, _p, _p, _i64, _i64, _i32, C.c_double, C.c_double, _p, _i64, _p, _p, _p,
                                         _p, C.c_size_t, _p]),
    "ivs_debug_stamps": (C.c_int, [_p, _i64]),
    "ivs_debug_last_grid": (_i64, []),
    "ivs_debug_mode_offset": (_i64, []),
    "ivs_surface_workspace_bytes": (_sz, [_i64, _i32]),
    "ivs_surface_batch_f64": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _i32, _p, _i64, _p, _i64, _i32, _p, _i64, _i32,
                                        _p, _p, _i32, _i32, _p, _sz, _p]),
}

_lib = None


def load():
    """Load libivs.so (no GPU needed for loading).  Raises EngineUnavailable if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch-ROCm ships its own libamdhip64; libivs.so must bind to THAT copy, so torch has to be loaded first.  Loading
    # libivs.so (-> /opt/rocm's runtime) before torch leaves two HIP runtimes in the process and torch sees no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise EngineUnavailable(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EngineUnavailable(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ivs_version() != ABI_VERSION:
        raise EngineUnavailable(f"libivs.so ABI {lib.ivs_version()} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise EngineError(f"{what} failed ({rc}): {load().ivs_last_error().decode()}")
