"""CPU: libivs.so loads and exports every symbol include/ivs.h declares (no compute calls)."""
import ctypes
import os
import re

from iv_interpolation_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ivs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ivs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ivs.h but not exported by libivs.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.ivs_version() == _lib.ABI_VERSION
    assert isinstance(lib.ivs_last_error(), bytes)
    assert lib.ivs_interp1d_workspace_bytes(100, 2, 3) >= 4 * 3 * 100 * 8


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    # unknown method -> IVS_EINVAL before anything touches the device
    rc = lib.ivs_surface_batch_f64(None, None, 0, 64, None, 0, 16, None, 1, None, 0, 64, None, 0, 16, None, None, 99, 0, None, 0, None)
    assert rc == -22 and b"unknown method" in lib.ivs_last_error()
    rc = lib.ivs_surface_batch_f64(None, None, 0, 64, None, 0, 16, None, 1, None, 0, 64, None, 0, 16, None, None, 0, 0, None, 0, None)
    assert rc == -22 and b"null pointer" in lib.ivs_last_error()
    rc = lib.ivs_interp1d_batch_f64(None, None, 0, None, 1, 1, 0, None, None, 0, None, 0, None, 7, None, 0, None)
    assert rc == -22


def test_shape_validation_codes_without_gpu():
    """Host-side validation returns errno-style codes before any launch (fake non-null pointers are never dereferenced)."""
    import ctypes as C
    lib = _lib.load()
    P = C.c_void_p(64)
    call = lambda **kw: lib.ivs_surface_batch_f64(  # noqa: E731
        P, kw.get("k_off"), kw.get("k_stride", 64), kw.get("nK", 64), P, 0, kw.get("nT", 16), P, kw.get("B", 1),
        P, 0, kw.get("mK", 64), P, 0, kw.get("mT", 16), P, None, kw.get("method", 0), 0, kw.get("ws"), kw.get("ws_bytes", 0), None)
    assert call(B=0) == 0 and call(mK=0) == 0                      # empty batches / grids are a no-op
    assert call(nT=33) == -34 and b"nT=33" in lib.ivs_last_error()  # IVS_ERANGE
    assert call(nT=0) == -34
    assert call(k_stride=10) == -22                                  # k_stride < nK
    assert call(B=-1) == -22
    assert call(nK=4000, k_stride=4000, method=1) == -34 and b"LDS" in lib.ivs_last_error()
    # the surface call never allocates: scratch is the caller's workspace, sized by ivs_surface_workspace_bytes
    need = lib.ivs_surface_workspace_bytes(1, 0)
    assert need >= 4096 and lib.ivs_surface_workspace_bytes(1000, 1) >= need + 1000 * 16
    assert lib.ivs_surface_workspace_bytes(1000, 0) == need                      # uniform batches: tables only
    assert call() == -12 and b"workspace" in lib.ivs_last_error()               # IVS_ENOMEM: no workspace
    assert call(ws=C.c_void_p(4096), ws_bytes=need - 1) == -12
    assert call(ws=C.c_void_p(4096 + 8), ws_bytes=need) == -22 and b"aligned" in lib.ivs_last_error()
    # int32 gather indices / stream offsets are range-checked instead of overflowing silently
    assert lib.ivs_ffill_index_batch(P, P, P, 1 << 31, 1, P, 1, 10, P, 10, None) == -34
    assert lib.ivs_interp1d_batch_f64(P, P, 5, P, 1, 1, 10, None, P, 10, P, 10, P, 0, P, 1 << 20, None) == -22   # stride < rows
    assert lib.ivs_interp1d_batch_f64(P, P, 10, P, 1, 1, 10, None, P, 10, P, 10, P, 0, P, 8, None) == -12        # workspace too small
    assert lib.ivs_candle_aggregate_f64(P, P, P, P, P, P, P, 1, 10, 0, P, P, P, P, P, P, P, None) == -22          # freq 0


def test_bridge_validation_codes_without_gpu():
    import ctypes as C
    lib = _lib.load()
    P = C.c_void_p(64)
    call = lambda **kw: lib.ivs_bridge_candles_f64(  # noqa: E731
        P, None, P, kw.get("S", 1), kw.get("rows", 100), kw.get("strategy", 0), 0.002, 1.5, P, kw.get("n_words", 1000),
        P, P, kw.get("tail", P), P, kw.get("ws", 1 << 20), None)
    assert call(strategy=7) == -22 and b"unknown strategy" in lib.ivs_last_error()
    assert call(rows=-1) == -22
    assert call(tail=None) == -22 and b"rng_tail" in lib.ivs_last_error()
    assert call(ws=8) == -12                                            # workspace too small
    assert call(n_words=1 << 31) == -34 and b"int32" in lib.ivs_last_error()
    assert lib.ivs_bridge_workspace_bytes(1000) >= 1000 * 20 and lib.ivs_bridge_workspace_bytes(-5) == lib.ivs_bridge_workspace_bytes(0)
    assert lib.ivs_mt19937_words_u32(1, None, -1, None) == -22
    assert lib.ivs_mt19937_words_u32(1, None, 0, None) == 0             # nothing to do
    assert lib.ivs_mt19937_words_u32(1, None, 10, None) == -22          # null buffer
