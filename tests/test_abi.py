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
    rc = lib.ivs_surface_batch_f64(None, None, 0, 64, None, 0, 16, None, 1, None, 0, 64, None, 0, 16, None, None, 99, 0, None)
    assert rc == -22 and b"unknown method" in lib.ivs_last_error()
    rc = lib.ivs_surface_batch_f64(None, None, 0, 64, None, 0, 16, None, 1, None, 0, 64, None, 0, 16, None, None, 0, 0, None)
    assert rc == -22 and b"null pointer" in lib.ivs_last_error()
    rc = lib.ivs_interp1d_batch_f64(None, None, 0, None, 1, 1, 0, None, None, 0, None, 0, None, 7, None, 0, None)
    assert rc == -22
