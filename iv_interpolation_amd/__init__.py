"""MI355X-native implied-volatility interpolation engine (hot path of liu-wei2021/IV_INTERPOLATION).

    from iv_interpolation_amd import IVInterpolator          # drop-in for interpolation.core.IVInterpolator
    from iv_interpolation_amd import engine                   # device-side batch API (surfaces, 1-D, ffill)

Numerics run in hand-written HIP kernels (iv_interpolation_amd/csrc) behind the C ABI in
include/ivs.h; torch-ROCm is used only for device buffers, streams and torch.distributed.
"""
from ._lib import (CUBIC, CUBICSPLINE, LINEAR, METHOD_CODES, SLINEAR, ST_OK, ST_TOO_FEW_KNOTS,  # noqa: F401
                   EngineError, EngineUnavailable)
from .core import IVInterpolator  # noqa: F401

__all__ = ["IVInterpolator", "EngineUnavailable", "EngineError", "METHOD_CODES",
           "LINEAR", "CUBIC", "CUBICSPLINE", "SLINEAR", "ST_OK", "ST_TOO_FEW_KNOTS"]
