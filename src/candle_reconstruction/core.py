"""Import-path shim for the reference's ``candle_reconstruction.core`` (MI355X engine underneath)."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from iv_interpolation_amd.candles import CandleReconstructor, MultiSymbolCandleReconstructor  # noqa: E402,F401
