"""Import-path shim: the reference's scripts do ``sys.path.append(<root>/src)`` and then
``from interpolation.core import IVInterpolator`` (reference complete_pipeline.py:30,34;
main.py:21,27; batch_processor.py:9).  Same path here, MI355X engine underneath."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from iv_interpolation_amd.core import IVInterpolator, logger  # noqa: E402,F401

__all__ = ["IVInterpolator"]
