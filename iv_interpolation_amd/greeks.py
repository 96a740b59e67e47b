"""Drop-in for the reference's ``src/interpolation/greeks.py`` (BlackScholesGreeks.calculate_greeks, :12-43) on the
MI355X engine.  NumPy arrays / scalars in -> NumPy out (one H2D/D2H round trip); torch CUDA tensors in -> tensors out."""
import numpy as np

from . import engine


class BlackScholesGreeks:
    @staticmethod
    def calculate_greeks(S, K, T, r, sigma, option_type="call"):
        torch = engine.require_device()
        on_device = all(hasattr(a, "is_cuda") and a.is_cuda for a in (S, K, T, r, sigma))
        if on_device:
            return engine.bs_greeks(S, K, T, r, sigma, default_is_put=(option_type != "call"))
        arrs = np.broadcast_arrays(*[np.asarray(a, np.float64) for a in (S, K, T, r, sigma)])
        shape = arrs[0].shape
        dev = [torch.from_numpy(np.ascontiguousarray(a).ravel()).cuda() for a in arrs]
        out = engine.bs_greeks(*dev, default_is_put=(option_type != "call"))
        res = {k: v.cpu().numpy().reshape(shape) for k, v in out.items()}
        if shape == ():
            res = {k: np.float64(v) for k, v in res.items()}
        return res
