"""ctypes loader of the C oracle (oracle/ivs_oracle_c.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libivs_oracle.so")
_P = C.c_void_p


class _Runner:
    def __init__(self, lib):
        self.lib = lib
        lib.ivs_oracle_threads.restype = C.c_int
        lib.ivs_oracle_surface_batch.restype = C.c_int
        lib.ivs_oracle_surface_batch.argtypes = [_P, _P, C.c_int64, C.c_int, _P, C.c_int64, C.c_int, _P, C.c_int64, _P,
                                                 C.c_int64, C.c_int, _P, C.c_int64, C.c_int, _P, _P, C.c_int]
        lib.ivs_oracle_interp1d_batch.restype = C.c_int
        lib.ivs_oracle_interp1d_batch.argtypes = [_P, _P, C.c_int64, _P, C.c_int64, C.c_int, _P, _P, _P, C.c_int64, _P, C.c_int]

    def threads(self):
        return int(self.lib.ivs_oracle_threads())

    def set_threads(self, n):
        self.lib.ivs_oracle_set_threads(int(n))

    def surface_batch(self, K, T, sigma, Kq, Tq, method, k_off=None):
        f = lambda a: np.ascontiguousarray(a, np.float64)   # noqa: E731
        K, T, sigma, Kq, Tq = f(K), f(T), f(sigma), f(Kq), f(Tq)
        if k_off is None:
            B, nT, nK = sigma.shape
            ks = 0 if K.ndim == 1 else nK
            ko = None
        else:
            k_off = np.ascontiguousarray(k_off, np.int64)
            B = len(k_off) - 1; nT = T.shape[-1]; nK = int(np.diff(k_off).max()); ks = 0
            ko = k_off.ctypes.data
        mK, mT = Kq.shape[-1], Tq.shape[-1]
        out = np.empty((B, mT, mK)); st = np.zeros(B, np.int32)
        rc = self.lib.ivs_oracle_surface_batch(K.ctypes.data, ko, ks, nK, T.ctypes.data, 0 if T.ndim == 1 else nT, nT,
                                               sigma.ctypes.data, B, Kq.ctypes.data, 0 if Kq.ndim == 1 else mK, mK,
                                               Tq.ctypes.data, 0 if Tq.ndim == 1 else mT, mT, out.ctypes.data,
                                               st.ctypes.data, int(method))
        if rc:
            raise ValueError("C oracle: size beyond NMAX" if rc == -1 else "C oracle: method not restated in C")
        return out, st


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (GPU boxes give a share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def load():
    if not os.path.exists(_PATH):
        raise FileNotFoundError(_PATH + " (run `make -C oracle`)")
    r = _Runner(C.CDLL(_PATH))
    r.set_threads(usable_cpus())
    return r
