"""CPU baselines of SURVEY section 8d / BASELINE.md section 3, timed by bench.py's cpu_baseline leg.  TEST / MEASUREMENT
INFRASTRUCTURE ONLY (never on the product path).

B1 "reference-shaped": the reference has no surface routine -- its unit of work is ``interpolate_symbol`` on ONE
   DataFrame (``/root/reference/src/interpolation/core.py:16-85``), fanned out one OS process per symbol
   (``batch_processor.py:234-239``).  A 64 x 16 surface posed that way is 16 strike series + 64 maturity series, each a
   DataFrame call through ``oracle/ref_symbol.py`` (this repo's restatement of that method, pinned by the golden
   frames).  Knots sit 4 minutes apart ("64 (4 min) -> 253-point grid"), ``min_points=2``.
   At ~10^2 series/s/core this runs for a bounded wall-clock budget and the rate is EXTRAPOLATED from the surfaces done.
B2 "vectorised NumPy": ``ivs_oracle.surface_batch`` (batched np.interp / not-a-knot Thomas over the whole chunk), one
   process per core.

Both use ``ProcessPoolExecutor(max_workers=cores)``.  They are started by bench.py BEFORE the process touches the GPU
(a forked/spawned worker of a GPU-initialised parent is what the pool's rules forbid).
"""
from __future__ import annotations

import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _setup_path():
    for p in (_HERE, _ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)


def _b2_worker(args):
    seed, n, nK, nT, mK, mT, method, budget = args
    _setup_path()
    import ivs_oracle as O
    from iv_interpolation_amd import synth
    d = synth.numpy_batch(n, nK, nT, seed=seed)
    Kq, Tq = synth.query_grids(mK, mT, nT)
    code = O.METHOD_CODES[method]
    O.surface_batch(d["K"][:64], d["T"], d["sigma"][:64], Kq, Tq, code)          # warm (imports, allocator)
    t0 = time.perf_counter(); done = 0
    while True:
        O.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, code); done += n
        if time.perf_counter() - t0 >= budget:
            break
    return done, time.perf_counter() - t0


def _b1_worker(args):
    seed, nK, nT, method, budget = args
    _setup_path()
    import warnings
    import pandas as pd
    import ref_symbol
    from iv_interpolation_amd import synth
    warnings.simplefilter("ignore")
    d = synth.numpy_batch(64, nK, nT, seed=seed)
    base = pd.Timestamp("2023-03-20 00:00:00")

    def series(values):
        n = len(values)
        return pd.DataFrame({"symbol": "s", "date": base + pd.to_timedelta(np.arange(n) * 4, unit="min"),
                             "iv": values, "underlying_price": 25000.0, "time_to_maturity": 0.1})

    t0 = time.perf_counter(); done = 0; b = 0
    while time.perf_counter() - t0 < budget:
        sig = d["sigma"][b % 64]                                    # [nT, nK]
        cols = []
        for t in range(nT):                                        # strike pass: nT series of nK knots
            r = ref_symbol.interpolate_symbol(series(sig[t]), method, min_points=2)
            cols.append(r["iv"].to_numpy()[:: max(1, (len(r) - 1) // max(nK - 1, 1))][:nK])
        Z = np.stack(cols)                                         # [nT, nK]
        for k in range(nK):                                        # maturity pass: nK series of nT knots
            ref_symbol.interpolate_symbol(series(Z[:, k]), method, min_points=2)
        done += 1; b += 1
    return done, time.perf_counter() - t0


def run_b2(method: str, nK: int, nT: int, mK: int, mT: int, budget_s: float = 6.0, per_proc: int = 2000, seed: int = 20230320):
    cores = usable_cpus()
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(_b2_worker, [(seed + i, per_proc, nK, nT, mK, mT, method, budget_s) for i in range(cores)]))
    wall = time.perf_counter() - t0
    rate = sum(n / t for n, t in res)
    return {"value": rate, "unit": "surfaces/s", "cores": cores, "kind": "port",
            "sample": f"B2 vectorised NumPy oracle (oracle/ivs_oracle.py), {cores} processes x {per_proc}-surface chunks for "
                      f"{budget_s:.0f} s each ({sum(n for n, _ in res)} surfaces, wall {wall:.1f} s), method {method}"}


def run_b1(method: str, nK: int, nT: int, budget_s: float = 8.0, seed: int = 20230320):
    cores = usable_cpus()
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=cores) as ex:
        res = list(ex.map(_b1_worker, [(seed + i, nK, nT, method, budget_s) for i in range(cores)]))
    wall = time.perf_counter() - t0
    n = sum(k for k, _ in res)
    rate = sum(k / t for k, t in res)
    return {"value": rate, "unit": "surfaces/s", "cores": cores, "kind": "port",
            "sample": f"B1 reference-shaped: one DataFrame call per 1-D series ({nT} strike + {nK} maturity series per surface) "
                      f"through oracle/ref_symbol.py, ProcessPoolExecutor({cores}) as batch_processor.py:234; EXTRAPOLATED from "
                      f"{n} surfaces in {budget_s:.0f} s per worker (wall {wall:.1f} s), method {method}"}
