"""Deterministic synthetic IV-surface batches (SURVEY.md section 8d generator).

Per snapshot: spot S ~ U(20000, 30000); moneyness grid m = linspace(0.70, 1.30, nK) with
per-snapshot jitter U(-0.2, 0.2)*dm (strictly increasing, non-uniform); maturities
T = [1,2,3,7,14,21,30,45,60,90,120,150,180,270,365,540]/365 (first nT); SVI-like vols
sigma = sqrt(a + b(rho(k-mu) + sqrt((k-mu)^2 + s^2))), k = ln(K/S)/sqrt(T), plus N(0, 0.002^2)
noise, clipped to [0.05, 3.0].  Strikes are handed to the engine in moneyness units K/S so
that the query grid Kq = linspace(0.72, 1.28, mK) is ONE shared array (the byte count of
section 8d excludes query grids); `absolute=True` gives K = S*m and per-surface Kq = S*linspace(..).

`numpy_batch` (host, numpy Generator, seed + rank) feeds parity tests; `torch_batch`
generates the same distribution directly in HBM for full-size runs (different RNG stream).
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 20230320
TENORS_DAYS = np.array([1, 2, 3, 7, 14, 21, 30, 45, 60, 90, 120, 150, 180, 270, 365, 540], np.float64)


def tenors(nT: int) -> np.ndarray:
    return TENORS_DAYS[:nT] / 365.0


def query_grids(mK: int, mT: int, nT: int = 16):
    Kq = np.linspace(0.72, 1.28, mK)
    Tq = np.geomspace(2.0 / 365.0, min(1.4, tenors(nT)[-1]), mT)
    return Kq, Tq


def numpy_batch(B: int, nK: int = 64, nT: int = 16, seed: int = BASE_SEED, nan_frac: float = 0.0,
                absolute: bool = False):
    """Uniform batch.  Returns dict(K [B,nK], T [nT], sigma [B,nT,nK], S [B])."""
    r = np.random.default_rng(seed)
    S = r.uniform(20000.0, 30000.0, B)
    dm = 0.6 / (nK - 1)
    m = np.linspace(0.70, 1.30, nK)[None, :] + r.uniform(-0.2, 0.2, (B, nK)) * dm
    T = tenors(nT)
    k = np.log(m)[:, None, :] / np.sqrt(T)[None, :, None]
    a = r.uniform(.15, .5, (B, 1, 1)); b = r.uniform(.05, .3, (B, 1, 1)); rho = r.uniform(-.7, .1, (B, 1, 1))
    mu = r.normal(0, .05, (B, 1, 1)); s = r.uniform(.1, .4, (B, 1, 1))
    sig = np.sqrt(a + b * (rho * (k - mu) + np.sqrt((k - mu) ** 2 + s ** 2))) + r.normal(0, 0.002, k.shape)
    sig = np.clip(sig, 0.05, 3.0)
    if nan_frac > 0:
        sig[r.random(sig.shape) < nan_frac] = np.nan
    K = m * S[:, None] if absolute else m
    return {"K": K, "T": T, "sigma": sig, "S": S}


def numpy_ragged_batch(B: int, nT: int = 16, lo: int = 8, hi: int = 128, seed: int = BASE_SEED):
    """Config 5: per-snapshot strike counts ~ randint(lo, hi+1), CSR layout.
    Returns dict(K [total], k_off [B+1], T [nT], sigma [nT*total] (surface b = [nT][nK_b]), nK_max)."""
    r = np.random.default_rng(seed)
    nk = r.integers(lo, hi + 1, B)
    k_off = np.concatenate([[0], np.cumsum(nk)]).astype(np.int64)
    T = tenors(nT)
    K = np.empty(int(k_off[-1])); sig = np.empty(nT * int(k_off[-1]))
    for b in range(B):
        n = int(nk[b])
        one = numpy_batch(1, n, nT, seed=int(r.integers(1 << 31)))
        K[k_off[b]:k_off[b + 1]] = one["K"][0]
        sig[nT * k_off[b]:nT * k_off[b + 1]] = one["sigma"][0].ravel()
    return {"K": K, "k_off": k_off, "T": T, "sigma": sig, "nK_max": int(nk.max()), "nk": nk}


def torch_batch(B: int, nK: int = 64, nT: int = 16, seed: int = BASE_SEED, device="cuda", chunk: int = 65536):
    """Same distribution generated in HBM (torch is plumbing here: RNG + elementwise setup, not the hot path)."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    f64 = dict(dtype=torch.float64, device=device)
    K = torch.empty((B, nK), **f64); sig = torch.empty((B, nT, nK), **f64)
    T = torch.tensor(tenors(nT), **f64)
    base = torch.linspace(0.70, 1.30, nK, **f64)
    dm = 0.6 / (nK - 1)
    u = lambda lo, hi, shape: torch.rand(shape, generator=g, **f64) * (hi - lo) + lo   # noqa: E731
    for a0 in range(0, B, chunk):
        n = min(chunk, B - a0)
        m = base[None, :] + u(-0.2, 0.2, (n, nK)) * dm
        k = torch.log(m)[:, None, :] / torch.sqrt(T)[None, :, None]
        a = u(.15, .5, (n, 1, 1)); b = u(.05, .3, (n, 1, 1)); rho = u(-.7, .1, (n, 1, 1))
        mu = torch.randn((n, 1, 1), generator=g, **f64) * .05; s = u(.1, .4, (n, 1, 1))
        v = torch.sqrt(a + b * (rho * (k - mu) + torch.sqrt((k - mu) ** 2 + s ** 2)))
        v = v + torch.randn(v.shape, generator=g, **f64) * 0.002
        sig[a0:a0 + n] = torch.clamp(v, 0.05, 3.0)
        K[a0:a0 + n] = m
    return {"K": K, "T": T, "sigma": sig}


def ragged_counts(B: int, lo: int = 8, hi: int = 128, seed: int = BASE_SEED):
    """Strike counts of a config-5 batch (host, deterministic in `seed`): every rank of a sharded run draws the SAME
    counts for the global batch and then builds only its own shard (torch_ragged_batch(nk=counts[lo:hi]))."""
    return np.random.default_rng(seed).integers(lo, hi + 1, B).astype(np.int64)


def torch_ragged_batch(B: int, nT: int = 16, lo: int = 8, hi: int = 128, seed: int = BASE_SEED, device="cuda", nk=None):
    """Config 5 in HBM: counts ~ randint(lo, hi+1) (or the given `nk`); strikes/vols from the same formulas (flat CSR)."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    f64 = dict(dtype=torch.float64, device=device)
    if nk is None:
        nk = torch.randint(lo, hi + 1, (B,), generator=g, device=device, dtype=torch.int64)
    else:
        nk = torch.as_tensor(np.asarray(nk, np.int64), device=device)
        B = int(nk.numel())
    k_off = torch.zeros(B + 1, dtype=torch.int64, device=device); k_off[1:] = torch.cumsum(nk, 0)
    total = int(k_off[-1])
    sid = torch.repeat_interleave(torch.arange(B, device=device), nk)           # surface of each strike
    j = torch.arange(total, device=device) - k_off[sid]                         # index within surface
    n1 = (nk[sid] - 1).to(torch.float64)
    u = lambda lo_, hi_, shape: torch.rand(shape, generator=g, **f64) * (hi_ - lo_) + lo_   # noqa: E731
    m = 0.70 + 0.60 * j.to(torch.float64) / n1 + u(-0.2, 0.2, (total,)) * (0.6 / n1)
    T = torch.tensor(tenors(nT), **f64)
    a = u(.15, .5, (B,)); b = u(.05, .3, (B,)); rho = u(-.7, .1, (B,))
    mu = torch.randn((B,), generator=g, **f64) * .05; s = u(.1, .4, (B,))
    # sigma flat: surface b occupies [nT*k_off[b], nT*k_off[b+1]) as [nT][nK_b]
    sig = torch.empty(nT * total, **f64)
    nkb = nk[sid]
    for t in range(nT):
        k = torch.log(m) / float(np.sqrt(tenors(nT)[t]))
        v = torch.sqrt(a[sid] + b[sid] * (rho[sid] * (k - mu[sid]) + torch.sqrt((k - mu[sid]) ** 2 + s[sid] ** 2)))
        v = torch.clamp(v + torch.randn((total,), generator=g, **f64) * 0.002, 0.05, 3.0)
        sig[nT * k_off[sid] + t * nkb + j] = v
    return {"K": m, "k_off": k_off, "T": T, "sigma": sig, "nK_max": int(nk.max()), "nk": nk}
