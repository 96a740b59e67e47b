#!/usr/bin/env python3
"""Headline benchmark: IV surfaces/s on the BASELINE.json workload (1M snapshots, 64x16 quote grid
-> 64x16 output grid, fp64), one process per GPU, batch sharded with no data-path collective.

    python bench.py [--gpus N --steps K --warmup W] [--method cubic|linear|...] [--workload cfg3|cfg4|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (one ivs_surface_batch_f64 launch) over the rank's whole
resident batch.  Inputs live in HBM before the timed region.  Rank 0 prints ONE JSON line.
Weak scaling: every rank holds `--batch` surfaces (default 1M), value = total surfaces / max-rank time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

WORKLOADS = {
    # name: (nK, nT, mK, mT, ragged, description)
    "cfg3": (64, 16, 64, 16, False, "1M snapshots x (64 strikes x 16 maturities) -> 64x16 output grid"),
    "cfg4": (64, 16, 256, 64, False, "1M snapshots x (64x16) -> dense 256x64 output grid"),
    "cfg5": (128, 16, 64, 16, True, "1M ragged snapshots, 8..128 strikes x 16 maturities -> 64x16"),
}


def algorithmic_bytes(B, nK, nT, mK, mT, total_strikes=None):
    """per surface sigma nT*nK*8 + K nK*8 + T nT*8 in, mT*mK*8 out (query grids shared,
    excluded); ragged adds the 8-byte CSR offset and uses the actual strike counts."""
    if total_strikes is None:
        return B * (8 * (nT * nK + nK + nT) + 8 * mT * mK)
    return 8 * (nT * total_strikes + total_strikes + nT * B) + B * (8 * mT * mK + 8)


def cpu_baseline(method, nK, nT, mK, mT, budget_s=20.0, sample=None):
    """The oracle ('port' of the reference's NumPy/SciPy arithmetic) timed on this box's host cores on a
    bounded sample of the same workload.  Uses the compiled C oracle with OpenMP when it is built,
    else the vectorised NumPy oracle on one core.  `sample` = (K, T, sigma, Kq, Tq, device results) of a few
    surfaces of the timed batch: they are recomputed here and compared (returned as the second value)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ivs_oracle as O
    from iv_interpolation_amd import synth
    code = O.METHOD_CODES[method]
    check = {}
    if sample is not None:
        Ks, Ts, sg, Kqs, Tqs, got = sample
        ref, _ = O.surface_batch(Ks, Ts, sg, Kqs, Tqs, code)
        check = {"surfaces": int(len(Ks)), "max_abs_diff_vs_oracle": float(np.nanmax(np.abs(got - ref))),
                 "bit_exact": bool(np.array_equal(got, ref, equal_nan=True))}
    Kq, Tq = synth.query_grids(mK, mT, nT)
    try:
        import c_oracle
        runner = c_oracle.load()
    except Exception:
        runner = None
    if runner is not None:
        cores = runner.threads()
        n = 20000
        d = synth.numpy_batch(n, nK, nT, seed=synth.BASE_SEED)
        runner.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, code)         # warm
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < budget_s / 2 or reps == 0:
            runner.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, code); reps += 1
        dt = time.perf_counter() - t0
        return {"value": n * reps / dt, "unit": "surfaces/s", "cores": cores, "kind": "port",
                "sample": f"{reps} x {n} surfaces of the same generator, C oracle (oracle/ivs_oracle_c.c, OpenMP), method {method}"}, check
    n = 2000
    d = synth.numpy_batch(n, nK, nT, seed=synth.BASE_SEED)
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < budget_s / 2 or reps == 0:
        O.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, code); reps += 1
    dt = time.perf_counter() - t0
    return {"value": n * reps / dt, "unit": "surfaces/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x {n} surfaces of the same generator, vectorised NumPy oracle (oracle/ivs_oracle.py), method {method}"}, check


def load_traffic(workload, method, kernel, batch):
    """HBM bytes per launch from the committed PMC pass (profiles/traffic.json), or None.  The pass was collected at the
    batch size recorded with it; other batch sizes of the same workload scale linearly (every surface is streamed once)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            e = json.load(f).get(f"{workload}:{method}:{kernel}")
        if not e:
            return None
        return e["hbm_bytes_per_launch"] * (batch / e.get("batch", batch))
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--method", default="cubic", choices=["linear", "cubic", "cubicspline", "slinear", "pchip", "akima"])
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1_000_000, help="surfaces per GPU (weak scaling)")
    ap.add_argument("--nk", type=int, default=0, help="override the strike count of a uniform workload (variable-shape kernel)")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", type=int, default=256, help="surfaces compared with the oracle after the timed region")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed region, also time the optional all-gather of the output shards (RCCL over xGMI); "
                         "reported separately as 'gather', never part of 'value'")
    a = ap.parse_args()

    import numpy as np
    import torch
    from iv_interpolation_amd import engine, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with python -m torch.distributed.run --nproc-per-node N")
    # one process per GPU; IVS_DIST_BACKEND=gloo + fewer devices than ranks is only for rehearsing the N>1 code
    # path on a 1-GPU box (ranks then share cuda:0)
    backend = os.environ.get("IVS_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local if backend == "nccl" else local % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1 or os.environ.get("IVS_FORCE_DIST") == "1":     # IVS_FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    nK, nT, mK, mT, ragged, desc = WORKLOADS[a.workload]
    if a.nk and not ragged:
        nK = a.nk
        desc = desc.replace("64 strikes", f"{nK} strikes").replace("(64x16)", f"({nK}x16)")
    B = a.batch
    seed = synth.BASE_SEED + rank
    Kq_h, Tq_h = synth.query_grids(mK, mT, nT)
    Kq = torch.from_numpy(Kq_h).cuda(); Tq = torch.from_numpy(Tq_h).cuda()
    if ragged:
        d = synth.torch_ragged_batch(B, nT, 8, 128, seed=seed)
        kw = dict(k_off=d["k_off"], nK_max=d["nK_max"], n_maturities=nT)
        total_strikes = int(d["k_off"][-1])
    else:
        d = synth.torch_batch(B, nK, nT, seed=seed)
        kw = {}
        total_strikes = None
    out = torch.empty((B, mT, mK), dtype=torch.float64, device="cuda")
    status = torch.empty((B,), dtype=torch.int32, device="cuda")

    ws = engine.surface_workspace(B, ragged)          # caller-owned scratch: the timed call allocates nothing

    def step():
        engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, a.method, out=out, status=status,
                             force_generic=a.force_generic, workspace=ws, **kw)

    # DVFS spin-up, before (and in addition to) the W warm-up steps: after any idle gap the first ~25 ms of launches
    # run 10-25 % slower while the clocks ramp (tools/warm_probe.py: 4.55, 4.56, 4.13, 4.00, 3.83, 3.77, 3.70 ms ...),
    # which would otherwise leak into the timed steps whenever W is small.  Untimed, fixed 100 ms of device time.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.1:
        step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    kernel = engine.last_kernel()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = [s.elapsed_time(e) for s, e in ev]
    if dist:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
    assert int(status.max()) == 0

    gather = None
    if a.gather and dist:
        # results normally stay sharded; the one optional collective is an all-gather of the output tiles
        from iv_interpolation_amd import sharding
        torch.cuda.synchronize(); dist.barrier()
        g0 = time.perf_counter()
        full = sharding.gather_outputs(out)
        torch.cuda.synchronize(); dist.barrier()
        gather = {"ms": (time.perf_counter() - g0) * 1e3, "bytes_per_rank": out.numel() * 8, "ranks": world,
                  "gathered_shape": list(full.shape)}
        del full

    # parity spot check: a few surfaces are handed to the cpu_baseline leg below, the only place where bench.py touches
    # the oracle (it is the checker there, never the thing measured)
    check = {}
    sample = None
    if rank == 0 and world == 1 and a.check > 0 and not ragged and not a.no_cpu_baseline:
        idx = torch.linspace(0, B - 1, a.check, device="cuda").long()
        sample = (d["K"][idx].cpu().numpy(), d["T"].cpu().numpy(), d["sigma"][idx].cpu().numpy(), Kq_h, Tq_h,
                  out[idx].cpu().numpy())

    copy_gbps = None
    if rank == 0 and world == 1:
        # context: what a plain device copy reaches on THIS device (read + write stream), same HIP-event timing
        nbytes = 2 << 30
        src = torch.empty(nbytes // 8, dtype=torch.float64, device="cuda").fill_(1.0); dst = torch.empty_like(src)
        dst.copy_(src); torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            dst.copy_(src)
        c1.record(); torch.cuda.synchronize()
        copy_gbps = 5 * 2 * nbytes / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst

    if rank == 0:
        total = B * world
        bytes_launch = algorithmic_bytes(B, nK, nT, mK, mT, total_strikes)
        avg_ms = sum(kern_ms) / len(kern_ms)
        achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
        res = {
            "metric": "IV surfaces/sec (1M-batch, 64x16 grid)", "value": total * a.steps / wall, "unit": "surfaces/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {desc}", "method": a.method, "surfaces_per_gpu": B,
                       "quote_grid": [nK, nT], "output_grid": [mK, mT], "kernel": kernel,
                       "sharding": f"{world} x contiguous shard, no collective", "seed": synth.BASE_SEED},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(a.workload, a.method, kernel, B),
                         "kernel": kernel, "kernel_ms_avg": avg_ms, "kernel_ms_min": min(kern_ms),
                         "kernel_ms_median": sorted(kern_ms)[len(kern_ms) // 2],
                         "device_copy_GBps": copy_gbps, "frac_of_device_copy": (achieved / copy_gbps) if copy_gbps else None,
                         "algorithmic_bytes_per_launch": bytes_launch, "timing": "HIP events around each launch on the launch stream"},
            "parity_check": check,
        }
        if gather:
            res["gather"] = gather
        if world == 1 and not a.no_cpu_baseline and not ragged:
            res["cpu_baseline"], res["parity_check"] = cpu_baseline(a.method, nK, nT, mK, mT, sample=sample)
        print(json.dumps(res), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
