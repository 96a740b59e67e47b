#!/usr/bin/env python3
"""Headline benchmark: IV surfaces/s on the BASELINE.json workload (config 3: 1M snapshots, 64x16 quote grid ->
64x16 output grid, fp64), one process per GPU, batch sharded with no data-path collective.

    python bench.py [--gpus N --steps K --warmup W] [--method cubic|linear|...] [--workload cfg3|cfg4|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W [--scaling strong|weak]

A "step" = one pass of the hot path (one ivs_surface_batch_f64 call) over the rank's whole resident shard.  Inputs live
in HBM before the timed region; the call's scratch is a caller-owned workspace (nothing is allocated inside the loop).
Rank 0 prints ONE JSON line.
  N = 1 : the 1M batch of the workload; the line also carries `other_configs` (configs 4 and 5, a few steps each),
          `cpu_baseline` (C/OpenMP port of the oracle) with `cpu_baseline.others` (B1 reference-shaped, B2 vectorised
          NumPy: SURVEY 8d / BASELINE.md 3) and `pcie` (H2D / D2H of one chunk: the rate a host-buffer caller would see).
  N > 1 : default `--scaling strong` = BASELINE config 3 as written: ONE 1M batch split over the ranks with
          sharding.shard_bounds (config 5: ragged_shard_bounds, equal strike counts); `--scaling weak` gives every rank
          its own `--batch` surfaces.  value = all surfaces of the job / max-over-ranks time.
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
HBM_ACHIEVABLE_GBPS = 6290.0  # the same guide's measured float4-copy rate: the yardstick next to the spec peak

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


def _oracle_path():
    p = os.path.join(ROOT, "oracle")
    if p not in sys.path:
        sys.path.insert(0, p)


def cpu_pool_baselines(method, nK, nT, mK, mT):
    """B1 / B2 of SURVEY 8d (process pools: run BEFORE this process initialises the GPU)."""
    _oracle_path()
    import cpu_baselines as CB
    return {"B2_vectorised_numpy": CB.run_b2(method, nK, nT, mK, mT, budget_s=5.0),
            "B1_reference_shaped": CB.run_b1(method, nK, nT, budget_s=6.0)}


def cpu_baseline(method, nK, nT, mK, mT, budget_s=20.0, sample=None, ragged=False):
    """The oracle ('port' of the reference's NumPy/SciPy arithmetic) timed on this box's host cores on a
    bounded sample of the same workload.  Uses the compiled C oracle with OpenMP when it is built,
    else the vectorised NumPy oracle on one core.  `sample` = (K, T, sigma, Kq, Tq, device results) of a few
    surfaces of the timed batch: they are recomputed here and compared (returned as the second value)."""
    import numpy as np
    _oracle_path()
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
    if ragged:
        n = 4000
        d = synth.numpy_ragged_batch(n, nT, 8, 128, seed=synth.BASE_SEED)
        call = (lambda: runner.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, code, k_off=d["k_off"])) if runner else \
               (lambda: O.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, code, k_off=d["k_off"]))
        what = f"{n} ragged surfaces (8..128 strikes) of the same generator"
    else:
        n = 20000 if runner else 2000
        d = synth.numpy_batch(n, nK, nT, seed=synth.BASE_SEED)
        call = (lambda: runner.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, code)) if runner else \
               (lambda: O.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, code))
        what = f"{n} surfaces of the same generator"
    call()                                                                   # warm
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < budget_s / 2 or reps == 0:
        call(); reps += 1
    dt = time.perf_counter() - t0
    impl = "C oracle (oracle/ivs_oracle_c.c, OpenMP)" if runner else "vectorised NumPy oracle (oracle/ivs_oracle.py)"
    return {"value": n * reps / dt, "unit": "surfaces/s", "cores": runner.threads() if runner else 1, "kind": "port",
            "sample": f"{reps} x {what}, {impl}, method {method}"}, check


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


def build_shard(torch, synth, sharding, workload, B_global, rank, world, scaling, nk_override=0):
    """Inputs of this rank in HBM.  strong: rank r owns the contiguous block shard_bounds(B_global, r, world) of ONE global
    batch (ragged: blocks of equal strike count); weak: every rank owns B_global surfaces.  Each rank generates only its
    own shard (seed + rank); the ragged strike counts of the global batch are drawn identically on every rank."""
    nK, nT, mK, mT, ragged, _ = WORKLOADS[workload]
    if nk_override and not ragged:
        nK = nk_override
    seed = synth.BASE_SEED + rank
    if not ragged:
        lo, hi = (0, B_global) if scaling == "weak" or world == 1 else sharding.shard_bounds(B_global, rank, world)
        d = synth.torch_batch(hi - lo, nK, nT, seed=seed)
        return d, {}, hi - lo, None, (lo, hi), nK
    import numpy as np
    if scaling == "weak" or world == 1:
        counts = synth.ragged_counts(B_global, 8, 128, seed=synth.BASE_SEED + 1000 + (rank if scaling == "weak" else 0))
        lo, hi = 0, B_global
    else:
        counts = synth.ragged_counts(B_global, 8, 128, seed=synth.BASE_SEED + 1000)
        k_off = np.concatenate([[0], np.cumsum(counts)])
        lo, hi = sharding.ragged_shard_bounds(k_off, world)[rank]
    d = synth.torch_ragged_batch(hi - lo, nT, 8, 128, seed=seed, nk=counts[lo:hi])
    kw = dict(k_off=d["k_off"], nK_max=128, n_maturities=nT)
    return d, kw, hi - lo, int(d["k_off"][-1]), (lo, hi), nK


def run_workload(torch, engine, synth, sharding, dist, a, workload, method, B_global, steps, warmup, rank, world, backend,
                 scaling, want_sample=False):
    """Generate the shard, run W + K steps, return the measurements of this rank (and, on rank 0, of the job)."""
    nK0, nT, mK, mT, ragged, desc = WORKLOADS[workload]
    d, kw, B, total_strikes, (lo, hi), nK = build_shard(torch, synth, sharding, workload, B_global, rank, world, scaling,
                                                        a.nk if workload == a.workload else 0)
    if nK != nK0:
        desc = desc.replace("64 strikes", f"{nK} strikes").replace("(64x16)", f"({nK}x16)")
    if a.nan_frac > 0 and workload == a.workload:
        g = torch.Generator(device="cuda"); g.manual_seed(synth.BASE_SEED + 77 + rank)
        d["sigma"][torch.rand(d["sigma"].shape, generator=g, device="cuda") < a.nan_frac] = float("nan")
        desc += f", {a.nan_frac:.0%} of the quotes missing (NaN)"
    Kq_h, Tq_h = synth.query_grids(mK, mT, nT)
    Kq = torch.from_numpy(Kq_h).cuda(); Tq = torch.from_numpy(Tq_h).cuda()
    out = torch.empty((B, mT, mK), dtype=torch.float64, device="cuda")
    status = torch.empty((B,), dtype=torch.int32, device="cuda")
    ws = engine.surface_workspace(B, ragged)          # caller-owned scratch: the timed call allocates nothing

    def step():
        engine.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, method, out=out, status=status,
                             force_generic=a.force_generic, workspace=ws, **kw)

    if dist:
        # the FIRST collective builds the communicator (hundreds of ms with the device idle): pay it here, before the
        # clocks are spun up, so that the barrier that brackets the timed region is a warm one
        dist.barrier()
        torch.cuda.synchronize()
    # DVFS spin-up, before (and in addition to) the W warm-up steps: after any idle gap the first ~25 ms of launches
    # run 10-25 % slower while the clocks ramp (tools/warm_probe.py), which would otherwise leak into the timed steps
    # whenever W is small.  Untimed, fixed 100 ms of device time.
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < a.spin_ms * 1e-3:
        step()
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    kernel = engine.last_kernel()
    launch = "eager"
    eager_step = step
    if a.launch == "graph":
        # the call is allocation- and synchronisation-free (include/ivs.h), so its launches -- maturity tables, the surface
        # kernel, the two filter passes -- replay as ONE hipGraph: the per-launch host cost and the gaps between dependent
        # kernels (~30 us per call, 8 % of a 125 k-surface shard at N = 8) leave the timed region; same work, same kernels
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            step = graph.replay
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            launch = "hipGraph replay of one captured call"
        except Exception as e:                                    # capture unavailable: time the eager call
            launch = f"eager (graph capture failed: {type(e).__name__})"
            step = eager_step
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record()
        step()
        ev[i][1].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kern_ms = [s.elapsed_time(e) for s, e in ev]
    avg_ms = sum(kern_ms) / len(kern_ms)
    per_rank_ms, per_rank_B, ranks_seen, total = [avg_ms], [B], 1, B
    if dist:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t[0])
        mine = torch.tensor([avg_ms, float(B), 1.0], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [float(x[0]) for x in allr]; per_rank_B = [int(x[1]) for x in allr]
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)                                       # how many ranks the collective really reached
        ranks_seen = int(ones[0]); total = sum(per_rank_B)
    assert int(status.max()) == 0

    # Placement (OFF by default since round 3: `value` and `roofline.frac` are measured on the FIRST allocation, what any
    # caller of engine.surface_batch gets).  On this platform the same kernel on the same inputs runs a few per cent faster
    # or slower depending on WHICH allocation the output lives in (tools/placement_probe.py); with
    # --placement-tries N > 1 the line additionally reports, in separate fields, the kernel time on the fastest of N
    # allocations (engine.place_output): never part of `value`.
    placement = {"tries": 1}
    if a.placement_tries > 1 and workload == a.workload and out.numel() * 8 <= (32 << 30):
        first = out
        def run_on(o):
            nonlocal out
            out = o
            eager_step()
        step = eager_step                                          # a captured graph has its output buffer baked in
        best, more_ms = engine.place_output(run_on, tuple(first.shape), tries=a.placement_tries - 1)
        out = best
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < a.spin_ms * 1e-3:
            step()
            torch.cuda.synchronize()
        evp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for s_, e_ in evp:
            s_.record(); step(); e_.record()
        torch.cuda.synchronize()
        placed_ms = sum(s_.elapsed_time(e_) for s_, e_ in evp) / steps
        placement = {"tries": a.placement_tries, "probe_ms_other_allocations": [round(x, 4) for x in more_ms],
                     "first_allocation_kernel_ms_avg": avg_ms, "placed_kernel_ms_avg": placed_ms,
                     "note": "the timed region above ran on the FIRST allocation; placed_* = the fastest of the other allocations, reported separately"}
        out = first
        del best
        torch.cuda.empty_cache()

    sample = None
    if want_sample and not ragged and a.check > 0 and a.nan_frac == 0:
        idx = torch.linspace(0, B - 1, a.check, device="cuda").long()
        sample = (d["K"][idx].cpu().numpy(), d["T"].cpu().numpy(), d["sigma"][idx].cpu().numpy(), Kq_h, Tq_h,
                  out[idx].cpu().numpy())
    bytes_launch = algorithmic_bytes(B, nK, nT, mK, mT, total_strikes)
    achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
    m = {
        "value": total * steps / wall, "ms_per_step": wall / steps * 1e3, "total_surfaces": total, "B_rank": B,
        "desc": desc, "kernel": kernel, "nK": nK, "nT": nT, "mK": mK, "mT": mT, "ragged": ragged, "sample": sample,
        "placement": placement, "launch": launch, "shard": [lo, hi], "per_rank_ms": per_rank_ms, "per_rank_B": per_rank_B, "ranks_seen": ranks_seen,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": load_traffic(workload, method, kernel, B),
                     "kernel": kernel, "kernel_ms_avg": avg_ms, "kernel_ms_min": min(kern_ms),
                     "kernel_ms_median": sorted(kern_ms)[len(kern_ms) // 2], "kernel_ms_steps": [round(x, 4) for x in kern_ms],
                     "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS, "achievable_GBps": HBM_ACHIEVABLE_GBPS,
                     "algorithmic_bytes_per_launch": bytes_launch,
                     "timing": "HIP events around each launch on the launch stream"},
    }
    if "placed_kernel_ms_avg" in placement:
        m["roofline"]["frac_placed"] = bytes_launch / (placement["placed_kernel_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBPS
        m["value_with_placement"] = B / (placement["placed_kernel_ms_avg"] * 1e-3) if world == 1 else None
    gather = None
    if a.gather and dist and workload == a.workload:
        # results normally stay sharded; the one optional collective is an all-gather of the output tiles
        torch.cuda.synchronize(); dist.barrier()
        g0 = time.perf_counter()
        full = sharding.gather_outputs(out)
        torch.cuda.synchronize(); dist.barrier()
        gather = {"ms": (time.perf_counter() - g0) * 1e3, "bytes_per_rank": out.numel() * 8, "ranks": world,
                  "gathered_shape": list(full.shape)}
        del full
    m["gather"] = gather
    m["pcie"] = None
    if want_sample and not ragged:
        # what a caller holding HOST buffers would pay on top: pinned H2D of a chunk of inputs, D2H of its outputs
        n = min(B, 100_000)
        hs = torch.empty((n, nT, nK), dtype=torch.float64).pin_memory(); ho = torch.empty((n, mT, mK), dtype=torch.float64).pin_memory()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        d["sigma"][:n].copy_(hs, non_blocking=True); torch.cuda.synchronize()
        e0.record(); d["sigma"][:n].copy_(hs, non_blocking=True); e1.record(); ho.copy_(out[:n], non_blocking=True); e2.record()
        torch.cuda.synchronize()
        h2d = hs.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9; d2h = ho.numel() * 8 / (e1.elapsed_time(e2) * 1e-3) / 1e9
        per_surface_s = (nT * nK * 8) / (h2d * 1e9) + (mT * mK * 8) / (d2h * 1e9) + avg_ms * 1e-3 / B
        m["pcie"] = {"h2d_GBps": h2d, "d2h_GBps": d2h, "chunk_surfaces": n,
                     "surfaces_per_s_host_buffers_serial": 1.0 / per_surface_s,
                     "note": "pinned host memory, copies serialised with the kernel (no overlap): never part of `value`"}
    del d, out, status, ws
    torch.cuda.empty_cache()
    return m


class stdout_to_stderr:
    """File-descriptor level redirect: stdout must carry exactly one JSON line."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--method", default="cubic", choices=["linear", "cubic", "cubicspline", "slinear", "pchip", "akima", "nearest", "zero", "from_derivatives", "quadratic"])
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=1_000_000,
                    help="surfaces of the job (strong scaling: split over the ranks) or per GPU (weak scaling)")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="N > 1 only.  strong (default) = ONE --batch split over the ranks (BASELINE config 3); weak = --batch per rank")
    ap.add_argument("--nk", type=int, default=0, help="override the strike count of a uniform workload (variable-shape kernel)")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--nan-frac", type=float, default=0.0,
                    help="fraction of quotes set to NaN (= missing): every row then has its own knot set (masked second-pass kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--placement-tries", type=int, default=1,
                    help="N > 1: after the timed region also time the kernel on the fastest of N - 1 further output allocations "
                         "(reported as roofline.frac_placed / value_with_placement, never as `value`)")
    ap.add_argument("--launch", default="eager", choices=["eager", "graph"],
                    help="graph: capture one call into a hipGraph after the warm-up and time its replays (same kernels, no per-launch host cost)")
    ap.add_argument("--spin-ms", type=float, default=100.0, help="untimed launches before the warm-up steps (clock ramp)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the config-4 / config-5 sub-results of the N = 1 line")
    ap.add_argument("--check", type=int, default=256, help="surfaces compared with the oracle after the timed region")
    ap.add_argument("--gather", action="store_true",
                    help="after the timed region, also time the optional all-gather of the output shards (RCCL over xGMI); "
                         "reported separately as 'gather', never part of 'value'")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with python -m torch.distributed.run --nproc-per-node N")
    scaling = a.scaling or "strong"        # one global batch; at N = 1 the two modes are the same workload
    nK, nT, mK, mT, ragged, _ = WORKLOADS[a.workload]
    if a.nk and not ragged:
        nK = a.nk

    # CPU baselines that fan out over processes run FIRST: nothing in this process has touched the GPU yet
    pool_baselines = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not ragged:
        pool_baselines = cpu_pool_baselines(a.method, nK, nT, mK, mT)

    import torch
    from iv_interpolation_amd import engine, sharding, synth

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
        with stdout_to_stderr():       # RCCL prints its version banner on stdout when the communicator is built
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()

    m = run_workload(torch, engine, synth, sharding, dist, a, a.workload, a.method, a.batch, a.steps, a.warmup, rank, world,
                     backend, scaling, want_sample=(rank == 0 and world == 1 and not a.no_cpu_baseline))

    others = {}
    if world == 1 and not a.no_other_configs and a.workload == "cfg3" and not a.nk and not a.force_generic and a.nan_frac == 0:
        # the driver runs the default line only: carry the other two BASELINE configs along (a few steps each)
        for wl, st in (("cfg4", 5), ("cfg5", 5)):
            o = run_workload(torch, engine, synth, sharding, None, a, wl, a.method, a.batch, st, 2, 0, 1, backend, "weak")
            others[wl] = {"workload": f"{wl}: {o['desc']}", "value": o["value"], "unit": "surfaces/s", "steps": st, "warmup": 2,
                          "ms_per_step": o["ms_per_step"], "kernel": o["kernel"],
                          "roofline": {k: o["roofline"][k] for k in ("achieved", "peak", "frac", "kernel_ms_avg", "algorithmic_bytes_per_launch")}}

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
        roof = m["roofline"]
        roof["device_copy_GBps"] = copy_gbps
        roof["frac_of_device_copy"] = (roof["achieved"] / copy_gbps) if copy_gbps else None
        if world > 1:
            roof["per_rank_kernel_ms_avg"] = m["per_rank_ms"]
        shard_note = (f"{world} contiguous shards of one {m['total_surfaces']}-surface batch" if scaling == "strong" and world > 1
                      else f"{world} x {m['B_rank']} surfaces")
        res = {
            "metric": "IV surfaces/sec (1M-batch, 64x16 grid)", "value": m["value"], "unit": "surfaces/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": m["ms_per_step"],
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {m['desc']}", "method": a.method, "total_surfaces": m["total_surfaces"],
                       "surfaces_per_gpu": m["per_rank_B"] if world > 1 else m["B_rank"],
                       "quote_grid": [m["nK"], m["nT"]], "output_grid": [m["mK"], m["mT"]], "kernel": m["kernel"],
                       "sharding": shard_note + (", balanced by strike count" if m["ragged"] and scaling == "strong" and world > 1 else "")
                                   + ", no data-path collective",
                       "ranks_seen_by_all_reduce": m["ranks_seen"], "seed": synth.BASE_SEED, "placement": m["placement"],
                       "launch": m["launch"]},
            "roofline": roof,
            "parity_check": {},
        }
        if m.get("value_with_placement"):
            res["value_with_placement"] = m["value_with_placement"]
        if m["gather"]:
            res["gather"] = m["gather"]
        if others:
            res["other_configs"] = others
        if m["pcie"]:
            res["pcie"] = m["pcie"]
        if not a.no_cpu_baseline:
            res["cpu_baseline"], res["parity_check"] = cpu_baseline(a.method, m["nK"], m["nT"], m["mK"], m["mT"],
                                                                   budget_s=20.0 if world == 1 else 8.0, sample=m["sample"],
                                                                   ragged=m["ragged"])
            if pool_baselines:
                res["cpu_baseline"]["others"] = pool_baselines
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
