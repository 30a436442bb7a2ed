#!/usr/bin/env python3
"""Benchmark of the WaveEnv integrator hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): TwoDim(15f0, 700), triple-ring design
space, RandomPosGaussianSource, dt = 1e-5, 100 integration steps per environment action, RandomDesignPolicy.
One bench "step" = one env(action) = 100 RK4 integration steps of the 700^2 grid = 49.0 M cell-updates, with the
wave-speed field assembly, source injection, energy reductions and frame capture that belong to it.  All inputs are
resident in HBM when the timed region starts (the only host->device traffic per action is the ~100 KB coefficient
table; the only device->host traffic is the 101x3 energy trace).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), each rank runs its own environment (weak scaling:
independent episodes); the design-space block is broadcast from rank 0 before the timed region and the
energy traces are all-gathered after the last step inside it.

Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

B_ALG = 104.0          # algorithmic bytes per cell-update: 48 r + 48 w state, 4 c, 4 source shape
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
N_GRID = 700
STEPS_PER_ACTION = 100


def baseline_metric() -> str:
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mcell-updates/s (700^2 grid, 100 integ steps) + achieved HBM GB/s vs peak"


def cpu_baseline(n_steps: int):
    """The C oracle ("port" of the reference algorithm, 1 core like the reference's single-threaded Julia CPU path) timed
    on this host on a bounded sample of the same workload."""
    import c_oracle as co
    import waves_oracle as wo
    f32 = np.float32
    dim = wo.TwoDim.from_size(15.0, N_GRID)
    rng = np.random.default_rng(0)
    ds = wo.build_triple_ring_design_space()
    a = wo.rand_design(ds, rng)
    b = ds(a, wo.rand_design(wo.build_action_space(a, 0.25), rng))
    flat = lambda d: np.concatenate([wo.stacked_cylinders(d).pos, wo.stacked_cylinders(d).r[:, None],
                                     wo.stacked_cylinders(d).c[:, None]], 1).astype(f32)
    G = wo.build_normal(wo.build_grid(dim), np.array([[-10.0, 3.0]]), np.array([0.3]), np.array([1.0]))
    sx = wo.build_pml_profile(dim.x, 2.0, 20000.0)
    ts = wo.build_tspan(0.0, 1e-5, n_steps)
    st = np.zeros((12, N_GRID, N_GRID), f32)
    t0 = time.perf_counter()
    co.integrate(dim.x, dim.y, sx, sx, wo.WATER, 1e-5, st, ts, G=wo.to_abi(G), freq=1000.0, d0=flat(a), d1=flat(b),
                 ti=0.0, tf=ts[-1], nthreads=1)
    dt = time.perf_counter() - t0
    out = {"value": round(N_GRID * N_GRID * n_steps / dt / 1e6, 3), "unit": "Mcell-updates/s", "cores": 1,
           "kind": "port",
           "sample": f"{n_steps} integration steps of the same 700^2 triple-ring workload ({dt:.1f} s), "
                     "oracle/waves_oracle.c single thread"}
    # for orientation only: the same restatement with its OpenMP loops on all the cores this process may use (the
    # reference's own CPU path is single-threaded: Julia sparse `*` and broadcast, no threads enabled in its scripts)
    try:
        nth = len(os.sched_getaffinity(0))
    except Exception:
        nth = os.cpu_count() or 1
    nth = min(nth, 16)  # the CPU share of a one-GPU box (its host may show hundreds of cores it will not give us)
    if nth > 1:
        st = np.zeros((12, N_GRID, N_GRID), f32)
        ts2 = wo.build_tspan(0.0, 1e-5, 2 * n_steps)
        t0 = time.perf_counter()
        co.integrate(dim.x, dim.y, sx, sx, wo.WATER, 1e-5, st, ts2, G=wo.to_abi(G), freq=1000.0, d0=flat(a), d1=flat(b),
                     ti=0.0, tf=ts2[-1], nthreads=nth)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": round(N_GRID * N_GRID * 2 * n_steps / dt2 / 1e6, 3), "cores": nth,
                            "sample": f"{2 * n_steps} steps, OpenMP ({dt2:.1f} s)"}
    return out


def batched_envs(w, dim, ds, dev, impl, n_envs, pml_width, actions=6):
    """BASELINE config 3 shape on one GPU: n_envs independent 700^2 environments, each on its own HIP stream, stepped
    concurrently (step_begin on all, then step_end on all).  Reported next to the headline, never as `value`."""
    import torch
    envs, pols = [], []
    for e in range(n_envs):
        src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                        rng=np.random.default_rng(500 + e))
        env = w.WaveEnv(dim, design_space=ds, source=src, integration_steps=STEPS_PER_ACTION, actions=actions + 4,
                        device=dev, impl=impl, rng=np.random.default_rng(600 + e), return_fields=False,
                        pml_width=pml_width)
        env.reset()
        envs.append(env)
        pols.append(w.RandomDesignPolicy(env.action_space(), np.random.default_rng(700 + e)))

    def sweep():  # at most 4 actions in flight at a time (w.step_all): the aggregate peaks there
        w.step_all(envs, [pol(env) for env, pol in zip(envs, pols)])

    sweep()
    sweep()
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(actions):
        sweep()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = len(dim.x)
    val = n_envs * actions * STEPS_PER_ACTION * n * n / dt / 1e6
    for env in envs:
        env.ctx.close()
    return {"envs_per_gpu": n_envs, "value": round(val, 2), "unit": "Mcell-updates/s",
            "whole_job_frac": round(B_ALG * val * 1e6 / (HBM_PEAK_GBS * 1e9), 4),
            "note": "independent envs overlapped on separate HIP streams, 4 in flight at a time (BASELINE config 3 shape); "
                    "not the headline"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed env actions (100 integration steps each)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--impl", default="auto", choices=["auto", "staged", "fused"])
    ap.add_argument("--cpu-steps", type=int, default=40, help="integration steps of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--grid", type=int, default=N_GRID, help="grid points per axis (700 = the metric's configuration)")
    ap.add_argument("--pml-width", type=float, default=2.0)
    ap.add_argument("--batch-envs", type=int, default=8,
                    help="extra (untimed-for-`value`) measurement: this many independent envs on the GPU, BASELINE config "
                         "3's 8-per-GPU shape, stepped four at a time on their HIP streams (0 = skip)")
    args = ap.parse_args()

    import torch  # first: the HIP runtime both torch and libwaves_amd use is then torch's
    import waves_jl_amd as w
    from waves_jl_amd import dist as wd

    rank, local_rank, world = wd.init()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    ngrid = args.grid
    dev = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(dev)

    # --- environment: rank 0 owns the design-space block, everyone else receives it over RCCL
    ds = w.build_triple_ring_design_space() if rank == 0 else None
    ds = wd.broadcast_design_space(ds, src=0)
    dim = w.TwoDim(15.0, ngrid)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(2 + 1000 * rank))
    total_actions = args.warmup + args.steps
    env = w.WaveEnv(dim, design_space=ds, source=src, integration_steps=STEPS_PER_ACTION, actions=total_actions + 8,
                    device=dev, impl=args.impl, rng=np.random.default_rng(1000 * rank), return_fields=False,
                    pml_width=args.pml_width)
    policy = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(1 + 1000 * rank))
    env.reset()

    for _ in range(args.warmup):
        env(policy(env))
    if world > 1:
        wd.gather_signals(env.signal)  # warm the communicator up outside the timed region
    # A full collection of the interpreter's cyclic GC takes 30-50 ms once torch is imported (millions of objects) and
    # would land in the middle of a 1.3 ms action every few hundred allocations: park everything allocated so far in the
    # permanent generation, as long-running Python services do.
    gc.collect()
    gc.freeze()

    wd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sigs = []
    dev_ms = 0.0
    kern_ms, kern_launches = 0.0, 0
    for _ in range(args.steps):
        env(policy(env))
        sigs.append(env.signal)
        tim_ = env.ctx.timing()
        dev_ms += tim_["total_ms"]
        kern_ms += tim_["step_kernel_ms"]
        kern_launches += tim_["step_kernel_launches"]
    all_sig = wd.gather_signals(np.stack(sigs))
    torch.cuda.synchronize()
    wd.barrier()
    elapsed = wd.max_over_ranks(time.perf_counter() - t0)

    cells = ngrid * ngrid
    cell_updates = world * cells * STEPS_PER_ACTION * args.steps
    value = cell_updates / elapsed / 1e6

    # --- roofline of the dominant kernel.  Its average launch duration is measured live with the HIP events the library
    # records on the ctx's stream around every wv_integrate call of the TIMED region above (first enqueue -> last kernel),
    # divided by the number of integrator launches in it:
    #   * resident path (the default whenever all tiles fit the device at once, e.g. 700^2): ONE launch of
    #     k_steps_resident per action does all 100 steps, so a launch processes 100 x cells cell-updates; the events sit
    #     directly around that launch;
    #   * single-step path (larger grids): 100 back-to-back k_step_fused launches per action between one pair of events,
    #     gaps included; agrees with the rocprofv3 kernel-trace average to ~2 %.
    # The event pair(s) placed directly around the integrator launch(es) (profiling mode) are reported for reference.
    out = None
    if rank == 0:
        tim = env.ctx.timing()
        impl = tim["impl"]
        resident = bool(tim.get("resident"))
        launches_per_action = 1 if resident else STEPS_PER_ACTION * (4 if impl == "staged" else 1)
        if kern_launches > 0 and kern_ms > 0.0:   # fused path: events directly around the integrator launch(es)
            avg_ms = kern_ms / kern_launches
        else:                                      # staged path: whole call / launches
            avg_ms = dev_ms / (args.steps * launches_per_action)
        env.ctx.set_profiling(True)
        kms, launches = 0.0, 0
        for _ in range(2):
            env(policy(env))
            t = env.ctx.timing()
            kms += t["step_kernel_ms"]
            launches += t["step_kernel_launches"]
        env.ctx.set_profiling(False)
        bracketed_us = kms / launches * 1e3
        if resident:
            units_per_launch = cells * STEPS_PER_ACTION
        else:
            units_per_launch = cells if impl == "fused" else cells / 4.0   # staged: one RK stage = 1/4 cell-update per cell
        alg_bytes = B_ALG * units_per_launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and (ngrid == N_GRID or (ngrid == 2048 and args.pml_width == 2.0)):
            try:
                key = "resident" if resident else impl
                traffic = json.load(open(tpath)).get(key if ngrid == N_GRID else f"{key}_{ngrid}")
            except Exception:
                traffic = None
        kname = "k_steps_resident" if resident else ("k_step_fused" if impl == "fused" else "k_stage")
        out = {
            "metric": baseline_metric(),
            "value": round(value, 2),
            "unit": "Mcell-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"TwoDim(15.0f0, {ngrid}) + triple-ring design_space, RandomPosGaussianSource, "
                                   f"{STEPS_PER_ACTION} integration steps per env action, RandomDesignPolicy",
                       "impl": impl, "envs_per_gpu": 1, "pml_width": args.pml_width,
                       "device_ms_per_step": round(dev_ms / args.steps, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": kname,
                         "avg_kernel_us": round(avg_ms * 1e3, 3), "event_bracketed_kernel_us": round(bracketed_us, 3),
                         "algorithmic_bytes_per_launch": alg_bytes, "steps_per_launch": STEPS_PER_ACTION if resident else 1,
                         "whole_job_frac": round(B_ALG * value * 1e6 / world / (HBM_PEAK_GBS * 1e9), 4)},
            "signal_checksum": float(np.sum(all_sig[0][-1])),
        }
        if world == 1 and args.batch_envs > 1:
            out["batched"] = batched_envs(w, dim, ds, dev, args.impl, args.batch_envs, args.pml_width)
        if world == 1 and args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps)
    wd.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    wd.finalize()


if __name__ == "__main__":
    main()
