#!/usr/bin/env python3
"""Benchmark of the WaveEnv integrator hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): TwoDim(15f0, 700), triple-ring design
space, RandomPosGaussianSource, dt = 1e-5, 100 integration steps per environment action, RandomDesignPolicy.
One bench "step" = one env(action) = 100 RK4 integration steps of the 700^2 grid = 49.0 M cell-updates, with the
wave-speed field assembly, source injection, energy reductions and frame capture that belong to it.  All inputs are
resident in HBM when the timed region starts (the only host->device traffic per action is the ~100 KB coefficient
table; the only device->host traffic is the 101x3 energy trace).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), each rank runs its own environment(s) (weak scaling:
independent episodes); the design-space block is broadcast from rank 0 before the timed region and the
energy traces are all-gathered after the last step inside it.  Launched either by torch.distributed.run (RANK /
LOCAL_RANK / WORLD_SIZE in the environment) or plainly as `python bench.py --gpus N`, which starts the N ranks itself.
`--envs-per-gpu 8` is BASELINE config 3's shape (64 episodes on 8 GPUs); `--steps 20` config 5's rollout length.

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
    nvis = nth
    nth = min(nth, 16)  # the CPU share of a one-GPU box (its host may show hundreds of cores it will not give us)
    # BASELINE config 1 (the CPU-path configuration: TwoDim(15, 256), single Gaussian source at (-10, 0), no design, 100
    # integration steps), same port, one core
    dim1 = wo.TwoDim.from_size(15.0, 256)
    G1 = wo.build_normal(wo.build_grid(dim1), np.array([[-10.0, 0.0]]), np.array([0.3]), np.array([1.0]))
    sx1 = wo.build_pml_profile(dim1.x, 2.0, 20000.0)
    ts1 = wo.build_tspan(0.0, 1e-5, 100)
    t0 = time.perf_counter()
    co.integrate(dim1.x, dim1.y, sx1, sx1, wo.WATER, 1e-5, np.zeros((12, 256, 256), f32), ts1, G=wo.to_abi(G1), freq=1000.0, nthreads=1)
    dt1 = time.perf_counter() - t0
    out["config1_256"] = {"value": round(256 * 256 * 100 / dt1 / 1e6, 3), "unit": "Mcell-updates/s", "cores": 1, "kind": "port",
                          "sample": f"all 100 steps of BASELINE config 1 ({dt1:.2f} s), oracle/waves_oracle.c single thread"}
    # R1 of SURVEY 8d: the reference-STRUCTURE restatement (numpy + scipy.sparse gradient matrices, the same temporaries,
    # concats and four wave-speed assemblies per step as the Julia source), one core, a bounded sample of both configurations
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import r1_baseline
        r1 = r1_baseline.bounded_sample(40, 6)
        out["r1_reference_structure"] = {"unit": "Mcell-updates/s", "cores": 1, "kind": "port (numpy + scipy.sparse, tools/r1_baseline.py)",
                                         **r1}
    except Exception as e:  # (scipy missing on some box: say so instead of failing the bench line)
        out["r1_reference_structure"] = {"error": repr(e)}
    try:
        model = next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    except Exception:
        model = "?"
    out["host"] = {"cpu": model, "cores_visible": nvis}
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


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n: int, argv=None, poll_s: float = 0.05) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this parent never touches a
    GPU, so nothing that initialised HIP is ever replaced), relay rank 0's JSON line, exit with the worst return code.
    ALL children are watched: when one exits non-zero the others -- which would sit in a collective until somebody's time
    limit -- are terminated (then killed) within seconds and the parent returns non-zero."""
    import subprocess
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(free_port())
    argv = [os.path.abspath(__file__)] + sys.argv[1:] if argv is None else list(argv)
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")  # (a file, not a pipe: nobody has to drain it while we poll)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc = max(rc, abs(code))
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for o in sorted(live):
                    procs[o].terminate()           # exactly the PIDs started above
                t_end = time.monotonic() + 5.0
                for o in sorted(live):
                    try:
                        procs[o].wait(timeout=max(0.1, t_end - time.monotonic()))
                    except subprocess.TimeoutExpired:
                        procs[o].kill()
                        procs[o].wait()
                live.clear()
                break
        if live:
            time.sleep(poll_s)
    out0.seek(0)
    text = out0.read()
    out0.close()
    if text:
        sys.stdout.write(text)
        sys.stdout.flush()
    return rc


def rollout_config5(w, dim, ds, dev, impl, pml_width, episodes=3, actions=20):
    """BASELINE config 5 on one GPU: rollouts of `actions` x 100 steps with RandomDesignPolicy, every rollout ONE device call
    (WaveEnv.steps_begin / wv_set_design_sequence: the policy does not read the wave state between actions), two calls in
    flight.  Reported next to the headline, never as `value`."""
    import torch
    env, policy = make_env(w, dim, ds, dev, impl, pml_width, actions * (episodes + 2), 4242)
    w.rollout_batched(env, policy, actions)   # warm-up: allocations, code, the step table of this shape
    gc.collect()
    gc.freeze()
    torch.cuda.synchronize()
    kern_ms, launches = 0.0, 0
    t0 = time.perf_counter()
    env.steps_begin([policy(env) for _ in range(actions)])
    for _ in range(episodes - 1):
        env.steps_begin([policy(env) for _ in range(actions)])
        env.steps_end()
        kern_ms += env.ctx.timing()["step_kernel_ms"]
        launches += 1
    env.steps_end()
    kern_ms += env.ctx.timing()["step_kernel_ms"]
    launches += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tim = env.ctx.timing()
    n = len(dim.x)
    steps = actions * STEPS_PER_ACTION
    avg_ms = kern_ms / launches
    resident = bool(tim["resident"])
    out = {"workload": f"{episodes} rollouts of {actions} actions x {STEPS_PER_ACTION} steps, RandomDesignPolicy, one device call per "
                       "rollout, two calls in flight", "kernel": "k_steps_resident" if resident else "k_step_fused",
           "value": round(episodes * steps * n * n / dt / 1e6, 2), "unit": "Mcell-updates/s",
           "ms_per_action": round(dt / (episodes * actions) * 1e3, 4),
           "whole_job_frac": round(B_ALG * episodes * steps * n * n / dt / (HBM_PEAK_GBS * 1e9), 4),
           "signal_checksum": float(np.sum(env.signal))}
    if resident:
        out.update({"avg_kernel_us": round(avg_ms * 1e3, 3), "steps_per_launch": steps,
                    "frac": round(B_ALG * n * n * steps / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    env.ctx.close()
    return out


def make_env(w, dim, ds, dev, impl, pml_width, actions, seed, **kw):
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(2 + seed))
    env = w.WaveEnv(dim, design_space=ds, source=src, integration_steps=STEPS_PER_ACTION, actions=actions, device=dev,
                    impl=impl, rng=np.random.default_rng(seed), return_fields=False, pml_width=pml_width, **kw)
    policy = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(1 + seed))
    env.reset()
    return env, policy


class StubCtx:
    """Stand-in for a device context (--stub-env): lets the rank logic of this script -- rendezvous, broadcast of the
    design space, sharding, timed loop, gather, max over ranks, the JSON line -- run on a machine without a GPU."""

    def timing(self):
        return {"total_ms": 1.0, "step_kernel_ms": 0.9, "step_kernel_launches": 1, "steps": STEPS_PER_ACTION, "impl": "fused",
                "resident": True}

    def set_profiling(self, on):
        pass

    def synchronize(self):
        pass

    def close(self):
        pass


class StubEnv:
    def __init__(self, seed):
        self.ctx, self.rng, self.signal, self.n = StubCtx(), np.random.default_rng(seed), None, 0

    def step_begin(self, action):
        self.n += 1

    def step_end(self):
        self.signal = np.full((STEPS_PER_ACTION + 1, 3), float(self.n), np.float32)

    def __call__(self, action):
        self.step_begin(action)
        self.step_end()


def sync_device():
    import torch
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_rollout(env, policy, n_actions, in_flight=2):
    """n_actions x env(policy(env)) with two actions in flight (w.rollout_pipelined spelled out to collect the timings);
    in_flight = 1 is the plain `env(action)` loop."""
    sigs, kern_ms, launches, dev_ms = [], 0.0, 0, 0.0
    job_us = timed_rollout.job_us = []
    fast = hasattr(env.ctx, "call_times_ms")   # resident calls: their durations are fetched once, behind the loop
    if fast:
        env.ctx.call_times_ms()                # (forget the warm-up's)

    def end():
        nonlocal kern_ms, launches, dev_ms
        env.step_end()
        sigs.append(env.signal)
        if fast:
            return
        t = env.ctx.timing()
        kern_ms += t["step_kernel_ms"]
        launches += t["step_kernel_launches"]
        dev_ms += t["total_ms"]
        job_us.append(t["step_kernel_ms"] * 1e3 / max(t["step_kernel_launches"], 1))

    # --with-state: state(env) in front of every action (src/data.jl:23) -- of the loops that run one action at a time
    with_state = getattr(timed_rollout, "with_state", False) and in_flight < 2
    prof = os.environ.get("WAVES_AMD_PYPROF")   # diagnostic: where the host's time per action goes (policy / begin / end)
    tp = [0.0, 0.0, 0.0]
    for k in range(n_actions):
        if prof:
            a = time.perf_counter()
            act = policy(env)
            b = time.perf_counter()
            env.step_begin(act)
            c_ = time.perf_counter()
            tp[0] += b - a
            tp[1] += c_ - b
            if k == 0:
                print(f"[bench pyprof] first action (us): policy {(b - a) * 1e6:.1f} | step_begin (starts the launch) {(c_ - b) * 1e6:.1f}", file=sys.stderr)
        else:
            if with_state:
                env.state()
            env.step_begin(policy(env))
        if k > 0 or in_flight < 2:
            a = time.perf_counter()
            end()
            tp[2] += time.perf_counter() - a
    if in_flight >= 2 and n_actions > 0:
        end()
    if fast:
        t = env.ctx.timing()
        times = env.ctx.call_times_ms()
        if t["resident"] and len(times) == n_actions:   # every action was a resident call: the kernel's own clock around each
            job_us.extend(v * 1e3 for v in times)
            kern_ms, launches, dev_ms = sum(times), n_actions, sum(times)
        else:                                            # single-step path (or a mixture): the last call's event times stand for all
            kern_ms = t["step_kernel_ms"] * n_actions
            launches = t["step_kernel_launches"] * n_actions
            dev_ms = t["total_ms"] * n_actions
    if prof and n_actions:
        print(f"[bench pyprof] per action (us): policy {tp[0] / n_actions * 1e6:.1f} | step_begin {tp[1] / n_actions * 1e6:.1f} | "
              f"step_end (incl. waiting) {tp[2] / n_actions * 1e6:.1f}", file=sys.stderr)
    return sigs, kern_ms, launches, dev_ms


def side_config(w, ds, dev, impl, grid, pml_width, actions, traffic_tab):
    """One more BASELINE configuration size measured in the same process after the headline (never `value`)."""
    import torch
    dim = w.TwoDim(15.0, grid)
    env, policy = make_env(w, dim, ds, dev, impl, pml_width, actions + 6, 9000 + grid)
    for _ in range(2):
        env(policy(env))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, kern_ms, launches, _ = timed_rollout(env, policy, actions)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tim = env.ctx.timing()
    resident = bool(tim["resident"])
    cells = grid * grid
    units = cells * (STEPS_PER_ACTION if resident else 1)
    avg_ms = kern_ms / max(launches, 1)
    key = ("resident" if resident else tim["impl"]) + ("" if grid == N_GRID else f"_{grid}")
    env.ctx.close()
    return {"workload": f"TwoDim(15.0f0, {grid}) + triple-ring design_space, pml_width {pml_width}, {actions} actions x "
                        f"{STEPS_PER_ACTION} steps", "kernel": "k_steps_resident" if resident else "k_step_fused",
            "value": round(actions * STEPS_PER_ACTION * cells / dt / 1e6, 2), "unit": "Mcell-updates/s",
            "avg_kernel_us": round(avg_ms * 1e3, 3), "steps_per_launch": STEPS_PER_ACTION if resident else 1,
            "frac": round(B_ALG * units / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "whole_job_frac": round(B_ALG * actions * STEPS_PER_ACTION * cells / dt / (HBM_PEAK_GBS * 1e9), 4),
            "traffic": traffic_tab.get(key) if (grid == N_GRID or pml_width == 2.0) else None}


def sweep_config(w, ds, dev, impl, grid, pml_width, nsteps, traffic_tab):
    """BASELINE config 4 as stated: the 2048^2 grid, one PML width of the sweep, ONE call of 500 integration steps (never `value`)."""
    import torch
    dim = w.TwoDim(15.0, grid)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(77))
    env = w.WaveEnv(dim, design_space=ds, source=src, integration_steps=nsteps, actions=4, device=dev, impl=impl,
                    rng=np.random.default_rng(78), return_fields=False, pml_width=pml_width)
    policy = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(79))
    env.reset()
    env(policy(env))        # warm-up: allocations, the step graph of this shape
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env(policy(env))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tim = env.ctx.timing()
    resident = bool(tim["resident"])
    cells = grid * grid
    launches = max(tim["step_kernel_launches"], 1)
    avg_ms = tim["step_kernel_ms"] / launches
    units = cells * (nsteps if resident else 1)
    sig = env.signal
    env.ctx.close()
    return {"workload": f"TwoDim(15.0f0, {grid}) + triple-ring design_space, pml_width {pml_width}, ONE call of {nsteps} integration steps",
            "kernel": "k_steps_resident" if resident else "k_step_fused",
            "value": round(nsteps * cells / dt / 1e6, 2), "unit": "Mcell-updates/s", "ms_per_call": round(dt * 1e3, 3),
            "avg_kernel_us": round(avg_ms * 1e3, 3), "steps_per_launch": nsteps if resident else 1,
            "frac": round(B_ALG * units / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "whole_job_frac": round(B_ALG * nsteps * cells / dt / (HBM_PEAK_GBS * 1e9), 4),
            "energy_end": [float(v) for v in sig[-1]],
            "traffic": traffic_tab.get(f"fused_{grid}") if pml_width == 2.0 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed env actions (100 integration steps each)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--impl", default="auto", choices=["auto", "staged", "fused"])
    ap.add_argument("--cpu-steps", type=int, default=160, help="integration steps of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--grid", type=int, default=N_GRID, help="grid points per axis (700 = the metric's configuration)")
    ap.add_argument("--pml-width", type=float, default=2.0)
    ap.add_argument("--envs-per-gpu", type=int, default=1,
                    help="independent environments per rank (BASELINE config 3: 8 per GPU, stepped four at a time on their "
                         "HIP streams); 1 = the headline: one environment, two of its actions in flight")
    ap.add_argument("--batch-envs", type=int, default=8,
                    help="extra (untimed-for-`value`) measurement at N=1: this many independent envs on the GPU (0 = skip)")
    ap.add_argument("--in-flight", type=int, default=2, choices=[1, 2],
                    help="actions of the one env in flight (2 = pipelined, the default; 1 = plain env(action) loop)")
    ap.add_argument("--with-state", action="store_true",
                    help="with --in-flight 1: read state(env) (the 128x128x4 observation, resized on the device) in front of every "
                         "action, as generate_episode! and the MPC script of the reference do")
    ap.add_argument("--side-configs", type=int, default=1, help="N=1: also time 2048^2 and 256^2 after the headline (0 = skip)")
    ap.add_argument("--stub-env", action="store_true", help=argparse.SUPPRESS)  # tests: rank logic without a GPU
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch  # first: the HIP runtime both torch and libwaves_amd use is then torch's
    import waves_jl_amd as w
    from waves_jl_amd import dist as wd

    # (a rehearsal with several ranks on ONE GPU cannot use RCCL -- it refuses two ranks on a device -- and takes gloo)
    shared = (os.environ.get("WAVES_AMD_ALLOW_SHARED_GPU") and
              0 < torch.cuda.device_count() < int(os.environ.get("WORLD_SIZE", "1")))
    if shared:  # no rank has the device to itself: the resident kernel (all its tiles co-resident) is nobody's to use
        os.environ["WAVES_AMD_FUSED_RESIDENT"] = "0"
    rank, local_rank, world = wd.init("gloo" if shared else None)
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    ngrid = args.grid
    ndev = torch.cuda.device_count()
    if ndev > 0 and local_rank >= ndev:
        # several ranks on one GPU cannot all hold the device for the cooperative resident kernel: rehearsals only
        if not os.environ.get("WAVES_AMD_ALLOW_SHARED_GPU"):
            print(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible); set WAVES_AMD_ALLOW_SHARED_GPU=1 to "
                  "rehearse with ranks sharing a device (single-step kernels)", file=sys.stderr)
            sys.exit(2)
        os.environ["WAVES_AMD_FUSED_RESIDENT"] = "0"
    dev = local_rank % max(ndev, 1)
    if ndev > 0:
        torch.cuda.set_device(dev)

    # --- environments: rank 0 owns the design-space block, everyone else receives it over RCCL
    ds = w.build_triple_ring_design_space() if rank == 0 else None
    ds = wd.broadcast_design_space(ds, src=0)
    dim = w.TwoDim(15.0, ngrid)
    E = max(1, args.envs_per_gpu)
    total_actions = args.warmup + 11 * args.steps   # (+ the ten times longer window behind the timed one: roofline "sustained")
    envs, pols = [], []
    for e in range(E):
        if args.stub_env:
            env, pol = StubEnv(1000 * rank + 17 * e), (lambda en: None)
        else:
            env, pol = make_env(w, dim, ds, dev, args.impl, args.pml_width, total_actions + 8, 1000 * rank + 17 * e)
        envs.append(env)
        pols.append(pol)
    env, policy = envs[0], pols[0]

    def sweep():
        return w.step_all(envs, [pol(en) for en, pol in zip(envs, pols)])

    assert len(ds.low.config) == 18 and len(ds.low.core) == 1  # every rank holds rank 0's triple-ring block

    if world > 1:
        # warm the communicator up outside the timed region, with a buffer of the very shape the timed gather will carry
        shape = (args.steps, STEPS_PER_ACTION + 1, 3) if E == 1 else (args.steps, E, STEPS_PER_ACTION + 1, 3)
        wd.gather_signals(np.zeros(shape, np.float32))
    # A full collection of the interpreter's cyclic GC takes 30-50 ms once torch is imported (millions of objects) and
    # would land in the middle of a 1 ms action every few hundred allocations: park everything allocated so far in the
    # permanent generation, as long-running Python services do.  In FRONT of the warm-up: the collection leaves the host's
    # caches cold and the device idle for tens of milliseconds, which is exactly what the warm-up steps are there to undo
    # (behind them it made the first timed action 0.3 ms slower on the host, and the first ten ran at the device's idle clocks).
    gc.collect()
    gc.freeze()
    timed_rollout.with_state = bool(E == 1 and args.with_state and args.in_flight == 1 and not args.stub_env)
    for _ in range(args.warmup):
        if E == 1:
            if timed_rollout.with_state:
                env.state()   # (the first one allocates the observation buffers)
            env(policy(env))
        else:
            sweep()
    for en in envs:   # the warm-up's resident launch leaves here: the timed region starts (and pays for) its own launch
        en.ctx.synchronize()
    wd.barrier()
    sync_device()
    t0 = time.perf_counter()
    dev_ms = 0.0
    kern_ms, kern_launches = 0.0, 0
    if E == 1:
        sigs, kern_ms, kern_launches, dev_ms = timed_rollout(env, policy, args.steps, args.in_flight)
    else:
        sigs = []
        for _ in range(args.steps):
            sweep()
            sigs.append(np.stack([en.signal for en in envs]))
            for en in envs:
                t_ = en.ctx.timing()
                dev_ms += t_["total_ms"]
                kern_ms += t_["step_kernel_ms"]
                kern_launches += t_["step_kernel_launches"]
    t_loop = time.perf_counter()
    for en in envs:          # (the library's own sync point: a resident launch that is waiting for further actions leaves now,
        en.ctx.synchronize()  # instead of being waited out by the collective's kernels and the device-wide synchronisation below)
    t_sync0 = time.perf_counter()
    # (one rank: nothing travels, and nothing is copied for the sake of it)
    all_sig = wd.gather_signals(np.stack(sigs)) if world > 1 else [sigs]
    t_gather = time.perf_counter()
    t_sync = t_gather
    sync_device()
    wd.barrier()
    if os.environ.get("WAVES_AMD_PYPROF") and rank == 0:
        print(f"[bench pyprof] timed region (ms): loop {1e3 * (t_loop - t0):.3f} | ctx.synchronize {1e3 * (t_sync0 - t_loop):.3f} | gather "
              f"{1e3 * (t_gather - t_sync0):.3f} | device sync + barrier {1e3 * (time.perf_counter() - t_sync):.3f}", file=sys.stderr)
    assert len(all_sig) == world
    elapsed = wd.max_over_ranks(time.perf_counter() - t0)

    cells = ngrid * ngrid
    cell_updates = world * E * cells * STEPS_PER_ACTION * args.steps
    value = cell_updates / elapsed / 1e6

    # --- roofline of the dominant kernel.  Its average launch duration is measured live with the HIP events the library
    # records on the ctx's stream directly around the integrator launch(es) of every wv_integrate call of the TIMED region
    # above, divided by the number of integrator launches in it:
    #   * resident path (the default whenever all tiles fit the device at once, e.g. 700^2): ONE launch of
    #     k_steps_resident per action does all 100 steps, so a launch processes 100 x cells cell-updates;
    #   * single-step path (larger grids, several envs per GPU): 100 back-to-back k_step_fused launches per action between
    #     one pair of events, gaps included; agrees with the rocprofv3 kernel-trace average to ~2 %.
    # The event pair(s) placed directly around the integrator launch(es) (profiling mode) are reported for reference.
    out = None
    if rank == 0:
        tim = env.ctx.timing()
        impl = tim["impl"]
        resident = bool(tim.get("resident"))
        launches_per_action = 1 if resident else STEPS_PER_ACTION * (4 if impl == "staged" else 1)
        if kern_launches > 0 and kern_ms > 0.0 and impl == "fused":   # events directly around the integrator launch(es)
            avg_ms = kern_ms / kern_launches
        else:                                      # staged path: whole call / launches
            avg_ms = dev_ms / (args.steps * E * launches_per_action)
        # The resident kernel serves the whole timed region as ONE launch (a job per action; ctx.synchronize() above made it
        # leave): its duration by the HIP events the launch carries, divided by the actions it served, is the average launch
        # time per action -- idle time between actions, if the host was ever late, included; rocprofv3 sees the same dispatch.
        # (If the launch left on its idle limit in between, several launches served the region: then the kernel's own clock
        # stamps around every job, summed, are used instead.)
        launch_info = None
        job_us = sorted(getattr(timed_rollout, "job_us", []) or [])
        if resident and E == 1 and tim.get("launch_jobs", 0) == args.steps and tim.get("launch_ms", 0.0) > 0.0:
            avg_ms = tim["launch_ms"] / tim["launch_jobs"]
            launch_info = {"launches_in_timed_region": 1, "actions_served": tim["launch_jobs"], "launch_ms": round(tim["launch_ms"], 4)}
        elif resident and E == 1:
            launch_info = {"launches_in_timed_region": "more than one (idle limit)", "actions_served": tim.get("launch_jobs", 0)}
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
        traffic, traffic_tab = None, {}
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic_tab = json.load(open(tpath))
            except Exception:
                traffic_tab = {}
        if ngrid == N_GRID or (ngrid == 2048 and args.pml_width == 2.0):
            key = "resident" if resident else impl
            traffic = traffic_tab.get(key if ngrid == N_GRID else f"{key}_{ngrid}")
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
                       "impl": impl, "envs_per_gpu": E, "pml_width": args.pml_width,
                       "in_flight": ("2 actions of the one env (host work of action k+1 under the kernel of action k)"
                                     if args.in_flight == 2 else "1 (plain env(action) loop)")
                                    if E == 1 else "4 envs at a time on their HIP streams",
                       "state_before_every_action": bool(getattr(timed_rollout, "with_state", False)),
                       "device_ms_per_step": round(dev_ms / args.steps, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": (traffic * (launch_info["actions_served"] if resident and launch_info and launch_info["launches_in_timed_region"] == 1 else 1)
                                     if traffic is not None else None),
                         "traffic_source": "profiles/traffic.json: rocprofv3 PMC bytes of this kernel (resident: per action, times the "
                                           "actions the launch served), collected by tools/collect_profiles.sh in a separate run "
                                           "(replayed here, not measured live)"
                                           if traffic is not None else None,
                         "kernel": kname,
                         "avg_kernel_us": round(avg_ms * 1e3, 3), "event_bracketed_kernel_us": round(bracketed_us, 3),
                         "avg_kernel_us_is": ("HIP-event duration of the one resident launch that served the timed region / actions served"
                                              if launch_info and launch_info["launches_in_timed_region"] == 1 else
                                              "mean over the timed region of the per-call durations the library reports"),
                         "launch": launch_info,
                         "job_us": ({"min": round(job_us[0], 2), "median": round(job_us[len(job_us) // 2], 2), "max": round(job_us[-1], 2),
                                     "n": len(job_us), "what": "per action, the kernel's own 100 MHz clock: job seen -> state, frames, trace complete"}
                                    if job_us and resident else None),
                         "algorithmic_bytes_per_launch": alg_bytes * (launch_info["actions_served"] if launch_info and launch_info["launches_in_timed_region"] == 1 else 1),
                         "steps_per_launch": (STEPS_PER_ACTION * (launch_info["actions_served"] if launch_info and launch_info["launches_in_timed_region"] == 1 else 1)) if resident else 1,
                         "algorithmic_bytes_per_action": alg_bytes if resident else None,
                         "whole_job_frac": round(B_ALG * value * 1e6 / world / (HBM_PEAK_GBS * 1e9), 4)},
            "signal_checksum": float(np.sum(all_sig[0][-1])),
        }
        out["ranks_gathered"] = len(all_sig)
        if world == 1 and args.side_configs and E == 1 and not args.stub_env:
            # The same loop over a window ten times as long (never `value`: the contract times exactly --steps actions): what
            # the headline is worth beyond a 19-ms window -- the launch's ramp and the region's ends weigh a tenth.
            n_long = 10 * args.steps
            sync_device()
            t0l = time.perf_counter()
            timed_rollout(env, policy, n_long, args.in_flight)
            env.ctx.synchronize()
            sync_device()
            dtl = time.perf_counter() - t0l
            out["sustained"] = {"actions": n_long, "ms_per_step": round(dtl / n_long * 1e3, 4),
                                "value": round(n_long * cells * STEPS_PER_ACTION / dtl / 1e6, 2), "unit": "Mcell-updates/s",
                                "whole_job_frac": round(B_ALG * n_long * cells * STEPS_PER_ACTION / dtl / (HBM_PEAK_GBS * 1e9), 4)}
        if world == 1 and args.side_configs and ngrid == N_GRID and E == 1 and not args.stub_env:
            for en in envs:  # (a context takes the resident path only when it has the device to itself)
                en.ctx.close()
            out["configs"] = {f"config4_2048_w{int(pw)}": sweep_config(w, ds, dev, args.impl, 2048, pw, 500, traffic_tab)
                              for pw in (1.0, 2.0, 4.0)}
            out["configs"]["config1_size_256"] = side_config(w, ds, dev, args.impl, 256, 2.0, 10, traffic_tab)
        if world == 1 and args.side_configs and ngrid == N_GRID and E == 1 and not args.stub_env:
            out["rollout_config5"] = rollout_config5(w, dim, ds, dev, args.impl, args.pml_width)
        if world == 1 and args.batch_envs > 1 and E == 1 and not args.stub_env:
            out["batched"] = batched_envs(w, dim, ds, dev, args.impl, args.batch_envs, args.pml_width)
        if world == 1 and args.cpu_steps > 0 and not args.stub_env:
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps)
    for en in envs:   # explicit: nothing is left for destructors that would run while the process (or a profiler) shuts down
        en.ctx.close()
    wd.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    wd.finalize()


if __name__ == "__main__":
    main()
