"""Where the host's time goes between two jobs of the plain `env(action)` loop (one action at a time), piece by piece with
perf_counter (cProfile inflates small calls).  Diagnostic, GPU box.  python tools/host_split.py [N 300]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import waves_jl_amd as w  # noqa: E402
from waves_jl_amd.designs import DesignInterpolator  # noqa: E402
from waves_jl_amd.env import build_tspan  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ds = w.build_triple_ring_design_space()
    env, policy = bench.make_env(w, w.TwoDim(15.0, 700), ds, 0, "fused", 2.0, n + 40, 5)
    for _ in range(10):
        env(policy(env))
    names = ["policy", "build_tspan", "design_space", "DesignInterpolator", "abi_args", "ctx.set_design", "ctx.integrate_begin", "bookkeeping",
             "next tspan", "wait + ctx.integrate_end"]
    rows = []
    t_all = time.perf_counter()
    for _ in range(n):
        t = [time.perf_counter()]
        a = policy(env); t.append(time.perf_counter())
        tspan = env.build_tspan(); ti = env.time(); t.append(time.perf_counter())
        cur = env.design; nxt = env.design_space(cur, a); t.append(time.perf_counter())
        interp = DesignInterpolator(cur, nxt, ti, tspan[-1]); t.append(time.perf_counter())
        args = interp.abi_args(); t.append(time.perf_counter())
        env.ctx.set_design(*args); t.append(time.perf_counter())
        env.ctx.integrate_begin(tspan, capture_frames=True, want_signal=True, want_fields=False); t.append(time.perf_counter())
        env._pending = getattr(env, "_pending", None) or []
        env._pending.append((tspan, interp)); env.design = nxt; env.time_step += env.integration_steps; t.append(time.perf_counter())
        build_tspan(env.time(), env.dt, env.integration_steps); t.append(time.perf_counter())
        env.step_end(); t.append(time.perf_counter())
        rows.append(np.diff(t))
    total = (time.perf_counter() - t_all) / n * 1e6
    env.ctx.synchronize()
    r = np.array(rows) * 1e6
    med = np.median(r, axis=0)
    for k, name in enumerate(names):
        print(f"{name:28s} median {med[k]:7.1f} us   mean {r[:, k].mean():7.1f}")
    jobs = np.asarray(env.ctx.call_times_ms()) * 1e3
    print(f"per action {total:.1f} us; job median {np.median(jobs):.1f} us; host pieces before the bell (policy .. integrate_begin) {med[:7].sum():.1f} us")


if __name__ == "__main__":
    main()
