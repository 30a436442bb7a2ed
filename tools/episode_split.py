"""Where an episode's time goes beyond its actions (diagnostic, GPU box): reset!(env), the first action (a launch starts), the rest,
at the headline configuration, 20 actions per episode, two in flight.  python tools/episode_split.py [episodes 12]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import waves_jl_amd as w  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    ds = w.build_triple_ring_design_space()
    env, policy = bench.make_env(w, w.TwoDim(15.0, 700), ds, 0, "fused", 2.0, 20, 5)
    rows = []
    for e in range(n + 2):
        t0 = time.perf_counter()
        env.reset()
        t1 = time.perf_counter()
        env.step_begin(policy(env))
        t2 = time.perf_counter()
        env.step_begin(policy(env))
        t3 = time.perf_counter()
        env.step_end()
        t4 = time.perf_counter()
        for k in range(18):
            env.step_begin(policy(env))
            env.step_end()
        env.step_end()
        t5 = time.perf_counter()
        if e >= 2:
            rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t5 - t0])
    r = np.median(np.array(rows), axis=0) * 1e6
    print(f"median over {n} episodes (us): reset {r[0]:.0f} | first step_begin {r[1]:.0f} | second step_begin {r[2]:.0f} | first step_end {r[3]:.0f} | "
          f"the other 19 actions {r[4]:.0f} ({r[4] / 19:.1f} each) | episode {r[5]:.0f} = {r[5] / 20:.1f} per action")
    t = env.ctx.timing()
    print("resident", t["resident"], "gave_up", t["gave_up"])


if __name__ == "__main__":
    main()
