"""Where the host's time goes in the plain `env(action)` loop (diagnostic, GPU box): cProfile around N actions at the headline
configuration.  python tools/host_profile.py [N]"""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import waves_jl_amd as w  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ds = w.build_triple_ring_design_space()
    env, policy = bench.make_env(w, w.TwoDim(15.0, 700), ds, 0, "fused", 2.0, n + 20, 5)
    for _ in range(10):
        env(policy(env))

    def loop():
        for _ in range(n):
            env(policy(env))

    pr = cProfile.Profile()
    pr.enable()
    loop()
    pr.disable()
    env.ctx.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
