"""Cost of returning trajectories from a 700^2 action (run on the GPU box): no fields / blocking copy in wv_integrate_end /
streamed on the copy stream under the next action, for several strides.  usage: python tools/stream_cost.py"""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np

import waves_jl_amd as w


def run(mode, stride, actions=12):
    dim = w.TwoDim(15.0, 700)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(1))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=100,
                    actions=actions + 8, rng=np.random.default_rng(2), return_fields=mode, trajectory_stride=stride)
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(3))
    env.reset()
    for _ in range(3):
        env(pol(env))
    t0 = time.perf_counter()
    if mode is True:            # the blocking copy needs the call ended before the next begins
        for _ in range(actions):
            env(pol(env))
    else:
        w.rollout_pipelined(env, pol, actions)
    dt = (time.perf_counter() - t0) / actions
    env.ctx.close()
    gc.collect()
    return dt


base = run(False, 1)
print(f"no fields                         {base * 1e3:7.3f} ms per action")
for stride in (50, 25, 10, 5, 1):
    mb = (100 // stride + 1) * 2 * 700 * 700 * 4 / 1e6
    a, b = run(True, stride), run("stream", stride)
    print(f"stride {stride:3d} ({mb:6.1f} MB per action): blocking copy {a * 1e3:7.3f} ms (+{(a / base - 1) * 100:5.1f} %)   "
          f"streamed {b * 1e3:7.3f} ms (+{(b / base - 1) * 100:5.1f} %)   [{mb / 1e3 / max(b - 0.0, 1e-9):5.1f} GB/s if transfer-bound]")
