"""The device's own account of a job (WAVES_AMD_JOBLOG) in the two loops: one action at a time and two in flight (diagnostic, GPU box)."""
import os
import sys

import numpy as np

os.environ["WAVES_AMD_JOBLOG"] = "1"
if len(sys.argv) > 1 and sys.argv[1] in "01":
    os.environ["WAVES_AMD_DEV_TABLES"] = sys.argv[1]   # 0: the host builds the tile tables of every call
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import waves_jl_amd as w  # noqa: E402

ds = w.build_triple_ring_design_space()
env, policy = bench.make_env(w, w.TwoDim(15.0, 700), ds, 0, "fused", 2.0, 1000, 5)
if len(sys.argv) > 2 and sys.argv[2] == "frozen":   # radii that hardly move: the launch order of the first call stays right
    env.action_speed = 1e-3
    policy = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(5))
print("== one action at a time", file=sys.stderr, flush=True)
for _ in range(14):
    env(policy(env))
env.ctx.synchronize()
print("== two in flight", file=sys.stderr, flush=True)
w.rollout_pipelined(env, policy, 14)
env.ctx.synchronize()
