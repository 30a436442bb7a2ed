"""Is the 2048^2 step time bimodal within a process or between processes?  (diagnostic, GPU box)
python tools/exp_2048_modes.py [pml_width] [calls] [grid]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waves_jl_amd as w  # noqa: E402

pw = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dim = w.TwoDim(15.0, grid)
src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0, rng=np.random.default_rng(77))
env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=500, actions=calls + 2, device=0,
                impl="fused", rng=np.random.default_rng(78), return_fields=False, pml_width=pw)
pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(79))
env.reset()
out = []
for k in range(calls):
    t0 = time.perf_counter()
    env(pol(env))
    dt = time.perf_counter() - t0
    t = env.ctx.timing()
    out.append((dt * 1e3, t["step_kernel_ms"] / max(t["step_kernel_launches"], 1) * 1e3))
print("grid", grid, "ns per cell-step (kernel)", " ".join(f"{b * 1e3 / grid / grid:.4f}" for a, b in out))
print("pml_width", pw, " per call: wall ms / kernel us per step:", " | ".join(f"{a:.1f} / {b:.1f}" for a, b in out), flush=True)
env.ctx.close()
