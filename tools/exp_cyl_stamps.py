"""Phase stamps of one step with a cylinder over the central third of the tiles (diagnostic, GPU box).
WAVES_AMD_STAMPS=<file> python tools/exp_cyl_stamps.py <radius>; then tools/stamps_resident.py <file>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waves_jl_amd as w  # noqa: E402

r = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0   # 0: the triple ring; < 0: a cylinder no tile sees
dim = w.TwoDim(15.0, int(os.environ.get("EXP_GRID", "700")))
src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0, rng=np.random.default_rng(2))
c = w.Cylinders([[0.0, 0.0]] if r > 0 else [[100.0, 100.0]], [abs(r)], [1500.0])
ds = w.build_triple_ring_design_space() if r == 0 else w.DesignSpace(c, c)
env = w.WaveEnv(dim, design_space=ds, source=src, integration_steps=100, actions=20, device=0, impl="fused",
                rng=np.random.default_rng(0), return_fields=False)
env.reset()
pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(1))
for _ in range(4):
    env(pol(env))
env.ctx.close()
