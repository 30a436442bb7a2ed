"""Where does the host time of one env action go?  (diagnostic, run on the GPU box)"""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import waves_jl_amd as w

dim = w.TwoDim(15.0, 700)
src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                rng=np.random.default_rng(2))
env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=100, actions=200,
                rng=np.random.default_rng(0), return_fields=False)
pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(1))
env.reset()
for _ in range(3):
    env(pol(env))
t0 = time.perf_counter()
n = 20
dev = 0.0
for _ in range(n):
    env(pol(env))
    dev += env.ctx.timing()["total_ms"]
t1 = time.perf_counter()
print(f"host {1e3*(t1-t0)/n:.3f} ms/action, device {dev/n:.3f} ms/action")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    env(pol(env))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
