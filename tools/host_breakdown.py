"""Where does the host time of one env action go?  (diagnostic, run on the GPU box)"""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import waves_jl_amd as w

dim = w.TwoDim(15.0, 700)
src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                rng=np.random.default_rng(2))
env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=100, actions=400,
                rng=np.random.default_rng(0), return_fields=False)
pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(1))
env.reset()
for _ in range(3):
    env(pol(env))
n = 40
t_pol = t_begin = t_end = 0.0
dev = 0.0
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    act = pol(env)
    b = time.perf_counter()
    env.step_begin(act)
    c = time.perf_counter()
    env.step_end()
    d = time.perf_counter()
    t_pol += b - a; t_begin += c - b; t_end += d - c
    dev += env.ctx.timing()["total_ms"]
t1 = time.perf_counter()
print(f"host {1e3*(t1-t0)/n:.3f} ms/action = policy {1e3*t_pol/n:.3f} + step_begin {1e3*t_begin/n:.3f} + step_end {1e3*t_end/n:.3f}; device {dev/n:.3f} ms/action")
ts = env.build_tspan()
interp = w.DesignInterpolator(env.design, env.design, ts[0], ts[-1])
a = time.perf_counter()
for _ in range(n):
    env.ctx.set_design(*interp.abi_args())
b = time.perf_counter()
for _ in range(n):
    env.build_tspan()
c = time.perf_counter()
print(f"set_design {1e3*(b-a)/n:.3f} ms, build_tspan {1e3*(c-b)/n:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    env(pol(env))
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
