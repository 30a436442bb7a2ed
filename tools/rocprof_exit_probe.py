"""Diagnostic: which ingredient of bench.py makes `rocprofv3 -- python3 bench.py` end in a SIGSEGV inside exit() handlers
(after the profiler has written its files)?  usage: rocprofv3 --kernel-trace --stats -- python3 tools/rocprof_exit_probe.py MODE
MODE is a '+'-joined set of: torch dist gcfreeze env profiling oracle"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
mode = set(sys.argv[1].split("+"))
if "torch" in mode:
    import torch
    torch.zeros(8, device="cuda").sum().item()
import numpy as np
import waves_jl_amd as w
if "dist" in mode:
    from waves_jl_amd import dist as wd
    wd.init()
if "oracle" in mode:
    import c_oracle  # noqa: F401  (loads libwaves_oracle.so: OpenMP runtime)
dim = w.TwoDim(15.0, 256)
src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0, rng=np.random.default_rng(1))
env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=30, actions=50,
                rng=np.random.default_rng(2), return_fields=False, resolution=(64, 64))
pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(3))
env.reset()
if "gcfreeze" in mode:
    gc.collect()
    gc.freeze()
if "env" in mode:
    w.rollout_pipelined(env, pol, 6)
if "profiling" in mode:
    env.ctx.set_profiling(True)
    env(pol(env))
    env.ctx.set_profiling(False)
env.ctx.close()
print("done", sys.argv[1])
