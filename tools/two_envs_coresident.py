"""Experiment: do two INDEPENDENT simulations sharing the CUs run faster than one simulation with
twice the tiles?  Two 500^2 environments (225 tiles each with WAVES_AMD_FUSED_AUTOTILE=0: one tile of each per CU) are
stepped concurrently with the resident kernel (WAVES_AMD_FORCE_RESIDENT=1: 450 blocks <= 512 slots), against one of them
alone and against one 707^2 environment (the same number of cells: ~2 tiles of the SAME simulation per CU).
Prints microseconds per integration step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("WAVES_AMD_FUSED_AUTOTILE", "0")
os.environ["WAVES_AMD_FORCE_RESIDENT"] = "1"
import numpy as np
import torch
import waves_jl_amd as w

STEPS, ACTIONS = 100, 12


def make(n, seed):
    dim = w.TwoDim(15.0, n)
    ds = w.build_triple_ring_design_space()
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(seed))
    env = w.WaveEnv(dim, design_space=ds, source=src, integration_steps=STEPS, actions=2 * ACTIONS + 8, device=0,
                    rng=np.random.default_rng(seed + 1), return_fields=False)
    env.reset()
    return env, w.RandomDesignPolicy(env.action_space(), np.random.default_rng(seed + 2))


def run(envs):
    for _ in range(2):
        for e, p in envs:
            e.step_begin(p(e))
        for e, p in envs:
            e.step_end()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern = 0.0
    for _ in range(ACTIONS):
        for e, p in envs:
            e.step_begin(p(e))
        for e, p in envs:
            e.step_end()
            kern += e.ctx.timing()["step_kernel_ms"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = [bool(e.ctx.timing()["resident"]) for e, _ in envs]
    return dt / (ACTIONS * STEPS) * 1e6, kern / (ACTIONS * len(envs) * STEPS) * 1e3, res


if __name__ == "__main__":
    a = make(500, 1)
    print("one 500^2 env alone        : wall %.2f us per step, kernel %.2f us per step, resident %s" % run([a]))
    b = make(500, 11)
    print("two 500^2 envs concurrently: wall %.2f us per step of BOTH, kernel %.2f us per step each, resident %s" % run([a, b]))
    a[0].ctx.close()
    b[0].ctx.close()
    c = make(707, 21)
    print("one 707^2 env (same cells) : wall %.2f us per step, kernel %.2f us per step, resident %s" % run([c]))
