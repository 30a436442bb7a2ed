"""A policy that computes on the same GPU between two actions (a small torch MLP on state(env)): what the waiting resident launch
costs it, and what the three ways out cost (diagnostic, GPU box).  python tools/exp_gpu_policy.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waves_jl_amd as w  # noqa: E402


def run(name, n_actions=30, sync_before_policy=False):
    dim = w.TwoDim(15.0, 700)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0, rng=np.random.default_rng(2))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=100, actions=n_actions + 10, device=0,
                    impl="fused", rng=np.random.default_rng(0), return_fields=False)
    env.reset()
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(1))
    dev = torch.device("cuda:0")
    net = torch.nn.Sequential(torch.nn.Linear(128 * 128 * 4, 1024), torch.nn.ReLU(), torch.nn.Linear(1024, 1024), torch.nn.ReLU(),
                              torch.nn.Linear(1024, 19)).to(dev)
    t_pol = 0.0

    def policy(env):
        nonlocal t_pol
        s = env.state().wave
        if sync_before_policy:
            env.ctx.synchronize()          # hand the GPU over: the waiting launch leaves now
        a = time.perf_counter()
        with torch.no_grad():
            y = net(torch.from_numpy(np.ascontiguousarray(s)).reshape(1, -1).to(dev))
            y.cpu()                         # (the policy's answer is needed on the host)
        t_pol += time.perf_counter() - a
        return pol(env)                     # (the action itself stays random: only the timing matters here)

    for _ in range(5):
        env(policy(env))
    torch.cuda.synchronize()
    t_pol = 0.0
    t0 = time.perf_counter()
    for _ in range(n_actions):
        env(policy(env))
    env.ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:58s} {dt / n_actions * 1e3:.3f} ms per action, of it the policy's GPU work {t_pol / n_actions * 1e3:.3f} ms", flush=True)
    env.ctx.close()


if __name__ == "__main__":
    run("waiting launch beside the policy (default, idle limit 1 ms)")
    os.environ["WAVES_AMD_IDLE_US"] = "50"
    run("idle limit 50 us")
    os.environ["WAVES_AMD_IDLE_US"] = "0"
    run("no waiting launch at all (WAVES_AMD_IDLE_US=0)")
    del os.environ["WAVES_AMD_IDLE_US"]
    run("ctx.synchronize() in front of the policy", sync_before_policy=True)
