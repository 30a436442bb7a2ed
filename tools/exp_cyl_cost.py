"""What the moving cylinders cost a resident action at 700^2 (diagnostic, GPU box): job durations by the kernel's own clock for
the triple-ring design, one cylinder and a cylinder no tile sees."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import waves_jl_amd as w  # noqa: E402


GRID = int(os.environ.get("EXP_GRID", "700"))


def run(name, ds, n=20):
    dim = w.TwoDim(15.0, GRID)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(2))
    env = w.WaveEnv(dim, design_space=ds, source=src, integration_steps=100, actions=n + 10, device=0, impl="fused",
                    rng=np.random.default_rng(0), return_fields=False)
    env.reset()
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(1))
    w.rollout_pipelined(env, pol, 5)
    env.ctx.call_times_ms()
    w.rollout_pipelined(env, pol, n)
    t = np.array(env.ctx.call_times_ms()) * 1e3
    print(f"{name:28s} jobs {len(t)}  min {t.min():.1f}  median {np.median(t):.1f}  max {t.max():.1f} us", flush=True)
    env.ctx.close()


if __name__ == "__main__":
    tr = w.build_triple_ring_design_space()
    far = w.Cylinders([[100.0, 100.0]], [0.1], [1500.0])   # outside the domain: culled from every tile
    far19 = w.Cylinders([[100.0 + k, 100.0] for k in range(19)], [0.1] * 19, [1500.0] * 19)
    big = w.Cylinders([[0.0, 0.0]], [40.0], [1500.0])       # covers the whole domain: every tile evaluates it
    mid = w.Cylinders([[0.0, 0.0]], [0.5], [1500.0])
    c10 = w.Cylinders([[0.0, 0.0]], [10.0], [1500.0])       # interior tiles only (no PML tile)
    c5 = w.Cylinders([[0.0, 0.0]], [5.0], [1500.0])
    cases = {"triple": ("triple ring (19 cylinders)", tr), "far": ("1 cylinder, out of reach", w.DesignSpace(far, far)),
             "far19": ("19 cylinders, out of reach", w.DesignSpace(far19, far19)), "big": ("1 cylinder over every tile", w.DesignSpace(big, big)),
             "c10": ("1 cylinder r=10 (interior)", w.DesignSpace(c10, c10)), "c5": ("1 cylinder r=5 (interior)", w.DesignSpace(c5, c5)),
             "mid": ("1 fixed cylinder r=0.5", w.DesignSpace(mid, mid))}
    which = sys.argv[1:] or list(cases)
    for rep in range(2):
        for k in which:
            run(*cases[k])
