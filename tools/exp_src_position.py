"""Does the source's position move the pace of an action (700^2, triple ring, two in flight)?  The tiles the source shape reaches run
the F_SRC variant; the pairing weights (plan_pair_order) do not know about them.  Diagnostic, GPU box.
  python tools/exp_src_position.py [actions per position 14]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waves_jl_amd as w  # noqa: E402


def main():
    nact = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    dim = w.TwoDim(15.0, 700)
    for mu_x in (-10.0,):
        for mu_y in np.linspace(-10.0, 10.0, 17):
            src = w.RandomPosGaussianSource(w.build_grid(dim), [[mu_x, mu_y]], [[mu_x, mu_y]], [0.3], [1.0], 1000.0, rng=np.random.default_rng(2))
            env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=100, actions=10 ** 6, device=0,
                            impl="fused", rng=np.random.default_rng(0), return_fields=False)
            env.reset()
            pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(1))
            w.rollout_pipelined(env, pol, 4)
            env.ctx.call_times_ms()
            w.rollout_pipelined(env, pol, nact)
            t = np.asarray(env.ctx.call_times_ms()) * 1e3
            print(f"source at ({mu_x:6.2f}, {mu_y:6.2f}): job median {np.median(t):7.1f} us  min {t.min():7.1f}  max {t.max():7.1f}", flush=True)
            env.ctx.close()


if __name__ == "__main__":
    main()
