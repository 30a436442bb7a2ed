"""Randomised soak of the resident launch's job protocol (GPU box): a mixed sequence of env / context calls -- plain actions,
two in flight, state(env), synchronize, reset, get/set_state, action sequences in one call, trajectories, rhs -- is run twice
with the same seeds: once with host pauses around a short idle limit of the launch (so that calls find it waiting, leaving or
gone), once undisturbed with the default limit.  Every trace, observation and the final frames must be the same bytes.

  python tools/soak.py [first_seed 0] [n_seeds 20] [actions 150] [grid 160]
"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import waves_jl_amd as w  # noqa: E402


def make_env(n, steps, actions, seed, **kw):
    dim = w.TwoDim(15.0, n)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0, rng=np.random.default_rng(seed + 2))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=steps, actions=actions,
                    rng=np.random.default_rng(seed), return_fields=False, **kw)
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(seed + 1))
    env.reset()
    return env, pol


def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass


def soak(n_actions, grid, steps, seed, sleeps):
    gc.collect()
    env, pol = make_env(grid, steps, 10 ** 6, seed)
    pr = np.random.default_rng(seed + 1000)
    out, modes = [], []
    k = 0
    while k < n_actions:
        mode = int(pr.choice(10, p=[0.22, 0.2, 0.16, 0.08, 0.05, 0.06, 0.08, 0.05, 0.05, 0.05]))
        pause = float(pr.uniform(0.0, 120.0))
        if sleeps:
            spin(pause)
        modes.append(mode)
        if mode == 0:                              # env(action)
            env(pol(env))
            out.append(env.signal.copy())
            k += 1
        elif mode == 1:                            # two in flight
            env.step_begin(pol(env))
            env.step_begin(pol(env))
            if sleeps:
                spin(pause / 2)
            env.step_end()
            out.append(env.signal.copy())
            if sleeps:
                spin(pause / 3)
            env.step_end()
            out.append(env.signal.copy())
            k += 2
        elif mode == 2:                            # state(env) in front of the action
            out.append(np.array(env.state().wave))
            env(pol(env))
            out.append(env.signal.copy())
            k += 1
        elif mode == 3:                            # the launch is told to leave
            env.ctx.synchronize()
        elif mode == 4:                            # reset!(env)
            env.reset()
        elif mode == 5:                            # the state leaves and comes back
            u = env.ctx.get_state()
            out.append(u[:, :, 0].copy())
            env.ctx.set_state(u)
        elif mode == 6:                            # three actions as one call
            sigs = env.steps_begin([pol(env) for _ in range(3)]) and env.steps_end()
            out += [s.copy() for s in sigs]
            k += 3
        elif mode == 7:                            # an action that returns its trajectories
            env.return_fields = True
            _, _, ut, ui = env(pol(env))
            env.return_fields = False
            out.append(env.signal.copy())
            out.append(ut[:, :, -1].copy())
            k += 1
        elif mode == 8:                            # another resolution of the observation, then the usual one
            out.append(env.ctx.observation(64, 64).copy())
            out.append(np.array(env.state().wave))
        else:                                      # three in a row, each begun as soon as the one before has ended
            for _ in range(3):
                env.step_begin(pol(env))
                env.step_end()
                out.append(env.signal.copy())
            k += 3
    out.append(np.array(env.ctx.get_frames()))
    res = env.ctx.timing()["resident"]
    env.ctx.close()
    return out, modes, res


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    actions = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    grid = int(sys.argv[4]) if len(sys.argv) > 4 else 160
    bad = 0
    for seed in range(first, first + nseeds):
        idle = [15, 30, 45, 60, 80, 110][seed % 6]
        os.environ["WAVES_AMD_IDLE_US"] = str(idle)
        a, modes, ra = soak(actions, grid, 30, seed, True)
        del os.environ["WAVES_AMD_IDLE_US"]
        b, _, rb = soak(actions, grid, 30, seed, False)
        diff = [i for i, (x, y) in enumerate(zip(a, b)) if not np.array_equal(x, y)]
        ok = not diff and len(a) == len(b)
        bad += 0 if ok else 1
        print(f"seed {seed} idle {idle} us: {len(a)} outputs, resident {ra}/{rb}: {'same bytes' if ok else 'DIFFERENT from output ' + str(diff[:5])}",
              flush=True)
    print("FAILED" if bad else "PASS", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
