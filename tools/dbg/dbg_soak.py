import os, sys, gc, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
os.environ["WAVES_AMD_OBS_IN_JOB"] = os.environ.get("WAVES_AMD_OBS_IN_JOB", "0")
import waves_jl_amd as w

def _env(n, steps, actions, seed, **kw):
    dim = w.TwoDim(15.0, n)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0, rng=np.random.default_rng(seed + 2))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=steps, actions=actions, rng=np.random.default_rng(seed), return_fields=False, **kw)
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(seed + 1))
    env.reset()
    return env, pol

def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us: pass

def soak(n_actions, seed, pattern_seed, sleeps):
    gc.collect()
    env, pol = _env(160, 30, n_actions + 4, seed)
    pr = np.random.default_rng(pattern_seed)
    sigs, modes = [], []
    k = 0
    while k < n_actions:
        mode = int(pr.integers(0, 4)); pause = float(pr.uniform(0.0, 90.0))
        if sleeps: spin(pause)
        if mode == 0 or k + 2 > n_actions:
            env(pol(env)); sigs.append(env.signal.copy()); modes.append(0); k += 1
        elif mode == 1:
            env.step_begin(pol(env)); env.step_begin(pol(env))
            if sleeps: spin(pause / 2)
            env.step_end(); sigs.append(env.signal.copy()); env.step_end(); sigs.append(env.signal.copy()); modes += [1, 1]; k += 2
        elif mode == 2:
            env.state(); env(pol(env)); sigs.append(env.signal.copy()); modes.append(2); k += 1
        else:
            env.ctx.synchronize(); env(pol(env)); sigs.append(env.signal.copy()); modes.append(3); k += 1
    env.ctx.close()
    return sigs, modes

idle, pat = sys.argv[1], int(sys.argv[2])
os.environ["WAVES_AMD_IDLE_US"] = idle
s1, m = soak(120, 300 + pat, pat, True)
del os.environ["WAVES_AMD_IDLE_US"]
s2, _ = soak(120, 300 + pat, pat, False)
bad = [i for i, (a, b) in enumerate(zip(s1, s2)) if not np.array_equal(a, b)]
print("idle", idle, "pattern", pat, "differing actions:", bad[:10], "modes around first:", (m[max(0, bad[0] - 4):bad[0] + 3] if bad else None))
if bad:
    i = bad[0]
    print(" row0 run1", s1[i][0], "run2", s2[i][0], " last row prev: run1", s1[i-1][-1] if i else None, "run2", s2[i-1][-1] if i else None)
    d = np.where(np.any(s1[i] != s2[i], axis=1))[0]
    print(" first differing rows of that action:", d[:8])
