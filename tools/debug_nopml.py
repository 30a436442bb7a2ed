import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import waves_jl_amd as w, waves_oracle as wo
f32 = np.float32
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dim = w.TwoDim(5.0, n); odim = wo.TwoDim.from_size(5.0, n)
it = w.Integrator(w.runge_kutta, w.AcousticDynamics(dim, w.WATER, 1.0, 0.0), 1e-5)
wave = w.build_wave(dim, 12)
ic = wo.build_normal(wo.build_grid(odim), np.array([[0.0, 0.0]]), np.array([0.3]), np.array([1.0]))
wave[:, :, 0] = ic; wave[:, :, 6] = ic
ts = it.build_tspan(0.0, 30)
oit = wo.Integrator(wo.runge_kutta, wo.AcousticDynamics.build(odim, wo.WATER, 1.0, 0.0), f32(1e-5))
for saves in ([1], [2], [10], [0, 10, 30]):
    sol = it(wave, ts, [w.UniformSpeed(w.WATER), w.NoSource()], save=saves)
    ref = oit(np.array(wave), ts, [lambda t: wo.WATER, wo.NoSource()], save=set(saves))
    d = np.abs(sol - ref)
    bad = np.argwhere(sol != ref)
    print("saves", saves, "equal", np.array_equal(sol, ref), "max diff", d.max(), "nbad", len(bad), "first bad", bad[:3].tolist(), "fields", sorted(set(bad[:, 2].tolist()))[:12] if len(bad) else [])
    if len(bad):
        i, j, f, k = bad[0]
        print("   got", sol[i, j, f, k], "want", ref[i, j, f, k], " x-range of bad", bad[:, 0].min(), bad[:, 0].max(), "y-range", bad[:, 1].min(), bad[:, 1].max())
