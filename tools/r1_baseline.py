"""R1 of BASELINE.md / SURVEY 8d: the "reference-structure" CPU restatement timed on one core.

The numpy oracle (oracle/waves_oracle.py: same slices, temporaries, concats and four wave-speed-field assemblies per
step as the Julia source) with its derivative operator replaced by what the reference literally does -- a scipy.sparse
CSC gradient matrix applied as `grad * u` and `(grad * u')'` (src/operators.jl:10-26,45-46) -- on ONE thread, like the
reference's own CPU path (Julia's SparseArrays `*` and broadcast are single-threaded; its scripts never start Julia with
threads).  It is a restatement written from the source text, NOT the reference binary (no Julia toolchain exists here).

  python tools/r1_baseline.py [steps_config1 = 100] [steps_config2 = 8]

Checks first, on a small grid, that the sparse-matrix form gives the same bits as the oracle's stencil form.
"""
import os
import sys
import time

for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[v] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import scipy.sparse as sp

import waves_oracle as wo

f32 = np.float32


def sparse_gradient(x):
    """gradient(x), src/operators.jl:10-22, as a CSC matrix (what build_gradient hands to Flux)."""
    return sp.csc_matrix(wo.gradient_dense(x, f32))


def install_sparse_operators(dim):
    """dx / dy of the oracle -> sparse matrix products, src/operators.jl:45-46."""
    G = sparse_gradient(dim.x)
    wo.dx = lambda g, u: np.asarray(G @ u, dtype=f32)
    wo.dy = lambda g, u: np.ascontiguousarray(np.asarray(G @ np.ascontiguousarray(u.T), dtype=f32).T)


def run_env(n, steps, design, seed=0):
    dim = wo.TwoDim.from_size(15.0, n)
    grid = wo.build_grid(dim)
    rng = np.random.default_rng(seed)
    if design:
        ds = wo.build_triple_ring_design_space()
        src = wo.RandomPosGaussianSource(grid, np.array([[-10.0, -10.0]], f32), np.array([[-10.0, 10.0]], f32),
                                         np.array([0.3], f32), np.array([1.0], f32), f32(1000.0))
        env = wo.WaveEnv(dim, design_space=ds, source=src, integration_steps=steps, actions=1, rng=rng, resolution=(64, 64))
        env.reset()
        action = wo.RandomDesignPolicy(env.action_space(), np.random.default_rng(seed + 1))(env)
        t0 = time.perf_counter()
        env(action, return_fields=False)
        return time.perf_counter() - t0, env.wave[:, :, :, -1], env.signal
    # config 1: no design (C(t) = c0), Source(build_normal(grid, [-10 0], [0.3], [1.0]), 1000)
    dyn = wo.AcousticDynamics.build(dim, wo.WATER, 2.0, 20000.0, f32)
    it = wo.Integrator(wo.runge_kutta, dyn, f32(1e-5))
    shape = wo.build_normal(grid, np.array([[-10.0, 0.0]]), np.array([0.3]), np.array([1.0]))
    src = wo.Source(shape, f32(1000.0))
    ts = wo.build_tspan(0.0, 1e-5, steps)
    u = np.zeros((n, n, 12), f32)
    t0 = time.perf_counter()
    sol = it(u, ts, [lambda t: wo.WATER, lambda t: src(t, f32)], save={steps})
    return time.perf_counter() - t0, sol[:, :, :, -1], None


def run_design_steps(n, steps, seed=0):
    """`steps` integration steps of the config-2 workload (triple ring moving between two random designs, Gaussian source)
    through the integrator alone -- no env bookkeeping, so that a bounded sample may be shorter than the 20 steps the env's
    frame capture needs.  Same closures as WaveEnv's step builds (src/env.jl:95-102)."""
    dim = wo.TwoDim.from_size(15.0, n)
    grid = wo.build_grid(dim)
    rng = np.random.default_rng(seed)
    ds = wo.build_triple_ring_design_space()
    a = wo.rand_design(ds, rng)
    b = ds(a, wo.rand_design(wo.build_action_space(a, 0.25), rng))
    ts = wo.build_tspan(0.0, 1e-5, steps)
    interp = wo.DesignInterpolator(a, b, f32(0.0), ts[-1])
    dyn = wo.AcousticDynamics.build(dim, wo.WATER, 2.0, 20000.0, f32)
    it = wo.Integrator(wo.runge_kutta, dyn, f32(1e-5))
    src = wo.Source(wo.build_normal(grid, np.array([[-10.0, 3.0]]), np.array([0.3]), np.array([1.0])), f32(1000.0))
    C = lambda t: wo.speed(interp(t, f32), grid, dyn.c0)
    F = lambda t: src(t, f32)
    u = np.zeros((n, n, 12), f32)
    t0 = time.perf_counter()
    it(u, ts, [C, F], save={steps})
    return time.perf_counter() - t0


def bounded_sample(steps1=10, steps2=2):
    """What bench.py reports as cpu_baseline.r1: a few steps of config 1 and config 2 on this host, one core."""
    out = {}
    stencil_dx, stencil_dy = wo.dx, wo.dy
    try:
        install_sparse_operators(wo.TwoDim.from_size(15.0, 256))
        dt1, _, _ = run_env(256, steps1, False)
        install_sparse_operators(wo.TwoDim.from_size(15.0, 700))
        dt2 = run_design_steps(700, steps2)
    finally:
        wo.dx, wo.dy = stencil_dx, stencil_dy
    out["config1_256"] = {"value": round(256 * 256 * steps1 / dt1 / 1e6, 4), "sample": f"{steps1} steps ({dt1:.1f} s)"}
    out["config2_700"] = {"value": round(700 * 700 * steps2 / dt2 / 1e6, 4), "sample": f"{steps2} steps ({dt2:.1f} s)"}
    return out


def main():
    s1 = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s2 = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    # --- the sparse-matrix operator gives the oracle's bits
    stencil_dx, stencil_dy = wo.dx, wo.dy
    _, ref, _ = run_env(96, 20, True)
    install_sparse_operators(wo.TwoDim.from_size(15.0, 96))
    _, got, _ = run_env(96, 20, True)
    assert np.array_equal(ref, got), "sparse-matrix gradient differs from the stencil form"
    print("sparse CSC gradient == stencil form of the oracle: bit-exact on 96^2 x 20 steps")
    try:
        ncore = len(os.sched_getaffinity(0))
    except Exception:
        ncore = os.cpu_count()
    cpu = next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    print(f"host: {cpu}, {ncore} cores visible, 1 used")
    for name, n, steps, design in (("config 1: 256^2, no design, Gaussian source", 256, s1, False),
                                   ("config 2: 700^2, triple ring, random-position source", 700, s2, True)):
        install_sparse_operators(wo.TwoDim.from_size(15.0, n))
        dt, _, _ = run_env(n, max(steps, 20) if design else steps, design)
        st = max(steps, 20) if design else steps
        print(f"R1 {name}: {st} steps in {dt:.2f} s  ->  {dt / st * 100:.1f} s per 100 steps, "
              f"{n * n * st / dt / 1e6:.3f} Mcell-updates/s (1 core)")
    wo.dx, wo.dy = stencil_dx, stencil_dy


if __name__ == "__main__":
    main()
