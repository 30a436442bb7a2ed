"""Generates the committed golden vectors of tests/golden/*.npz with the CPU oracle (oracle/waves_oracle.py + C twin).

The reference itself (Julia) cannot run in this pipeline and holds no fixtures for the 2-D path, so these vectors pin
OUR restatement against regressions and give the GPU tests inputs/outputs that do not depend on the oracle code being
present (they are data: inputs and expected outputs).  They are "parity unpinned" with respect to the reference in
the sense of oracle/waves_oracle.py's header.   Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import c_oracle as co  # noqa: E402
import waves_oracle as wo  # noqa: E402

f32 = np.float32


def flat(d):
    c = wo.stacked_cylinders(d)
    return np.concatenate([c.pos, c.r[:, None], c.c[:, None]], axis=1).astype(f32)


def case_moving_design_64():
    """64^2 x 20 steps, 3-cylinder moving design (radii, positions), Gaussian source, random non-zero initial state."""
    rng = np.random.default_rng(2024)
    n, steps = 64, 20
    dim = wo.TwoDim.from_size(15.0, n)
    grid = wo.build_grid(dim)
    pos0 = rng.uniform(-4, 4, (3, 2)).astype(f32)
    d0 = wo.Cylinders(pos0, rng.uniform(1.0, 2.5, 3).astype(f32), rng.uniform(700, 2400, 3).astype(f32))
    d1 = wo.Cylinders((pos0 + rng.uniform(-0.4, 0.4, (3, 2))).astype(f32), (d0.r + rng.uniform(-0.3, 0.3, 3)).astype(f32), d0.c.copy())
    G = wo.build_normal(grid, np.array([[-2.0, 1.0]]), np.array([0.6]), np.array([1.0]))
    u0 = (rng.standard_normal((n, n, 12)) * 0.05).astype(f32)
    ts = wo.build_tspan(f32(0.001), 1e-5, steps)
    # numpy oracle (the literal restatement) ...
    dyn = wo.AcousticDynamics.build(dim, wo.WATER, 2.0, 20000.0)
    it = wo.Integrator(wo.runge_kutta, dyn, f32(1e-5))
    interp = wo.DesignInterpolator(d0, d1, ts[0], ts[-1])
    src = wo.Source(G, f32(1000.0))
    sig = np.zeros((steps + 1, 3), f32)
    dO = f32(wo.get_dx(dim) * wo.get_dy(dim))
    sol = it(u0, ts, [lambda t: wo.speed(interp(t), grid, wo.WATER), lambda t: src(t)], save={0, 10, 20},
             on_state=lambda i, u: sig.__setitem__(i, wo.energies(u[:, :, 0], u[:, :, 6], dO)))
    # ... cross-checked against the C twin before anything is written
    sx = wo.build_pml_profile(dim.x, 2.0, 20000.0)
    st, es, fr = co.integrate(dim.x, dim.y, sx, sx, wo.WATER, 1e-5, wo.to_abi(u0), ts, G=wo.to_abi(G), freq=1000.0,
                              d0=flat(d0), d1=flat(d1), ti=ts[0], tf=ts[-1], frame_steps=(0, 10, 20))
    assert np.array_equal(wo.to_abi(sol), fr) and np.array_equal(es.astype(f32) * dO, sig)
    return dict(n=n, grid_size=f32(15.0), x=dim.x, tspan=ts, c0=wo.WATER, dt=f32(1e-5), pml_width=f32(2.0),
                pml_scale=f32(20000.0), source_shape=G, freq=f32(1000.0), pos0=d0.pos, r0=d0.r, c_0=d0.c, pos1=d1.pos,
                r1=d1.r, c_1=d1.c, u0=u0, frames=sol, signal=sig, sigma=sx,
                speed_mid=wo.speed(interp(f32(0.0011)), grid, wo.WATER))


def case_config1_256():
    """BASELINE config 1: TwoDim(15, 256), Gaussian source at (-10, 0), no design, 100 steps from rest.
    Stored: the U_tot / U_inc planes of the final state and the energy trace (the full state would be 3 MB)."""
    n, steps = 256, 100
    dim = wo.TwoDim.from_size(15.0, n)
    grid = wo.build_grid(dim)
    G = wo.build_normal(grid, np.array([[-10.0, 0.0]]), np.array([0.3]), np.array([1.0]))
    ts = wo.build_tspan(0.0, 1e-5, steps)
    sx = wo.build_pml_profile(dim.x, 2.0, 20000.0)
    st, es, _ = co.integrate(dim.x, dim.y, sx, sx, wo.WATER, 1e-5, np.zeros((12, n, n), f32), ts, G=wo.to_abi(G), freq=1000.0)
    dO = f32(wo.get_dx(dim) * wo.get_dy(dim))
    fin = wo.from_abi(st)
    return dict(n=n, x=dim.x, tspan=ts, source_shape=G, freq=f32(1000.0), u_tot=np.ascontiguousarray(fin[:, :, 0]),
                vx_tot=np.ascontiguousarray(fin[:, :, 1]), psi_x=np.ascontiguousarray(fin[:, :, 3]),
                u_inc=np.ascontiguousarray(fin[:, :, 6]), signal=es.astype(f32) * dO, checksum=np.float64(st.astype(np.float64).sum()))


def case_gradient_1d():
    """The reference's own KAT inputs (test/operators.jl:4-30): OneDim(25f0, 1024), grad*y for x^2, sin, exp."""
    dim = wo.OneDim.from_size(25.0, 1024)
    g = wo.build_gradient(dim.x)
    out = dict(x=dim.x, dx=wo.get_dx(dim))
    for name, fn in (("x2", lambda x: x * x), ("sin", np.sin), ("exp", np.exp)):
        y = fn(dim.x).astype(f32)
        out["y_" + name] = y
        out["d_" + name] = wo.dx(g, y)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "moving_design_64.npz"), **case_moving_design_64())
    np.savez_compressed(os.path.join(HERE, "config1_256.npz"), **case_config1_256())
    np.savez_compressed(os.path.join(HERE, "gradient_1d_1024.npz"), **case_gradient_1d())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
