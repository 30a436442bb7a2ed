"""Shared helpers of the test-suite: case builders that feed the SAME seeded inputs to the oracle and to the device."""
import numpy as np

import c_oracle as co
import waves_oracle as wo

f32 = np.float32


def flat_design(design):
    """stacked cylinders -> (M, 4) px, py, r, c for the C oracle."""
    c = wo.stacked_cylinders(design)
    return np.concatenate([c.pos, c.r[:, None], c.c[:, None]], axis=1).astype(f32)


def oracle_to_mirror_design(w, d):
    """oracle design object -> host-mirror design object (same numbers)."""
    if isinstance(d, wo.NoDesign):
        return w.NoDesign()
    if isinstance(d, wo.Cylinders):
        return w.Cylinders(d.pos, d.r, d.c)
    if isinstance(d, wo.Cloak):
        return w.Cloak(oracle_to_mirror_design(w, d.config), oracle_to_mirror_design(w, d.core))
    if isinstance(d, wo.AdjustablePositionScatterers):
        return w.AdjustablePositionScatterers(oracle_to_mirror_design(w, d.cylinders))
    return w.AdjustableRadiiScatterers(oracle_to_mirror_design(w, d.cylinders))


def random_state(rng, nx, ny, scale=1.0, aux=True):
    """(nx, ny, 12) Fortran-ordered random state; aux=False leaves Psi_x, Psi_y, Omega zero."""
    u = (rng.standard_normal((nx, ny, 12)) * scale).astype(f32)
    if not aux:
        u[:, :, [3, 4, 5, 9, 10, 11]] = 0
    return np.asfortranarray(u)


def small_moving_design(rng, n_cyl=3, centre=(0.0, 0.0), spread=3.0):
    """Two Cylinders designs (initial, final) with moving radii AND positions and distinct speeds."""
    pos0 = (rng.uniform(-spread, spread, (n_cyl, 2)) + np.array(centre)).astype(f32)
    pos1 = (pos0 + rng.uniform(-0.3, 0.3, (n_cyl, 2))).astype(f32)
    r0 = rng.uniform(0.4, 1.5, n_cyl).astype(f32)
    r1 = (r0 + rng.uniform(-0.3, 0.3, n_cyl)).astype(f32)
    c = rng.uniform(600.0, 2500.0, n_cyl).astype(f32)
    return wo.Cylinders(pos0, r0, c), wo.Cylinders(pos1, r1, c.copy())


def oracle_integrate(dim, state_abi, tspan, *, c0=wo.WATER, dt=1e-5, pml=(2.0, 20000.0), G=None, freq=0.0, d0=None,
                     d1=None, ti=0.0, tf=0.0, frame_steps=(), nthreads=8):
    """C oracle run.  state_abi: (12, ny, nx).  Returns (final (12,ny,nx), signal (n+1,3) f32 scaled, frames)."""
    sx = wo.build_pml_profile(dim.x, pml[0], pml[1])
    st, es, fr = co.integrate(dim.x, dim.y, sx, sx, c0, dt, state_abi, tspan, G=None if G is None else wo.to_abi(G),
                              freq=freq, d0=d0, d1=d1, ti=ti, tf=tf, frame_steps=frame_steps, nthreads=nthreads)
    dO = f32(wo.get_dx(dim) * wo.get_dy(dim))
    return st, es.astype(f32) * dO, fr


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
