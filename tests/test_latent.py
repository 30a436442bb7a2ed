"""the batched 1-D latent dynamics of the surrogate models (src/dynamics.jl:190-222 under RK4), device path
(wv_latent_integrate) against the numpy restatement oracle/latent_oracle.py -- bit for bit.  PARITY UNPINNED by the
reference (it holds no fixture for this path); the restatement is pinned by known answers below."""
import numpy as np
import pytest

import latent_oracle as lo
import waves_jl_amd as w
import waves_oracle as wo

f32 = np.float32


def test_linear_interp_known_answers():
    X = np.array([[0.0, 0.0], [1.0, 2.0], [3.0, 4.0]], f32)                 # (K = 3, B = 2)
    Y = np.zeros((2, 3, 2), f32)
    Y[0, :, 0], Y[1, :, 0] = [0, 10, 30], [1, 1, 5]
    Y[0, :, 1], Y[1, :, 1] = [2, 4, 8], [0, -2, -2]
    C = lo.LinearInterpolation(X, Y)
    assert np.array_equal(C(np.array([0.0, 0.0], f32)), Y[:, 0, :])         # at the first knot
    assert np.array_equal(C(np.array([3.0, 4.0], f32)), Y[:, 2, :])         # at the final knot (the `final_step` branch)
    assert np.allclose(C(np.array([0.5, 3.0], f32)), [[5.0, 6.0], [1.0, -2.0]])
    assert np.array_equal(C(np.array([9.0, 9.0], f32)), np.zeros((2, 2), f32))   # outside: every mask is false


def test_latent_oracle_reduces_to_the_wave_equation():
    """C = 1, no source, no damping: total and incident sets solve the same equation (up to the rounding of where c0 is
    multiplied in), and a Gaussian pulse splits into two halves of half the amplitude that travel c0*t."""
    n, steps = 1024, 300
    x = wo.OneDim.from_size(15.0, n).x
    dyn = lo.LatentDynamics(x, 1531.0, 5.0, 10000.0)
    t = wo.build_tspan(0.0, 1e-5, steps)[:, None]
    C = lo.LinearInterpolation(t[[0, -1], :], np.ones((n, 2, 1), f32))
    F = lo.Source1D(np.zeros((n, 1), f32), 1.0)
    z0 = np.zeros((n, 4, 1), f32)
    z0[:, 0, 0] = z0[:, 2, 0] = np.exp(-(x / 0.3) ** 2)
    z = lo.integrate(dyn, z0, t, [C, F, np.zeros((n, 1), f32)], 1e-5)
    assert z.shape == (n, 4, 1, steps + 1)
    assert np.abs(z[:, 0] - z[:, 2]).max() < 1e-4
    peak = x[np.argmax(z[n // 2:, 0, 0, -1]) + n // 2]
    assert abs(peak - 1531.0 * 300e-5) < 0.1                                 # the right-going half sits at c0 * t
    assert abs(z[:, 0, 0, -1].max() - 0.5) < 0.01                            # ... with half the amplitude


def _case(rng, n, B, K, steps):
    x = wo.OneDim.from_size(15.0, n).x
    t0 = rng.uniform(0, 1e-3, B).astype(f32)
    t = np.stack([wo.build_tspan(t0[b], 1e-5, steps) for b in range(B)], axis=1)
    X = np.stack([np.linspace(t[0, b], t[-1, b], K).astype(f32) for b in range(B)], axis=1)
    X[-1, :] = t[-1, :]
    Y = (1.0 + 0.3 * rng.standard_normal((n, K, B))).astype(f32)
    shape = (0.1 * rng.standard_normal((n, B))).astype(f32)
    PML = rng.uniform(0, 1, (n, B)).astype(f32) ** 2
    z0 = (0.1 * rng.standard_normal((n, 4, B))).astype(f32)
    return x, t, X, Y, shape, PML, z0


@pytest.mark.gpu
@pytest.mark.parametrize("n,B,K,steps", [(1024, 3, 2, 60), (257, 2, 4, 33), (64, 5, 3, 20), (3, 1, 2, 4)])
def test_device_latent_integrator_bit_exact(n, B, K, steps):
    rng = np.random.default_rng(n + B)
    x, t, X, Y, shape, PML, z0 = _case(rng, n, B, K, steps)
    dyn = lo.LatentDynamics(x, 1531.0, 5.0, 10000.0)
    ref = lo.integrate(dyn, z0, t, [lo.LinearInterpolation(X, Y), lo.Source1D(shape, 1000.0), PML], 1e-5)
    dim = w.OneDim(15.0, n)
    assert np.array_equal(dim.x, x)
    it = w.LatentIntegrator(dim, 1531.0, 5.0, 10000.0, 1e-5)
    got = it(z0, t, [w.LinearInterpolation(X, Y), w.LatentSource(shape, 1000.0), PML])
    assert got.shape == ref.shape == (n, 4, B, steps + 1)
    assert np.array_equal(got, ref, equal_nan=True)
    assert np.array_equal(w.compute_latent_energy(got, wo.get_dx(wo.OneDim(x))), lo.compute_latent_energy(ref, wo.get_dx(wo.OneDim(x))))


@pytest.mark.gpu
def test_device_latent_integrator_reference_script_setup():
    """scripts/adjoint_sensitivity.jl:9-30 (forward part): OneDim(15, 1024), dt 1e-5, N = 300, C = ones interpolated over
    [t0, tN], F = Source(zeros, 1), PML = dyn.pml / maximum(dyn.pml)."""
    n, steps = 1024, 300
    x = wo.OneDim.from_size(15.0, n).x
    dyn = lo.LatentDynamics(x, wo.WATER, 5.0, 10000.0)
    t = wo.build_tspan(0.0, 1e-5, steps)[:, None]
    X, Y = t[[0, -1], :], np.ones((n, 2, 1), f32)
    PML = (dyn.pml / dyn.pml.max())[:, None].astype(f32)
    z0 = np.zeros((n, 4, 1), f32)
    z0[:, 0, 0] = z0[:, 2, 0] = np.exp(-(x / 0.3) ** 2)
    ref = lo.integrate(dyn, z0, t, [lo.LinearInterpolation(X, Y), lo.Source1D(np.zeros((n, 1), f32), 1.0), PML], 1e-5)
    got = w.LatentIntegrator(w.OneDim(15.0, n), wo.WATER, 5.0, 10000.0, 1e-5)(
        z0, t, [w.LinearInterpolation(X, Y), w.LatentSource(np.zeros((n, 1), f32), 1.0), PML])
    assert np.array_equal(got, ref)
    assert np.abs(got[:, 0, 0, -1]).max() > 0.3
