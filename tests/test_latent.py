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


# ---- adjoint_sensitivity (src/dynamics.jl:97-128) ---------------------------------------------------------------------
def _small_case(T, seed=3, n=16, B=2, K=3):
    rng = np.random.default_rng(seed)
    x = np.linspace(-5, 5, n)
    dyn = lo.LatentDynamics(x, 1.3, 1.0, 5.0, T=T)
    X = np.stack([np.linspace(0, 0.06, K)] * B, axis=1).astype(T)
    Y = (1 + 0.3 * rng.standard_normal((n, K, B))).astype(T)
    sh = rng.standard_normal((n, B)).astype(T)
    PML = rng.random((n, B)).astype(T)
    u = rng.standard_normal((n, 4, B)).astype(T)
    lam = rng.standard_normal((n, 4, B)).astype(T)
    return rng, dyn, X, Y, sh, PML, u, lam


def test_step_pullback_against_finite_differences_fp64():
    """The hand-written pullback of one runge_kutta call (what the reference takes from Zygote), in the oracle's fp64 twin,
    against central differences of the forward oracle: EVERY entry of z, C.Y, F.shape and PML (edge cells included), at
    stage times that fall into two different knot intervals."""
    T = np.float64
    rng, dyn, X, Y, sh, PML, u, lam = _small_case(T)
    t = np.array([0.013, 0.027])        # t + dt/2 crosses the knot at 0.03 for the second batch element
    dt = 1e-2

    def L(u, Y, sh, PML):
        th = [lo.LinearInterpolation(X, Y, T), lo.Source1D(sh, 3.0, T), PML]
        return float(np.sum(lam * lo.runge_kutta(dyn, u, t, th, dt)))

    th = [lo.LinearInterpolation(X, Y, T), lo.Source1D(sh, 3.0, T), PML]
    grads = lo._vjp_runge_kutta(dyn, u, t, th, dt, lam, T)
    args = [u, Y, sh, PML]
    for which, G in enumerate(grads):
        scale = np.abs(G).max()
        for idx in np.ndindex(*G.shape):
            a = [v.copy() for v in args]
            a[which][idx] += 1e-6
            p = L(*a)
            a[which][idx] -= 2e-6
            m = L(*a)
            assert abs((p - m) / 2e-6 - G[idx]) <= 1e-6 * scale + 1e-9, (which, idx)


def test_adjoint_sweep_is_the_reference_loop():
    """The sweep as written at src/dynamics.jl:101-116 (fp64 twin): mu_i = (I + J_i')(a_i + mu_{i+1}) over ALL saved times
    -- checked against explicit Jacobians built column by column from the step pullback."""
    T = np.float64
    rng, dyn, X, Y, sh, PML, u, _ = _small_case(T, seed=5, n=8, B=1)
    dt, steps = 1e-2, 3
    t = (np.arange(steps + 1) * dt)[:, None]
    th = [lo.LinearInterpolation(X, Y, T), lo.Source1D(sh, 3.0, T), PML]
    z = [u]
    for i in range(steps):
        z.append(z[-1] + lo.runge_kutta(dyn, z[-1], t[i], th, dt))
    z = np.stack(z, axis=3)
    adj = rng.standard_normal(z.shape)
    got = lo.adjoint_sensitivity(dyn, z, t, th, adj, dt, T)[0]
    mu = np.zeros_like(u)
    for i in reversed(range(steps + 1)):
        v = adj[..., i] + mu
        mu = v + lo._vjp_runge_kutta(dyn, z[..., i], t[i], th, dt, v, T)[0]
    assert np.allclose(got, mu, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n,B,K,steps", [(1024, 2, 2, 40), (257, 3, 4, 25), (64, 5, 3, 20), (3, 1, 2, 4)])
def test_device_adjoint_against_the_restatement(n, B, K, steps):
    """wv_latent_adjoint against oracle/latent_oracle.adjoint_sensitivity (fp32, same formulas; the order in which a
    cell's contributions are added differs in two places, hence a tolerance: 2e-5 of the largest entry)."""
    rng = np.random.default_rng(7 * n + B)
    x, t, X, Y, shape, PML, z0 = _case(rng, n, B, K, steps)
    dyn = lo.LatentDynamics(x, 1531.0, 5.0, 10000.0)
    theta_o = [lo.LinearInterpolation(X, Y), lo.Source1D(shape, 1000.0), PML]
    it = w.LatentIntegrator(w.OneDim(15.0, n), 1531.0, 5.0, 10000.0, 1e-5)
    theta = [w.LinearInterpolation(X, Y), w.LatentSource(shape, 1000.0), PML]
    z, back = it.rrule(z0, t, theta)
    assert np.array_equal(z, lo.integrate(dyn, z0, t, theta_o, 1e-5), equal_nan=True)
    adj = rng.standard_normal(z.shape).astype(f32)
    gz0, g = back(adj)
    rz0, rY, rsh, rp = lo.adjoint_sensitivity(dyn, z, t, theta_o, adj, 1e-5)
    for name, got, ref in (("z0", gz0, rz0), ("Y", g["Y"], rY), ("shape", g["shape"], rsh), ("PML", g["PML"], rp)):
        assert got.shape == ref.shape, name
        assert np.isfinite(ref).all(), name
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), (name, np.abs(got - ref).max(), np.abs(ref).max())


@pytest.mark.gpu
def test_device_adjoint_reference_script_setup():
    """scripts/adjoint_sensitivity.jl:9-45: OneDim(15, 1024), dt 1e-5, N = 300, C = ones, F = Source(zeros, 1),
    PML = dyn.pml / maximum(dyn.pml), loss = mse(z[:, 1, 1, end], target): dL/dz is non-zero at the last saved time only.
    The device gradient with respect to z0 equals the restatement's, and a step along it lowers the loss (what the
    script's optimisation loop relies on)."""
    n, steps = 1024, 300
    x = wo.OneDim.from_size(15.0, n).x
    dyn = lo.LatentDynamics(x, wo.WATER, 5.0, 10000.0)
    t = wo.build_tspan(0.0, 1e-5, steps)[:, None]
    X, Y = t[[0, -1], :], np.ones((n, 2, 1), f32)
    shape = np.zeros((n, 1), f32)
    PML = (dyn.pml / dyn.pml.max())[:, None].astype(f32)
    target = (f32(1.0) / (f32(0.3) * np.sqrt(f32(2.0) * f32(np.pi))) * np.exp(-(x ** 2) / (f32(2.0) * f32(0.3) ** 2))).astype(f32)  # build_normal(x, [0], [0.3], [1]), src/utils.jl:4-10
    rng = np.random.default_rng(11)
    z0 = (0.05 * rng.standard_normal((n, 4, 1))).astype(f32)
    z0[:, 0, 0] += np.exp(-((x + 2.0) / 0.5) ** 2).astype(f32)
    it = w.LatentIntegrator(w.OneDim(15.0, n), wo.WATER, 5.0, 10000.0, 1e-5)
    theta = [w.LinearInterpolation(X, Y), w.LatentSource(shape, 1.0), PML]

    def loss_and_adj(z):
        r = z[:, 0, 0, -1] - target
        adj = np.zeros_like(z)
        adj[:, 0, 0, -1] = (f32(2.0) / f32(n)) * r          # d mse / d z[:, 1, 1, end]
        return float(np.mean(r.astype(np.float64) ** 2)), adj

    z, back = it.rrule(z0, t, theta)
    l0, adj = loss_and_adj(z)
    gz0, g = back(adj)
    ref = lo.adjoint_sensitivity(dyn, z, t, [lo.LinearInterpolation(X, Y), lo.Source1D(shape, 1.0), PML], adj, 1e-5)
    assert np.abs(gz0 - ref[0]).max() <= 2e-5 * np.abs(ref[0]).max()
    assert np.abs(g["PML"] - ref[3]).max() <= 2e-5 * np.abs(ref[3]).max()
    step = f32(0.2 * l0 / float(np.sum(gz0.astype(np.float64) ** 2)))
    l1, _ = loss_and_adj(it(z0 - step * gz0, t, theta))
    assert l1 < l0 * 0.9, (l0, l1)
