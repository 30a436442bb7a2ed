"""Pins the oracle against the ONLY automated test the reference holds for this path (test/operators.jl:4-30), and adds
the known-answer tests SURVEY 8c lists for the parts the reference leaves unpinned.  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

import c_oracle as co
import waves_oracle as wo

f32 = np.float32


# ---- the reference's own test, verbatim in structure: test/operators.jl:4-30 ---------------------------------
# NOTE on the third @test (y = exp(x) on [-25, 25]): a second-order stencil has truncation error ~ dx^2/6 * e^x, which
# is ~3e7 at x = 25 -- `all(abs.(e) .< dx)` cannot hold there for ANY correct implementation of operators.jl:10-22 (the
# reference has no test runner or CI, so the assertion was evidently never executed).  It is reproduced as written on
# the part of the domain where the truncation bound is below dx, and as a relative bound on the whole domain.
@pytest.mark.parametrize("name,fn,dfn", [("x2", lambda x: x * x, lambda x: f32(2.0) * x), ("sin", np.sin, np.cos),
                                         ("exp", np.exp, np.exp)])
def test_reference_gradient_testset(name, fn, dfn):
    dim = wo.OneDim.from_size(25.0, 1024)
    dx = wo.get_dx(dim)
    grad = wo.build_gradient(dim.x)
    y = fn(dim.x).astype(f32)
    dydx_numerical = wo.dx(grad, y)
    dydx_true = dfn(dim.x).astype(f32)
    e = dydx_numerical - dydx_true
    if name != "exp":
        assert np.all(np.abs(e) < dx)
    else:
        ok = dim.x < np.log(3.0 / dx) - 1.0
        assert ok.sum() > 500 and np.all(np.abs(e[ok]) < dx)
        assert np.all(np.abs(e) / dydx_true < dx)
        assert not np.all(np.abs(e) < dx)   # the assertion as written is unsatisfiable (see NOTE)


@pytest.mark.parametrize("axis", [0, 1])
def test_reference_gradient_testset_2d_both_axes(axis):
    """the same assertion carried to both axes of the 2-D stencil, edge rows included (SURVEY 8c KAT 1)"""
    n = 1024
    dim = wo.TwoDim.from_size(25.0, n)
    g = wo.build_gradient(dim.x)
    dx = wo.get_dx(dim)
    X, Y = np.meshgrid(dim.x, dim.y, indexing="ij")
    coord = X if axis == 0 else Y
    for fn, dfn, rel in ((lambda x: x * x, lambda x: f32(2.0) * x, False), (np.sin, np.cos, False), (np.exp, np.exp, True)):
        u = fn(coord).astype(f32)
        d = wo.dx(g, u) if axis == 0 else wo.dy(g, u)
        true = dfn(coord).astype(f32)
        err = np.abs(d - true) / (true if rel else f32(1))
        assert np.all(err < dx)
        dc = co.gradient(dim.x, axis, wo.to_abi(u))
        assert np.array_equal(wo.from_abi(dc), d)


def test_stencil_is_the_reference_sparse_matrix():
    """gradient_dense follows operators.jl:10-22 literally; the stencil form must give what `grad * u` gives with
    scipy's CSC kernel (same ascending-column accumulation as SparseArrays)."""
    n = 97
    x = wo.julia_range_f32(-3.0, 3.0, n)
    dense = wo.gradient_dense(x)
    A = sp.csc_matrix(dense)
    assert A.nnz == 2 * n + 2
    rng = np.random.default_rng(1)
    u = rng.standard_normal((n, 5)).astype(f32)
    ref = np.asarray(A @ u, dtype=f32)
    assert (A @ u).dtype == np.float32
    got = wo.dx(wo.build_gradient(x), u)
    assert np.array_equal(ref, got)
    # rows are exactly the stencils
    g = wo.build_gradient(x)
    assert dense[0, 0] == g.fwd[0] and dense[0, 1] == g.fwd[1] and dense[0, 2] == g.fwd[2]
    assert dense[n - 1, n - 3] == g.bwd[0] and dense[n - 1, n - 2] == g.bwd[1] and dense[n - 1, n - 1] == g.bwd[2]
    assert dense[5, 4] == g.cm and dense[5, 6] == g.cp and g.cm == -g.cp


def test_gradient_second_order_convergence():
    errs = []
    for n in (65, 129, 257):
        dim = wo.TwoDim.from_size(1.0, n)
        g = wo.build_gradient(dim.x, np.float64)
        X, Y = np.meshgrid(dim.x.astype(np.float64), dim.y.astype(np.float64), indexing="ij")
        u = np.sin(2 * X) * np.cos(3 * Y)
        ex = np.abs(wo.dx(g, u) - 2 * np.cos(2 * X) * np.cos(3 * Y)).max()
        ey = np.abs(wo.dy(g, u) + 3 * np.sin(2 * X) * np.sin(3 * Y)).max()
        errs.append(max(ex, ey))
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5


def test_julia_range_is_exactly_rounded_and_hits_endpoints():
    for n in (256, 700, 2048):
        x = wo.julia_range_f32(-15.0, 15.0, n)
        assert x[0] == f32(-15) and x[-1] == f32(15) and x.dtype == np.float32
        ref = (-15.0 + 30.0 * np.arange(n) / (n - 1)).astype(f32)   # float64 evaluation, rounded once
        assert np.array_equal(x, ref)
        assert np.array_equal(x, -x[::-1])
    t = wo.build_tspan(0.0, 1e-5, 100)
    assert t[0] == 0 and t[-1] == f32(f32(100) * f32(1e-5)) and len(t) == 101


def test_pml_profile_hand_values():
    """src/pml.jl:21-29 at hand-computed cells (SURVEY 8c KAT 5)"""
    dim = wo.TwoDim.from_size(15.0, 256)
    p = wo.build_pml_profile(dim.x, 2.0, 20000.0)
    ax = np.abs(dim.x)
    inside = ax > f32(15.0 - 2.0)
    assert np.all(p[~inside] == 0) and np.all(p[inside][1:-1] >= 0)
    m = ax[inside].min()
    i = 3  # a cell inside the left layer
    v = f32(f32(ax[i] - m) / f32(2.0))
    assert p[i] == f32(f32(f32(v * v) * v) * f32(20000.0))
    assert p[0] == p[-1] and p[0] < 20000.0 and np.array_equal(p, p[::-1])
    assert np.array_equal(p, co.pml_profile(dim.x, 2.0, 20000.0))
    field = wo.build_pml(dim, 2.0, 20000.0)
    assert field.shape == (256, 256) and np.array_equal(field[:, 7], p)


def test_speed_matches_bruteforce_and_only_takes_design_values():
    """SURVEY 8c KAT 6"""
    rng = np.random.default_rng(3)
    dim = wo.TwoDim.from_size(15.0, 128)
    grid = wo.build_grid(dim)
    ds = wo.build_triple_ring_design_space()
    d = wo.rand_design(ds, rng)
    c = wo.speed(d, grid, wo.WATER)
    assert set(np.unique(c)) <= {f32(1531.0), f32(1032.0)}
    cyl = wo.stacked_cylinders(d)
    brute = np.full((128, 128), f32(1531.0), dtype=f32)
    for i in range(128):
        for j in range(128):
            for m in range(len(cyl)):
                ddx = f32(dim.x[i] - cyl.pos[m, 0]); ddy = f32(dim.y[j] - cyl.pos[m, 1])
                if f32(f32(ddx * ddx) + f32(ddy * ddy)) < f32(cyl.r[m] * cyl.r[m]):
                    brute[i, j] = cyl.c[m]
    assert np.array_equal(c, brute)
    assert (c == f32(1032.0)).sum() > 50
    cc = co.speed_field(dim.x, dim.y, np.concatenate([cyl.pos, cyl.r[:, None], cyl.c[:, None]], 1), wo.WATER)
    assert np.array_equal(wo.from_abi(cc), c)
    # overlapping cylinders ADD their speeds (designs.jl:114) and switch the ambient term off
    two = wo.Cylinders(np.array([[0, 0], [0.2, 0]], f32), np.array([1.0, 1.0], f32), np.array([100.0, 30.0], f32))
    c2 = wo.speed(two, grid, wo.WATER)
    assert f32(130.0) in np.unique(c2)


def test_build_normal_integrates_to_amplitude():
    """SURVEY 8c KAT 7"""
    dim = wo.TwoDim.from_size(15.0, 700)
    g = wo.build_normal(wo.build_grid(dim), np.array([[-10.0, 3.0]]), np.array([0.3]), np.array([1.0]))
    integral = float(g.astype(np.float64).sum()) * float(wo.get_dx(dim)) * float(wo.get_dy(dim))
    assert abs(integral - 1.0) < 1e-4
    assert g.dtype == np.float32 and g.max() < 1.0 / (2 * np.pi * 0.09) * 1.0001


def test_design_interpolator_order_and_endpoints():
    rng = np.random.default_rng(5)
    ds = wo.build_triple_ring_design_space()
    a = wo.rand_design(ds, rng)
    act = wo.rand_design(wo.build_action_space(a, 0.25), rng)
    b = ds(a, act)
    assert np.all(b.config.cylinders.r >= f32(0.2)) and np.all(b.config.cylinders.r <= f32(1.0))
    assert np.array_equal(b.config.cylinders.pos, a.config.cylinders.pos)
    it = wo.DesignInterpolator(a, b, f32(0.001), f32(0.002))
    assert np.array_equal(it(f32(0.0005)).stacked().r, a.stacked().r)       # clamped below ti
    mid = it(f32(0.0015)).stacked()
    ra, rb = a.stacked().r, b.stacked().r
    inv = f32(1.0) / f32(f32(0.002) - f32(0.001))
    tau = f32(f32(0.0015) - f32(0.001))
    assert np.array_equal(mid.r, ra + ((rb + ra * f32(-1)) * inv) * tau)
    flat = lambda d: np.concatenate([d.stacked().pos, d.stacked().r[:, None], d.stacked().c[:, None]], 1)
    got = co.design_at(flat(a), flat(b), f32(0.001), f32(0.002), f32(0.0015))
    assert np.array_equal(got, flat(it(f32(0.0015))))


def test_no_design_invariance_and_energy_bookkeeping():
    """SURVEY 8c KATs 3 and 8 on the numpy oracle: scalar c keeps total == incident bit for bit; env bookkeeping."""
    rng = np.random.default_rng(0)
    dim = wo.TwoDim.from_size(15.0, 64)
    grid = wo.build_grid(dim)
    src = wo.RandomPosGaussianSource(grid, np.array([[-10., -10.]], f32), np.array([[-10., 10.]], f32),
                                     np.array([0.3], f32), np.array([1.0], f32), f32(1000.0))
    src.reset(rng)
    env = wo.WaveEnv(dim, design_space=wo.build_triple_ring_design_space(), source=src, integration_steps=20, actions=2,
                     rng=rng, resolution=(32, 32))
    pol = wo.RandomDesignPolicy(env.action_space(), rng)
    assert not env.is_terminated()
    env(pol(env))
    first = env.signal.copy()
    assert first.shape == (21, 3) and env.time_step == 20 and env.wave.shape == (64, 64, 12, 3)
    last_state = env.wave[:, :, :, 2].copy()
    env(pol(env))
    assert env.is_terminated() and env.time_step == 40
    # row 1 of a step equals the last row of the previous step
    assert np.array_equal(env.signal[0], first[-1])
    # frames are the states at steps n-20, n-10, n: frame 0 of the second action is the state it started from
    assert np.array_equal(env.wave[:, :, :, 0], last_state)
    # no design: total and incident sets stay identical, scattered energy is exactly 0
    it = wo.Integrator(wo.runge_kutta, wo.AcousticDynamics.build(dim, wo.WATER, 2.0, 20000.0), f32(1e-5))
    sig = []
    sol = it(wo.build_wave(dim, 12), wo.build_tspan(0.0, 1e-5, 30), [lambda t: wo.WATER, src], save={30},
             on_state=lambda i, u: sig.append(wo.energies(u[:, :, 0], u[:, :, 6], f32(1))))
    u = sol[..., 0]
    assert np.abs(u[:, :, 0]).max() > 0 and np.array_equal(u[:, :, :6], u[:, :, 6:])
    assert all(s[2] == 0 for s in sig)
    with pytest.raises(IndexError):
        wo.WaveEnv(dim, design_space=wo.build_triple_ring_design_space(), source=src, integration_steps=10, rng=rng,
                   resolution=(32, 32))(pol(env))


def test_fp32_oracle_vs_fp64_twin_sets_the_tolerance():
    """SURVEY 8c KAT 9: the fp32 restatement against the same equations in fp64 -- this is the round-off floor any
    fp32 implementation with a different (legal) operation order would sit at, and it bounds the tolerance quoted in
    the GPU tests (which are in fact bit-exact)."""
    dim = wo.TwoDim.from_size(15.0, 96)
    grid = wo.build_grid(dim)
    G = wo.build_normal(grid, np.array([[-2.0, 0.5]]), np.array([0.5]), np.array([1.0]))
    res = {}
    for T in (np.float32, np.float64):
        it = wo.Integrator(wo.runge_kutta, wo.AcousticDynamics.build(dim, wo.WATER, 2.0, 20000.0, T), f32(1e-5))
        src = wo.Source(G, f32(1000.0))
        cyl = wo.Cylinders(np.array([[1.0, 0.0]], f32), np.array([1.2], f32), np.array([1032.0], f32))
        gridT = grid.astype(T)
        C = lambda t: wo.speed(cyl, gridT, T(wo.WATER))
        F = lambda t: src(t, T)
        res[T] = it(wo.build_wave(dim, 12, T), wo.build_tspan(0.0, 1e-5, 60), [C, F], save={60})[..., 0]
    scale = np.abs(res[np.float64][:, :, 0]).max()
    err = np.abs(res[np.float32].astype(np.float64) - res[np.float64])[:, :, [0, 6]].max() / scale
    assert scale > 1e-3 and err < 2e-5, err


def test_rk4_fourth_order_in_dt():
    """SURVEY 8c KAT 4 on the fp64 twin: halving dt divides the time-discretisation error by ~16."""
    dim = wo.TwoDim.from_size(4.0, 48)
    grid = wo.build_grid(dim, np.float64)
    u0 = wo.build_wave(dim, 12, np.float64)
    pulse = np.exp(-((grid[:, :, 0]) ** 2 + (grid[:, :, 1]) ** 2) / 0.5)
    u0[:, :, 0] = pulse
    u0[:, :, 6] = pulse

    def run(nsteps):
        dt = 4e-4 / nsteps
        dyn = wo.AcousticDynamics.build(dim, 1531.0, 1.0, 0.0, np.float64)
        it = wo.Integrator(wo.runge_kutta, dyn, np.float64(dt))
        ts = np.linspace(0.0, 4e-4, nsteps + 1)
        return it(u0, ts, [lambda t: np.float64(1531.0), lambda t: np.float64(0.0)], save={nsteps})[..., 0]

    ref = run(64)
    e1 = np.abs(run(4) - ref).max()
    e2 = np.abs(run(8) - ref).max()
    assert 10 < e1 / e2 < 22, e1 / e2


def test_pml_absorbs_a_pulse():
    """SURVEY 8c KAT 5 (scripts/pml.jl:10-13 pulse): with the PML on, far less energy is left after the pulse has hit
    the boundary than with pml_scale = 0."""
    dim = wo.TwoDim.from_size(5.0, 96)
    grid = wo.build_grid(dim)
    ic = wo.build_normal(grid, np.array([[0.0, 0.0]]), np.array([0.3]), np.array([1.0]))
    left = {}
    for scale in (0.0, 20000.0):
        st = np.zeros((12, 96, 96), f32)
        st[0] = wo.to_abi(ic)
        st[6] = wo.to_abi(ic)
        sx = wo.build_pml_profile(dim.x, 1.0, scale)
        ts = wo.build_tspan(0.0, 1e-5, 600)
        _, es, _ = co.integrate(dim.x, dim.y, sx, sx, wo.WATER, 1e-5, st, ts, nthreads=4)
        left[scale] = es[-1, 0] / es[0, 0]
    assert left[20000.0] < 0.05 * left[0.0], left


# ---- the observation path state(env) = imresize(cat(u_tot frames, source shape), resolution) ------
def test_imresize_linear_restated_rule():
    """imresize is Images.jl's (third-party, unpinned): PARITY UNPINNED.  What is pinned here is the rule as restated in
    waves_oracle.imresize_linear: pixel-centre aligned linear interpolation, Float64 evaluation rounded once."""
    f32 = np.float32
    rng = np.random.default_rng(3)
    a = rng.standard_normal((40, 30, 4)).astype(f32)
    assert np.array_equal(wo.imresize_linear(a, (40, 30)), a)                     # same size: identity
    c = np.full((50, 70, 2), 3.25, f32)
    assert np.array_equal(wo.imresize_linear(c, (16, 9)), np.full((16, 9, 2), 3.25, f32))   # constants survive
    # an affine field is reproduced exactly at the pixel-centre aligned sample positions
    i, j = np.meshgrid(np.arange(1, 65), np.arange(1, 49), indexing="ij")
    lin = (0.5 * i - 0.25 * j + 2.0).astype(f32)                                     # exactly representable
    r = wo.imresize_linear(lin, (16, 12))
    xo = (64 / 16) * (np.arange(1, 17) - 0.5) + 0.5
    yo = (48 / 12) * (np.arange(1, 13) - 0.5) + 0.5
    want = (0.5 * xo[:, None] - 0.25 * yo[None, :] + 2.0).astype(f32)
    assert np.array_equal(r, want)
    # 2:1 in both axes: the sample point is the centre of a 2x2 block -> its mean
    b = rng.standard_normal((8, 6)).astype(f32)
    m = wo.imresize_linear(b, (4, 3))
    blk = b.astype(np.float64).reshape(4, 2, 3, 2)
    want = (0.5 * (0.5 * blk[:, 0, :, 0] + 0.5 * blk[:, 1, :, 0]) + 0.5 * (0.5 * blk[:, 0, :, 1] + 0.5 * blk[:, 1, :, 1]))
    assert np.array_equal(m, want.astype(f32))
    # the env wires it up: (128, 128, 4) from three U_tot frames and the source shape
    dim = wo.TwoDim.from_size(15.0, 160)
    src = wo.RandomPosGaussianSource(wo.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0)
    env = wo.WaveEnv(dim, design_space=wo.build_triple_ring_design_space(), source=src, integration_steps=20, actions=2,
                     rng=np.random.default_rng(0))
    env.reset()
    x = env.state()
    assert x.shape == (128, 128, 4) and x.dtype == f32
    assert np.array_equal(x[:, :, :3], np.zeros((128, 128, 3), f32))
    assert np.array_equal(x[:, :, 3], wo.imresize_linear(src.shape, (128, 128))) and x[:, :, 3].max() > 0
