"""Parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle on the same seeded
inputs.  Fields, RHS, gradient, wave-speed mask and PML profile are compared BIT-EXACTLY (np.array_equal; +0 == -0):
the kernels keep the reference's fp32 operation order and are built with -ffp-contract=off.  The stated fp32
tolerance of the north star therefore only applies to the two reductions/transcendentals whose order/rounding the
reference itself leaves open:  energy traces rel. 1e-5 (sum order), Gaussian source shape 2 ulp (exp)."""
import numpy as np
import pytest

import c_oracle as co
import waves_jl_amd as w
import waves_oracle as wo
from helpers import (flat_design, oracle_integrate, oracle_to_mirror_design, random_state, rel_err,
                     small_moving_design)

pytestmark = pytest.mark.gpu
f32 = np.float32
IMPLS = ["staged", "fused"]
ENERGY_RTOL = 1e-5


def make_ctx(n, impl="auto", pml=(2.0, 20000.0), size=15.0, dt=1e-5, c0=wo.WATER):
    dim = wo.TwoDim.from_size(size, n)
    ctx = w._ffi.Context(dim.x, dim.y, c0=c0, dt=dt, pml_width=pml[0], pml_scale=pml[1], device=0, impl=impl)
    return dim, ctx


def set_design(ctx, d0, d1, ti, tf):
    a, b = wo.stacked_cylinders(d0), wo.stacked_cylinders(d1)
    ctx.set_design((a.pos, a.r, a.c), (b.pos, b.r, b.c), ti, tf)


def test_device_is_gfx950_and_library_loaded():
    assert w.device_count() >= 1
    dim, ctx = make_ctx(64)
    assert ctx.cell_area() == f32(wo.get_dx(dim) * wo.get_dy(dim))
    ctx.close()


@pytest.mark.parametrize("axis", [0, 1])
def test_reference_gradient_testset_on_device(axis):
    """test/operators.jl:4-30 carried to both axes of the device stencil (see the NOTE in test_oracle_reference_kat)"""
    n = 1024
    dim, ctx = make_ctx(n, size=25.0)
    dx = wo.get_dx(dim)
    X, Y = np.meshgrid(dim.x, dim.y, indexing="ij")
    coord = X if axis == 0 else Y
    g = wo.build_gradient(dim.x)
    for fn, dfn, rel in ((lambda x: x * x, lambda x: f32(2.0) * x, False), (np.sin, np.cos, False), (np.exp, np.exp, True)):
        u = fn(coord).astype(f32)
        d = ctx.gradient(axis, u)
        true = dfn(coord).astype(f32)
        assert np.all(np.abs(d - true) / (true if rel else f32(1)) < dx)
        assert np.array_equal(d, wo.dx(g, u) if axis == 0 else wo.dy(g, u))   # bit-exact vs the oracle
    ctx.close()


def test_pml_profile_bit_exact():
    for n, width in ((256, 2.0), (700, 2.0), (300, 4.0), (128, 1.0)):
        dim, ctx = make_ctx(n, pml=(width, 20000.0))
        sx, sy = ctx.pml()
        ref = wo.build_pml_profile(dim.x, width, 20000.0)
        assert np.array_equal(sx, ref) and np.array_equal(sy, ref)
        ctx.close()


def test_speed_field_bit_exact_moving_design():
    rng = np.random.default_rng(11)
    dim, ctx = make_ctx(700)
    grid = wo.build_grid(dim)
    ds = wo.build_triple_ring_design_space()
    a = wo.rand_design(ds, rng)
    b = ds(a, wo.rand_design(wo.build_action_space(a, 0.25), rng))
    it = wo.DesignInterpolator(a, b, f32(0.003), f32(0.004))
    set_design(ctx, a, b, it.ti, it.tf)
    flips = 0
    prev = None
    for t in (0.0029, 0.003, 0.003005, 0.00333, 0.0037775, 0.004, 0.0041):
        c = ctx.speed_field(t)
        ref = wo.speed(it(f32(t)), grid, wo.WATER)
        assert np.array_equal(c, ref)
        if prev is not None:
            flips += int((c != prev).sum())
        prev = c
    assert flips > 100   # the radii really moved cells across the mask
    # overlapping cylinders add (designs.jl:114), position-moving designs
    d0, d1 = small_moving_design(rng, 5, spread=1.0)
    set_design(ctx, d0, d1, 0.0, 1e-3)
    it2 = wo.DesignInterpolator(d0, d1, f32(0.0), f32(1e-3))
    for t in (0.0, 4.5e-4, 1e-3):
        assert np.array_equal(ctx.speed_field(t), wo.speed(it2(f32(t)), grid, wo.WATER))
    ctx.set_design(None, None, 0.0, 0.0)
    assert np.all(ctx.speed_field(0.0) == wo.WATER)
    ctx.close()


def test_gaussian_source_shape_and_time_factor():
    dim, ctx = make_ctx(700)
    grid = wo.build_grid(dim)
    mu = np.array([[-10.0, 3.3], [2.0, -1.0]], f32)
    sig = np.array([0.3, 0.7], f32)
    a = np.array([1.0, 0.5], f32)
    ctx.set_gaussian_source(mu, sig, a, 1000.0)
    G = ctx.source_shape()
    ref = wo.build_normal(grid, mu, sig, a)
    ulp = np.spacing(np.abs(ref).astype(f32)) + np.finfo(f32).tiny
    assert np.all(np.abs(G - ref) <= 2 * ulp * 2)      # exp rounding: <= 2 ulp per term
    assert rel_err(G, ref) < 3e-7
    for t in (0.0, 1e-5, 0.00123, 0.0199):
        s = wo.source_time_factor(f32(t), f32(1000.0))
        assert np.array_equal(ctx.source_field(t), G * s)
        assert s == co.source_factor(t, 1000.0)
    ctx.close()


@pytest.mark.parametrize("n", [64, 257])
def test_rhs_bit_exact(n):
    """dyn(x, t, theta), src/dynamics.jl:179-188, on a random state with every field populated"""
    rng = np.random.default_rng(n)
    dim, ctx = make_ctx(n)
    grid = wo.build_grid(dim)
    d0, d1 = small_moving_design(rng, 4)
    G = wo.build_normal(grid, np.array([[0.5, -0.5]]), np.array([0.8]), np.array([2.0]))
    ctx.set_source_shape(G, 1000.0)
    set_design(ctx, d0, d1, 0.0, 1e-3)
    x = random_state(rng, n, n)
    t = f32(3.3e-4)
    k = ctx.rhs(x, t)
    dyn = wo.AcousticDynamics.build(dim, wo.WATER, 2.0, 20000.0)
    it = wo.DesignInterpolator(d0, d1, f32(0.0), f32(1e-3))
    src = wo.Source(G, f32(1000.0))
    ref = dyn(x, t, [lambda tt: wo.speed(it(tt), grid, wo.WATER), lambda tt: src(tt)])
    assert np.array_equal(k, ref)
    ctx.close()


@pytest.mark.parametrize("impl", IMPLS)
def test_config1_256_no_design_100_steps(impl):
    """BASELINE config 1: TwoDim(15, 256), single Gaussian source at (-10, 0), no design, 100 steps"""
    dim, ctx = make_ctx(256, impl)
    grid = wo.build_grid(dim)
    G = wo.build_normal(grid, np.array([[-10.0, 0.0]]), np.array([0.3]), np.array([1.0]))
    ctx.set_source_shape(G, 1000.0)
    ctx.set_design(None, None, 0.0, 0.0)
    ts = wo.build_tspan(0.0, 1e-5, 100)
    sig, ut, ui = ctx.integrate(ts, capture_frames=True, want_signal=True, want_fields=True)
    st, rsig, fr = oracle_integrate(dim, np.zeros((12, 256, 256), f32), ts, G=G, freq=1000.0, frame_steps=(80, 90, 100))
    frames = ctx.get_frames()
    for k in range(3):
        assert np.array_equal(wo.to_abi(frames[:, :, :, k]), fr[k])
    assert np.abs(st[0]).max() > 0.5
    assert np.array_equal(frames[:, :, :6, 2], frames[:, :, 6:, 2])      # no design: total == incident
    assert np.all(sig[:, 2] == 0)
    assert rel_err(sig[:, :2], rsig[:, :2]) < ENERGY_RTOL
    assert np.array_equal(ut[:, :, 100], frames[:, :, 0, 2]) and np.array_equal(ui[:, :, 90], frames[:, :, 6, 1])
    assert np.all(ut[:, :, 0] == 0)
    t = ctx.timing()
    assert t["impl"] == impl and t["steps"] == 100 and t["total_ms"] > 0
    ctx.close()


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("n,steps", [(96, 40), (200, 30)])
def test_moving_design_from_random_state(impl, n, steps):
    """moving radii + positions, source on, every one of the 12 fields (PML auxiliaries included) non-zero at t0"""
    rng = np.random.default_rng(n + steps)
    dim, ctx = make_ctx(n, impl)
    grid = wo.build_grid(dim)
    d0, d1 = small_moving_design(rng, 4, spread=4.0)
    G = wo.build_normal(grid, np.array([[-3.0, 1.0]]), np.array([0.5]), np.array([1.0]))
    u0 = random_state(rng, n, n, scale=0.1)
    ts = wo.build_tspan(f32(0.002), 1e-5, steps)
    ctx.set_source_shape(G, 1000.0)
    set_design(ctx, d0, d1, ts[0], ts[-1])
    ctx.set_state(u0)
    sig, _, _ = ctx.integrate(ts, capture_frames=True)
    st, rsig, fr = oracle_integrate(dim, wo.to_abi(u0), ts, G=G, freq=1000.0, d0=flat_design(d0), d1=flat_design(d1),
                                    ti=ts[0], tf=ts[-1], frame_steps=(steps - 20, steps - 10, steps))
    frames = ctx.get_frames()
    for k in range(3):
        assert np.array_equal(wo.to_abi(frames[:, :, :, k]), fr[k]), f"frame {k}"
    assert np.array_equal(wo.to_abi(ctx.get_state()), st)
    assert not np.array_equal(st[0], st[6])
    assert rel_err(sig, rsig) < ENERGY_RTOL
    ctx.close()


@pytest.mark.parametrize("impl", IMPLS)
def test_env_rollout_matches_oracle_env(impl):
    """WaveEnv mirror vs the oracle's WaveEnv: 3 actions of 25 steps, triple-ring cloak, source next to the rings so
    that the scattered field is non-trivial.  Same seeded designs/actions are injected on both sides."""
    n, steps = 160, 25
    odim = wo.TwoDim.from_size(15.0, n)
    ogrid = wo.build_grid(odim)
    mu = np.array([[0.5, 0.3]], f32)
    osrc = wo.RandomPosGaussianSource(ogrid, mu, mu, np.array([0.3], f32), np.array([1.0], f32), f32(1000.0))
    orng = np.random.default_rng(42)
    oenv = wo.WaveEnv(odim, design_space=wo.build_triple_ring_design_space(), source=osrc, integration_steps=steps,
                      actions=3, rng=orng, resolution=(32, 32))
    oenv.reset()
    dim = w.TwoDim(15.0, n)
    src = w.RandomPosGaussianSource(w.build_grid(dim), mu, mu, [0.3], [1.0], 1000.0, rng=np.random.default_rng(1))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=steps,
                    actions=3, rng=np.random.default_rng(2), resolution=(32, 32), impl=impl)
    env.reset()
    env.design = oracle_to_mirror_design(w, oenv.design)
    # the device builds the Gaussian itself; feed the oracle the device's shape so the fields can be compared exactly
    osrc.shape = np.array(env.source.shape)
    assert rel_err(osrc.shape, wo.build_normal(ogrid, mu, osrc.sigma, osrc.a)) < 3e-7
    pol = wo.RandomDesignPolicy(oenv.action_space(), np.random.default_rng(3))
    while not oenv.is_terminated():
        assert not env.is_terminated()
        oact = pol(oenv)
        ots, ointerp, outot, ouinc = oenv(oact)
        ts, interp, utot, uinc = env(oracle_to_mirror_design(w, oact))
        assert np.array_equal(ts, ots) and env.time_step == oenv.time_step
        assert np.array_equal(env.wave, oenv.wave)
        assert np.array_equal(utot, outot) and np.array_equal(uinc, ouinc)
        assert rel_err(env.signal, oenv.signal) < ENERGY_RTOL
        assert np.array_equal(env.design.stacked().r, wo.stacked_cylinders(oenv.design).r)
    assert env.is_terminated() and oenv.signal[-1, 2] > 1e-6 * oenv.signal[-1, 0] > 0
    assert abs(env.reward() - oenv.reward()) <= ENERGY_RTOL * abs(oenv.reward())
    env.ctx.close()


@pytest.mark.parametrize("impl", IMPLS)
def test_config2_700_triple_ring_100_steps(impl):
    """BASELINE config 2 (the benchmark workload): TwoDim(15, 700), triple-ring design, Gaussian source, 100 steps.
    Run twice: from the zero state (the env's first action) and continued from the resulting state."""
    rng = np.random.default_rng(0)
    dim, ctx = make_ctx(700, impl)
    grid = wo.build_grid(dim)
    ds = wo.build_triple_ring_design_space()
    a = wo.rand_design(ds, rng)
    b = ds(a, wo.rand_design(wo.build_action_space(a, 0.25), np.random.default_rng(1)))
    mu = np.array([[-10.0, np.random.default_rng(2).uniform(-10, 10)]], f32)
    G = wo.build_normal(grid, mu, np.array([0.3], f32), np.array([1.0], f32))
    ctx.set_source_shape(G, 1000.0)
    state = np.zeros((12, 700, 700), f32)
    for act in range(2):
        ts = wo.build_tspan(f32(f32(100 * act) * f32(1e-5)), 1e-5, 100)
        d0, d1 = (a, b) if act == 0 else (b, a)
        set_design(ctx, d0, d1, ts[0], ts[-1])
        sig, _, _ = ctx.integrate(ts, capture_frames=True)
        state, rsig, fr = oracle_integrate(dim, state, ts, G=G, freq=1000.0, d0=flat_design(d0), d1=flat_design(d1),
                                           ti=ts[0], tf=ts[-1], frame_steps=(80, 90, 100))
        frames = ctx.get_frames()
        for k in range(3):
            assert np.array_equal(wo.to_abi(frames[:, :, :, k]), fr[k]), f"action {act} frame {k}"
        assert rel_err(sig[:, :2], rsig[:, :2]) < ENERGY_RTOL
        assert np.abs(sig[:, 2] - rsig[:, 2]).max() <= ENERGY_RTOL * rsig[:, 0].max()
    ctx.close()


@pytest.mark.parametrize("impl", IMPLS)
def test_pulse_through_design_scattering(impl):
    """scripts/pml.jl-style initial pulse placed next to a cylinder: total != incident, scattered energy grows"""
    n = 192
    dim, ctx = make_ctx(n, impl, size=5.0, pml=(1.0, 20000.0))
    grid = wo.build_grid(dim)
    ic = wo.build_normal(grid, np.array([[-1.0, 0.0]]), np.array([0.3]), np.array([1.0]))
    u0 = np.zeros((n, n, 12), f32, order="F")
    u0[:, :, 0] = ic
    u0[:, :, 6] = ic
    cyl = wo.Cylinders(np.array([[0.8, 0.1]], f32), np.array([0.7], f32), np.array([1032.0], f32))
    ctx.set_source_shape(None, 0.0)
    set_design(ctx, cyl, cyl, 0.0, 1.0)
    ctx.set_state(u0)
    ts = wo.build_tspan(0.0, 1e-5, 150)
    sig, _, _ = ctx.integrate(ts)
    st, rsig, _ = oracle_integrate(dim, wo.to_abi(u0), ts, pml=(1.0, 20000.0), d0=flat_design(cyl), d1=flat_design(cyl),
                                   ti=0.0, tf=1.0)
    assert np.array_equal(wo.to_abi(ctx.get_state()), st)
    assert rel_err(sig, rsig) < ENERGY_RTOL
    assert sig[-1, 2] > 1e-3 * sig[-1, 0] and sig[0, 2] == 0
    ctx.close()


def test_uniform_non_ambient_speed_and_integrator_mirror():
    """Integrator-level use without an env (scripts/pml.jl:4-18): iter(wave, tspan, [t -> c, NoSource()])"""
    n = 96
    dim = w.TwoDim(5.0, n)
    odim = wo.TwoDim.from_size(5.0, n)
    it = w.Integrator(w.runge_kutta, w.AcousticDynamics(dim, w.WATER, 1.0, 0.0), 1e-5)
    wave = w.build_wave(dim, 12)
    ic = wo.build_normal(wo.build_grid(odim), np.array([[0.0, 0.0]]), np.array([0.3]), np.array([1.0]))
    wave[:, :, 0] = ic
    wave[:, :, 6] = ic
    ts = it.build_tspan(0.0, 30)
    sol = it(wave, ts, [w.UniformSpeed(w.WATER), w.NoSource()], save=[0, 10, 30])
    assert sol.shape == (n, n, 12, 3)
    oit = wo.Integrator(wo.runge_kutta, wo.AcousticDynamics.build(odim, wo.WATER, 1.0, 0.0), f32(1e-5))
    ref = oit(np.array(wave), ts, [lambda t: wo.WATER, wo.NoSource()], save={0, 10, 30})
    assert np.array_equal(sol, ref)
    # a different uniform speed for the total field only
    sol2 = it(wave, ts, [w.UniformSpeed(1200.0), w.NoSource()], save=[30])
    ref2 = oit(np.array(wave), ts, [lambda t: f32(1200.0), wo.NoSource()], save={30})
    assert np.array_equal(sol2, ref2) and not np.array_equal(sol2[:, :, 0], sol2[:, :, 6])


def test_error_behaviour():
    dim, ctx = make_ctx(64)
    ts = wo.build_tspan(0.0, 1e-5, 10)
    with pytest.raises(w.WavesAmdError) as ei:          # Julia: BoundsError at src/env.jl:116
        ctx.integrate(ts, capture_frames=True)
    assert ei.value.status == w._ffi.WV_ERR_INVALID
    ctx.integrate(ts, capture_frames=False)              # Integrator-level call is fine
    ctx.integrate_begin(ts)
    with pytest.raises(w.WavesAmdError) as ei:
        ctx.set_state(np.zeros((64, 64, 12), f32))
    assert ei.value.status == w._ffi.WV_ERR_STATE
    ctx.integrate_end()
    with pytest.raises(w.WavesAmdError):
        ctx.integrate_end()
    ctx.close()
    with pytest.raises(AssertionError):                  # src/env.jl:52
        w.WaveEnv(w.TwoDim(15.0, 64), design_space=w.build_triple_ring_design_space(), resolution=(128, 128))


def test_recycled_device_memory_does_not_leak_into_results():
    """contexts created after others were destroyed get recycled (non-zero) device memory: the reduced field sets of the
    fused kernel must not depend on what a buffer held before (regression: scratch states are zeroed at wv_create)"""
    rng = np.random.default_rng(5)
    for rep in range(3):
        dim, ctx = make_ctx(96, "fused", size=5.0, pml=(1.0, 0.0))
        junk = random_state(rng, 96, 96, scale=1e3)
        ctx.set_frames(np.stack([junk] * 3, axis=3))
        ctx.integrate(wo.build_tspan(0.0, 1e-5, 25), capture_frames=True, want_signal=False)   # dirties every buffer
        ctx.close()
    dim, ctx = make_ctx(96, "fused", size=5.0, pml=(1.0, 0.0))
    ic = wo.build_normal(wo.build_grid(dim), np.array([[0.0, 0.0]]), np.array([0.3]), np.array([1.0]))
    u0 = np.zeros((96, 96, 12), f32, order="F")
    u0[:, :, 0] = ic
    u0[:, :, 6] = ic
    ctx.set_source_shape(None, 0.0)
    ctx.set_state(u0)
    ts = wo.build_tspan(0.0, 1e-5, 30)
    ctx.integrate(ts, want_signal=False)
    st, _, _ = oracle_integrate(dim, wo.to_abi(u0), ts, pml=(1.0, 0.0))
    assert np.array_equal(wo.to_abi(ctx.get_state()), st)
    ctx.close()


# ---- the resident kernel (all steps of a call in one cooperative launch, tagged halo exchange) ------------------------
def _run_resident_case(monkeypatch, resident, n, steps, *, capture, fields, aux, seed):
    import gc
    gc.collect()  # contexts earlier tests left to the collector: the resident path needs the device to itself
    monkeypatch.setenv("WAVES_AMD_FUSED_RESIDENT", "1" if resident else "0")
    rng = np.random.default_rng(seed)
    dim, ctx = make_ctx(n, "fused")
    grid = wo.build_grid(dim)
    d0, d1 = small_moving_design(rng, 5, spread=4.0)
    G = wo.build_normal(grid, np.array([[-2.0, 1.5]]), np.array([0.5]), np.array([1.0]))
    u0 = random_state(rng, n, n, scale=0.1, aux=aux)
    ts = wo.build_tspan(f32(0.001), 1e-5, steps)
    ctx.set_source_shape(G, 1000.0)
    set_design(ctx, d0, d1, ts[0], ts[-1])
    ctx.set_state(u0)
    sig, ut, ui = ctx.integrate(ts, capture_frames=capture, want_fields=fields)
    tim = ctx.timing()
    # a second action on top of the first: the exchange tags keep growing, the energy row is carried over
    ts2 = wo.build_tspan(ts[-1], 1e-5, steps)
    set_design(ctx, d1, d0, ts2[0], ts2[-1])
    sig2, _, _ = ctx.integrate(ts2, capture_frames=capture)
    out = dict(sig=sig, sig2=sig2, ut=ut, ui=ui, frames=ctx.get_frames(), state=ctx.get_state(), resident=tim["resident"],
               launches=None)
    ctx.close()
    return out, (dim, u0, ts, G, d0, d1)


@pytest.mark.parametrize("n,steps,capture,fields,aux", [(200, 30, True, False, True), (160, 21, True, True, False),
                                                         (96, 2, False, True, False), (330, 25, True, False, False)])
def test_resident_kernel_equals_single_step_kernels_and_oracle(monkeypatch, n, steps, capture, fields, aux):
    """Same inputs through k_steps_resident (one launch, state in registers, tagged halo exchange) and through the
    per-step launches: every output bit-identical, incl. trajectories, frames, a follow-up action; and == the oracle."""
    a, case = _run_resident_case(monkeypatch, True, n, steps, capture=capture, fields=fields, aux=aux, seed=n + steps)
    b, _ = _run_resident_case(monkeypatch, False, n, steps, capture=capture, fields=fields, aux=aux, seed=n + steps)
    assert a["resident"] is True and b["resident"] is False
    for k in ("sig", "sig2", "frames", "state"):
        assert np.array_equal(a[k], b[k]), k
    if fields:
        assert np.array_equal(a["ut"], b["ut"]) and np.array_equal(a["ui"], b["ui"])
    dim, u0, ts, G, d0, d1 = case
    st, rsig, _ = oracle_integrate(dim, wo.to_abi(u0), ts, G=G, freq=1000.0, d0=flat_design(d0), d1=flat_design(d1),
                                   ti=ts[0], tf=ts[-1])
    assert rel_err(a["sig"], rsig) < ENERGY_RTOL
    if fields:  # u_tot[:, :, k] is the U_tot plane after k steps
        assert np.array_equal(np.asarray(a["ut"])[:, :, -1].T, st[0])
        assert np.array_equal(np.asarray(a["ui"])[:, :, -1].T, st[6])


def test_resident_kernel_is_the_default_at_700_and_not_used_when_tiles_do_not_fit(monkeypatch):
    import gc
    gc.collect()
    monkeypatch.delenv("WAVES_AMD_FUSED_RESIDENT", raising=False)
    for n, expect in ((700, True), (1500, False)):
        dim, ctx = make_ctx(n, "fused")
        ctx.set_gaussian_source([[3.0, 0.5]], [0.3], [1.0], 1000.0)
        ctx.integrate(wo.build_tspan(0.0, 1e-5, 4))
        assert ctx.timing()["resident"] is expect, n
        ctx.close()


def test_resident_kernel_only_when_the_context_has_the_device_to_itself(monkeypatch):
    """Several environments on one GPU step concurrently on their streams: they use the single-step kernels (a resident
    kernel occupies the whole device for the duration of a call).  Same bits either way."""
    import gc
    gc.collect()
    monkeypatch.delenv("WAVES_AMD_FUSED_RESIDENT", raising=False)
    ts = wo.build_tspan(0.0, 1e-5, 24)

    def go(ctx):
        ctx.set_gaussian_source([[1.0, -2.0]], [0.4], [1.0], 1000.0)
        ctx.reset()
        sig, _, _ = ctx.integrate(ts, capture_frames=True)
        return sig, ctx.get_frames(), ctx.timing()["resident"]

    dim, a = make_ctx(300, "fused")
    sa, fa, ra = go(a)
    dim, b = make_ctx(300, "fused")
    sb, fb, rb = go(b)
    sa2, fa2, ra2 = go(a)
    b.close()
    sa3, fa3, ra3 = go(a)
    a.close()
    assert (ra, rb, ra2, ra3) == (True, False, False, True)
    for s_, f_ in ((sb, fb), (sa2, fa2), (sa3, fa3)):
        assert np.array_equal(s_, sa) and np.array_equal(f_, fa)


def _ring_run(ctx, steps, calls):
    out = []
    ctx.set_gaussian_source([[1.0, -2.0]], [0.4], [1.0], 1000.0)
    ctx.reset()
    t0 = 0.0
    for _ in range(calls):
        ts = wo.build_tspan(f32(t0), 1e-5, steps)
        sig, _, _ = ctx.integrate(ts, capture_frames=True)
        out.append((sig, ctx.get_frames()))
        t0 = ts[-1]
    return out


def test_resident_exchange_tags_wrap_around(monkeypatch):
    """The exchange words carry a 32-bit step tag that grows from call to call; shortly before it would wrap the buffer
    is cleared and the count restarts.  Start it 16 steps below the threshold and cross it: same bits as a fresh one."""
    import gc
    gc.collect()
    monkeypatch.delenv("WAVES_AMD_FUSED_RESIDENT", raising=False)
    dim, ref = make_ctx(200, "fused")
    want = _ring_run(ref, 30, 3)
    ref.close()
    monkeypatch.setenv("WAVES_AMD_TAG_BASE", hex(0xFFFF0000 - 46))
    dim, ctx = make_ctx(200, "fused")
    got = _ring_run(ctx, 30, 3)
    assert ctx.timing()["resident"] is True
    ctx.close()
    for (sa, fa), (sb, fb) in zip(want, got):
        assert np.array_equal(sa, sb) and np.array_equal(fa, fb)


def test_resident_give_up_is_transparent(monkeypatch):
    """VERDICT r2 item 2.  A resident tile that never sees its neighbours' words must neither hang the device nor cost the
    caller its state: with the poll budget forced to one some wave gives up in the first step, the launch drains without
    having touched the call's initial condition (barrier A of k_steps_resident), the library runs the SAME call again on
    the single-step kernels and returns WV_OK with the oracle's bits -- as the reference's env(action) never loses
    env.wave (src/env.jl:102-116: the state is only rebound once `sol` exists).  The context then stays on the
    single-step kernels."""
    import gc
    gc.collect()
    monkeypatch.delenv("WAVES_AMD_FUSED_RESIDENT", raising=False)
    dim, ref = make_ctx(300, "fused")
    want = _ring_run(ref, 40, 2)
    assert ref.timing()["resident"] is True and ref.timing()["gave_up"] is False
    ref.close()
    dim, ctx = make_ctx(300, "fused")
    monkeypatch.setenv("WAVES_AMD_WAIT_POLLS", "1")
    got = _ring_run(ctx, 40, 1)
    t = ctx.timing()
    assert t["gave_up"] is True and t["resident"] is False
    monkeypatch.delenv("WAVES_AMD_WAIT_POLLS")
    assert np.array_equal(want[0][0], got[0][0]) and np.array_equal(want[0][1], got[0][1])
    got2 = _ring_run(ctx, 40, 2)   # reset() + fresh calls on the same context
    t = ctx.timing()
    assert t["resident"] is False and t["gave_up"] is False   # ... which stays on the single-step kernels after a give-up
    ctx.close()
    for (sa, fa), (sb, fb) in zip(want, got2):
        assert np.array_equal(sa, sb) and np.array_equal(fa, fb)
    # the fields are the oracle's: 40 steps from the zero state, the source of _ring_run, no design
    x = dim.x
    sx = wo.build_pml_profile(x, 2.0, 20000.0)
    G = wo.to_abi(wo.build_normal(wo.build_grid(dim), np.array([[1.0, -2.0]]), np.array([0.4]), np.array([1.0])))
    from c_oracle import integrate as c_integrate
    st, es, _ = c_integrate(x, x, sx, sx, wo.WATER, 1e-5, np.zeros((12, 300, 300), f32), wo.build_tspan(f32(0.0), 1e-5, 40),
                            G=G, freq=1000.0, nthreads=4)
    assert np.array_equal(wo.to_abi(got[0][1][:, :, :, 2]), st)


def test_two_pending_calls_both_recover_from_a_give_up(monkeypatch):
    """Two calls in flight when the resident launch gives the first one up: the second one was meant for the same launch and
    never ran.  Both are run again, in order, by the single-step kernels; both _end calls return the right bits.  Also
    ADVICE r2 (_ffi.py): the Python mirror of the pending queue stays in line with the library's (begin/end of different
    lengths afterwards)."""
    import gc
    gc.collect()
    monkeypatch.delenv("WAVES_AMD_FUSED_RESIDENT", raising=False)

    def run(ctx, force):
        ctx.set_gaussian_source([[1.0, -2.0]], [0.4], [1.0], 1000.0)
        ctx.reset()
        ts1 = wo.build_tspan(f32(0.0), 1e-5, 30)
        ts2 = wo.build_tspan(ts1[-1], 1e-5, 24)
        ts3 = wo.build_tspan(ts2[-1], 1e-5, 21)
        sig0, _, _ = ctx.integrate(ts1, capture_frames=True)       # (a first call: the launch is there, nothing on the stream)
        if force:
            monkeypatch.setenv("WAVES_AMD_WAIT_POLLS", "1")
        ctx.integrate_begin(ts2, capture_frames=True)
        ctx.integrate_begin(ts3, capture_frames=True)
        if force:
            monkeypatch.delenv("WAVES_AMD_WAIT_POLLS")
        assert ctx.pending() == 2
        sig1, _, _ = ctx.integrate_end()
        t1 = ctx.timing()
        sig2, _, _ = ctx.integrate_end()
        t2 = ctx.timing()
        assert ctx.pending() == 0
        return (sig0, sig1, sig2, ctx.get_frames()), (t1, t2)

    dim, ref = make_ctx(300, "fused")
    want, (t1, t2) = run(ref, False)
    assert t1["resident"] and t2["resident"] and not t1["gave_up"]
    ref.close()
    dim, ctx = make_ctx(300, "fused")
    got, (t1, t2) = run(ctx, True)
    assert t1["gave_up"] and not t1["resident"] and not t2["resident"]
    for a, b in zip(want, got):
        assert np.array_equal(a, b)
    # the queue mirror: an _end with nothing pending is the library's WV_ERR_STATE, and the context is still usable
    with pytest.raises(w._ffi.WavesAmdError):
        ctx.integrate_end()
    sig, _, _ = ctx.integrate(wo.build_tspan(f32(0.0), 1e-5, 20), capture_frames=True)
    assert sig.shape == (21, 3)
    ctx.close()


# ---- the observation path -------------------------------------------------------------------------
@pytest.mark.parametrize("n,res", [(700, (128, 128)), (256, (128, 128)), (257, (100, 90)), (64, (64, 63)), (96, (1, 5))])
def test_observation_resized_on_the_device_equals_the_restated_rule(n, res):
    """wv_observation == imresize_linear(cat(U_tot frames, source shape)) bit for bit (Float64 evaluation, one rounding).
    Parity with Images.jl itself is unpinned (third-party, absent): see waves_oracle.imresize_linear."""
    rng = np.random.default_rng(n)
    dim, ctx = make_ctx(n, "fused")
    wave = np.asfortranarray(rng.standard_normal((n, n, 12, 3)).astype(f32))
    ctx.set_frames(wave)
    G = wo.build_normal(wo.build_grid(dim), np.array([[2.0, -3.0]]), np.array([0.7]), np.array([1.0]))
    ctx.set_source_shape(G, 1000.0)
    got = ctx.observation(*res)
    want = wo.imresize_linear(np.concatenate([wave[:, :, 0, :], G[:, :, None]], axis=2), res)
    assert got.shape == res + (4,) and np.array_equal(got, want)
    ctx.set_source_shape(None, 0.0)                      # NoSource: the shape channel is zero
    assert np.array_equal(ctx.observation(*res)[:, :, 3], np.zeros(res, f32))
    with pytest.raises(w._ffi.WavesAmdError):
        ctx.observation(n + 1, 4)
    ctx.close()


def test_env_state_is_the_resized_observation():
    dim = w.TwoDim(15.0, 300)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(5))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=30, actions=3,
                    rng=np.random.default_rng(6), return_fields=False)
    env.reset()
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(7))
    env(pol(env))
    s = env.state()
    frames = env.ctx.get_frames()[:, :, 0, :]
    full = np.concatenate([frames, env.ctx.source_shape()[:, :, None]], axis=2)
    assert s.wave.shape == (128, 128, 4)
    assert np.array_equal(s.wave, wo.imresize_linear(full, (128, 128)))
    assert np.abs(s.wave[:, :, 2]).max() > 0
    env.ctx.close()


def test_episode_with_observations_windowing_and_file_round_trip(tmp_path):
    """SURVEY 8f ranks 1-2 end to end on the device: generate_episode! recording state(env) before every action
    (src/data.jl:12-33), prepare_data (:35-57), save / load."""
    dim = w.TwoDim(15.0, 160)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(15))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=20, actions=4,
                    rng=np.random.default_rng(16), return_fields=False)
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(17))
    ep = w.generate_episode(pol, env, with_states=True)
    assert len(ep) == 4 and len(ep.s) == 4 and ep.s[0].wave.shape == (128, 128, 4)
    assert not ep.s[0].wave[:, :, :3].any() and ep.s[0].wave[:, :, 3].max() > 0      # zero wave, source shape present
    assert np.abs(ep.s[3].wave[:, :, 2]).max() > 0
    s, a, t, y = w.prepare_data(ep, 2)
    assert len(y) == 3 and y[0].shape == (41, 3) and t[0].shape == (41,)
    assert np.array_equal(y[1][:21], ep.y[1]) and np.array_equal(y[1][21:], ep.y[2][1:])
    p = str(tmp_path / "ep.npz")
    ep.save(p)
    back = w.Episode.load(p)
    assert all(np.array_equal(b.wave, e.wave) for b, e in zip(back.s, ep.s))
    assert all(np.array_equal(b, e) for b, e in zip(back.y, ep.y))
    env.ctx.close()


def test_strided_trajectories_are_a_subsampling_of_the_full_ones(monkeypatch):
    """SURVEY 8f rank 3 (the byte-saving half): with wv_set_trajectory_stride(k) the u_tot / u_inc outputs hold the saved
    times 0, k, 2k, ...; everything else is unchanged.  Both step kernels."""
    import gc
    for resident in ("1", "0"):
        gc.collect()
        monkeypatch.setenv("WAVES_AMD_FUSED_RESIDENT", resident)
        outs = {}
        for k in (1, 4, 7):
            dim, ctx = make_ctx(150, "fused")
            ctx.set_gaussian_source([[1.0, -2.0]], [0.4], [1.0], 1000.0)
            ctx.set_trajectory_stride(k)
            sig, ut, ui = ctx.integrate(wo.build_tspan(0.0, 1e-5, 30), capture_frames=True, want_fields=True)
            outs[k] = (sig, ut, ui, ctx.get_frames())
            ctx.close()
        full = outs[1]
        assert full[1].shape == (150, 150, 31)
        for k in (4, 7):
            sig, ut, ui, fr = outs[k]
            assert ut.shape == (150, 150, 30 // k + 1)
            assert np.array_equal(ut, full[1][:, :, ::k]) and np.array_equal(ui, full[2][:, :, ::k])
            assert np.array_equal(sig, full[0]) and np.array_equal(fr, full[3])
