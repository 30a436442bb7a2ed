"""More parity tests through the C ABI (round 2): the headline kernel pinned to the oracle on the headline configuration,
BASELINE config 4 (2048^2) against the oracle and at its full length, seeded slices of the randomised stress runs, two
actions in flight, and the regressions of the round-1 review."""
import ctypes
import gc
import os
import sys

import numpy as np
import pytest

import waves_jl_amd as w
import waves_oracle as wo
from helpers import flat_design, oracle_integrate, random_state, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
f32 = np.float32
ENERGY_RTOL = 1e-5


def make_ctx(n, impl="auto", pml=(2.0, 20000.0), size=15.0, dt=1e-5, c0=wo.WATER):
    dim = wo.TwoDim.from_size(size, n)
    ctx = w._ffi.Context(dim.x, dim.y, c0=c0, dt=dt, pml_width=pml[0], pml_scale=pml[1], device=0, impl=impl)
    return dim, ctx


def set_design(ctx, d0, d1, ti, tf):
    a, b = wo.stacked_cylinders(d0), wo.stacked_cylinders(d1)
    ctx.set_design((a.pos, a.r, a.c), (b.pos, b.r, b.c), ti, tf)


def triple_ring_pair(seed=0):
    ds = wo.build_triple_ring_design_space()
    a = wo.rand_design(ds, np.random.default_rng(seed))
    b = ds(a, wo.rand_design(wo.build_action_space(a, 0.25), np.random.default_rng(seed + 1)))
    return a, b


@pytest.mark.parametrize("resident", [True, False])
def test_headline_kernel_on_the_headline_config_vs_oracle(monkeypatch, resident):
    """BASELINE config 2 (700^2, triple ring, Gaussian source, 100 steps, two consecutive actions) with the step kernel
    ASSERTED: k_steps_resident (the bench's kernel) and, separately, the single-step k_step_fused."""
    gc.collect()  # contexts other tests left to the collector would keep this one off the resident path
    monkeypatch.setenv("WAVES_AMD_FUSED_RESIDENT", "1" if resident else "0")
    dim, ctx = make_ctx(700, "fused")
    G = wo.build_normal(wo.build_grid(dim), np.array([[-10.0, 3.7]], f32), np.array([0.3], f32), np.array([1.0], f32))
    ctx.set_source_shape(G, 1000.0)
    a, b = triple_ring_pair(0)
    state = np.zeros((12, 700, 700), f32)
    for act in range(2):
        ts = wo.build_tspan(f32(f32(100 * act) * f32(1e-5)), 1e-5, 100)
        d0, d1 = (a, b) if act == 0 else (b, a)
        set_design(ctx, d0, d1, ts[0], ts[-1])
        sig, _, _ = ctx.integrate(ts, capture_frames=True)
        assert ctx.timing()["resident"] is resident and ctx.timing()["impl"] == "fused"
        state, rsig, fr = oracle_integrate(dim, state, ts, G=G, freq=1000.0, d0=flat_design(d0), d1=flat_design(d1),
                                           ti=ts[0], tf=ts[-1], frame_steps=(80, 90, 100))
        frames = ctx.get_frames()
        for k in range(3):
            assert np.array_equal(wo.to_abi(frames[:, :, :, k]), fr[k]), f"action {act} frame {k}"
        assert rel_err(sig[:, :2], rsig[:, :2]) < ENERGY_RTOL
        assert np.abs(sig[:, 2] - rsig[:, 2]).max() <= ENERGY_RTOL * rsig[:, 0].max()
    ctx.close()


@pytest.mark.parametrize("width", [1.0, 4.0])
def test_config4_2048_ten_steps_vs_oracle(width):
    """BASELINE config 4's grid (2048^2) at PML widths 1 and 4, triple ring + source, 10 steps from a random state with
    the auxiliary fields confined to the PML, against the C oracle: every field of the final state bit for bit."""
    n = 2048
    dim, ctx = make_ctx(n, "fused", pml=(width, 20000.0))
    G = wo.build_normal(wo.build_grid(dim), np.array([[-10.0, 0.0]], f32), np.array([0.3], f32), np.array([1.0], f32))
    ctx.set_source_shape(G, 1000.0)
    a, b = triple_ring_pair(5)
    ts = wo.build_tspan(f32(0.0), 1e-5, 10)
    set_design(ctx, a, b, ts[0], ts[-1])
    rng = np.random.default_rng(11)
    u0 = np.zeros((n, n, 12), f32, order="F")
    for f in (0, 1, 2, 6, 7, 8):
        u0[:, :, f] = (rng.standard_normal((n, n)) * 0.1).astype(f32)
    ctx.set_state(u0)
    sig, _, _ = ctx.integrate(ts)
    assert ctx.timing()["impl"] == "fused" and not ctx.timing()["resident"]   # 3 700 tiles: single-step kernels
    st, rsig, _ = oracle_integrate(dim, wo.to_abi(u0), ts, pml=(width, 20000.0), G=G, freq=1000.0, d0=flat_design(a),
                                   d1=flat_design(b), ti=ts[0], tf=ts[-1], nthreads=16)
    assert np.array_equal(wo.to_abi(ctx.get_state()), st)
    assert rel_err(sig[:, :2], rsig[:, :2]) < ENERGY_RTOL
    ctx.close()


def test_config4_2048_500_steps_width_sweep_properties():
    """BASELINE config 4 at its full length: 2048^2, 500 steps, PML widths 1 / 2 / 4.  The oracle would need minutes, so
    size-independent properties: the fused kernel is deterministic (two runs, same bits), equals the staged kernels bit
    for bit on the final state (width 2), energies stay finite and the incident energy decays once the source's first
    periods have left through the PML."""
    n, steps = 2048, 500
    ts = wo.build_tspan(f32(0.0), 1e-5, steps)
    a, b = triple_ring_pair(9)
    finals = {}
    for width, impls in ((1.0, ["fused"]), (2.0, ["fused", "fused", "staged"]), (4.0, ["fused"])):
        for k, impl in enumerate(impls):
            dim, ctx = make_ctx(n, impl, pml=(width, 20000.0))
            ctx.set_gaussian_source([[-10.0, 0.0]], [0.3], [1.0], 1000.0)
            set_design(ctx, a, b, ts[0], ts[-1])
            sig, _, _ = ctx.integrate(ts)
            assert np.isfinite(sig).all() and sig[-1, 0] > 0 and sig[-1, 2] >= 0
            finals[(width, k)] = (ctx.get_state(), sig)
            ctx.close()
            gc.collect()
    assert np.array_equal(finals[(2.0, 0)][0], finals[(2.0, 1)][0]) and np.array_equal(finals[(2.0, 0)][1], finals[(2.0, 1)][1])
    assert np.array_equal(finals[(2.0, 0)][0], finals[(2.0, 2)][0])          # fused == staged after 500 steps
    assert rel_err(finals[(2.0, 0)][1], finals[(2.0, 2)][1]) < ENERGY_RTOL
    # a wider PML changes nothing in the interior before the first reflections could return: the three runs agree on the
    # early part of the trace to the energy tolerance (the PML cells themselves differ)
    e1, e2, e4 = (finals[(wd, 0)][1] for wd in (1.0, 2.0, 4.0))
    assert np.abs(e1[:30, 0] - e2[:30, 0]).max() <= 0.1 * e2[:30, 0].max()
    assert np.abs(e4[:30, 0] - e2[:30, 0]).max() <= 0.1 * e2[:30, 0].max()


def test_seeded_slice_of_the_oracle_stress_run():
    """tools/stress_oracle.py, 30 seeded cases: random small grids (8 ... 260, odd tile widths included), PML widths,
    designs, sources, initial states, either step kernel -- fields bit-exact against the C oracle, energies to 1e-5."""
    import stress_oracle
    gc.collect()
    bad, nres = stress_oracle.run_cases(30, 1234, verbose=False)
    assert bad == 0 and nres >= 5


def test_seeded_slice_of_the_resident_stress_run():
    """tools/stress_resident.py, 24 seeded cases up to 760^2: the resident kernel against the single-step kernels, every
    output (frames, trajectories, energy traces) bit-identical over 1-3 consecutive calls."""
    import stress_resident
    gc.collect()
    bad, nres = stress_resident.run_cases(24, 4321, verbose=False)
    assert bad == 0 and nres >= 12


def test_rhs_between_integrates_does_not_leak_into_the_single_step_path(monkeypatch):
    """wv_rhs stages its argument in a scratch state; the single-step kernel's reduced field sets rely on the auxiliary
    planes of the scratch states being zero outside the PML (round-1 advisor finding)."""
    monkeypatch.setenv("WAVES_AMD_FUSED_RESIDENT", "0")
    n = 160
    dim, ctx = make_ctx(n, "fused")
    a, b = triple_ring_pair(3)
    ts = wo.build_tspan(f32(0.0), 1e-5, 25)
    set_design(ctx, a, b, ts[0], ts[-1])
    G = wo.build_normal(wo.build_grid(dim), np.array([[0.5, 0.3]], f32), np.array([0.3], f32), np.array([1.0], f32))
    ctx.set_source_shape(G, 1000.0)
    sig, _, _ = ctx.integrate(ts, capture_frames=True)
    st, _, _ = oracle_integrate(dim, np.zeros((12, n, n), f32), ts, G=G, freq=1000.0, d0=flat_design(a), d1=flat_design(b),
                                ti=ts[0], tf=ts[-1])
    assert np.array_equal(wo.to_abi(ctx.get_state()), st)
    ctx.rhs(random_state(np.random.default_rng(5), n, n, aux=True), 0.0)      # non-zero Psi / Omega everywhere
    ts2 = wo.build_tspan(ts[-1], 1e-5, 25)
    set_design(ctx, b, a, ts2[0], ts2[-1])
    ctx.integrate(ts2, capture_frames=True)
    assert not ctx.timing()["resident"]
    st2, _, _ = oracle_integrate(dim, st, ts2, G=G, freq=1000.0, d0=flat_design(b), d1=flat_design(a), ti=ts2[0], tf=ts2[-1])
    assert np.array_equal(wo.to_abi(ctx.get_state()), st2)
    ctx.close()


def test_exchange_granules_are_never_seen_torn():
    """What the resident kernel's halo exchange relies on beyond the ISA's promises (csrc/fused_body.h): 16-byte-aligned
    16-byte agent-scope accesses are not torn.  Writers and readers on all XCDs, three address patterns."""
    dim, ctx = make_ctx(700)
    checked, torn = ctx.selftest_granules(20000)
    assert torn == 0 and checked > 10 ** 8
    ctx.close()


def _env(n, steps, actions, seed, **kw):
    dim = w.TwoDim(15.0, n)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(seed))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=steps,
                    actions=actions, rng=np.random.default_rng(seed + 1), return_fields=False, resolution=(64, 64), **kw)
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(seed + 2))
    env.reset()
    return env, pol


@pytest.mark.parametrize("n,resident", [(700, True), (300, True), (300, False)])
def test_two_actions_in_flight_equal_one_at_a_time(monkeypatch, n, resident):
    """env.step_begin(k+1) before env.step_end(k) (w.rollout_pipelined): the host prepares action k+1 while action k runs.
    Signals of every action and the final env.wave must be bit-identical to the plain `env(policy(env))` loop."""
    gc.collect()
    monkeypatch.setenv("WAVES_AMD_FUSED_RESIDENT", "1" if resident else "0")
    steps, actions = 40, 5
    env, pol = _env(n, steps, actions, 77)
    ref = []
    while not env.is_terminated():
        env(pol(env))
        ref.append(env.signal)
    assert env.ctx.timing()["resident"] is resident
    wave_ref, design_ref = env.wave, env.design.stacked().r.copy()
    env.ctx.close()
    gc.collect()
    env, pol = _env(n, steps, actions, 77)
    got = w.rollout_pipelined(env, pol, actions)
    assert env.is_terminated() and len(got) == actions
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    assert np.array_equal(env.wave, wave_ref) and np.array_equal(env.design.stacked().r, design_ref)
    # a third call while two are pending is refused, and so is a setter; the queue stays intact
    env2, pol2 = env, pol
    env2.actions += 3
    env2.step_begin(pol2(env2))
    env2.step_begin(pol2(env2))
    with pytest.raises(w.WavesAmdError) as ei:
        env2.ctx.integrate_begin(env2.build_tspan(), capture_frames=True)
    assert ei.value.status == w._ffi.WV_ERR_STATE
    with pytest.raises(w.WavesAmdError):
        env2.ctx.reset()
    env2.step_end()
    env2.step_end()
    env2.ctx.close()


@pytest.mark.parametrize("n,resident,per", [(700, True, None), (300, True, 2), (300, False, None)])
def test_action_sequence_in_one_call_equals_one_call_per_action(monkeypatch, n, resident, per):
    """w.rollout_batched / WaveEnv.steps_begin (wv_set_design_sequence): several actions of a state-independent policy in
    ONE device call.  Every action's signal, the final env.wave and the bookkeeping must be bit-identical to the plain
    `env(policy(env))` loop (src/data.jl:22-27)."""
    gc.collect()
    monkeypatch.setenv("WAVES_AMD_FUSED_RESIDENT", "1" if resident else "0")
    steps, actions = 40, 5
    env, pol = _env(n, steps, actions, 91)
    ref = []
    while not env.is_terminated():
        env(pol(env))
        ref.append(env.signal)
    wave_ref, design_ref, ts_ref = env.wave, env.design.stacked().r.copy(), env.time_step
    env.ctx.close()
    gc.collect()
    env, pol = _env(n, steps, actions, 91)
    got = w.rollout_batched(env, pol, actions, per_launch=per)
    assert env.ctx.timing()["resident"] is resident
    assert env.is_terminated() and len(got) == actions and env.time_step == ts_ref
    for a, b in zip(ref, got):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert np.array_equal(env.signal, ref[-1])
    assert np.array_equal(env.wave, wave_ref) and np.array_equal(env.design.stacked().r, design_ref)
    # the context is back to one design: a plain action after the sequence continues the rollout
    env.actions += 1
    env(pol(env))
    assert np.all(np.isfinite(env.signal)) and env.signal.shape == (steps + 1, 3)
    # a sequence whose length does not match the call is refused and consumed
    c = env.design.stacked()
    env.ctx.set_design_sequence([(c.pos, c.r, c.c)] * 3, [(0.0, 1.0)] * 2, steps)
    with pytest.raises(w.WavesAmdError) as ei:
        env.ctx.integrate_begin(env.build_tspan(), capture_frames=True)
    assert ei.value.status == w._ffi.WV_ERR_INVALID
    env.ctx.close()


@pytest.mark.parametrize("resident", [True, False])
def test_action_sequence_keeps_the_frames_of_every_action(monkeypatch, resident):
    """steps_begin(..., keep_frames=True) (capture_frames == 2): env.wave and state(env) after EVERY action of the one
    device call, bit-identical to what the plain `env(policy(env))` loop sees between its actions (src/data.jl:22-27)."""
    gc.collect()
    monkeypatch.setenv("WAVES_AMD_FUSED_RESIDENT", "1" if resident else "0")
    n, steps, actions = 300, 40, 4
    env, pol = _env(n, steps, actions, 123)
    waves, obs, designs, tspans, sigs = [], [], [], [], []
    while not env.is_terminated():
        env(pol(env))
        waves.append(env.wave)
        st = env.state()
        obs.append(st.wave); designs.append(st.design.stacked().r.copy()); tspans.append(np.array(st.tspan))
        sigs.append(env.signal)
    env.ctx.close()
    gc.collect()
    env, pol = _env(n, steps, actions, 123)
    env.steps_begin([pol(env) for _ in range(actions)], keep_frames=True)
    with pytest.raises(w.WavesAmdError):   # not overlapped with another call
        env.ctx.integrate_begin(env.build_tspan(), capture_frames=True)
    got = env.steps_end()
    assert env.ctx.timing()["resident"] is resident
    for k in range(actions):
        assert np.array_equal(got[k], sigs[k])
        assert np.array_equal(env.ctx.get_frames_action(k), waves[k])
        st = env.state_after(k)
        assert np.array_equal(st.wave, obs[k]) and np.array_equal(st.design.stacked().r, designs[k])
        assert np.array_equal(st.tspan, tspans[k])
    assert np.array_equal(env.wave, waves[-1])
    with pytest.raises(w.WavesAmdError):
        env.ctx.get_frames_action(actions)
    # any later integrate call ends the validity of the kept frames
    env.actions += 1
    env(pol(env))
    with pytest.raises(w.WavesAmdError):
        env.ctx.observation_action(0, 32, 32)
    env.ctx.close()


def test_generate_episode_in_one_launch_equals_the_plain_loop():
    """generate_episode(..., with_states=True, per_launch=n): the episode of src/data.jl:12-33 (states in front of every
    action, actions, tspans, signals) from ONE device call per n actions, equal to the plain loop's."""
    gc.collect()
    n, steps, actions = 300, 40, 5

    def episode(**kw):
        env, pol = _env(n, steps, actions, 321)
        ep = w.generate_episode(pol, env, reset=False, with_states=True, **kw)
        env.ctx.close()
        gc.collect()
        return ep

    ref, got = episode(in_flight=1), episode(per_launch=3)
    assert len(ref) == len(got) == actions and len(got.s) == actions
    for k in range(actions):
        assert np.array_equal(ref.s[k].wave, got.s[k].wave) and np.array_equal(ref.s[k].tspan, got.s[k].tspan)
        assert np.array_equal(ref.s[k].design.stacked().r, got.s[k].design.stacked().r)
        assert np.array_equal(ref.a[k].stacked().r, got.a[k].stacked().r)
        assert np.array_equal(ref.t[k], got.t[k]) and np.array_equal(ref.y[k], got.y[k])


def test_state_written_through_the_raw_device_pointer_is_looked_at_again():
    """wv_device_frames hands out env.wave's device pointer; until wv_release_device_frames every integrate re-derives
    what it otherwise caches about the state (field-set precondition, initial energies)."""
    n = 160
    dim, ctx = make_ctx(n, "fused")
    ts = wo.build_tspan(f32(0.0), 1e-5, 20)
    ctx.integrate(ts)                                     # zero state: reduced field sets, cached energy partials
    ptr, nbytes = ctx.device_frames()
    assert nbytes == 3 * 12 * n * n * 4
    hip = ctypes.CDLL("libamdhip64.so")
    u = random_state(np.random.default_rng(8), n, n, aux=True)          # auxiliary fields non-zero OUTSIDE the PML
    ctx.integrate(wo.build_tspan(ts[-1], 1e-5, 20))      # (an integrate between the hand-out and the write)
    dst = ctypes.c_void_p(ptr + 2 * 12 * n * n * 4)      # the last frame = the integrator's initial condition
    assert hip.hipMemcpy(dst, u.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(u.nbytes), 1) == 0
    sig, _, _ = ctx.integrate(ts)
    st, rsig, _ = oracle_integrate(dim, wo.to_abi(u), ts)
    assert np.array_equal(wo.to_abi(ctx.get_state()), st)
    assert rel_err(sig[:, :2], rsig[:, :2]) < ENERGY_RTOL
    ctx.release_device_frames()
    ctx.close()


@pytest.mark.parametrize("n,stride,resident", [(300, 1, True), (700, 10, True), (300, 7, False)])
def test_streamed_trajectories_equal_the_copied_ones(monkeypatch, n, stride, resident):
    """want_fields = "stream": the saved u_tot / u_inc planes of an action go to pinned host memory on a copy stream
    while the next action computes.  Same bits as the device-buffer-then-copy path; two streamed actions in
    flight; views stay valid while the next action runs."""
    gc.collect()
    monkeypatch.setenv("WAVES_AMD_FUSED_RESIDENT", "1" if resident else "0")
    steps = 40

    def run(mode, pipelined):
        env, pol = _env(n, steps, 3, 5, trajectory_stride=stride)
        env.return_fields = mode
        outs = []
        if pipelined:
            env.step_begin(pol(env))
            env.step_begin(pol(env))
            outs.append(env.step_end())
            env.step_begin(pol(env))
            first = (outs[0][2].copy(), outs[0][3].copy())
            outs.append(env.step_end())
            outs.append(env.step_end())
            assert np.array_equal(outs[0][2], first[0]) and np.array_equal(outs[0][3], first[1])   # view survived one more begin
        else:
            for _ in range(3):
                outs.append(env(pol(env)))
        res = [(np.array(o[2]), np.array(o[3])) for o in outs]
        assert env.ctx.timing()["resident"] is resident
        env.ctx.close()
        gc.collect()
        return res

    ref = run(True, False)
    assert ref[0][0].shape == (n, n, steps // stride + 1)
    got = run("stream", False)
    got2 = run("stream", True)
    for a, b, c in zip(ref, got, got2):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])


def test_streamed_planes_are_transferred_under_the_next_action():
    """700^2 actions that stream every 10th saved plane (11 x 2 planes = 43 MB per action) against actions that return no
    fields, both as pipelined rollouts: the device-to-host copy of action k runs on its own stream while action k+1
    computes.  Measured on the box of the round (tools/stream_cost.py, profiles/r02/stream_cost.txt): +8 % at stride 10
    (the blocking copy of wv_integrate_end: +410 %); from stride 5 on the PCIe link (~50 GB/s) is the bound."""
    import time
    gc.collect()
    times = {}
    for mode in (False, "stream"):
        env, pol = _env(700, 100, 30, 9, trajectory_stride=10)
        env.return_fields = mode
        for _ in range(3):
            env(pol(env))
        t0 = time.perf_counter()
        w.rollout_pipelined(env, pol, 12)
        times[mode] = time.perf_counter() - t0
        env.ctx.close()
        gc.collect()
    # (round 3: the plain rollout no longer pays a kernel launch per action -- its resident launch stays on the device --
    # while a streamed call still has one launch of its own, behind whose end the copy stream waits: the ratio rose from
    # 1.08 to ~1.2 because the denominator fell from 0.95 to 0.90 ms per action)
    assert times["stream"] < 1.3 * times[False], times
