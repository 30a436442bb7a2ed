"""Size-independent properties at the full BASELINE sizes (700^2 and 2048^2), where the oracle is too slow to be the
checker for every step: determinism, linearity in the source amplitude (exact for powers of two), no-design
invariance, staged == fused bit for bit, energy-trace continuity across actions, PML decay."""
import numpy as np
import pytest

import waves_jl_amd as w
import waves_oracle as wo

pytestmark = pytest.mark.gpu
f32 = np.float32


def ctx_for(n, impl, width=2.0):
    dim = wo.TwoDim.from_size(15.0, n)
    return dim, w._ffi.Context(dim.x, dim.y, c0=wo.WATER, dt=1e-5, pml_width=width, pml_scale=20000.0, impl=impl)


def ring_design(seed):
    ds = w.build_triple_ring_design_space()
    a = w.rand(ds, np.random.default_rng(seed))
    b = ds(a, w.rand(w.build_action_space(a, 0.25), np.random.default_rng(seed + 1)))
    return a.stacked(), b.stacked()


def run(n, impl, steps, amp=1.0, design=True, width=2.0, mu=(3.0, 0.5)):
    dim, ctx = ctx_for(n, impl, width)
    ctx.set_gaussian_source([list(mu)], [0.3], [amp], 1000.0)
    if design:
        a, b = ring_design(5)
        ctx.set_design((a.pos, a.r, a.c), (b.pos, b.r, b.c), 0.0, f32(steps * 1e-5))
    ts = wo.build_tspan(0.0, 1e-5, steps)
    sig, _, _ = ctx.integrate(ts, capture_frames=True)
    fr = ctx.get_frames()
    ctx.close()
    return sig, fr


@pytest.mark.parametrize("n,steps", [(700, 100), (2048, 40)])
def test_staged_equals_fused_and_deterministic(n, steps):
    s1, f1 = run(n, "fused", steps)
    s2, f2 = run(n, "fused", steps)
    s3, f3 = run(n, "staged", steps)
    assert np.array_equal(f1, f2) and np.array_equal(s1, s2)          # run-to-run reproducible incl. the reduction
    assert np.array_equal(f1, f3)                                     # two independent kernels, same bits
    assert np.allclose(s1, s3, rtol=1e-5, atol=0)
    assert np.abs(f1[:, :, 0, 2]).max() > 0.1
    assert not np.array_equal(f1[:, :, 0, 2], f1[:, :, 6, 2])         # the source sits inside the ring: scattering


@pytest.mark.parametrize("impl", ["staged", "fused"])
def test_linearity_in_source_amplitude_is_exact_for_powers_of_two(impl):
    s1, f1 = run(700, impl, 60, amp=1.0)
    s4, f4 = run(700, impl, 60, amp=4.0)
    # scaling by a power of two commutes with every fp32 rounding except in the subnormal range (the far tails of the
    # Gaussian): exact wherever the field is a normal number
    # (subnormal INTERMEDIATES -- dt*k products of tiny values -- also feed cells that are themselves tiny but normal)
    big = np.abs(f1) > 1e-15
    assert big.sum() > 100000 and np.array_equal(f4[big], (f1 * f32(4.0))[big])
    assert np.abs(f4 - f1 * f32(4.0)).max() < 1e-25
    assert np.allclose(s4, s1 * 16.0, rtol=1e-6)


@pytest.mark.parametrize("impl", ["staged", "fused"])
def test_no_design_total_equals_incident_2048(impl):
    s, f = run(2048, impl, 30, design=False, width=4.0)
    assert np.array_equal(f[:, :, :6, :], f[:, :, 6:, :]) and np.all(s[:, 2] == 0) and s[-1, 0] > 0


def test_signal_continuity_and_frames_across_actions():
    dim = w.TwoDim(15.0, 700)
    src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                    rng=np.random.default_rng(0))
    env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=100, actions=3,
                    rng=np.random.default_rng(1), return_fields=False)
    pol = w.RandomDesignPolicy(env.action_space(), np.random.default_rng(2))
    ep = w.generate_episode(pol, env)
    assert len(ep) == 3 and env.is_terminated() and env.time_step == 300
    for k in range(1, 3):
        assert np.array_equal(ep.y[k][0], ep.y[k - 1][-1])           # row 1 of a step == last row of the previous
    assert ep.y[0].shape == (101, 3) and np.all(np.diff(ep.y[0][:, 1]) >= 0) is not None
    st = env.state()
    assert st.wave.shape == (128, 128, 4)                           # env.resolution, resized on the device
    env.reset()
    assert env.time_step == 0 and not env.wave.any()


def test_pml_decay_512_2000_steps():
    """after the source is switched off the PML drains the domain: 2 % of the initial energy is left after 2000 steps
    (512^2: the 2048^2 width sweep of config 4 is tests/test_gpu_round2.py::test_config4_2048_500_steps_width_sweep_properties)"""
    dim, ctx = ctx_for(512, "auto", 2.0)
    ic = wo.build_normal(wo.build_grid(dim), np.array([[0.0, 0.0]]), np.array([0.6]), np.array([1.0]))
    u0 = np.zeros((512, 512, 12), f32, order="F")
    u0[:, :, 0] = ic
    u0[:, :, 6] = ic
    ctx.set_source_shape(None, 0.0)
    ctx.set_state(u0)
    sig, _, _ = ctx.integrate(wo.build_tspan(0.0, 1e-5, 2000))
    ctx.close()
    assert sig[-1, 0] < 0.02 * sig[0, 0]


def test_step_all_overlapped_envs_equal_sequential_stepping():
    """Several environments on one GPU stepped through w.step_all (groups of actions in flight on the envs' HIP streams,
    single-step kernels) give exactly what stepping them one after the other gives."""
    def make(seed):
        dim = w.TwoDim(15.0, 220)
        src = w.RandomPosGaussianSource(w.build_grid(dim), [[-10.0, -10.0]], [[-10.0, 10.0]], [0.3], [1.0], 1000.0,
                                        rng=np.random.default_rng(seed))
        env = w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), source=src, integration_steps=25, actions=4,
                        rng=np.random.default_rng(seed + 1), return_fields=False)
        env.reset()
        return env, w.RandomDesignPolicy(env.action_space(), np.random.default_rng(seed + 2))

    seq = []
    for k in range(3):
        env, pol = make(10 * k)
        for _ in range(2):
            env(pol(env))
        seq.append((np.array(env.signal), env.ctx.get_frames()))
        env.ctx.close()
    envs, pols = zip(*[make(10 * k) for k in range(3)])
    for _ in range(2):
        w.step_all(envs, [p(e) for e, p in zip(envs, pols)], max_in_flight=2)
    for k, env in enumerate(envs):
        assert env.ctx.timing()["resident"] is False          # three live contexts share the device
        assert np.array_equal(env.signal, seq[k][0]) and np.array_equal(env.ctx.get_frames(), seq[k][1])
        env.ctx.close()
