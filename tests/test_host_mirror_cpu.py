"""Host logic of the mirror (design algebra, spaces, interpolator arguments, tspan) against the oracle's restatement of
the same reference code.  CPU only."""
import numpy as np
import pytest

import waves_jl_amd as w
import waves_oracle as wo
from helpers import oracle_to_mirror_design

f32 = np.float32


def _eq(a, b):
    a, b = a.stacked(), wo.stacked_cylinders(b)
    return np.array_equal(a.pos, b.pos) and np.array_equal(a.r, b.r) and np.array_equal(a.c, b.c)


def test_grid_and_tspan_match_oracle():
    for n in (256, 700):
        d, o = w.TwoDim(15.0, n), wo.TwoDim.from_size(15.0, n)
        assert np.array_equal(d.x, o.x) and np.array_equal(d.y, o.y)
        assert w.get_dx(d) == wo.get_dx(o)
    assert np.array_equal(w.build_grid(w.TwoDim(15.0, 32)), wo.build_grid(wo.TwoDim.from_size(15.0, 32)))
    for ts in (0, 100, 1900):
        ti = f32(f32(ts) * f32(1e-5))
        assert np.array_equal(w.build_tspan(ti, 1e-5, 100), wo.build_tspan(ti, 1e-5, 100))
    assert np.array_equal(w.build_dirichlet(w.TwoDim(15.0, 16)), wo.build_dirichlet(wo.TwoDim.from_size(15.0, 16)))


def test_design_spaces_and_algebra_match_oracle():
    ds, os_ = w.build_triple_ring_design_space(), wo.build_triple_ring_design_space()
    assert _eq(ds.low, os_.low) and _eq(ds.high, os_.high)
    assert len(ds.low.stacked()) == 19
    a = w.rand(ds, np.random.default_rng(7))
    oa = wo.rand_design(os_, np.random.default_rng(7))
    assert _eq(a, oa)
    asp = w.build_action_space(a, 0.25)
    oasp = wo.build_action_space(oa, 0.25)
    act = w.rand(asp, np.random.default_rng(8))
    oact = wo.rand_design(oasp, np.random.default_rng(8))
    assert np.array_equal(act.cylinders.r, oact.cylinders.r) and np.all(act.cylinders.pos == 0)
    b, ob = ds(a, act), os_(oa, oact)
    assert _eq(b, ob)
    it, oit = w.DesignInterpolator(a, b, 0.001, 0.002), wo.DesignInterpolator(oa, ob, f32(0.001), f32(0.002))
    for t in (0.0005, 0.001, 0.00137, 0.002, 0.003):
        assert _eq(it(t), oit(f32(t)))
    (p0, r0, c0), (p1, r1, c1), ti, tf = it.abi_args()
    assert p0.shape == (19, 2) and r1.shape == (19,) and ti == f32(0.001) and tf == f32(0.002)
    assert (f32(0.0015) in it) and not (f32(0.0025) in it)
    # vector-space identities of src/designs.jl:47-53
    d = (b - a)
    assert np.array_equal(d.stacked().r, b.stacked().r + a.stacked().r * f32(-1))
    h = d / f32(4)
    assert np.array_equal(h.stacked().r, d.stacked().r * (f32(1) / f32(4)))


def test_nodesign_has_no_clamp_like_the_reference():
    sp = w.DesignSpace(w.NoDesign(), w.NoDesign())
    assert isinstance(w.rand(sp, np.random.default_rng(0)), w.NoDesign)
    with pytest.raises(TypeError):
        sp(w.NoDesign(), w.NoDesign())


def test_position_scatterers_action_space():
    cyl = w.Cylinders([[0, 0], [1, 1]], [0.5, 0.5], [1000, 1000])
    sp = w.build_action_space(w.AdjustablePositionScatterers(cyl), 0.1)
    assert np.all(sp.low.cylinders.pos == f32(-0.1)) and np.all(sp.high.cylinders.r == 0)
    o = wo.build_action_space(wo.AdjustablePositionScatterers(wo.Cylinders(cyl.pos, cyl.r, cyl.c)), 0.1)
    assert np.array_equal(sp.low.cylinders.pos, o.low.cylinders.pos)


def test_oracle_to_mirror_roundtrip():
    os_ = wo.build_triple_ring_design_space()
    m = oracle_to_mirror_design(w, os_.low)
    assert isinstance(m, w.Cloak) and _eq(m, os_.low)


# ---- Episode container, prepare_data windowing, on-disk episode format (src/data.jl, src/utils.jl) ----
def _toy_episode(A=5, n=6, with_states=True):
    rng = np.random.default_rng(11)
    ds = w.build_triple_ring_design_space()
    dim = w.TwoDim(15.0, 16)
    s, a, t, y = [], [], [], []
    t0 = 0.0
    design = w.rand(ds, rng)
    for k in range(A):
        ts = (t0 + 1e-5 * np.arange(n)).astype(np.float32)
        if with_states:
            s.append(w.env.WaveEnvState(dim, ts, np.asfortranarray(rng.standard_normal((8, 8, 4)).astype(np.float32)), design))
        a.append(w.rand(w.build_action_space(design, 0.25), rng))
        t.append(ts)
        sig = rng.standard_normal((n, 3)).astype(np.float32)
        if y:
            sig[0] = y[-1][-1]          # row 1 of a step equals the last row of the previous step (src/env.jl:105-111)
        y.append(sig)
        t0 = float(ts[-1])
    return w.Episode(s, a, t, y)


def test_flatten_repeated_last_dim_follows_the_reference():
    """src/utils.jl:20-35: first segment whole, later segments without their (repeated) first sample, segment-major."""
    x = np.arange(2 * 4 * 3, dtype=np.float32).reshape(2, 4, 3)          # (rows, n = 4 samples, k = 3 segments)
    got = w.flatten_repeated_last_dim(x)
    want = np.stack([np.concatenate([x[r, :, 0], x[r, 1:, 1], x[r, 1:, 2]]) for r in range(2)])
    assert got.shape == (2, 4 + 3 * 2) and np.array_equal(got, want)
    v = np.stack([np.array([0, 1, 2, 3], np.float32), np.array([3, 4, 5, 6], np.float32)], axis=1)   # hcat of two tspans
    assert np.array_equal(w.flatten_repeated_last_dim(v), np.arange(7, dtype=np.float32))
    both = w.flatten_repeated_last_dim([x[0], x[1]])                       # Vector{<:AbstractMatrix} method: hcat
    assert both.shape == (10, 2) and np.array_equal(both[:, 1], want[1])


def test_prepare_data_windows_like_the_reference():
    ep = _toy_episode(A=5, n=6)
    s, a, t, y = w.prepare_data(ep, 3)
    assert len(s) == len(a) == len(t) == len(y) == 3                       # length(ep) - (horizon - 1)
    for i in range(3):
        assert s[i] is ep.s[i] and a[i] == list(ep.a[i:i + 3])
        assert t[i].shape == (6 + 5 * 2,) and y[i].shape == (6 + 5 * 2, 3)
        assert np.array_equal(t[i], np.concatenate([ep.t[i], ep.t[i + 1][1:], ep.t[i + 2][1:]]))
        assert np.array_equal(y[i], np.concatenate([ep.y[i], ep.y[i + 1][1:], ep.y[i + 2][1:]]))
        assert np.all(np.diff(t[i]) > 0)                                    # no repeated boundary sample left
    s2, a2, t2, y2 = w.prepare_data([ep, ep], 3)                            # vcat over episodes (src/data.jl:59-62)
    assert len(y2) == 6 and np.array_equal(y2[3], y[0])
    assert w.prepare_data(ep, 1)[3][0].shape == (6, 3)


def test_episode_file_round_trip_without_pickle(tmp_path):
    ep = _toy_episode(A=4, n=5)
    p = str(tmp_path / "episode.npz")
    ep.save(p)
    z = np.load(p, allow_pickle=False)                                      # plain arrays + a JSON manifest only
    assert z["y"].shape == (4, 5, 3) and z["s_wave"].shape == (4, 8, 8, 4)
    back = w.Episode.load(p)
    assert len(back) == 4
    for k in range(4):
        assert np.array_equal(back.y[k], ep.y[k]) and np.array_equal(back.t[k], ep.t[k])
        assert np.array_equal(back.s[k].wave, ep.s[k].wave) and np.array_equal(back.s[k].tspan, ep.s[k].tspan)
        for d0, d1 in ((back.a[k], ep.a[k]), (back.s[k].design, ep.s[k].design)):
            assert type(d0) is type(d1)
            c0, c1 = d0.stacked(), d1.stacked()
            assert np.array_equal(c0.pos, c1.pos) and np.array_equal(c0.r, c1.r) and np.array_equal(c0.c, c1.c)
    assert np.array_equal(back.s[0].dim.x, ep.s[0].dim.x)
    bare = _toy_episode(A=2, n=5, with_states=False)
    bare.save(p)
    assert w.Episode.load(p).s == [] and len(w.Episode.load(p)) == 2


class _RecordingCtx:
    """Stands in for the device context: records what WaveEnv hands to the ABI."""

    def __init__(self):
        self.calls, self._rows = [], []

    def set_design(self, initial, final, ti, tf):
        self.calls.append(("design", initial, final, np.float32(ti), np.float32(tf)))

    def integrate_begin(self, tspan, **kw):
        self.calls.append(("begin", np.array(tspan, np.float32)))
        self._rows.append(len(tspan))

    def set_design_sequence(self, designs, ti_tf, steps_per_action):
        self.calls.append(("sequence", designs, np.array(ti_tf, np.float32), steps_per_action))

    def integrate_sequence_begin(self, tspans, **kw):
        ts = np.array(tspans, np.float32)
        self.calls.append(("sequence_begin", ts))
        self._rows.append(ts.shape[0] * (ts.shape[1] - 1) + 1)

    def integrate_end(self):
        n = self._rows.pop(0)
        return np.arange(3 * n, dtype=np.float32).reshape(n, 3), None, None


def _bare_env(seed, steps=30, actions=4):
    env = object.__new__(w.WaveEnv)   # (the constructor opens a device context: not on a CPU-only machine)
    env.rng = np.random.default_rng(seed)
    env.design_space = w.build_triple_ring_design_space()
    env.design = w.rand(env.design_space, env.rng)
    env.ctx = _RecordingCtx()
    env.signal = np.zeros(steps + 1, np.float32)
    env.time_step, env.dt, env.integration_steps, env.actions = 0, np.float32(1e-5), steps, actions
    env.action_speed, env.return_fields, env._pending = np.float32(250.0), False, None
    return env, w.RandomDesignPolicy(env.action_space(), np.random.default_rng(seed + 1))


def test_action_sequence_hands_the_abi_what_the_per_action_loop_does():
    """WaveEnv.steps_begin (one device call for n actions, wv_set_design_sequence) against n x step_begin: the same designs,
    interpolation intervals and tspans reach the ABI, the bookkeeping ends in the same place, steps_end cuts the
    (n*steps + 1)-row trace into the n overlapping (steps + 1)-row signals."""
    n, steps = 4, 30
    a, pa = _bare_env(5, steps, n)
    for _ in range(n):
        a.step_begin(pa(a))
    b, pb = _bare_env(5, steps, n)
    sigs = w.rollout_batched(b, pb, n)
    designs_a = [c for c in a.ctx.calls if c[0] == "design"]
    begins_a = [c for c in a.ctx.calls if c[0] == "begin"]
    (_, designs_b, titf_b, sps), (_, tspans_b) = b.ctx.calls
    assert sps == steps and len(designs_b) == n + 1 and tspans_b.shape == (n, steps + 1)
    for k in range(n):
        _, ini, fin, ti, tf = designs_a[k]
        for got, want in ((designs_b[k], ini), (designs_b[k + 1], fin)):
            assert all(np.array_equal(g, x) for g, x in zip(got, want))
        assert titf_b[k, 0] == ti and titf_b[k, 1] == tf
        assert np.array_equal(tspans_b[k], begins_a[k][1])
    assert a.time_step == b.time_step == n * steps
    assert np.array_equal(a.design.stacked().r, b.design.stacked().r)
    assert len(sigs) == n and all(s.shape == (steps + 1, 3) for s in sigs)
    for k in range(n):   # rows k*steps .. (k+1)*steps of the whole trace; neighbours share their boundary row
        assert sigs[k][0, 0] == 3 * k * steps and sigs[k][-1, 0] == 3 * (k + 1) * steps
    assert np.array_equal(b.signal, sigs[-1])


def test_generate_episode_pipelines_only_policies_that_say_they_do_not_read_the_state():
    """the reference's loop (src/data.jl:22-27) is strictly sequential; two actions in flight are only valid for a
    policy that does not look at the wave state.  The default depth follows the policy's own `reads_state` attribute
    (RandomDesignPolicy: False); an exception in a step leaves nothing pending in the env."""
    import waves_jl_amd as w

    class FakeEnv:
        def __init__(self, fail_at=None):
            self.actions, self.integration_steps, self.time_step, self.return_fields = 4, 20, 0, False
            self.log, self.signal, self._pending, self.fail_at = [], np.zeros((21, 3), np.float32), [], fail_at

        def is_terminated(self):
            return self.time_step >= self.actions * self.integration_steps

        def reset(self):
            self.time_step = 0

        def build_tspan(self):
            return np.arange(21, dtype=np.float32)

        def step_begin(self, a):
            if self.fail_at is not None and len([x for x in self.log if x == "b"]) == self.fail_at:
                raise RuntimeError("boom")
            self.log.append("b")
            self._pending.append(a)
            self.time_step += self.integration_steps

        def step_end(self):
            self.log.append("e")
            self._pending.pop(0)

    class StatePolicy:          # says nothing: assumed to read the state
        def __call__(self, env):
            assert not env._pending, "policy called while an action is pending"
            return 0

    class BlindPolicy:
        reads_state = False

        def __call__(self, env):
            return 0

    env = FakeEnv()
    w.generate_episode(StatePolicy(), env)
    assert "".join(env.log) == "bebebebe"
    env = FakeEnv()
    w.generate_episode(BlindPolicy(), env)
    assert "".join(env.log) == "bbebebee"
    env = FakeEnv()
    w.generate_episode(BlindPolicy(), env, in_flight=1)
    assert "".join(env.log) == "bebebebe"
    assert w.RandomDesignPolicy.reads_state is False
    env = FakeEnv(fail_at=2)
    with pytest.raises(RuntimeError):
        w.generate_episode(BlindPolicy(), env)
    assert not env._pending and env.return_fields is False
