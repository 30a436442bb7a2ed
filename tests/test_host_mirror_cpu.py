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
