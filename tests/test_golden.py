"""Committed golden vectors (tests/golden/*.npz, produced by tests/golden/make_golden.py with the oracle).
CPU: the oracle still reproduces them bit for bit.  GPU: the HIP path, through the C ABI, reproduces them without the
oracle being involved at all."""
import os

import numpy as np
import pytest

import waves_jl_amd as w

f32 = np.float32
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def test_golden_files_are_small_data_only():
    for f in os.listdir(G):
        if f.endswith(".npz"):
            assert os.path.getsize(os.path.join(G, f)) < 2 << 20
            d = load(f)
            assert all(d[k].dtype.kind in "fiu" for k in d.files)


def test_oracle_reproduces_golden_vectors():
    import c_oracle as co
    import waves_oracle as wo
    g = load("gradient_1d_1024.npz")
    grad = wo.build_gradient(g["x"])
    for k in ("x2", "sin", "exp"):
        assert np.array_equal(wo.dx(grad, g["y_" + k]), g["d_" + k])
    assert np.array_equal(wo.OneDim.from_size(25.0, 1024).x, g["x"])
    d = load("moving_design_64.npz")
    n = int(d["n"])
    dim = wo.TwoDim.from_size(float(d["grid_size"]), n)
    assert np.array_equal(dim.x, d["x"])
    sx = wo.build_pml_profile(dim.x, 2.0, 20000.0)
    assert np.array_equal(sx, d["sigma"])
    flat = lambda p, r, c: np.concatenate([p, r[:, None], c[:, None]], 1).astype(f32)
    st, es, fr = co.integrate(dim.x, dim.y, sx, sx, float(d["c0"]), float(d["dt"]), wo.to_abi(d["u0"]), d["tspan"],
                              G=wo.to_abi(d["source_shape"]), freq=float(d["freq"]), d0=flat(d["pos0"], d["r0"], d["c_0"]),
                              d1=flat(d["pos1"], d["r1"], d["c_1"]), ti=d["tspan"][0], tf=d["tspan"][-1], frame_steps=(0, 10, 20))
    assert np.array_equal(wo.from_abi(fr), d["frames"])
    assert np.array_equal(es.astype(f32) * f32(wo.get_dx(dim) * wo.get_dy(dim)), d["signal"])
    c = load("config1_256.npz")
    assert np.array_equal(wo.build_tspan(0.0, 1e-5, 100), c["tspan"])


@pytest.mark.gpu
def test_hip_reproduces_reference_gradient_vectors():
    """test/operators.jl:4-30 inputs: the 1-D vectors are laid along x (rows identical) and along y"""
    g = load("gradient_1d_1024.npz")
    ctx = w._ffi.Context(g["x"], g["x"], c0=1531.0, dt=1e-5, pml_width=2.0, pml_scale=20000.0)
    for k in ("x2", "sin", "exp"):
        u = np.repeat(g["y_" + k][:, None], 1024, axis=1)
        assert np.array_equal(ctx.gradient(0, u), np.repeat(g["d_" + k][:, None], 1024, axis=1))
        assert np.array_equal(ctx.gradient(1, u.T), np.repeat(g["d_" + k][None, :], 1024, axis=0))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["staged", "fused"])
def test_hip_reproduces_moving_design_golden(impl):
    d = load("moving_design_64.npz")
    ctx = w._ffi.Context(d["x"], d["x"], c0=float(d["c0"]), dt=float(d["dt"]), pml_width=float(d["pml_width"]),
                         pml_scale=float(d["pml_scale"]), impl=impl)
    assert np.array_equal(ctx.pml()[0], d["sigma"])
    ctx.set_source_shape(d["source_shape"], float(d["freq"]))
    ts = d["tspan"]
    ctx.set_design((d["pos0"], d["r0"], d["c_0"]), (d["pos1"], d["r1"], d["c_1"]), ts[0], ts[-1])
    assert np.array_equal(ctx.speed_field(0.0011), d["speed_mid"])
    ctx.set_state(d["u0"])
    sig, _, _ = ctx.integrate(ts, capture_frames=True)
    fr = ctx.get_frames()
    assert np.array_equal(fr, d["frames"])          # states at steps 0, 10, 20  ==  sol[:, :, :, end-20:10:end]
    assert np.allclose(sig, d["signal"], rtol=1e-5, atol=0)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("impl", ["staged", "fused"])
def test_hip_reproduces_config1_golden(impl):
    c = load("config1_256.npz")
    ctx = w._ffi.Context(c["x"], c["x"], c0=1531.0, dt=1e-5, pml_width=2.0, pml_scale=20000.0, impl=impl)
    ctx.set_source_shape(c["source_shape"], float(c["freq"]))
    sig, _, _ = ctx.integrate(c["tspan"])
    u = ctx.get_state()
    assert np.array_equal(u[:, :, 0], c["u_tot"]) and np.array_equal(u[:, :, 6], c["u_inc"])
    assert np.array_equal(u[:, :, 1], c["vx_tot"]) and np.array_equal(u[:, :, 3], c["psi_x"])
    assert np.allclose(sig, c["signal"], rtol=1e-5, atol=0)
    assert abs(float(u.astype(np.float64).sum()) - float(c["checksum"])) <= 1e-9 * abs(float(c["checksum"])) + 1e-12
    ctx.close()
