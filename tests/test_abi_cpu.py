"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/waves_amd.h declares,
and refuses to compute without a GPU (no fallback).  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

import waves_jl_amd as w
from waves_jl_amd import _ffi


def _declared_symbols():
    src = open(_ffi.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wv_[a-z_0-9]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    path = w.build()
    assert os.path.exists(path)
    L = ctypes.CDLL(path)
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/waves_amd.h but not exported"
    assert _ffi.lib().wv_abi_version() == 3


def test_ffi_binds_every_declared_symbol():
    L = _ffi.lib()
    for s in _declared_symbols():
        f = getattr(L, s)
        assert f.argtypes is not None, f"{s} has no ctypes signature in _ffi.py"


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_ffi.wv_config) == 32
    assert ctypes.sizeof(_ffi.wv_timing) == 56   # ABI 3: + gave_up, launch_ms, launch_jobs


def test_no_cpu_fallback(gpu_available):
    if gpu_available:
        pytest.skip("GPU present")
    dim = w.TwoDim(15.0, 64)
    with pytest.raises(w.WavesAmdError) as ei:
        w.Integrator(w.runge_kutta, w.AcousticDynamics(dim, w.WATER, 2.0, 20000.0), 1e-5)
    assert ei.value.status == _ffi.WV_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)
    with pytest.raises(w.WavesAmdError):
        w.WaveEnv(dim, design_space=w.build_triple_ring_design_space(), resolution=(32, 32))


def test_create_argument_validation():
    L = _ffi.lib()
    h = ctypes.c_void_p()
    x = np.linspace(-1, 1, 16).astype(np.float32)
    cfg = _ffi.wv_config(16, 12, 1531.0, 1e-5, 2.0, 2e4, 0, 0)
    rc = L.wv_create(ctypes.byref(cfg), _ffi.fptr(x), _ffi.fptr(x), ctypes.byref(h))
    assert rc == _ffi.WV_ERR_INVALID and b"nx must equal ny" in L.wv_last_error(None)
    cfg = _ffi.wv_config(4, 4, 1531.0, 1e-5, 2.0, 2e4, 0, 0)
    assert L.wv_create(ctypes.byref(cfg), _ffi.fptr(x), _ffi.fptr(x), ctypes.byref(h)) == _ffi.WV_ERR_INVALID
    assert L.wv_create(None, None, None, None) == _ffi.WV_ERR_INVALID


def test_product_never_imports_the_oracle():
    root = os.path.dirname(_ffi.__file__)
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "waves_oracle" not in txt and "c_oracle" not in txt and "oracle/" not in txt, f


def test_c_host_example_builds_against_the_header_alone():
    """examples/host_loop.cpp -- the env(action) loop written against include/waves_amd.h with plain g++ (no HIP headers on
    the caller's side) -- compiles and links against the library (it is RUN in the GPU tests)."""
    import subprocess
    csrc = os.path.dirname(w.build())
    r = subprocess.run(["make", "-C", csrc, "example"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(os.path.join(os.path.dirname(os.path.dirname(csrc)), "examples", "host_loop"))
