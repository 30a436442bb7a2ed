"""ctypes binding of oracle/libwaves_oracle.so (C restatement of the reference hot path).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/waves_oracle.py.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwaves_oracle.so")
_lib = None

_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    src = os.path.join(_HERE, "waves_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "libwaves_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.wo_build_pml_profile.argtypes = [C.c_int, _fp, C.c_float, C.c_float, _fp]
        L.wo_design_at.argtypes = [C.c_int, _fp, _fp, C.c_float, C.c_float, C.c_float, _fp]
        L.wo_speed_field.argtypes = [C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, C.c_float, _fp]
        L.wo_source_factor.argtypes = [C.c_float, C.c_float]
        L.wo_source_factor.restype = C.c_float
        L.wo_rhs.argtypes = [C.c_int, C.c_int, _fp, _fp, _fp, C.c_float, _fp, _fp, _fp, C.c_float, _fp]
        L.wo_gradient.argtypes = [C.c_int, C.c_int, _fp, C.c_int, _fp, _fp]
        L.wo_integrate.argtypes = [C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_float, C.c_float, _fp, _fp, C.c_int, _fp,
                                   C.c_float, C.c_int, _fp, _fp, C.c_float, C.c_float, _dp, _fp, _ip, C.c_int, C.c_int]
        L.wo_integrate.restype = C.c_int
        L.wo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _f(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a: Optional[np.ndarray]):
    return a.ctypes.data_as(_fp) if a is not None else None


def pml_profile(xs, width, scale) -> np.ndarray:
    xs = _f(xs)
    out = np.empty_like(xs)
    lib().wo_build_pml_profile(len(xs), _p(xs), width, scale, _p(out))
    return out


def design_at(d0, d1, ti, tf, t) -> np.ndarray:
    d0, d1 = _f(d0), _f(d1)
    out = np.empty_like(d0)
    lib().wo_design_at(d0.shape[0], _p(d0), _p(d1), ti, tf, t, _p(out))
    return out


def speed_field(x, y, cyl, c0) -> np.ndarray:
    """cyl: (M,4) px,py,r,c.  Returns (ny, nx) C-order == Julia (nx, ny) column-major."""
    x, y, cyl = _f(x), _f(y), _f(cyl)
    out = np.empty((len(y), len(x)), dtype=np.float32)
    lib().wo_speed_field(len(x), len(y), _p(x), _p(y), cyl.shape[0], _p(cyl), c0, _p(out))
    return out


def source_factor(t, freq) -> np.float32:
    return np.float32(lib().wo_source_factor(t, freq))


def rhs(x, sx, sy, c0, state, cfield=None, G=None, sfac=0.0) -> np.ndarray:
    """state: (12, ny, nx).  Returns k (12, ny, nx)."""
    x, sx, sy, state = _f(x), _f(sx), _f(sy), _f(state)
    ny, nx = state.shape[1:]
    cfield = _f(cfield) if cfield is not None else None
    G = _f(G) if G is not None else None
    out = np.empty_like(state)
    lib().wo_rhs(nx, ny, _p(x), _p(sx), _p(sy), c0, _p(state), _p(cfield), _p(G), sfac, _p(out))
    return out


def gradient(x, axis, u) -> np.ndarray:
    """u: (ny, nx) C-order; axis 0 = along x, 1 = along y."""
    x, u = _f(x), _f(u)
    out = np.empty_like(u)
    lib().wo_gradient(u.shape[1], u.shape[0], _p(x), axis, _p(u), _p(out))
    return out


def integrate(x, y, sx, sy, c0, dt, state, tspan, G=None, freq=0.0, d0=None, d1=None, ti=0.0, tf=0.0,
              frame_steps: Sequence[int] = (), want_energy=True, nthreads=1):
    """Returns (final_state (12,ny,nx), esum (nsteps+1,3) float64 raw sums or None, frames (nf,12,ny,nx))."""
    x, y, sx, sy, tspan = _f(x), _f(y), _f(sx), _f(sy), _f(tspan)
    state = _f(state).copy()
    ny, nx = state.shape[1:]
    nsteps = len(tspan) - 1
    G = _f(G) if G is not None else None
    M = 0 if d0 is None else int(np.asarray(d0).shape[0])
    d0a = _f(d0) if M else None
    d1a = _f(d1) if M else None
    esum = np.zeros((nsteps + 1, 3), dtype=np.float64) if want_energy else None
    fs = np.asarray(list(frame_steps), dtype=np.int32)
    frames = np.zeros((len(fs), 12, ny, nx), dtype=np.float32) if len(fs) else None
    rc = lib().wo_integrate(nx, ny, _p(x), _p(y), _p(sx), _p(sy), c0, dt, _p(state), _p(tspan), nsteps, _p(G), freq, M,
                            _p(d0a), _p(d1a), ti, tf, esum.ctypes.data_as(_dp) if esum is not None else None,
                            _p(frames), fs.ctypes.data_as(_ip) if len(fs) else None, len(fs), nthreads)
    if rc != 0:
        raise MemoryError("wo_integrate failed")
    return state, esum, frames


def max_threads() -> int:
    return int(lib().wo_max_threads())
