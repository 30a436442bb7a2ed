"""Dynamics + time integrator -- host mirror of the reference's src/dynamics.jl (2-D path).

`AcousticDynamics` + `Integrator` own one device context (wv_ctx): the finite-difference operators, PML profile,
Dirichlet mask and the Runge-Kutta loop all execute as HIP kernels inside libwaves_amd.so.
"""
from __future__ import annotations

import functools
from typing import Optional, Sequence

import numpy as np

from . import _ffi
from .designs import AbstractDesign, Cylinders, DesignInterpolator, NoDesign
from .dims import TwoDim, _range_f32
from .sources import NoSource

f32 = np.float32


@functools.lru_cache(maxsize=4096)
def _tspan_cached(ti: float, dt: float, steps: int):
    stop = f32(f32(ti) + f32(f32(steps) * f32(dt)))
    t = _range_f32(f32(ti), stop, steps + 1)
    t.setflags(write=False)
    return t


def build_tspan(ti, dt, steps: int) -> np.ndarray:
    """src/dynamics.jl:5-7: collect(range(ti, ti + steps*dt, steps + 1)) in Float32.  (Memoised: every episode walks
    through the same (ti, dt, steps) triples.)"""
    return _tspan_cached(float(f32(ti)), float(f32(dt)), int(steps)).copy()


def runge_kutta(*_args, **_kw):
    """src/dynamics.jl:9-16.  Marker for `Integrator(runge_kutta, dyn, dt)`: classical RK4 is the integration function
    the device kernels implement (the four stages are fused in one kernel); it cannot be called on host arrays."""
    raise RuntimeError("runge_kutta runs on the device inside Integrator; call the Integrator")


class AcousticDynamics:
    """src/dynamics.jl:130-149.  Holds the constructor arguments; the operators are built on the device when an
    `Integrator` binds a time step to it (`wv_create`)."""

    def __init__(self, dim: TwoDim, c0, pml_width, pml_scale, *, device: int = 0, impl: str = "auto"):
        if not isinstance(dim, TwoDim):
            raise TypeError("only AcousticDynamics{TwoDim} is on the MI355X hot path (SURVEY 8a a8)")
        self.dim = dim
        self.c0 = f32(c0)
        self.pml_width = f32(pml_width)
        self.pml_scale = f32(pml_scale)
        self.device = device
        self.impl = impl


class UniformSpeed:
    """`C = t -> c` with a scalar c (scripts/pml.jl:16).  c == dyn.c0 is NoDesign; any other value is expressed as one
    cylinder that covers the whole grid."""

    def __init__(self, c):
        self.c = f32(c)


class Integrator:
    """src/dynamics.jl:18-53."""

    def __init__(self, integration_function, dynamics: AcousticDynamics, dt):
        if integration_function is not runge_kutta:
            raise NotImplementedError("only runge_kutta (src/dynamics.jl:9-16) is implemented on the device")
        self.integration_function = integration_function
        self.dynamics = dynamics
        self.dt = f32(dt)
        d = dynamics
        self.ctx = _ffi.Context(d.dim.x, d.dim.y, c0=d.c0, dt=self.dt, pml_width=d.pml_width, pml_scale=d.pml_scale,
                                device=d.device, impl=d.impl)

    def build_tspan(self, ti, steps: int):  # :26
        return build_tspan(ti, self.dt, steps)

    # --- theta = [C, F] ---------------------------------------------------------------------------------------
    def _bind_theta(self, theta, t_first, t_last):
        C, F = theta
        dyn = self.dynamics
        if C is None or isinstance(C, NoDesign):
            self.ctx.set_design(None, None, t_first, t_last)
        elif isinstance(C, UniformSpeed) or np.isscalar(C):
            c = C.c if isinstance(C, UniformSpeed) else f32(C)
            if c == dyn.c0:
                self.ctx.set_design(None, None, t_first, t_last)
            else:
                big = f32(4.0) * f32(max(np.abs(dyn.dim.x).max(), np.abs(dyn.dim.y).max()) + 1.0)
                cyl = ([[0.0, 0.0]], [big], [c])
                self.ctx.set_design(cyl, cyl, t_first, t_last)
        elif isinstance(C, DesignInterpolator):
            self.ctx.set_design(*C.abi_args())
        elif isinstance(C, AbstractDesign):  # a static design
            s = C.stacked()
            self.ctx.set_design((s.pos, s.r, s.c), (s.pos, s.r, s.c), t_first, t_last)
        else:
            raise TypeError("C must be a DesignInterpolator, a design, NoDesign/None or a scalar speed: closures cannot "
                            "cross the C ABI (include/waves_amd.h)")
        (F if F is not None else NoSource()).attach(self.ctx)

    def __call__(self, ui, tspan, theta, *, save: Optional[Sequence[int]] = None):
        """`iter(ui, tspan, theta)`: returns the states at the saved time indices stacked on a new last axis
        (all of them, like the reference, when `save` is None -- one device round trip per step: meant for tests;
        WaveEnv uses the single-launch-sequence path with energies and frame capture instead)."""
        ts = np.ascontiguousarray(tspan, np.float32).reshape(-1)
        n = len(ts) - 1
        self._bind_theta(theta, ts[0], ts[-1])
        self.ctx.set_state(ui)
        keep = set(range(n + 1)) if save is None else set(int(s) for s in save)
        out = {}
        if 0 in keep:
            out[0] = np.asfortranarray(ui, dtype=np.float32).copy(order="F")
        i = 0
        for stop in sorted(k for k in keep if k > 0):
            self.ctx.integrate(ts[i:stop + 1], want_signal=False)
            out[stop] = self.ctx.get_state()
            i = stop
        if i < n:
            self.ctx.integrate(ts[i:], want_signal=False)
        ks = sorted(out)
        return np.stack([out[k] for k in ks], axis=3) if ks else None

    def rhs(self, x, t, theta):
        """`dyn(x, t, theta)` (src/dynamics.jl:179-188) evaluated on the device."""
        self._bind_theta(theta, f32(t), f32(t))
        if isinstance(theta[0], DesignInterpolator):
            self.ctx.set_design(*theta[0].abi_args())
        return self.ctx.rhs(x, t)
