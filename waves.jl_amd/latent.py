"""Host mirror of the batched 1-D latent dynamics the reference's surrogate models integrate:
`iter(z0, t, [C, F, PML])` with `iter = Integrator(runge_kutta, AcousticDynamics(latent_dim::OneDim, ...), dt)`
(src/model/acoustic_energy_model.jl:89-107; src/dynamics.jl:190-222).  The integration runs on the device
(wv_latent_integrate, csrc/kernels_latent.hip); there is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from .dims import _range_f32

f32 = np.float32


class OneDim:
    """src/dims.jl:6-9, 48-50."""

    def __init__(self, grid_size, n: int):
        self.x = _range_f32(-float(grid_size), float(grid_size), int(n))

    def size(self):
        return (len(self.x),)


class LinearInterpolation:
    """src/utils.jl:88-98: X (K, B) knots, Y (n, K, B) values."""

    def __init__(self, X, Y):
        self.X = np.asfortranarray(X, dtype=f32)
        self.Y = np.asfortranarray(Y, dtype=f32)
        assert self.Y.shape[1:] == self.X.shape


class LatentSource:
    """Source(shape (n, B), freq), src/sources.jl:10-23."""

    def __init__(self, shape, freq):
        self.shape = np.asfortranarray(shape, dtype=f32)
        self.freq = f32(freq)


class LatentIntegrator:
    """Integrator(runge_kutta, AcousticDynamics(latent_dim, c0, pml_width, pml_scale), dt) for a OneDim latent space."""

    def __init__(self, dim: OneDim, c0, pml_width, pml_scale, dt, device=0):
        self.dim, self.c0, self.pml_width, self.pml_scale, self.dt, self.device = dim, f32(c0), f32(pml_width), f32(pml_scale), f32(dt), device

    def __call__(self, z0, t, theta):
        """z0 (n, 4, B), t (steps + 1, B), theta = [C, F, PML] -> z (n, 4, B, steps + 1)."""
        Cint, F, PML = theta
        n = len(self.dim.x)
        z0 = np.asfortranarray(z0, dtype=f32)
        t = np.asfortranarray(t, dtype=f32)
        B, steps = z0.shape[2], t.shape[0] - 1
        PML = np.asfortranarray(PML, dtype=f32)
        assert z0.shape == (n, 4, B) and t.shape == (steps + 1, B) and PML.shape == (n, B)
        assert Cint.Y.shape[0] == n and Cint.X.shape[1] == B and F.shape.shape == (n, B)
        cfg = _ffi.wv_latent_config(n, B, Cint.X.shape[0], steps, float(self.c0), float(self.dt), float(self.pml_width),
                                    float(self.pml_scale), float(F.freq), int(self.device))
        z = np.empty((n, 4, B, steps + 1), f32, order="F")
        L = _ffi.lib()
        x = np.ascontiguousarray(self.dim.x, f32)
        rc = L.wv_latent_integrate(C.byref(cfg), _ffi.fptr(x), _ffi.fptr(Cint.X), _ffi.fptr(Cint.Y), _ffi.fptr(F.shape),
                                   _ffi.fptr(PML), _ffi.fptr(z0), _ffi.fptr(t), _ffi.fptr(z))
        if rc != _ffi.WV_OK:
            raise _ffi.WavesAmdError(rc, (L.wv_last_error(None) or b"").decode())
        return z


    def adjoint_sensitivity(self, z, t, theta, dL_dz):
        """adjoint_sensitivity(iter, z, t, theta, dL_dz), src/dynamics.jl:97-121 -> (dL_dz0 (n, 4, B), grads) with
        grads = {"Y": (n, K, B), "shape": (n, B), "PML": (n, B)}: the parts of theta the reference's models train."""
        Cint, F, PML = theta
        n = len(self.dim.x)
        z = np.asfortranarray(z, dtype=f32)
        adj = np.asfortranarray(dL_dz, dtype=f32)
        t = np.asfortranarray(t, dtype=f32)
        B, steps = z.shape[2], t.shape[0] - 1
        PML = np.asfortranarray(PML, dtype=f32)
        assert z.shape == (n, 4, B, steps + 1) and adj.shape == z.shape and t.shape == (steps + 1, B) and PML.shape == (n, B)
        assert Cint.Y.shape[0] == n and Cint.X.shape[1] == B and F.shape.shape == (n, B)
        K = Cint.X.shape[0]
        cfg = _ffi.wv_latent_config(n, B, K, steps, float(self.c0), float(self.dt), float(self.pml_width),
                                    float(self.pml_scale), float(F.freq), int(self.device))
        gz0 = np.empty((n, 4, B), f32, order="F")
        gY = np.empty((n, K, B), f32, order="F")
        gsh = np.empty((n, B), f32, order="F")
        gp = np.empty((n, B), f32, order="F")
        L = _ffi.lib()
        x = np.ascontiguousarray(self.dim.x, f32)
        rc = L.wv_latent_adjoint(C.byref(cfg), _ffi.fptr(x), _ffi.fptr(Cint.X), _ffi.fptr(Cint.Y), _ffi.fptr(F.shape),
                                 _ffi.fptr(PML), _ffi.fptr(z), _ffi.fptr(t), _ffi.fptr(adj), _ffi.fptr(gz0), _ffi.fptr(gY),
                                 _ffi.fptr(gsh), _ffi.fptr(gp))
        if rc != _ffi.WV_OK:
            raise _ffi.WavesAmdError(rc, (L.wv_last_error(None) or b"").decode())
        return gz0, {"Y": gY, "shape": gsh, "PML": gp}

    def rrule(self, z0, t, theta):
        """Flux.ChainRulesCore.rrule(iter::Integrator, z0, t, theta), src/dynamics.jl:123-128: (z, back)."""
        z = self(z0, t, theta)
        return z, (lambda adj: self.adjoint_sensitivity(z, t, theta, adj))


def compute_latent_energy(z, dx):
    """src/model/acoustic_energy_model.jl:6-15 -> (steps + 1, 3, B)."""
    tot, inc = z[:, 0, :, :], z[:, 2, :, :]
    sc = tot - inc
    e = lambda a: (np.sum((a * a).astype(np.float64), axis=0).astype(f32) * f32(dx)).astype(f32)
    return np.transpose(np.stack([e(tot), e(inc), e(sc)], axis=0), (2, 0, 1))
