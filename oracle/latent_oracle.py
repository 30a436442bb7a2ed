"""CPU ORACLE of the batched 1-D latent dynamics -- TEST INFRASTRUCTURE ONLY, not part of the product.

numpy restatement, from the source text, of
  * `(dyn::AcousticDynamics{OneDim})(x, t, theta)`        /root/reference/src/dynamics.jl:190-222
  * `linear_interp` / `LinearInterpolation`               /root/reference/src/utils.jl:69-98
  * `(source::Source)(t::AbstractVector)`                 /root/reference/src/sources.jl:21-23
  * `build_pml(::OneDim, ...)`, `build_dirichlet(::OneDim)` /root/reference/src/pml.jl:6-15, src/dims.jl:111-115
  * `runge_kutta`, `Integrator` (matrix tspan)            /root/reference/src/dynamics.jl:9-16, 37-49
as the surrogate models drive them (`theta = [C, F, PML]`, /root/reference/src/model/acoustic_energy_model.jl:89-107).
fp32, same temporaries and operation order, no FMA.

PARITY UNPINNED: the reference holds no fixture for this path and cannot be run here (Julia).  Two spots where the Julia
text leaves the rounding to library versions that are not pinned (no Manifest):
  * `dyn.c0 * grad * (U_inc .+ f)` (dynamics.jl:207) -- restated as the generic left fold `(c0 * grad) * (...)`: every
    matrix coefficient is scaled (and rounded) first.  LinearAlgebra >= 1.7 may instead dispatch a 3-argument method;
  * `sum(...; dims)` in linear_interp adds one non-zero term to zeros: exact in any order.
Only tests/ may import this file.
"""
from __future__ import annotations

import numpy as np

import waves_oracle as wo

f32 = np.float32


def build_pml_1d(x: np.ndarray, width, scale) -> np.ndarray:
    """src/pml.jl:6-15."""
    ax = np.abs(x.astype(f32))
    start = f32(min(ax[0], ax[-1]) - f32(width))
    pml = np.maximum(ax - start, f32(0.0)) / f32(width)
    pml = np.clip(pml, f32(0.0), f32(1.0)).astype(f32)
    return ((pml * pml) * pml * f32(scale)).astype(f32)      # pml .^ 3 * scale  (literal_pow: x*x*x)


def build_dirichlet_1d(n: int) -> np.ndarray:
    """src/dims.jl:111-115."""
    bc = np.ones(n, f32)
    bc[[0, -1]] = 0
    return bc


def linear_interp(X: np.ndarray, Y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """src/utils.jl:69-86.  X (K, B), Y (n, K, B), x (B) -> (n, B)."""
    x_row = x[None, :].astype(f32)
    d = (X - x_row).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        dYdX = (np.diff(Y, axis=1) / np.diff(d, axis=0)[None, :, :]).astype(f32)
    l, r = X[:-1, :], X[1:, :]
    final_step = (r == r[[-1], :]) & (r[[-1], :] == x_row)        # r .== r[[end], :] .== x_row  (chained comparison)
    mask = ((l <= x_row) & (x_row < r)) | final_step
    m = mask.astype(f32)
    x0 = np.sum(X[:-1, :] * m, axis=0, keepdims=True, dtype=f32)
    with np.errstate(invalid="ignore"):
        y0 = np.sum(Y[:, :-1, :] * m[None], axis=1, dtype=f32)
        dydx = np.sum(dYdX * m[None], axis=1, dtype=f32)
        return (y0 + (x_row - x0) * dydx).astype(f32)


class LinearInterpolation:
    """src/utils.jl:88-98."""

    def __init__(self, X, Y):
        self.X, self.Y = np.asarray(X, f32), np.asarray(Y, f32)

    def __call__(self, t):
        return linear_interp(self.X, self.Y, np.asarray(t, f32))


class Source1D:
    """Source with an (n, B) shape called with a time vector, src/sources.jl:21-23."""

    def __init__(self, shape, freq):
        self.shape, self.freq = np.asarray(shape, f32), f32(freq)

    def __call__(self, t):
        t = np.asarray(t, f32)
        arg = ((f32(6.2831855) * t[None, :]).astype(f32) * self.freq).astype(f32)   # 2.0f0 * pi * permutedims(t) * freq
        return (self.shape * np.sin(arg.astype(np.float64)).astype(f32)).astype(f32)   # accurately rounded sin(::Float32)


class LatentDynamics:
    """AcousticDynamics{OneDim}: ctor src/dynamics.jl:141-149, call :190-222."""

    def __init__(self, x, c0, pml_width, pml_scale):
        self.x = np.asarray(x, f32)
        self.c0 = f32(c0)
        self.grad = wo.build_gradient(self.x, f32)
        self.pml = build_pml_1d(self.x, pml_width, pml_scale)
        self.bc = build_dirichlet_1d(len(self.x))

    def __call__(self, x, t, theta):
        C, F, PML = theta
        g, c0 = self.grad, self.c0
        sigma = (self.pml[[0]][:, None] * PML).astype(f32)           # pml_scale .* PML
        U_tot, V_tot, U_inc, V_inc = x[:, 0, :], x[:, 1, :], x[:, 2, :], x[:, 3, :]
        c = C(t)
        f = F(t)
        a = (c0 * c).astype(f32)
        dU_tot = a * wo._grad_axis0(g, V_tot) - sigma * U_tot
        dV_tot = a * wo._grad_axis0(g, (U_tot + f).astype(f32)) - sigma * V_tot
        dU_inc = c0 * wo._grad_axis0(g, V_inc) - sigma * U_inc
        gs = wo.Gradient(n=g.n, cm=f32(c0 * g.cm), cp=f32(c0 * g.cp), fwd=(c0 * g.fwd).astype(f32), bwd=(c0 * g.bwd).astype(f32))
        dV_inc = wo._grad_axis0(gs, (U_inc + f).astype(f32)) - sigma * V_inc      # (c0 * grad) * (U_inc .+ f)
        bc = self.bc[:, None]
        return np.stack([dU_tot * bc, dV_tot, dU_inc * bc, dV_inc], axis=1).astype(f32)


def runge_kutta(f, u, t, theta, dt):
    """src/dynamics.jl:9-16 with a time vector."""
    dt = f32(dt)
    hdt = f32(f32(0.5) * dt)
    k1 = f(u, t, theta)
    k2 = f(u + hdt * k1, (t + hdt).astype(f32), theta)
    k3 = f(u + hdt * k2, (t + hdt).astype(f32), theta)
    k4 = f(u + dt * k3, (t + dt).astype(f32), theta)
    du = f32(f32(1) / f32(6.0)) * (((k1 + f32(2) * k2) + f32(2) * k3) + k4)
    return (du * dt).astype(f32)


def integrate(dyn: LatentDynamics, z0, t, theta, dt):
    """(iter::Integrator)(ui, tspan::AbstractMatrix, theta), src/dynamics.jl:37-49.  z0 (n, 4, B), t (steps + 1, B)
    -> (n, 4, B, steps + 1)."""
    with np.errstate(invalid="ignore", over="ignore"):
        u = np.asarray(z0, f32)
        out = [u]
        for i in range(t.shape[0] - 1):
            u = (u + runge_kutta(dyn, u, np.asarray(t[i, :], f32), theta, dt)).astype(f32)
            out.append(u)
        return np.stack(out, axis=3)


def compute_latent_energy(z, dx):
    """src/model/acoustic_energy_model.jl:6-15 -> (steps + 1, 3, B)  (sums in float64, rounded once)."""
    tot, inc = z[:, 0, :, :], z[:, 2, :, :]
    sc = tot - inc
    e = lambda a: (np.sum((a * a).astype(np.float64), axis=0).astype(f32) * f32(dx)).astype(f32)
    return np.transpose(np.stack([e(tot), e(inc), e(sc)], axis=0), (2, 0, 1))
