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
    """src/utils.jl:69-86.  X (K, B), Y (n, K, B), x (B) -> (n, B).  (dtype of X: fp32 as the reference, or the fp64 twin)"""
    f32 = X.dtype.type
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

    def __init__(self, X, Y, T=f32):
        self.X, self.Y = np.asarray(X, T), np.asarray(Y, T)

    def __call__(self, t):
        return linear_interp(self.X, self.Y, np.asarray(t, self.X.dtype))


class Source1D:
    """Source with an (n, B) shape called with a time vector, src/sources.jl:21-23."""

    def __init__(self, shape, freq, T=f32):
        self.shape, self.freq = np.asarray(shape, T), T(freq)

    def __call__(self, t):
        f32 = self.shape.dtype.type
        t = np.asarray(t, f32)
        arg = ((f32(6.2831855) * t[None, :]).astype(f32) * self.freq).astype(f32)   # 2.0f0 * pi * permutedims(t) * freq
        return (self.shape * np.sin(arg.astype(np.float64)).astype(f32)).astype(f32)   # accurately rounded sin(::Float32)


class LatentDynamics:
    """AcousticDynamics{OneDim}: ctor src/dynamics.jl:141-149, call :190-222."""

    def __init__(self, x, c0, pml_width, pml_scale, T=f32):
        self.T = T
        self.x = np.asarray(x, T)
        self.c0 = T(c0)
        self.grad = wo.build_gradient(self.x, T)
        self.pml = build_pml_1d(np.asarray(x, f32), pml_width, pml_scale).astype(T)
        self.bc = build_dirichlet_1d(len(self.x)).astype(T)

    def __call__(self, x, t, theta):
        f32 = self.T   # (the fp64 twin only serves the finite-difference checks of the adjoint)
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
    f32 = getattr(f, "T", np.float32)
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


# ----------------------------------------------------------------------------------------------------------------------
# adjoint_sensitivity  (/root/reference/src/dynamics.jl:97-128)
#
# The reference obtains the vector-Jacobian products from Zygote (`Flux.pullback` of one `runge_kutta` call).  What is
# restated here is the mathematics of that pullback, derived by hand from the forward text above -- the discrete adjoint
# of the RK4 increment -- in fp32 with the forward's own temporaries.  PARITY UNPINNED twice over: Zygote's accumulation
# order is an implementation detail of a library version that is not pinned, and no fixture exists.  The restatement is
# pinned instead by finite differences of the forward oracle in float64 (tests/test_latent.py).
# Gradients are produced for what the reference's models train through this call: z0, C.Y, F.shape and PML
# (LinearInterpolation trains Y only, utils.jl:94; X and freq come from the time grid / a constant).
def _gradT_axis0(g, w):
    """`grad' * w` for w of shape (n, ...): column i collects row i-1 (cp), row i+1 (cm) and the one-sided rows 0 / n-1."""
    out = np.zeros_like(w)
    out[2:] = out[2:] + g.cp * w[1:-1]            # rows 1 .. n-2, entry (j, j+1)
    out[:-2] = out[:-2] + g.cm * w[1:-1]          # rows 1 .. n-2, entry (j, j-1)
    for k in range(3):
        out[k] = out[k] + g.fwd[k] * w[0]
        out[-3 + k] = out[-3 + k] + g.bwd[k] * w[-1]
    return out


def _interp_weights(X, t):
    """mask (K-1, B), x0 (1, B), dX (K-1, B) of linear_interp at the time vector t."""
    x_row = t[None, :].astype(X.dtype)
    l, r = X[:-1, :], X[1:, :]
    final_step = (r == r[[-1], :]) & (r[[-1], :] == x_row)
    mask = (((l <= x_row) & (x_row < r)) | final_step).astype(X.dtype)
    x0 = np.sum(X[:-1, :] * mask, axis=0, keepdims=True, dtype=X.dtype)
    d = X - x_row
    return mask, x0, np.diff(d, axis=0)


def _vjp_dynamics(dyn, x, t, theta, q, T=f32):
    """Pullback of `dyn(x, t, theta)` for the cotangent q (n, 4, B): (xbar, Ybar, shapebar, PMLbar)."""
    C, F, PML = theta
    g, c0 = dyn.grad, T(dyn.c0)
    sigma = (dyn.pml[[0]][:, None] * PML).astype(T)
    U_tot, V_tot, U_inc, V_inc = x[:, 0, :], x[:, 1, :], x[:, 2, :], x[:, 3, :]
    c = C(t)
    f = F(t)
    a = (c0 * c).astype(T)
    bc = dyn.bc[:, None]
    g1, q1, g3, q3 = q[:, 0, :] * bc, q[:, 1, :], q[:, 2, :] * bc, q[:, 3, :]
    gs = wo.Gradient(n=g.n, cm=T(c0 * g.cm), cp=T(c0 * g.cp), fwd=(c0 * g.fwd).astype(T), bwd=(c0 * g.bwd).astype(T))
    GT_w0 = _gradT_axis0(g, (a * q1).astype(T))
    GT_w3 = _gradT_axis0(gs, q3)
    Ut = GT_w0 - sigma * g1
    Vt = _gradT_axis0(g, (a * g1).astype(T)) - sigma * q1
    Ui = GT_w3 - sigma * g3
    Vi = _gradT_axis0(g, (c0 * g3).astype(T)) - sigma * q3
    fbar = GT_w0 + GT_w3
    abar = g1 * wo._grad_axis0(g, V_tot) + q1 * wo._grad_axis0(g, (U_tot + f).astype(T))
    cbar = (c0 * abar).astype(T)
    sbar = -(((g1 * U_tot + q1 * V_tot) + g3 * U_inc) + q3 * V_inc)
    # c = y0 + (t - x0) * dydx,  y0 = Y[:, k],  dydx = (Y[:, k+1] - Y[:, k]) / dX[k]
    mask, x0, dX = _interp_weights(C.X, np.asarray(t, T))
    with np.errstate(divide="ignore", invalid="ignore"):
        wr = (cbar * (np.asarray(t, T)[None, :] - x0))[:, None, :] / dX[None, :, :]     # d c / d Y[:, k+1] * cbar
    wr = np.where(mask[None] != 0, wr, T(0)).astype(T)
    Ybar = np.zeros_like(C.Y)
    Ybar[:, 1:, :] += wr
    Ybar[:, :-1, :] += np.where(mask[None] != 0, cbar[:, None, :], T(0)) - wr
    arg = ((T(6.2831855) * np.asarray(t, T)[None, :]).astype(T) * T(F.freq)).astype(T)
    shapebar = (fbar * np.sin(arg.astype(np.float64)).astype(T)).astype(T)
    PMLbar = (dyn.pml[[0]][:, None] * sbar).astype(T)
    return np.stack([Ut, Vt, Ui, Vi], axis=1).astype(T), Ybar.astype(T), shapebar, PMLbar


def _vjp_runge_kutta(dyn, u, t, theta, dt, lam, T=f32):
    """Pullback of `runge_kutta(dyn, u, t, theta, dt)` (the increment du) for the cotangent lam."""
    dt = T(dt)
    hdt = T(T(0.5) * dt)
    ts = [t, (t + hdt).astype(T), (t + hdt).astype(T), (t + dt).astype(T)]
    k1 = dyn(u, ts[0], theta)
    y2 = (u + hdt * k1).astype(T)
    k2 = dyn(y2, ts[1], theta)
    y3 = (u + hdt * k2).astype(T)
    k3 = dyn(y3, ts[2], theta)
    y4 = (u + dt * k3).astype(T)
    sb = ((lam * dt) * T(T(1) / T(6.0))).astype(T)
    kb = [sb, (T(2) * sb).astype(T), (T(2) * sb).astype(T), sb]
    ys = [u, y2, y3, y4]
    zb = np.zeros_like(u)
    Yb = shb = pb = None
    for S in (3, 2, 1, 0):
        yb, Y_, s_, p_ = _vjp_dynamics(dyn, ys[S], ts[S], theta, kb[S], T)
        zb = (zb + yb).astype(T)
        if S > 0:
            kb[S - 1] = (kb[S - 1] + (dt if S == 3 else hdt) * yb).astype(T)
        Yb = Y_ if Yb is None else (Yb + Y_).astype(T)
        shb = s_ if shb is None else (shb + s_).astype(T)
        pb = p_ if pb is None else (pb + p_).astype(T)
    return zb, Yb, shb, pb


def adjoint_sensitivity(dyn: LatentDynamics, z, t, theta, adj, dt, T=f32):
    """src/dynamics.jl:97-121: the reverse sweep over ALL saved times (the loop includes the last one, as written).
    z, adj (n, 4, B, steps + 1), t (steps + 1, B) -> (dL/dz0 (n, 4, B), dL/dY (n, K, B), dL/dshape (n, B), dL/dPML (n, B))."""
    with np.errstate(invalid="ignore", over="ignore"):
        lam = (adj[:, :, :, -1] * T(0)).astype(T)
        gY = gsh = gp = None
        for i in reversed(range(z.shape[3])):
            lam = (lam + adj[:, :, :, i]).astype(T)
            zb, Yb, shb, pb = _vjp_runge_kutta(dyn, z[:, :, :, i].astype(T), np.asarray(t[i, :], T), theta, dt, lam, T)
            lam = (lam + zb).astype(T)
            gY = Yb if gY is None else (gY + Yb).astype(T)
            gsh = shb if gsh is None else (gsh + shb).astype(T)
            gp = pb if gp is None else (gp + pb).astype(T)
        return lam, gY, gsh, gp
