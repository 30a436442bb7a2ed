"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

numpy restatement of the WaveEnv hot path of gladisor/Waves.jl, written from
the reference's source text (the reference is Julia; no Julia toolchain exists
in this pipeline, so nothing from the reference was ever executed).

PARITY STATUS: pinned only for the finite-difference gradient operator (the
reference's single automated test, test/operators.jl:4-30, is reproduced in
tests/test_oracle_reference_kat.py).  Everything else on the 2-D path
(d/dy, PML, acoustic_dynamics, RK4, speed, build_normal, energies, env
stepping) is **parity unpinned**: the reference holds no golden vector, KAT or
fixture for it.  See DESIGN.md "Oracle".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product (waves.jl_amd/) must never import it.

Every function cites the reference file:line it follows (paths relative to
/root/reference/).  Arithmetic is done in the dtype `T` (np.float32 for the
oracle proper, np.float64 for the round-off twin) with the reference's
operation ORDER: Julia's fused broadcast evaluates `a .+ b .+ c` as
(a+b)+c, `a .* b .* c` as (a*b)*c, never contracts a*b+c into an FMA on the
CPU, and SparseArrays' `A*B` accumulates each output element in ascending
column order starting from zero.
"""
from __future__ import annotations

from dataclasses import dataclass
from fractions import Fraction
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np

f32 = np.float32
f64 = np.float64

# src/designs.jl:8-13
ALUMINIUM = f32(3100.0)
COPPER = f32(2260.0)
BRASS = f32(2120.0)
AIR = f32(344.0)
WATER = f32(1531.0)

FRAMESKIP = 10  # src/env.jl:90


# ----------------------------------------------------------------------------
# Julia range semantics
# ----------------------------------------------------------------------------
def _round_fraction_f32(q: Fraction) -> np.float32:
    """Nearest float32 to an exact rational (round-half-even via float64 guard)."""
    d = float(q)  # correctly rounded to f64
    s = np.float32(d)
    # guard against double rounding: compare the two neighbouring f32 candidates exactly
    lo = np.nextafter(s, np.float32(-np.inf))
    hi = np.nextafter(s, np.float32(np.inf))
    best = s
    bestd = abs(Fraction(float(s)) - q)
    for cand in (lo, hi):
        dd = abs(Fraction(float(cand)) - q)
        if dd < bestd:
            best, bestd = cand, dd
    return np.float32(best)


def julia_range_f32(start, stop, n: int) -> np.ndarray:
    """`collect(range(start::Float32, stop::Float32, n))`.

    Julia evaluates Float32 ranges in twice precision (base/twiceprecision.jl):
    element i is the exact affine interpolant rounded once to Float32, and the
    endpoints are hit exactly.  Restated as exact rational arithmetic on the
    Float32 endpoint values, rounded to nearest.
    """
    a = Fraction(float(np.float32(start)))
    b = Fraction(float(np.float32(stop)))
    if n == 1:
        return np.array([np.float32(start)], dtype=np.float32)
    out = np.empty(n, dtype=np.float32)
    for i in range(n):
        out[i] = _round_fraction_f32(a + (b - a) * Fraction(i, n - 1))
    return out


# ----------------------------------------------------------------------------
# src/dims.jl
# ----------------------------------------------------------------------------
@dataclass
class OneDim:
    """src/dims.jl:6-9, ctor :48-50 (`OneDim(grid_size, n)`)."""
    x: np.ndarray

    @staticmethod
    def from_size(grid_size, n: int) -> "OneDim":
        return OneDim(julia_range_f32(-f32(grid_size), f32(grid_size), n))


@dataclass
class TwoDim:
    """src/dims.jl:12-15, ctor :56-60 (`TwoDim(grid_size::Float32, n::Int)`)."""
    x: np.ndarray
    y: np.ndarray

    @staticmethod
    def from_size(grid_size, n: int) -> "TwoDim":
        x = julia_range_f32(-f32(grid_size), f32(grid_size), n)
        return TwoDim(x, x.copy())

    def size(self) -> Tuple[int, int]:  # dims.jl:70-72
        return (len(self.x), len(self.y))


def build_grid(dim: TwoDim, T=f32) -> np.ndarray:
    """src/dims.jl:92-97 -> (nx, ny, 2): g[i,j,0] = x[i], g[i,j,1] = y[j]."""
    nx, ny = dim.size()
    g = np.empty((nx, ny, 2), dtype=T)
    g[:, :, 0] = dim.x.astype(T)[:, None]
    g[:, :, 1] = dim.y.astype(T)[None, :]
    return g


def build_wave(dim: TwoDim, fields: int, T=f32) -> np.ndarray:
    """src/dims.jl:107-109."""
    return np.zeros(dim.size() + (fields,), dtype=T)


def build_dirichlet(dim: TwoDim, T=f32) -> np.ndarray:
    """src/dims.jl:117-124 (one(dim) :103-105 = ones)."""
    bc = np.ones(dim.size(), dtype=T)
    bc[:, 0] = 0
    bc[0, :] = 0
    bc[:, -1] = 0
    bc[-1, :] = 0
    return bc


def _mean_diff(x: np.ndarray, T=f32):
    """`Flux.mean(diff(x))` in T: sum of the n-1 differences / (n-1).

    Julia's `mean` of a Float32 vector is sum(x)/length with a pairwise sum; the
    sum order is not reproducible bit-for-bit, so accumulate in float64 and
    round once (differs from any fp32 summation order by <= 1 ulp, and the value
    only scales the energy trace).
    """
    d = np.diff(x.astype(T))
    return T(np.sum(d.astype(np.float64)) / (len(x) - 1))


def get_dx(dim, T=f32):  # src/dims.jl:126
    return _mean_diff(dim.x, T)


def get_dy(dim, T=f32):  # src/dims.jl:127
    return _mean_diff(dim.y, T)


# ----------------------------------------------------------------------------
# src/operators.jl
# ----------------------------------------------------------------------------
FORWARD_DIFF_COEF = np.array([-3.0, 4.0, -1.0], dtype=f32)   # operators.jl:3
BACKWARD_DIFF_COEF = np.array([1.0, -4.0, 3.0], dtype=f32)   # operators.jl:4
CENTRAL_DIFF_COEF = np.array([-1.0, 1.0], dtype=f32)         # operators.jl:5


def gradient_dense(x: np.ndarray, T=f32) -> np.ndarray:
    """src/operators.jl:10-22, literally: dense matrix, columns filled, divided by
    2*Delta element-wise, then transposed.  Returned DENSE (n x n); rows are the
    stencils.  Only for small n (tests); the stencil form below is the same
    arithmetic without the matrix."""
    n = len(x)
    grad = np.zeros((n, n), dtype=T)
    delta = T((T(x[-1]) - T(x[0])) / T(n - 1))
    grad[[0, 1, 2], 0] = FORWARD_DIFF_COEF.astype(T)
    grad[[n - 3, n - 2, n - 1], n - 1] = BACKWARD_DIFF_COEF.astype(T)
    for i in range(1, n - 1):
        grad[[i - 1, i + 1], i] = CENTRAL_DIFF_COEF.astype(T)
    return (grad / (T(2) * delta)).T.copy()


@dataclass
class Gradient:
    """The nonzeros of `gradient(x)` (operators.jl:10-22) as stencil coefficients.

    Each coefficient is individually coef / (2*Delta) in T (element-wise matrix
    division at operators.jl:21), so cm == -cp exactly but 4/(2D) is its own
    rounding."""
    n: int
    cm: np.floating   # -1/(2D)   (row i, col i-1)
    cp: np.floating   # +1/(2D)   (row i, col i+1)
    fwd: np.ndarray   # [-3, 4, -1]/(2D)  row 0, cols 0..2
    bwd: np.ndarray   # [ 1,-4,  3]/(2D)  row n-1, cols n-3..n-1
    T: type = f32


def build_gradient(x: np.ndarray, T=f32) -> Gradient:
    """src/operators.jl:24-26 -> gradient(dim.x)."""
    n = len(x)
    delta = T((T(x[-1]) - T(x[0])) / T(n - 1))
    two_d = T(2) * delta
    return Gradient(
        n=n,
        cm=T(T(-1.0) / two_d),
        cp=T(T(1.0) / two_d),
        fwd=(FORWARD_DIFF_COEF.astype(T) / two_d).astype(T),
        bwd=(BACKWARD_DIFF_COEF.astype(T) / two_d).astype(T),
        T=T,
    )


def _grad_axis0(g: Gradient, u: np.ndarray) -> np.ndarray:
    """`grad * u` for u of shape (n, ...): SparseArrays CSC * dense accumulates
    C[i] += A[i,j]*B[j] for ascending j, C zero-initialised (0 + p == p), with
    separately rounded products (no muladd)."""
    out = np.empty_like(u)
    out[1:-1] = g.cm * u[:-2] + g.cp * u[2:]
    out[0] = (g.fwd[0] * u[0] + g.fwd[1] * u[1]) + g.fwd[2] * u[2]
    out[-1] = (g.bwd[0] * u[-3] + g.bwd[1] * u[-2]) + g.bwd[2] * u[-1]
    return out


def dx(g: Gradient, u: np.ndarray) -> np.ndarray:
    """`∂x(∇, u) = ∇ * u`  (operators.jl:45)."""
    return _grad_axis0(g, u)


def dy(g: Gradient, u: np.ndarray) -> np.ndarray:
    """`∂y(∇, u) = (∇ * u')'`  (operators.jl:46)."""
    return _grad_axis0(g, u.T).T


# ----------------------------------------------------------------------------
# src/pml.jl
# ----------------------------------------------------------------------------
def build_pml_profile(xs: np.ndarray, width, scale, T=f32) -> np.ndarray:
    """src/pml.jl:21-29, the 1-D profile that `repeat(x, 1, ny)` tiles along y."""
    x = np.abs(xs.astype(T))
    width = T(width)
    scale = T(scale)
    pml_start = T(x[0] - width)
    region = x > pml_start
    x = x.copy()
    x[~region] = 0
    if region.any():
        x[region] = (x[region] - x[region].min()) / width
    return ((x * x) * x) * scale   # `x .^ 3 * scale`: literal_pow(^,x,Val(3)) = x*x*x


def build_pml(dim: TwoDim, width, scale, T=f32) -> np.ndarray:
    """src/pml.jl:21-29 -> (nx, ny) field sigma_x[i, j] = profile(x_i)."""
    p = build_pml_profile(dim.x, width, scale, T)
    return np.repeat(p[:, None], len(dim.y), axis=1)


# ----------------------------------------------------------------------------
# src/utils.jl  build_normal (2-D)
# ----------------------------------------------------------------------------
def _exp(a: np.ndarray, T) -> np.ndarray:
    """Julia's exp(::Float32) is computed in higher precision and rounded
    (<1 ulp).  Restated as double-precision exp rounded to T."""
    return np.exp(a.astype(np.float64)).astype(T)


def build_normal(grid: np.ndarray, mu: np.ndarray, sigma: np.ndarray, a: np.ndarray, T=f32) -> np.ndarray:
    """src/utils.jl:12-18.  grid (nx,ny,2); mu (K,2); sigma (K,); a (K,)."""
    grid = grid.astype(T)
    mu = np.asarray(mu, dtype=T).reshape(-1, 2)
    sigma = np.asarray(sigma, dtype=T).reshape(-1)
    a = np.asarray(a, dtype=T).reshape(-1)
    two_pi = T(f32(2.0) * f32(np.pi))          # `2.0f0 * π` -> Float32 6.2831855
    out = np.zeros(grid.shape[:2], dtype=T)
    for k in range(len(sigma)):
        ddx = grid[:, :, 0] - mu[k, 0]
        ddy = grid[:, :, 1] - mu[k, 1]
        d2 = ddx * ddx + ddy * ddy             # sum((x .- mu).^2, dims=3)
        s2 = sigma[k] * sigma[k]
        coef = T(1.0) / (two_pi * s2)          # 1 ./ (2f0*pi*sigma.^2)
        e = _exp((-d2) / (T(2.0) * s2), T)
        f = (coef * a[k]) * e                  # (coef .* a) .* exp(...)
        out = out + f                          # sum(f, dims=3), ascending k
    return out


# ----------------------------------------------------------------------------
# src/designs.jl
# ----------------------------------------------------------------------------
class NoDesign:
    """src/designs.jl:55-63."""

    def __add__(self, o):
        return NoDesign()

    def scale(self, n):
        return NoDesign()


@dataclass
class Cylinders:
    """src/designs.jl:69-88."""
    pos: np.ndarray  # (M, 2)
    r: np.ndarray    # (M,)
    c: np.ndarray    # (M,)

    def __post_init__(self):
        self.pos = np.asarray(self.pos).reshape(-1, 2)
        self.r = np.asarray(self.r).reshape(-1)
        self.c = np.asarray(self.c).reshape(-1)

    def astype(self, T):
        return Cylinders(self.pos.astype(T), self.r.astype(T), self.c.astype(T))

    def __add__(self, o):  # :80-81
        if isinstance(o, Cylinders):
            return Cylinders(self.pos + o.pos, self.r + o.r, self.c + o.c)
        return Cylinders(self.pos + o, self.r + o, self.c + o)

    def scale(self, n):  # :82  `cylinders * n`
        return Cylinders(self.pos * n, self.r * n, self.c * n)

    def clamp(self, low, high):  # :87
        return Cylinders(np.clip(self.pos, low.pos, high.pos), np.clip(self.r, low.r, high.r),
                         np.clip(self.c, low.c, high.c))

    def __len__(self):
        return len(self.r)


def stack(c1: Cylinders, c2: Cylinders) -> Cylinders:
    """src/designs.jl:133-138."""
    return Cylinders(np.vstack([c1.pos, c2.pos]), np.concatenate([c1.r, c2.r]), np.concatenate([c1.c, c2.c]))


@dataclass
class AdjustableRadiiScatterers:
    """src/designs.jl:179-192 (algebra :147-173)."""
    cylinders: Cylinders

    def __add__(self, o):
        if isinstance(o, type(self)):
            return type(self)(self.cylinders + o.cylinders)
        return type(self)(self.cylinders + o)

    def scale(self, n):
        return type(self)(self.cylinders.scale(n))

    def clamp(self, low, high):
        return type(self)(self.cylinders.clamp(low.cylinders, high.cylinders))

    def astype(self, T):
        return type(self)(self.cylinders.astype(T))

    def stacked(self) -> Cylinders:
        return self.cylinders


class AdjustablePositionScatterers(AdjustableRadiiScatterers):
    """src/designs.jl:194-208."""


@dataclass
class Cloak:
    """src/designs.jl:210-228."""
    config: AdjustableRadiiScatterers
    core: Cylinders

    def __add__(self, o):
        if isinstance(o, Cloak):                      # :219
            return Cloak(self.config + o.config, self.core + o.core)
        if isinstance(o, AdjustableRadiiScatterers):  # :218  Cloak + action
            return Cloak(self.config + o, self.core)
        return Cloak(self.config + o, self.core + o)  # :220

    def scale(self, n):  # :221
        return Cloak(self.config.scale(n), self.core.scale(n))

    def clamp(self, low, high):  # :226
        return Cloak(self.config.clamp(low.config, high.config), self.core.clamp(low.core, high.core))

    def astype(self, T):
        return Cloak(self.config.astype(T), self.core.astype(T))

    def stacked(self) -> Cylinders:  # :228
        return stack(self.config.cylinders, self.core)


def design_sub(d1, d2, T=f32):
    """`d1 - d2 = d1 + (-1.0f0 * d2)`  (designs.jl:51)."""
    return d1 + d2.scale(T(-1.0))


def design_div(d, n, T=f32):
    """`design / n = design * (1.0f0/Float32(n))`  (designs.jl:52)."""
    return d.scale(T(1.0) / T(n))


def stacked_cylinders(design) -> Optional[Cylinders]:
    if isinstance(design, NoDesign):
        return None
    if isinstance(design, Cylinders):
        return design
    return design.stacked()


@dataclass
class DesignSpace:
    """src/designs.jl:23-33."""
    low: object
    high: object

    def __call__(self, design, action):  # :31-33
        return (design + action).clamp(self.low, self.high)


def _uniform_array_sample(rng, l, r):
    """src/designs.jl:243-251: rand(eltype, size) .* (r .- l) .+ l.  The RNG is
    numpy's (Julia's is not reproducible outside Julia: SURVEY 8a a16)."""
    eps = rng.random(l.shape, dtype=np.float32)
    return eps * (r - l) + l


def rand_design(space: DesignSpace, rng):
    """src/designs.jl:253-269."""
    low, high = space.low, space.high
    if isinstance(low, NoDesign):
        return NoDesign()
    if isinstance(low, Cylinders):
        return Cylinders(_uniform_array_sample(rng, low.pos, high.pos),
                         _uniform_array_sample(rng, low.r, high.r),
                         _uniform_array_sample(rng, low.c, high.c))
    if isinstance(low, Cloak):
        return Cloak(rand_design(DesignSpace(low.config, high.config), rng),
                     rand_design(DesignSpace(low.core, high.core), rng))
    return type(low)(rand_design(DesignSpace(low.cylinders, high.cylinders), rng))


def build_action_space(design, scale) -> DesignSpace:
    """src/designs.jl:90-94, 187-192, 203-208, 227."""
    scale = f32(scale)
    if isinstance(design, Cloak):
        return build_action_space(design.config, scale)
    if isinstance(design, Cylinders):
        one = lambda a: np.ones_like(a)
        low = Cylinders(one(design.pos) * -scale, one(design.r) * -scale, one(design.c) * -scale)
        high = Cylinders(one(design.pos) * scale, one(design.r) * scale, one(design.c) * scale)
        return DesignSpace(low, high)
    s = build_action_space(design.cylinders, scale)
    z = f32(0.0)
    if isinstance(design, AdjustablePositionScatterers):
        low = AdjustablePositionScatterers(Cylinders(s.low.pos, s.low.r * z, s.low.c * z))
        high = AdjustablePositionScatterers(Cylinders(s.high.pos, s.high.r * z, s.high.c * z))
    else:
        low = AdjustableRadiiScatterers(Cylinders(s.low.pos * z, s.low.r, s.low.c * z))
        high = AdjustableRadiiScatterers(Cylinders(s.high.pos * z, s.high.r, s.high.c * z))
    return DesignSpace(low, high)


@dataclass
class DesignInterpolator:
    """src/designs.jl:274-292."""
    initial: object
    final: object
    ti: np.floating
    tf: np.floating

    def __call__(self, t, T=f32):
        if isinstance(self.initial, NoDesign):
            return NoDesign()
        ti, tf, t = T(self.ti), T(self.tf), T(t)
        dt_ = T(tf - ti)
        dt_ = dt_ if dt_ > T(0.0) else T(1.0)
        init = self.initial.astype(T)
        fin = self.final.astype(T)
        dy_ = design_sub(fin, init, T)
        tau = T(min(max(t, ti), tf) - ti)
        return init + design_div(dy_, dt_, T).scale(tau)


def location_mask(cyls: Cylinders, grid: np.ndarray) -> np.ndarray:
    """src/designs.jl:99-104 -> Bool (nx, ny, M)."""
    T = grid.dtype.type
    M = len(cyls)
    pos = cyls.pos.astype(T).T.reshape(1, 1, 2, M)
    r2 = cyls.r.astype(T).reshape(1, 1, M)
    r2 = r2 * r2
    d = grid[:, :, :, None] - pos
    d = d * d
    s = d[:, :, 0, :] + d[:, :, 1, :]
    return s < r2


def speed(design, grid: np.ndarray, ambient_speed):
    """src/designs.jl:110-116 (Cylinders), :176 (scatterers), :228 (Cloak), :63 (NoDesign)."""
    T = grid.dtype.type
    cyls = stacked_cylinders(design)
    if cyls is None:
        return T(ambient_speed)
    mask = location_mask(cyls, grid)
    ambient_mask = mask.sum(axis=2) == 0
    C0 = ambient_mask.astype(T) * T(ambient_speed)
    C_design = np.zeros(grid.shape[:2], dtype=T)
    cc = cyls.c.astype(T)
    for m in range(len(cyls)):                      # sum(mask .* c, dims=3), ascending m
        C_design = C_design + mask[:, :, m].astype(T) * cc[m]
    return C0 + C_design


def hexagon_ring(r) -> np.ndarray:
    """src/designs.jl:303-311.  `(i-1) * 2pi/6.0f0`: Int * (2*pi::Float64 / Float32) is
    Float64; r::Float32 * cos(::Float64) is Float64; Matrix{Float32}(...) rounds once."""
    r = float(f32(r))
    pos = []
    for i in range(1, 7):
        ang = (i - 1) * (2 * np.pi) / 6.0
        pos.append([r * np.cos(ang), r * np.sin(ang)])
    return np.array(pos, dtype=np.float64).astype(f32)


def build_2d_rotation_matrix(theta) -> np.ndarray:
    """src/designs.jl:313-319 (Float64)."""
    alpha = theta * np.pi / 180.0
    return np.array([[np.cos(alpha), -np.sin(alpha)], [np.sin(alpha), np.cos(alpha)]], dtype=np.float64)


def _matmul_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Float32 (6x2)*(2x2): each output = a1*b1 + a2*b2 in fp32.  (Julia's small
    matmul may use muladd; the ring positions are constants of the design space and
    are passed through the ABI as data, so this rounding never reaches the kernels.)"""
    out = np.empty((a.shape[0], b.shape[1]), dtype=f32)
    for i in range(a.shape[0]):
        for j in range(b.shape[1]):
            out[i, j] = f32(a[i, 0] * b[0, j]) + f32(a[i, 1] * b[1, j])
    return out


def build_radii_design_space(pos: np.ndarray) -> DesignSpace:
    """src/designs.jl:337-351."""
    DESIGN_SPEED = f32(3) * AIR
    M = pos.shape[0]
    r_low = np.full(M, 0.2, dtype=f32)
    r_high = np.full(M, 1.0, dtype=f32)
    c = np.full(M, DESIGN_SPEED, dtype=f32)
    core = Cylinders(np.array([[5.0, 0.0]], dtype=f32), np.array([2.0], dtype=f32), np.array([DESIGN_SPEED], dtype=f32))
    low = Cloak(AdjustableRadiiScatterers(Cylinders(pos.copy(), r_low, c.copy())), core)
    high = Cloak(AdjustableRadiiScatterers(Cylinders(pos.copy(), r_high, c.copy())), core)
    return DesignSpace(low, high)


def build_triple_ring_design_space() -> DesignSpace:
    """src/designs.jl:353-365."""
    rot = build_2d_rotation_matrix(30).astype(f32)
    rings = np.vstack([hexagon_ring(3.5), _matmul_f32(hexagon_ring(4.75), rot), hexagon_ring(6.0)])
    pos = rings + np.array([[5.0, 0.0]], dtype=f32)
    return build_radii_design_space(pos.astype(f32))


# ----------------------------------------------------------------------------
# src/sources.jl
# ----------------------------------------------------------------------------
def _sin(arg, T):
    """Julia's sin(::Float32): argument reduction + kernel in Float64, rounded once."""
    return T(np.sin(np.float64(arg)))


def source_time_factor(t, freq, T=f32):
    """`sin.(2.0f0 * pi * permutedims(t) * freq)`  (sources.jl:21-22, 67-69):
    ((2f0*pi) * t) * freq in T, then sin."""
    two_pi = T(f32(2.0) * f32(np.pi))
    arg = T(T(two_pi * T(t)) * T(freq))
    return _sin(arg, T)


class NoSource:
    """src/sources.jl:7-8."""
    shape = None
    freq = f32(0)

    def __call__(self, t, T=f32):
        return T(0.0)

    def reset(self, rng):
        return None


@dataclass
class Source:
    """src/sources.jl:10-23."""
    shape: np.ndarray
    freq: np.floating

    def __call__(self, t, T=f32):
        return self.shape.astype(T) * source_time_factor(t, self.freq, T)

    def reset(self, rng):
        return None


@dataclass
class RandomPosGaussianSource:
    """src/sources.jl:25-69."""
    grid: np.ndarray
    mu_low: np.ndarray
    mu_high: np.ndarray
    sigma: np.ndarray
    a: np.ndarray
    freq: np.floating
    shape: Optional[np.ndarray] = None
    mu: Optional[np.ndarray] = None

    def reset(self, rng):  # :41-51
        lo = np.asarray(self.mu_low, dtype=f32).reshape(-1, 2)
        hi = np.asarray(self.mu_high, dtype=f32).reshape(-1, 2)
        eps = rng.random(lo.shape, dtype=np.float32)
        self.mu = (hi - lo) * eps + lo
        self.shape = build_normal(self.grid, self.mu, self.sigma, self.a)

    def __call__(self, t, T=f32):  # :67-69
        return self.shape.astype(T) * source_time_factor(t, self.freq, T)


# ----------------------------------------------------------------------------
# src/dynamics.jl
# ----------------------------------------------------------------------------
def build_tspan(ti, dt, steps: int) -> np.ndarray:
    """src/dynamics.jl:5-7: collect(range(ti, ti + steps*dt, steps+1)) in Float32."""
    ti = f32(ti)
    stop = f32(ti + f32(f32(steps) * f32(dt)))
    return julia_range_f32(ti, stop, steps + 1)


@dataclass
class AcousticDynamics:
    """src/dynamics.jl:130-149."""
    dim: TwoDim
    c0: np.floating
    grad: Gradient
    pml: np.ndarray   # (nx, ny) sigma_x
    bc: np.ndarray    # (nx, ny)
    T: type = f32

    @staticmethod
    def build(dim: TwoDim, c0, pml_width, pml_scale, T=f32) -> "AcousticDynamics":
        return AcousticDynamics(dim, T(c0), build_gradient(dim.x, T), build_pml(dim, pml_width, pml_scale, T),
                                build_dirichlet(dim, T), T)

    def __call__(self, x: np.ndarray, t, theta) -> np.ndarray:
        """src/dynamics.jl:179-188."""
        C, F = theta
        c = C(t)
        f = F(t)
        dtot = acoustic_dynamics(x[:, :, 0:6], c, f, self.grad, self.pml, self.bc)
        dinc = acoustic_dynamics(x[:, :, 6:12], self.c0, f, self.grad, self.pml, self.bc)
        return np.concatenate([dtot, dinc], axis=2)


def acoustic_dynamics(x, c, f, grad: Gradient, pml, bc) -> np.ndarray:
    """src/dynamics.jl:151-177, same temporaries and evaluation order."""
    U = x[:, :, 0]
    Vx = x[:, :, 1]
    Vy = x[:, :, 2]
    Psix = x[:, :, 3]
    Psiy = x[:, :, 4]
    Om = x[:, :, 5]

    b = c * c                       # c .^ 2

    sx = pml
    sy = pml.T                      # sigma_x'

    Vxx = dx(grad, Vx)
    Vyy = dy(grad, Vy)
    Uf = U + f
    Ux = dx(grad, Uf)
    Uy = dy(grad, Uf)

    dU = (((b * (Vxx + Vyy) + Psix) + Psiy) - (sx + sy) * U) - Om
    dVx = Ux - sx * Vx
    dVy = Uy - sy * Vy
    dPsix = (b * sx) * Vyy
    dPsiy = (b * sy) * Vxx
    dOm = (sx * sy) * U

    return np.stack([bc * dU, dVx, dVy, dPsix, dPsiy, dOm], axis=2)


def runge_kutta(f: Callable, u: np.ndarray, t, theta, dt):
    """src/dynamics.jl:9-16."""
    T = u.dtype.type
    dt = T(dt)
    half = T(0.5)
    hdt = T(half * dt)
    k1 = f(u, t, theta)
    k2 = f(u + hdt * k1, T(t + hdt), theta)
    k3 = f(u + hdt * k2, T(t + hdt), theta)
    k4 = f(u + dt * k3, T(t + dt), theta)
    sixth = T(T(1) / T(6.0))
    du = sixth * (((k1 + T(2) * k2) + T(2) * k3) + k4)
    return du * dt


@dataclass
class Integrator:
    """src/dynamics.jl:18-53."""
    integration_function: Callable
    dynamics: AcousticDynamics
    dt: np.floating

    def __call__(self, ui: np.ndarray, tspan: np.ndarray, theta, save: Optional[Sequence[int]] = None,
                 on_state: Optional[Callable] = None):
        """Returns the states concatenated on a new last axis (dynamics.jl:45-48).
        `save` restricts which time indices are kept (None = all, like the reference);
        `on_state(i, u)` is called for every i (used for energies without a trajectory)."""
        T = ui.dtype.type
        u = ui
        out = {}
        nt = len(tspan)
        if on_state is not None:
            on_state(0, u)
        if save is None or 0 in save:
            out[0] = u
        for i in range(nt - 1):
            du = self.integration_function(self.dynamics, u, T(tspan[i]), theta, self.dt)
            u = u + du
            if on_state is not None:
                on_state(i + 1, u)
            if save is None or (i + 1) in save:
                out[i + 1] = u
        keys = sorted(out)
        return np.stack([out[k] for k in keys], axis=3)


# ----------------------------------------------------------------------------
# src/env.jl
# ----------------------------------------------------------------------------
def energies(u_tot: np.ndarray, u_inc: np.ndarray, dOmega, T=f32) -> np.ndarray:
    """src/env.jl:105-111 for ONE time point -> [tot, inc, sc].  Squares and the
    difference are formed in T exactly as the reference does; the spatial sum (whose
    order Julia leaves unspecified: @simd / GPU tree) is accumulated in float64 and
    rounded once to T, then scaled by dOmega in T."""
    u_sc = u_tot - u_inc
    s = lambda a: T(np.sum((a * a).astype(np.float64)))
    d = T(dOmega)
    return np.array([s(u_tot) * d, s(u_inc) * d, s(u_sc) * d], dtype=T)



def imresize_linear(w, resolution):
    """`imresize(w, resolution)` of src/env.jl:135 for an (nx, ny[, k]) array resized on its first two axes.

    imresize is Images.jl / ImageTransformations (third-party; version unpinned, Manifest git-ignored; not under
    /root/reference).  Its published rule, restated: linear B-spline interpolation of the original, flat beyond the edge
    pixels, sampled pixel-centre aligned at x_o = (n / r) * (i - 0.5) + 0.5 (1-based) on every resized axis, evaluated
    in Float64 and rounded once to the element type; axes that keep their size are copied.  The summation order below
    (x inside, y outside) is this project's definition.  PARITY UNPINNED: no fixture of the reference covers it."""
    w = np.asarray(w)
    nx, ny = w.shape[:2]
    rx, ry = resolution
    xo = np.clip((nx / rx) * (np.arange(1, rx + 1, dtype=np.float64) - 0.5) + 0.5 - 1.0, 0.0, nx - 1.0)
    yo = np.clip((ny / ry) * (np.arange(1, ry + 1, dtype=np.float64) - 0.5) + 0.5 - 1.0, 0.0, ny - 1.0)
    i0 = np.floor(xo).astype(np.int64)
    j0 = np.floor(yo).astype(np.int64)
    i1 = np.minimum(i0 + 1, nx - 1)
    j1 = np.minimum(j0 + 1, ny - 1)
    fx = (xo - i0).reshape((rx, 1) + (1,) * (w.ndim - 2))
    fy = (yo - j0).reshape((1, ry) + (1,) * (w.ndim - 2))
    wd = w.astype(np.float64)
    lo = (1.0 - fx) * wd[i0][:, j0] + fx * wd[i1][:, j0]
    hi = (1.0 - fx) * wd[i0][:, j1] + fx * wd[i1][:, j1]
    return ((1.0 - fy) * lo + fy * hi).astype(w.dtype)


class WaveEnv:
    """src/env.jl:14-121."""

    def __init__(self, dim: TwoDim, *, design_space: DesignSpace, action_speed=250.0, source=None, c0=WATER,
                 pml_width=2.0, pml_scale=20000.0, resolution=(128, 128), dt=1e-5, integration_steps=100,
                 actions=10, rng=None, T=f32):
        assert all(s > r for s, r in zip(dim.size(), resolution)), "Resolution must be less than finite element grid."
        self.T = T
        self.rng = rng if rng is not None else np.random.default_rng(0)
        self.dim = dim
        self.design_space = design_space
        self.design = rand_design(design_space, self.rng)
        self.wave = np.zeros(dim.size() + (12, 3), dtype=T)
        self.source = source if source is not None else NoSource()
        self.iter = Integrator(runge_kutta, AcousticDynamics.build(dim, c0, pml_width, pml_scale, T), f32(dt))
        self.signal = np.zeros(integration_steps + 1, dtype=T)
        self.time_step = 0
        self.resolution = resolution
        self.action_speed = f32(action_speed)
        self.dt = f32(dt)
        self.integration_steps = integration_steps
        self.actions = actions

    def time(self):  # :69-71
        return f32(f32(self.time_step) * self.dt)

    def build_tspan(self):  # :73-75
        return build_tspan(self.time(), self.dt, self.integration_steps)

    def is_terminated(self):  # :77-79
        return self.time_step >= self.actions * self.integration_steps

    def reset(self):  # :81-88
        self.time_step = 0
        self.wave = self.wave * self.T(0.0)
        self.design = rand_design(self.design_space, self.rng)
        self.signal = self.signal * self.T(0.0)
        self.source.reset(self.rng)

    def action_space(self):  # :143-145
        return build_action_space(rand_design(self.design_space, self.rng),
                                  f32(f32(self.action_speed * self.dt) * f32(self.integration_steps)))

    def reward(self):  # :147-149
        return self.signal.sum()

    def state(self):  # :132-137  (x only; dim, tspan and design ride along unchanged)
        shape = getattr(self.source, "shape", None)
        if shape is None:
            shape = np.zeros(self.dim.size(), dtype=self.T)
        w = np.concatenate([self.wave[:, :, 0, :], np.asarray(shape, dtype=self.T)[:, :, None]], axis=2)
        return imresize_linear(w, self.resolution)

    def __call__(self, action, return_fields=True):
        """src/env.jl:91-121."""
        T = self.T
        tspan = self.build_tspan()
        ti = self.time()
        current_design = self.design
        next_design = self.design_space(current_design, action)
        interp = DesignInterpolator(current_design, next_design, ti, tspan[-1])
        grid = build_grid(self.dim, T)
        c0 = self.iter.dynamics.c0
        C = lambda t: speed(interp(t, T), grid, c0)
        F = lambda t: self.source(t, T)

        n = self.integration_steps
        if n < 2 * FRAMESKIP:
            raise IndexError("BoundsError: sol[:, :, :, end-20:10:end] needs integration_steps >= 20")
        frames = {n - 2 * FRAMESKIP, n - FRAMESKIP, n}
        dOmega = T(get_dx(self.dim, T) * get_dy(self.dim, T))
        sig = np.zeros((n + 1, 3), dtype=T)
        utot = np.zeros(self.dim.size() + (n + 1,), dtype=T) if return_fields else None
        uinc = np.zeros(self.dim.size() + (n + 1,), dtype=T) if return_fields else None

        def on_state(i, u):
            sig[i] = energies(u[:, :, 0], u[:, :, 6], dOmega, T)
            if return_fields:
                utot[:, :, i] = u[:, :, 0]
                uinc[:, :, i] = u[:, :, 6]

        sol = self.iter(self.wave[:, :, :, -1], tspan, [C, F], save=frames, on_state=on_state)
        self.signal = sig
        self.design = next_design
        self.wave = sol
        self.time_step += self.integration_steps
        return tspan, interp, utot, uinc


class RandomDesignPolicy:
    """src/env.jl:151-157."""

    def __init__(self, a_space: DesignSpace, rng):
        self.a_space = a_space
        self.rng = rng

    def __call__(self, env):
        return rand_design(self.a_space, self.rng)


# ----------------------------------------------------------------------------
# conversions to the C-ABI memory layout: Julia (x, y, field) column-major ==
# C-order (field, y, x)
# ----------------------------------------------------------------------------
def to_abi(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(np.transpose(a, tuple(range(a.ndim - 1, -1, -1))))


def from_abi(a: np.ndarray) -> np.ndarray:
    return np.transpose(a, tuple(range(a.ndim - 1, -1, -1)))
