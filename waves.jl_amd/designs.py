"""Scatterer designs -- host mirror of the reference's src/designs.jl.

Designs are tiny parameter vectors (19 cylinders in the triple-ring space); their vector-space algebra, clamping and
sampling stay on the host in fp32 with the reference's operation order.  What reaches the GPU is the pair
(initial, final) of a `DesignInterpolator`, through `wv_set_design`; the wave-speed field `speed(design, grid, c0)` is
assembled per Runge-Kutta stage inside the integrator kernels.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32

# src/designs.jl:8-13
ALUMINIUM = f32(3100.0)
COPPER = f32(2260.0)
BRASS = f32(2120.0)
AIR = f32(344.0)
WATER = f32(1531.0)


def _f(a, shape=None):
    a = np.array(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


class AbstractDesign:
    """src/designs.jl:35-53: `-`, scalar `*`, `/` derive from `+` and `design * Float32`."""

    def __sub__(self, other):  # :51  d1 + (-1.0f0 * d2)
        return self + other * f32(-1.0)

    def __rmul__(self, n):  # :50
        return self * n

    def __truediv__(self, n):  # :52  design * (1.0f0 / Float32(n))
        return self * (f32(1.0) / f32(n))


class NoDesign(AbstractDesign):
    """src/designs.jl:55-63."""

    def __add__(self, other):
        return NoDesign()

    def __mul__(self, n):
        return NoDesign()

    def zero(self):
        return self

    def clamp(self, low, high):
        # the reference has no clamp(::NoDesign, ...) method: env(action) would throw a MethodError
        raise TypeError("MethodError: no method matching clamp(::NoDesign, ::NoDesign, ::NoDesign)")

    def stacked(self):
        return None

    def __repr__(self):
        return "NoDesign()"


class Cylinders(AbstractDesign):
    """src/designs.jl:69-88.  pos (M, 2), r (M,), c (M,)."""

    def __init__(self, pos, r, c):
        self.pos = _f(pos).reshape(-1, 2)
        self.r = _f(r).reshape(-1)
        self.c = _f(c).reshape(-1)
        if not (len(self.pos) == len(self.r) == len(self.c)):
            raise ValueError("DimensionMismatch: pos, r and c must describe the same number of cylinders")

    @classmethod
    def _raw(cls, pos, r, c):
        """Results of float32 array arithmetic on valid Cylinders: already (M, 2) / (M,) float32 -- no copies, no checks
        (the design algebra builds eight of these per env action)."""
        o = object.__new__(cls)
        o.pos, o.r, o.c = pos, r, c
        return o

    def __add__(self, o):
        if isinstance(o, Cylinders):  # :80
            return Cylinders._raw(self.pos + o.pos, self.r + o.r, self.c + o.c)
        o = f32(o)  # :81
        return Cylinders._raw(self.pos + o, self.r + o, self.c + o)

    __radd__ = __add__

    def __mul__(self, o):
        if isinstance(o, Cylinders):  # :83
            return Cylinders._raw(self.pos * o.pos, self.r * o.r, self.c * o.c)
        o = f32(o)  # :82
        return Cylinders._raw(self.pos * o, self.r * o, self.c * o)

    def zero(self):  # :85
        return self * f32(0.0)

    def __len__(self):  # :86
        return len(self.r)

    def clamp(self, low, high):  # :87
        # clamp.(x, lo, hi) == min(max(x, lo), hi); the ufunc pair costs a quarter of np.clip on 19-element arrays
        return Cylinders._raw(np.minimum(np.maximum(self.pos, low.pos), high.pos), np.minimum(np.maximum(self.r, low.r), high.r),
                              np.minimum(np.maximum(self.c, low.c), high.c))

    def vec(self):  # :88
        return np.concatenate([self.pos.ravel(order="F"), self.r, self.c])

    def stacked(self):
        return self

    def __repr__(self):
        return f"Cylinders(M={len(self)})"


def stack(c1: Cylinders, c2: Cylinders) -> Cylinders:
    """src/designs.jl:133-138."""
    return Cylinders._raw(np.vstack([c1.pos, c2.pos]), np.concatenate([c1.r, c2.r]), np.concatenate([c1.c, c2.c]))


class AbstractScatterers(AbstractDesign):
    """src/designs.jl:147-177."""

    def __init__(self, cylinders: Cylinders):
        self.cylinders = cylinders

    def __add__(self, o):
        if isinstance(o, type(self)):
            return type(self)(self.cylinders + o.cylinders)
        return type(self)(self.cylinders + f32(o))

    __radd__ = __add__

    def __mul__(self, o):
        if isinstance(o, type(self)):
            return type(self)(self.cylinders * o.cylinders)
        return type(self)(self.cylinders * f32(o))

    def zero(self):
        return type(self)(self.cylinders.zero())

    def clamp(self, low, high):
        return type(self)(self.cylinders.clamp(low.cylinders, high.cylinders))

    def __len__(self):
        return len(self.cylinders)

    def stacked(self):
        return self.cylinders

    def __repr__(self):
        return f"{type(self).__name__}({self.cylinders!r})"


class AdjustableRadiiScatterers(AbstractScatterers):
    """src/designs.jl:179-192."""

    def vec(self):
        return self.cylinders.r


class AdjustablePositionScatterers(AbstractScatterers):
    """src/designs.jl:194-208."""

    def vec(self):
        return self.cylinders.pos.ravel(order="F")


class Cloak(AbstractDesign):
    """src/designs.jl:210-228."""

    def __init__(self, config: AbstractScatterers, core: Cylinders):
        self.config = config
        self.core = core

    def vec(self):  # :217
        return self.config.vec()

    def __add__(self, o):
        if isinstance(o, Cloak):  # :219
            return Cloak(self.config + o.config, self.core + o.core)
        if isinstance(o, AbstractScatterers):  # :218  cloak + action
            return Cloak(self.config + o, self.core)
        return Cloak(self.config + f32(o), self.core + f32(o))  # :220

    def __mul__(self, o):
        if isinstance(o, Cloak):  # :222
            return Cloak(self.config * o.config, self.core * o.core)
        return Cloak(self.config * f32(o), self.core * f32(o))  # :221

    def zero(self):  # :225
        return Cloak(self.config.zero(), self.core.zero())

    def clamp(self, low, high):  # :226
        return Cloak(self.config.clamp(low.config, high.config), self.core.clamp(low.core, high.core))

    def stacked(self):  # :228  (designs are immutable values: the stacked form is built once per object)
        st = getattr(self, "_stacked", None)
        if st is None:
            st = self._stacked = stack(self.config.cylinders, self.core)
        return st

    def __repr__(self):
        return f"Cloak({self.config!r}, core={self.core!r})"


class DesignSpace:
    """src/designs.jl:23-33."""

    def __init__(self, low, high):
        self.low = low
        self.high = high

    def __call__(self, design, action):  # :31-33
        return (design + action).clamp(self.low, self.high)

    def rand(self, rng):
        return rand(self, rng)


def _uniform_array_sample(rng, l, r):
    """src/designs.jl:243-251."""
    eps = rng.random(l.shape, dtype=np.float32)
    return eps * (r - l) + l


def rand(space: DesignSpace, rng: np.random.Generator):
    """`rand(::DesignSpace)`: src/designs.jl:253-269 (scripts/data.jl:40-42 for NoDesign).  Julia's RNG stream cannot be
    reproduced outside Julia; the draw is the same distribution from a seeded numpy Generator."""
    low, high = space.low, space.high
    if isinstance(low, NoDesign):
        return NoDesign()
    if isinstance(low, Cylinders):
        return Cylinders(_uniform_array_sample(rng, low.pos, high.pos), _uniform_array_sample(rng, low.r, high.r),
                         _uniform_array_sample(rng, low.c, high.c))
    if isinstance(low, Cloak):
        return Cloak(rand(DesignSpace(low.config, high.config), rng), rand(DesignSpace(low.core, high.core), rng))
    return type(low)(rand(DesignSpace(low.cylinders, high.cylinders), rng))


def build_action_space(design, scale) -> DesignSpace:
    """src/designs.jl:90-94 (Cylinders), :187-192 (radii), :203-208 (positions), :227 (Cloak); scripts/data.jl:44-46."""
    scale = f32(scale)
    if isinstance(design, NoDesign):
        return DesignSpace(NoDesign(), NoDesign())
    if isinstance(design, Cloak):
        return build_action_space(design.config, scale)
    if isinstance(design, Cylinders):
        one = np.ones_like
        return DesignSpace(Cylinders(one(design.pos) * -scale, one(design.r) * -scale, one(design.c) * -scale),
                           Cylinders(one(design.pos) * scale, one(design.r) * scale, one(design.c) * scale))
    s = build_action_space(design.cylinders, scale)
    z = f32(0.0)
    if isinstance(design, AdjustablePositionScatterers):
        return DesignSpace(AdjustablePositionScatterers(Cylinders(s.low.pos, s.low.r * z, s.low.c * z)),
                           AdjustablePositionScatterers(Cylinders(s.high.pos, s.high.r * z, s.high.c * z)))
    return DesignSpace(AdjustableRadiiScatterers(Cylinders(s.low.pos * z, s.low.r, s.low.c * z)),
                       AdjustableRadiiScatterers(Cylinders(s.high.pos * z, s.high.r, s.high.c * z)))


class DesignInterpolator:
    """src/designs.jl:274-292.  On the device path only (initial, final, ti, tf) are used: the interpolation itself is
    done per stage time inside libwaves_amd (`wv_set_design`); `__call__` is the host evaluation for inspection."""

    def __init__(self, initial, final=None, ti=0.0, tf=0.0):
        if final is None:  # :283-285
            final, ti, tf = initial.zero(), 0.0, 0.0
        self.initial, self.final = initial, final
        self.ti, self.tf = f32(ti), f32(tf)

    def __call__(self, t):
        if isinstance(self.initial, NoDesign):
            return NoDesign()
        dt = f32(self.tf - self.ti)
        dt = dt if dt > f32(0.0) else f32(1.0)
        dy = self.final - self.initial
        tau = f32(min(max(f32(t), self.ti), self.tf) - self.ti)
        return self.initial + (dy / dt) * tau

    def __contains__(self, t):  # :294-296
        return self.ti <= f32(t) <= self.tf

    def abi_args(self):
        """(initial, final, ti, tf) in the form Context.set_design takes."""
        a, b = _abi_of(self.initial), _abi_of(self.final)
        if a is None:
            return None, None, self.ti, self.tf
        return a, b, self.ti, self.tf


def _parts(design):
    """The Cylinders a design stacks to, in order (src/designs.jl:133-138, 228), without building the stacked object."""
    if isinstance(design, Cylinders):
        return (design,)
    if isinstance(design, AbstractScatterers):
        return (design.cylinders,)
    if isinstance(design, Cloak):
        return (design.config.cylinders, design.core)
    return None


def _abi_of(design):
    """The design as Context.set_design takes it (unpacks as (pos, r, c)); built once per design object: designs are immutable
    values and each is passed twice, as `final` and then as `initial`."""
    t = getattr(design, "_abi", None)
    if t is None:
        parts = _parts(design)
        if parts is None:
            return None   # NoDesign
        from . import _ffi
        t = design._abi = _ffi._DesignAbi([(q.pos, q.r, q.c) for q in parts])
    return t


def hexagon_ring(r) -> np.ndarray:
    """src/designs.jl:303-311 (Float64 trigonometry, rounded once to Float32)."""
    r = float(f32(r))
    return np.array([[r * np.cos(i * 2 * np.pi / 6.0), r * np.sin(i * 2 * np.pi / 6.0)] for i in range(6)]).astype(np.float32)


def build_2d_rotation_matrix(theta) -> np.ndarray:
    """src/designs.jl:313-319."""
    alpha = theta * np.pi / 180.0
    return np.array([[np.cos(alpha), -np.sin(alpha)], [np.sin(alpha), np.cos(alpha)]])


def _cloak_space(pos, c_value, core_c):
    M = pos.shape[0]
    core = Cylinders([[5.0, 0.0]], [2.0], [core_c])
    low = Cloak(AdjustableRadiiScatterers(Cylinders(pos, np.full(M, 0.2, np.float32), np.full(M, c_value, np.float32))), core)
    high = Cloak(AdjustableRadiiScatterers(Cylinders(pos, np.full(M, 1.0, np.float32), np.full(M, c_value, np.float32))), core)
    return DesignSpace(low, high)


def build_simple_radii_design_space() -> DesignSpace:
    """src/designs.jl:322-335."""
    return _cloak_space(np.zeros((1, 2), np.float32), AIR, AIR)


def build_radii_design_space(pos) -> DesignSpace:
    """src/designs.jl:337-351."""
    speed = f32(3) * AIR
    return _cloak_space(np.asarray(pos, np.float32), speed, speed)


def build_triple_ring_design_space() -> DesignSpace:
    """src/designs.jl:353-365."""
    rot = build_2d_rotation_matrix(30).astype(np.float32)
    ring2 = hexagon_ring(4.75)
    ring2 = (ring2[:, 0:1] * rot[0:1, :]).astype(np.float32) + (ring2[:, 1:2] * rot[1:2, :]).astype(np.float32)
    rings = np.vstack([hexagon_ring(3.5), ring2, hexagon_ring(6.0)])
    return build_radii_design_space(rings + np.array([[5.0, 0.0]], np.float32))
