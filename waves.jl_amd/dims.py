"""Grid geometry -- host mirror of the reference's src/dims.jl (only what the WaveEnv hot path uses)."""
from __future__ import annotations

from fractions import Fraction

import numpy as np

f32 = np.float32


def _range_f32(start, stop, n: int) -> np.ndarray:
    """`collect(range(start::Float32, stop::Float32, n))`: Julia evaluates Float32 ranges in twice precision, i.e.
    each element is the exact affine interpolant of the two Float32 endpoints rounded once to Float32.  Evaluated in
    float64 (vectorised); the rare elements that land within 1e-9 ulp of a Float32 rounding boundary are redone in exact
    rational arithmetic so the single rounding is honoured."""
    a64, b64 = float(f32(start)), float(f32(stop))
    if n == 1:
        return np.array([f32(start)], dtype=np.float32)
    i = np.arange(n, dtype=np.float64)
    x = a64 + (b64 - a64) * i / (n - 1)
    out = x.astype(np.float32)
    # distance of the float64 value to the midpoint between the two neighbouring float32 values
    other = np.where(out.astype(np.float64) > x, np.nextafter(out, np.float32(-np.inf)), np.nextafter(out, np.float32(np.inf)))
    mid = 0.5 * (out.astype(np.float64) + other.astype(np.float64))
    ulp = np.abs(out.astype(np.float64) - other.astype(np.float64))
    risky = np.abs(x - mid) <= 1e-9 * ulp
    risky[0] = risky[-1] = False
    out[0], out[-1] = f32(start), f32(stop)
    if risky.any():
        # exact comparison of the rational a + (b-a)*k/(n-1) with the midpoint, in integer arithmetic (every float is an
        # integer ratio with a power-of-two denominator)
        an, ad = a64.as_integer_ratio()
        bn, bd = b64.as_integer_ratio()
        for k in np.nonzero(risky)[0]:
            k = int(k)
            # q = (an/ad)*(n-1-k)/(n-1) + (bn/bd)*k/(n-1)  ->  numerator / denominator
            qn = an * bd * (n - 1 - k) + bn * ad * k
            qd = ad * bd * (n - 1)
            mn, md = float(mid[k]).as_integer_ratio()
            lhs, rhs = qn * md, mn * qd          # q ? mid   <=>   lhs ? rhs   (qd, md > 0)
            if lhs == rhs:
                continue                         # an exact tie: float64 -> float32 already rounded half-to-even
            lo_, hi_ = (out[k], other[k]) if out[k] < other[k] else (other[k], out[k])
            out[k] = lo_ if lhs < rhs else hi_
    return out


class TwoDim:
    """src/dims.jl:12-15.  `TwoDim(grid_size, n)` (:56-60), `TwoDim(x, y)`."""

    def __init__(self, a, b):
        if np.isscalar(a) and isinstance(b, (int, np.integer)):
            x = _range_f32(-f32(a), f32(a), int(b))
            self.x, self.y = x, x.copy()
        else:
            self.x = np.ascontiguousarray(a, dtype=np.float32)
            self.y = np.ascontiguousarray(b, dtype=np.float32)

    def size(self, i=None):
        s = (len(self.x), len(self.y))  # src/dims.jl:70-72
        return s if i is None else s[i]

    def __repr__(self):
        return f"TwoDim(nx={len(self.x)}, ny={len(self.y)}, x=[{self.x[0]}, {self.x[-1]}])"


def build_grid(dim: TwoDim) -> np.ndarray:
    """src/dims.jl:92-97 -> (nx, ny, 2)."""
    nx, ny = dim.size()
    g = np.empty((nx, ny, 2), dtype=np.float32, order="F")
    g[:, :, 0] = dim.x[:, None]
    g[:, :, 1] = dim.y[None, :]
    return g


def build_wave(dim: TwoDim, fields: int) -> np.ndarray:
    """src/dims.jl:107-109."""
    return np.zeros(dim.size() + (fields,), dtype=np.float32, order="F")


def build_dirichlet(dim: TwoDim) -> np.ndarray:
    """src/dims.jl:117-124.  (The kernels apply this mask as an index test; this array is for inspection.)"""
    bc = np.ones(dim.size(), dtype=np.float32, order="F")
    bc[:, 0] = 0
    bc[0, :] = 0
    bc[:, -1] = 0
    bc[-1, :] = 0
    return bc


def _mean_diff(x):
    return f32(np.sum(np.diff(x).astype(np.float64)) / (len(x) - 1))


def get_dx(dim) -> np.float32:  # src/dims.jl:126
    return _mean_diff(dim.x)


def get_dy(dim) -> np.float32:  # src/dims.jl:127
    return _mean_diff(dim.y)
