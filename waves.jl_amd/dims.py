"""Grid geometry -- host mirror of the reference's src/dims.jl (only what the WaveEnv hot path uses)."""
from __future__ import annotations

from fractions import Fraction

import numpy as np

f32 = np.float32


def _range_f32(start, stop, n: int) -> np.ndarray:
    """`collect(range(start::Float32, stop::Float32, n))`: Julia evaluates Float32 ranges in twice precision, i.e.
    each element is the exact affine interpolant of the two Float32 endpoints rounded once to Float32."""
    a, b = Fraction(float(f32(start))), Fraction(float(f32(stop)))
    out = np.empty(n, dtype=np.float32)
    if n == 1:
        out[0] = f32(start)
        return out
    for i in range(n):
        q = a + (b - a) * Fraction(i, n - 1)
        s = f32(float(q))
        best, bd = s, abs(Fraction(float(s)) - q)
        for cand in (np.nextafter(s, f32(-np.inf)), np.nextafter(s, f32(np.inf))):
            d = abs(Fraction(float(cand)) - q)
            if d < bd:
                best, bd = cand, d
        out[i] = best
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
