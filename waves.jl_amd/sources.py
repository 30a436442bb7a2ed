"""Sources of acoustic energy -- host mirror of the reference's src/sources.jl.

The spatial shape lives on the GPU (it is read by every Runge-Kutta stage); the time factor sin(2f0*pi*t*freq) is one
scalar per stage, computed by libwaves_amd on the host and passed to the kernels as an argument.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


class NoSource:
    """src/sources.jl:7-8."""
    freq = f32(0.0)

    def reset(self, rng=None):
        return None

    def attach(self, ctx):
        ctx.set_source_shape(None, 0.0)

    def __call__(self, t):
        return f32(0.0)


class Source:
    """src/sources.jl:10-23: static shape (nx, ny) times sin(2 pi t freq)."""

    def __init__(self, shape, freq):
        self.shape = np.asfortranarray(shape, dtype=np.float32)
        self.freq = f32(freq)
        self._ctx = None

    def reset(self, rng=None):
        return None

    def attach(self, ctx):
        self._ctx = ctx
        ctx.set_source_shape(self.shape, self.freq)

    def __call__(self, t):
        if self._ctx is None:
            raise RuntimeError("Source is not attached to a device environment")
        return self._ctx.source_field(t)


class RandomPosGaussianSource:
    """src/sources.jl:25-69.  `grid` is accepted for signature parity (the kernels take coordinates from the ctx)."""

    def __init__(self, grid, mu_low, mu_high, sigma, a, freq, rng=None):
        self.grid = grid
        self.mu_low = np.asarray(mu_low, np.float32).reshape(-1, 2)
        self.mu_high = np.asarray(mu_high, np.float32).reshape(-1, 2)
        self.sigma = np.asarray(sigma, np.float32).reshape(-1)
        self.a = np.asarray(a, np.float32).reshape(-1)
        self.freq = f32(freq)
        self.rng = rng if rng is not None else np.random.default_rng()
        self._ctx = None
        self.mu = self.mu_high.copy()  # ctor builds with mu_high then calls reset! (:61-64)
        self.reset()

    def reset(self, rng=None):  # :41-51
        rng = rng if rng is not None else self.rng
        eps = rng.random(self.mu_low.shape, dtype=np.float32)
        self.mu = (self.mu_high - self.mu_low) * eps + self.mu_low
        if self._ctx is not None:
            self._ctx.set_gaussian_source(self.mu, self.sigma, self.a, self.freq)

    def attach(self, ctx):
        self._ctx = ctx
        ctx.set_gaussian_source(self.mu, self.sigma, self.a, self.freq)

    @property
    def shape(self):
        if self._ctx is None:
            raise RuntimeError("RandomPosGaussianSource is not attached to a device environment")
        return self._ctx.source_shape()

    def __call__(self, t):  # :67-69
        if self._ctx is None:
            raise RuntimeError("RandomPosGaussianSource is not attached to a device environment")
        return self._ctx.source_field(t)
