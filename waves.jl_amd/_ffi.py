"""ctypes binding of libwaves_amd.so -- the C ABI declared in include/waves_amd.h.

There is NO fallback: if the HIP library is missing or no gfx950 device is usable, every compute entry point raises
(`WavesAmdError`).  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("WAVES_AMD_LIB") or os.path.join(CSRC, "libwaves_amd.so")  # override: A/B builds
HEADER_PATH = os.path.normpath(os.path.join(_HERE, "..", "include", "waves_amd.h"))

WV_OK, WV_ERR_INVALID, WV_ERR_HIP, WV_ERR_NO_DEVICE, WV_ERR_NOMEM, WV_ERR_STATE = range(6)
WV_IMPL_AUTO, WV_IMPL_STAGED, WV_IMPL_FUSED = 0, 1, 2
IMPLS = {"auto": WV_IMPL_AUTO, "staged": WV_IMPL_STAGED, "fused": WV_IMPL_FUSED}


class WavesAmdError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__(f"libwaves_amd status {status}: {msg}")
        self.status = status


class wv_config(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("c0", C.c_float), ("dt", C.c_float), ("pml_width", C.c_float),
                ("pml_scale", C.c_float), ("device", C.c_int), ("impl", C.c_int)]


class wv_latent_config(C.Structure):
    _fields_ = [("n", C.c_int), ("batch", C.c_int), ("knots", C.c_int), ("steps", C.c_int), ("c0", C.c_float),
                ("dt", C.c_float), ("pml_width", C.c_float), ("pml_scale", C.c_float), ("freq", C.c_float), ("device", C.c_int)]


class wv_timing(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("step_kernel_ms", C.c_double), ("step_kernel_launches", C.c_int),
                ("steps", C.c_int), ("impl", C.c_int), ("resident", C.c_int), ("gave_up", C.c_int), ("launch_ms", C.c_double),
                ("launch_jobs", C.c_int)]


_fp = C.POINTER(C.c_float)
_vp = C.c_void_p
_lib = None


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source of the library for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout)
        print(r.stderr)
    if r.returncode != 0:
        raise RuntimeError("building libwaves_amd.so failed")
    return LIB_PATH


def _sig(L, name, argtypes, restype=C.c_int):
    f = getattr(L, name)
    f.argtypes = argtypes
    f.restype = restype
    return f


def lib():
    """Load libwaves_amd.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WavesAmdError(-1, f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    ctx = _vp
    _sig(L, "wv_abi_version", [])
    _sig(L, "wv_last_error", [ctx], C.c_char_p)
    _sig(L, "wv_device_count", [C.POINTER(C.c_int)])
    _sig(L, "wv_create", [C.POINTER(wv_config), _fp, _fp, C.POINTER(ctx)])
    _sig(L, "wv_destroy", [ctx])
    _sig(L, "wv_get_pml", [ctx, _fp, _fp])
    _sig(L, "wv_set_pml", [ctx, _fp, _fp])
    _sig(L, "wv_get_cell_area", [ctx, _fp])
    _sig(L, "wv_set_frames", [ctx, _fp])
    _sig(L, "wv_get_frames", [ctx, _fp])
    _sig(L, "wv_set_state", [ctx, _fp])
    _sig(L, "wv_get_state", [ctx, _fp])
    _sig(L, "wv_reset", [ctx])
    _sig(L, "wv_set_source_shape", [ctx, _fp, C.c_float])
    _sig(L, "wv_set_gaussian_source", [ctx, C.c_int, _fp, _fp, _fp, C.c_float])
    _sig(L, "wv_get_source_shape", [ctx, _fp])
    _sig(L, "wv_observation", [ctx, C.c_int, C.c_int, _fp])
    # (the six array arguments as plain addresses: no pointer objects to build per action, see _DesignAbi)
    _sig(L, "wv_set_design", [ctx, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float])
    _sig(L, "wv_set_design_sequence", [ctx, C.c_int, C.c_int, C.c_int, _fp, _fp])
    _sig(L, "wv_observation_action", [ctx, C.c_int, C.c_int, C.c_int, _fp])
    _sig(L, "wv_get_frames_action", [ctx, C.c_int, _fp])
    _sig(L, "wv_speed_field", [ctx, C.c_float, _fp])
    _sig(L, "wv_source_field", [ctx, C.c_float, _fp])
    _sig(L, "wv_gradient", [ctx, C.c_int, _fp, _fp])
    _sig(L, "wv_rhs", [ctx, _fp, C.c_float, _fp])
    _sig(L, "wv_integrate", [ctx, _fp, C.c_int, C.c_int, _fp, _fp, _fp])
    _sig(L, "wv_integrate_begin", [ctx, _fp, C.c_int, C.c_int, C.c_int, C.c_int])
    _sig(L, "wv_integrate_end", [ctx, _fp, _fp, _fp])
    _sig(L, "wv_pending", [ctx, C.POINTER(C.c_int)])
    _sig(L, "wv_integrate_end_view", [ctx, _fp, C.POINTER(_fp), C.POINTER(_fp), C.POINTER(C.c_int)])
    _sig(L, "wv_set_trajectory_stride", [ctx, C.c_int])
    _sig(L, "wv_set_profiling", [ctx, C.c_int])
    _sig(L, "wv_get_timing", [ctx, C.POINTER(wv_timing)])
    _sig(L, "wv_get_call_times", [ctx, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)])
    _sig(L, "wv_set_stream", [ctx, _vp])
    _sig(L, "wv_synchronize", [ctx])
    _sig(L, "wv_device_frames", [ctx, C.POINTER(_vp), C.POINTER(C.c_size_t)])
    _sig(L, "wv_release_device_frames", [ctx])
    _sig(L, "wv_latent_integrate", [C.POINTER(wv_latent_config), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp])
    _sig(L, "wv_latent_adjoint", [C.POINTER(wv_latent_config)] + [_fp] * 12)
    _sig(L, "wv_selftest_granules", [ctx, C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)])
    _sig(L, "wv_device_source_shape", [ctx, C.POINTER(_vp), C.POINTER(C.c_size_t)])
    _lib = L
    return L


def fptr(a):
    return a.ctypes.data_as(_fp) if a is not None else None


def f32c(a) -> np.ndarray:
    """float32, contiguous in whatever order the array already has (C or F)."""
    a = np.asarray(a, dtype=np.float32)
    if a.flags.f_contiguous or a.flags.c_contiguous:
        return a
    return np.ascontiguousarray(a)


def fortran(a, shape=None) -> np.ndarray:
    """float32 array in the reference's (column-major) memory layout."""
    a = np.asfortranarray(a, dtype=np.float32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


class _DesignAbi:
    """One stacked design in the form wv_set_design takes: pos (M, 2) column-major, r (M,), c (M,) -- ONE float32 buffer
    [px | py | r | c] filled straight from the design's parts (a Cloak: its configuration, then its core; src/designs.jl:133-138,
    228), and the three addresses.  Designs are immutable values and every design is handed over twice (as `final`, then as
    `initial`): built once.  (The stacked Cylinders object, three concatenations and three pointer casts per action that this
    replaces were a quarter of the host's time between two actions of a state-dependent loop.)"""
    __slots__ = ("buf", "M", "ptrs")

    def __init__(self, parts):
        M = 0
        for q in parts:
            M += len(q[1])
        buf = np.empty(4 * M, np.float32)
        o = 0
        for pos, r, c in parts:
            m = len(r)
            if m == 0:
                continue
            pos = np.asarray(pos, np.float32).reshape(-1, 2)
            buf[o:o + m] = pos[:, 0]
            buf[M + o:M + o + m] = pos[:, 1]
            buf[2 * M + o:2 * M + o + m] = r
            buf[3 * M + o:3 * M + o + m] = c
            o += m
        self.buf, self.M = buf, M
        a = buf.ctypes.data
        self.ptrs = [a, a + 8 * M, a + 12 * M]

    def __iter__(self):   # (pos, r, c) = abi: what Context.set_design also accepts as a plain tuple
        return iter((self.pos, self.r, self.c))

    def __getitem__(self, k):
        return (self.pos, self.r, self.c)[k]

    def __len__(self):
        return 3

    @property
    def pos(self):
        return self.buf[:2 * self.M].reshape(2, self.M).T   # (M, 2), column-major like the reference's

    @property
    def r(self):
        return self.buf[2 * self.M:3 * self.M]

    @property
    def c(self):
        return self.buf[3 * self.M:]


def design_abi(pos, r, c) -> "_DesignAbi":
    return _DesignAbi([(pos, r, c)])


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().wv_device_count(C.byref(n))
    if rc != WV_OK:
        return 0
    return n.value


class Context:
    """One wv_ctx: one environment's device state on one GPU."""

    def __init__(self, x, y, *, c0, dt, pml_width, pml_scale, device=0, impl="auto"):
        L = lib()
        self._L = L
        self.x = np.ascontiguousarray(x, dtype=np.float32)
        self.y = np.ascontiguousarray(y, dtype=np.float32)
        self.nx, self.ny = len(self.x), len(self.y)
        cfg = wv_config(self.nx, self.ny, float(c0), float(dt), float(pml_width), float(pml_scale), int(device),
                        IMPLS[impl] if isinstance(impl, str) else int(impl))
        h = _vp()
        rc = L.wv_create(C.byref(cfg), fptr(self.x), fptr(self.y), C.byref(h))
        if rc != WV_OK:
            raise WavesAmdError(rc, (L.wv_last_error(None) or b"").decode())
        self._h = h
        self.device = int(device)

    def _ck(self, rc):
        if rc != WV_OK:
            raise WavesAmdError(rc, (self._L.wv_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.wv_destroy(self._h)
            self._h = None

    def __del__(self):
        # At interpreter shutdown the HIP runtime (and a profiler's tool library) may already be finalising: freeing
        # device memory from a destructor then runs inside exit handlers and has been seen to crash them.  Contexts that
        # matter are closed explicitly; the driver reclaims the rest with the process.
        try:
            if sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass

    # --- geometry-derived quantities
    def pml(self):
        sx = np.empty(self.nx, np.float32)
        sy = np.empty(self.ny, np.float32)
        self._ck(self._L.wv_get_pml(self._h, fptr(sx), fptr(sy)))
        return sx, sy

    def set_pml(self, sx, sy):
        sx = np.ascontiguousarray(sx, np.float32)
        sy = np.ascontiguousarray(sy, np.float32)
        assert sx.shape == (self.nx,) and sy.shape == (self.ny,)
        self._ck(self._L.wv_set_pml(self._h, fptr(sx), fptr(sy)))

    def cell_area(self) -> np.float32:
        v = C.c_float(0)
        self._ck(self._L.wv_get_cell_area(self._h, C.byref(v)))
        return np.float32(v.value)

    # --- state.  Arrays use the reference's shapes, Fortran order: (nx, ny, 12[, 3])
    def set_frames(self, wave):
        w = fortran(wave, (self.nx, self.ny, 12, 3))
        self._ck(self._L.wv_set_frames(self._h, fptr(w)))

    def get_frames(self):
        w = np.empty((self.nx, self.ny, 12, 3), np.float32, order="F")
        self._ck(self._L.wv_get_frames(self._h, fptr(w)))
        return w

    def set_state(self, u):
        w = fortran(u, (self.nx, self.ny, 12))
        self._ck(self._L.wv_set_state(self._h, fptr(w)))

    def get_state(self):
        w = np.empty((self.nx, self.ny, 12), np.float32, order="F")
        self._ck(self._L.wv_get_state(self._h, fptr(w)))
        return w

    def reset(self):
        self._ck(self._L.wv_reset(self._h))

    # --- coefficients
    def set_source_shape(self, shape, freq):
        s = fortran(shape, (self.nx, self.ny)) if shape is not None else None
        self._ck(self._L.wv_set_source_shape(self._h, fptr(s), float(freq)))

    def set_gaussian_source(self, mu, sigma, a, freq):
        mu = fortran(np.asarray(mu, np.float32).reshape(-1, 2))
        sigma = np.ascontiguousarray(sigma, np.float32).reshape(-1)
        a = np.ascontiguousarray(a, np.float32).reshape(-1)
        K = len(sigma)
        assert mu.shape == (K, 2) and a.shape == (K,)
        self._ck(self._L.wv_set_gaussian_source(self._h, K, fptr(mu), fptr(sigma), fptr(a), float(freq)))

    def source_shape(self):
        s = np.empty((self.nx, self.ny), np.float32, order="F")
        self._ck(self._L.wv_get_source_shape(self._h, fptr(s)))
        return s

    def set_design(self, initial, final, ti, tf):
        """initial/final: (pos (M,2), r (M,), c (M,)) tuples or None (NoDesign)."""
        if initial is None:
            self._ck(self._L.wv_set_design(self._h, 0, None, None, None, None, None, None, float(ti), float(tf)))
            return
        arrs, ptrs = [], []
        for d in (initial, final):
            ab = d if type(d) is _DesignAbi else design_abi(*d)
            arrs.append(ab)
            ptrs += ab.ptrs
        M = arrs[0].M
        assert arrs[1].M == M
        self._ck(self._L.wv_set_design(self._h, M, *ptrs, float(ti), float(tf)))

    def set_design_sequence(self, designs, ti_tf, steps_per_action):
        """The designs of n actions that the NEXT integrate call runs in one launch: designs = n + 1 tuples
        (pos (M,2), r (M,), c (M,)) -- design k in force before action k, k + 1 after it --, ti_tf = n pairs (ti, tf)."""
        rows = [np.concatenate([np.asarray(pos, np.float32).reshape(-1, 2), np.asarray(r, np.float32).reshape(-1, 1),
                                np.asarray(c, np.float32).reshape(-1, 1)], axis=1) for pos, r, c in designs]
        d = np.ascontiguousarray(np.stack(rows), np.float32)          # (n + 1, M, 4): px, py, r, c
        tt = np.ascontiguousarray(ti_tf, np.float32).reshape(-1, 2)
        assert d.shape[0] == tt.shape[0] + 1
        self._ck(self._L.wv_set_design_sequence(self._h, tt.shape[0], int(steps_per_action), d.shape[1], fptr(d), fptr(tt)))

    def integrate_sequence_begin(self, tspans, *, capture_frames=True, want_signal=True):
        """tspans: (n, steps + 1) -- every action's own tspan; after set_design_sequence.  Ended by integrate_end, whose
        signal has n * steps + 1 rows.  capture_frames="all": the frames of every action are kept on the device
        (observation_action / get_frames_action)."""
        ts = np.ascontiguousarray(tspans, np.float32)
        n, per = ts.shape[0], ts.shape[1] - 1
        cap = 2 if capture_frames == "all" else int(bool(capture_frames))
        self._ck(self._L.wv_integrate_begin(self._h, fptr(ts), n * per, cap, int(bool(want_signal)), 0))
        if not hasattr(self, "_pend"):
            self._pend = []
        self._pend.append((n * per, bool(want_signal), False))

    def observation(self, rx, ry):
        """state(env)'s x: (rx, ry, 4) = imresize(cat(u_tot frames, source shape), (rx, ry)), resized on the device."""
        o = np.empty((int(rx), int(ry), 4), np.float32, order="F")
        self._ck(self._L.wv_observation(self._h, int(rx), int(ry), fptr(o)))
        return o

    def observation_action(self, action, rx, ry):
        """observation() as it was after action `action` of the last sequence call begun with capture_frames="all"."""
        o = np.empty((int(rx), int(ry), 4), np.float32, order="F")
        self._ck(self._L.wv_observation_action(self._h, int(action), int(rx), int(ry), fptr(o)))
        return o

    def get_frames_action(self, action):
        """get_frames() as it was after action `action` of the last sequence call begun with capture_frames="all"."""
        wv = np.empty((self.nx, self.ny, 12, 3), np.float32, order="F")
        self._ck(self._L.wv_get_frames_action(self._h, int(action), fptr(wv)))
        return wv

    def speed_field(self, t):
        o = np.empty((self.nx, self.ny), np.float32, order="F")
        self._ck(self._L.wv_speed_field(self._h, float(t), fptr(o)))
        return o

    def source_field(self, t):
        o = np.empty((self.nx, self.ny), np.float32, order="F")
        self._ck(self._L.wv_source_field(self._h, float(t), fptr(o)))
        return o

    def gradient(self, axis, u):
        u = fortran(u, (self.nx, self.ny))
        o = np.empty((self.nx, self.ny), np.float32, order="F")
        self._ck(self._L.wv_gradient(self._h, int(axis), fptr(u), fptr(o)))
        return o

    def rhs(self, x, t):
        x = fortran(x, (self.nx, self.ny, 12))
        k = np.empty((self.nx, self.ny, 12), np.float32, order="F")
        self._ck(self._L.wv_rhs(self._h, fptr(x), float(t), fptr(k)))
        return k

    # --- integration
    def integrate(self, tspan, *, capture_frames=False, want_signal=True, want_fields=False):
        self.integrate_begin(tspan, capture_frames=capture_frames, want_signal=want_signal, want_fields=want_fields)
        return self.integrate_end()

    def set_trajectory_stride(self, stride: int):
        """u_tot / u_inc of later integrate calls hold every `stride`-th saved time (0, stride, ...)."""
        self._ck(self._L.wv_set_trajectory_stride(self._h, int(stride)))
        self._traj_stride = int(stride)

    def integrate_begin(self, tspan, *, capture_frames=False, want_signal=True, want_fields=False):
        """Enqueue one integrate call and return.  A second call may be begun before the first is ended (the host then
        prepares call k+1 while call k runs); integrate_end ends the oldest pending call."""
        ts = np.ascontiguousarray(tspan, np.float32).reshape(-1)
        n = len(ts) - 1
        wf = 2 if want_fields == "stream" else int(bool(want_fields))   # "stream": planes go to pinned host memory as they are produced
        self._ck(self._L.wv_integrate_begin(self._h, fptr(ts), n, int(bool(capture_frames)), int(bool(want_signal)), wf))
        if not hasattr(self, "_pend"):
            self._pend = []
        self._pend.append((n, bool(want_signal), want_fields if want_fields == "stream" else bool(want_fields)))

    def pending(self) -> int:
        return len(getattr(self, "_pend", ()))

    def _end_ck(self, rc):
        """Status check of an integrate_end: on failure the mirror of the library's queue of pending calls is brought
        back in line with the library's own count (an _end that failed after the device work was waited for HAS ended
        its call; an argument error has not) before the error is raised."""
        if rc != WV_OK:
            n = C.c_int(0)
            if self._L.wv_pending(self._h, C.byref(n)) == WV_OK:
                while len(self._pend) > n.value:
                    self._pend.pop(0)
        self._ck(rc)

    def integrate_end(self):
        if not getattr(self, "_pend", None):  # nothing pending: let the library say so (WV_ERR_STATE)
            self._ck(self._L.wv_integrate_end(self._h, None, None, None))
        n, ws, wf = self._pend[0]
        sig = np.empty((n + 1, 3), np.float32) if ws else None
        if wf == "stream":   # zero-copy views of the library's pinned planes (valid until the second integrate_begin from now)
            pt, pi, npl = _fp(), _fp(), C.c_int(0)
            self._end_ck(self._L.wv_integrate_end_view(self._h, fptr(sig), C.byref(pt), C.byref(pi), C.byref(npl)))
            self._pend.pop(0)
            shape = (npl.value, self.ny, self.nx)
            ut = np.ctypeslib.as_array(pt, shape).transpose(2, 1, 0)
            ui = np.ctypeslib.as_array(pi, shape).transpose(2, 1, 0)
            return sig, ut, ui
        planes = n // getattr(self, "_traj_stride", 1) + 1
        ut = np.empty((self.nx, self.ny, planes), np.float32, order="F") if wf else None
        ui = np.empty((self.nx, self.ny, planes), np.float32, order="F") if wf else None
        self._end_ck(self._L.wv_integrate_end(self._h, fptr(sig), fptr(ut), fptr(ui)))
        self._pend.pop(0)
        return sig, ut, ui

    # --- measurement / plumbing
    def set_profiling(self, on: bool):
        self._ck(self._L.wv_set_profiling(self._h, int(bool(on))))

    def timing(self) -> dict:
        t = wv_timing()
        self._ck(self._L.wv_get_timing(self._h, C.byref(t)))
        return {"total_ms": t.total_ms, "step_kernel_ms": t.step_kernel_ms,
                "step_kernel_launches": t.step_kernel_launches, "steps": t.steps,
                "impl": {1: "staged", 2: "fused"}.get(t.impl, str(t.impl)), "resident": bool(t.resident),
                "gave_up": bool(t.gave_up), "launch_ms": t.launch_ms, "launch_jobs": t.launch_jobs}

    def call_times_ms(self, cap=4096):
        """Durations (ms, the resident kernel's own clock) of the resident calls ended since the last query, oldest first."""
        buf = (C.c_double * int(cap))()
        n = C.c_int(0)
        self._ck(self._L.wv_get_call_times(self._h, buf, int(cap), C.byref(n)))
        return [buf[k] for k in range(n.value)]

    def set_stream(self, stream_handle):
        self._ck(self._L.wv_set_stream(self._h, _vp(stream_handle) if stream_handle else None))

    def synchronize(self):
        self._ck(self._L.wv_synchronize(self._h))

    def device_frames(self):
        p, n = _vp(), C.c_size_t(0)
        self._ck(self._L.wv_device_frames(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def release_device_frames(self):
        """The raw pointer of device_frames() is no longer written by the caller."""
        self._ck(self._L.wv_release_device_frames(self._h))

    def selftest_granules(self, iters=20000):
        """(granules checked, granules found torn) of the 16-byte exchange-granule self-test."""
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self._ck(self._L.wv_selftest_granules(self._h, int(iters), C.byref(a), C.byref(b)))
        return a.value, b.value

    def device_source_shape(self):
        p, n = _vp(), C.c_size_t(0)
        self._ck(self._L.wv_device_source_shape(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value
