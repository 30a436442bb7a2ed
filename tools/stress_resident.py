"""Randomised cross-check (run on the GPU box): the resident step kernel against the single-step kernels on random grids,
PML widths, step counts, designs, sources and initial states -- every output must be bit-identical.
usage: python tools/stress_resident.py [n_cases] [seed]"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import waves_jl_amd as w
import waves_oracle as wo

f32 = np.float32


def run(resident, cfg):
    os.environ["WAVES_AMD_FUSED_RESIDENT"] = "1" if resident else "0"
    rng = np.random.default_rng(cfg["seed"])
    n = cfg["n"]
    dim = wo.TwoDim.from_size(cfg["size"], n)
    ctx = w._ffi.Context(dim.x, dim.y, c0=wo.WATER, dt=cfg["dt"], pml_width=cfg["pml"], pml_scale=cfg["scale"], impl="fused")
    M = cfg["M"]
    if M:
        pos0 = rng.uniform(-0.8 * cfg["size"], 0.8 * cfg["size"], (M, 2)).astype(f32)
        pos1 = (pos0 + rng.uniform(-0.4, 0.4, (M, 2))).astype(f32)
        r0 = rng.uniform(0.05, cfg["rmax"], M).astype(f32)
        r1 = np.abs(r0 + rng.uniform(-0.3, 0.3, M)).astype(f32)
        c = rng.uniform(500.0, 3000.0, M).astype(f32)
    if cfg["source"]:
        ctx.set_gaussian_source([[float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))]], [float(rng.uniform(0.2, 1.0))], [1.0], 1000.0)
    u0 = (rng.standard_normal((n, n, 12)) * 0.1).astype(f32)
    if not cfg["aux"]:
        u0[:, :, [3, 4, 5, 9, 10, 11]] = 0
    ctx.set_state(np.asfortranarray(u0))
    outs = []
    t0 = 0.001
    for call in range(cfg["calls"]):
        ts = wo.build_tspan(f32(t0), cfg["dt"], cfg["steps"])
        if M:
            a, b = ((pos0, r0, c), (pos1, r1, c)) if call % 2 == 0 else ((pos1, r1, c), (pos0, r0, c))
            ctx.set_design(a, b, ts[0], ts[-1])
        cap = cfg["capture"] and cfg["steps"] >= 20
        sig, ut, ui = ctx.integrate(ts, capture_frames=cap, want_fields=cfg["fields"])
        outs.append((sig, ut, ui, ctx.get_frames() if cap else ctx.get_state()))
        t0 = float(ts[-1])
    res = ctx.timing()["resident"]
    ctx.close()
    return outs, res


def run_cases(ncases, seed, sizes=(8, 9, 17, 33, 57, 64, 65, 100, 128, 191, 256, 300, 420, 511, 640, 700, 730, 760), verbose=True):
    """-> (number of mismatching cases, number of cases that ran on the resident kernel)"""
    rng = np.random.default_rng(seed)
    bad = 0
    nres = 0
    for k in range(ncases):
        n = int(rng.choice(list(sizes)))
        cfg = dict(seed=int(rng.integers(1 << 30)), n=n, size=float(rng.choice([5.0, 15.0, 40.0])), dt=1e-5,
                   pml=float(rng.choice([0.0, 0.3, 1.0, 2.0, 5.0])), scale=float(rng.choice([0.0, 20000.0])),
                   M=int(rng.choice([0, 1, 3, 19, 40])), rmax=float(rng.choice([0.5, 2.0, 6.0])), source=bool(rng.integers(2)),
                   aux=bool(rng.integers(2)), steps=int(rng.choice([2, 3, 7, 20, 21, 33, 60])), calls=int(rng.choice([1, 2, 3])),
                   capture=bool(rng.integers(2)), fields=bool(rng.integers(2)) and n <= 300)
        gc.collect()
        a, ra = run(True, cfg)
        gc.collect()
        b, rb = run(False, cfg)
        ok = True
        for (sa, uta, uia, fa), (sb, utb, uib, fb) in zip(a, b):
            ok = ok and np.array_equal(sa, sb, equal_nan=True) and np.array_equal(fa, fb, equal_nan=True)
            if cfg["fields"]:
                ok = ok and np.array_equal(uta, utb, equal_nan=True) and np.array_equal(uia, uib, equal_nan=True)
        nres += int(ra)
        bad += 0 if ok else 1
        if verbose or not ok:
            print(("ok  " if ok else "BAD ") + f"resident={ra} (reference path resident={rb}) " + str(cfg), flush=True)
    os.environ.pop("WAVES_AMD_FUSED_RESIDENT", None)
    return bad, nres


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad, nres = run_cases(ncases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{ncases} cases, {nres} ran resident, {bad} mismatching")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
