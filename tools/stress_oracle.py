"""Randomised parity check (run on the GPU box): the HIP path through the C ABI against the C oracle on random small
grids, PML widths, designs, sources and initial states -- fields bit-exact, energy traces to 1e-5.
usage: python tools/stress_oracle.py [n_cases] [seed]"""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import waves_jl_amd as w
import waves_oracle as wo
from helpers import oracle_integrate, rel_err

f32 = np.float32


def run_cases(ncases, seed, verbose=True):
    """-> (number of mismatching cases, number of cases that ran on the resident kernel)"""
    rng = np.random.default_rng(seed)
    bad = 0
    nres = 0
    for k in range(ncases):
        n = int(rng.choice([8, 9, 16, 33, 57, 64, 65, 96, 128, 150, 200, 260]))
        size = float(rng.choice([5.0, 15.0]))
        pml = (float(rng.choice([0.3, 1.0, 2.0, 4.0])), float(rng.choice([0.0, 20000.0])))
        steps = int(rng.choice([1, 2, 5, 20, 21, 33]))
        M = int(rng.choice([0, 1, 4, 19]))
        source = bool(rng.integers(2))
        aux = bool(rng.integers(2))
        resident = bool(rng.integers(2))
        os.environ["WAVES_AMD_FUSED_RESIDENT"] = "1" if resident else "0"
        gc.collect()
        dim = wo.TwoDim.from_size(size, n)
        ctx = w._ffi.Context(dim.x, dim.y, c0=wo.WATER, dt=1e-5, pml_width=pml[0], pml_scale=pml[1], impl="fused")
        d0 = d1 = None
        ts = wo.build_tspan(f32(0.001), 1e-5, steps)
        if M:
            pos0 = rng.uniform(-0.8 * size, 0.8 * size, (M, 2)).astype(f32)
            pos1 = (pos0 + rng.uniform(-0.3, 0.3, (M, 2))).astype(f32)
            r0 = rng.uniform(0.05, 0.3 * size, M).astype(f32)
            r1 = np.abs(r0 + rng.uniform(-0.2, 0.2, M)).astype(f32)
            c = rng.uniform(500.0, 3000.0, M).astype(f32)
            d0 = np.concatenate([pos0, r0[:, None], c[:, None]], 1).astype(f32)
            d1 = np.concatenate([pos1, r1[:, None], c[:, None]], 1).astype(f32)
            ctx.set_design((pos0, r0, c), (pos1, r1, c), ts[0], ts[-1])
        G = None
        if source:
            G = wo.build_normal(wo.build_grid(dim), np.array([[rng.uniform(-3, 3), rng.uniform(-3, 3)]]),
                                np.array([rng.uniform(0.2, 1.0)]), np.array([1.0]))
            ctx.set_source_shape(G, 1000.0)
        u0 = (rng.standard_normal((n, n, 12)) * 0.1).astype(f32)
        if not aux:
            u0[:, :, [3, 4, 5, 9, 10, 11]] = 0
        u0 = np.asfortranarray(u0)
        ctx.set_state(u0)
        cap = steps >= 20
        sig, _, _ = ctx.integrate(ts, capture_frames=cap)
        st, rsig, _ = oracle_integrate(dim, wo.to_abi(u0), ts, pml=pml, G=G, freq=1000.0 if source else 0.0, d0=d0, d1=d1,
                                       ti=ts[0], tf=ts[-1])
        got = wo.to_abi(ctx.get_state())
        ok = np.array_equal(got, st, equal_nan=True) and (not np.isfinite(rsig).all() or rel_err(sig, rsig) < 1e-5)
        was_res = ctx.timing()["resident"]
        ctx.close()
        bad += 0 if ok else 1
        nres += int(was_res)
        if verbose or not ok:
            print(("ok  " if ok else "BAD ") + f"n={n} size={size} pml={pml} steps={steps} M={M} source={source} aux={aux} resident={was_res}",
                  flush=True)
    os.environ.pop("WAVES_AMD_FUSED_RESIDENT", None)
    return bad, nres


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad, nres = run_cases(ncases, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"{ncases} cases, {nres} on the resident kernel, {bad} mismatching")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
