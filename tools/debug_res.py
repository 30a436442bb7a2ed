"""Diagnostic: first step at which the resident kernel and the single-step kernels differ (golden 64^2 case)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np
if len(sys.argv) > 1:
    import waves_jl_amd as w
    d = np.load(os.path.join(ROOT, "tests", "golden", "moving_design_64.npz"))
    out = {}
    for n in (2, 3, 4, 6):
        ctx = w._ffi.Context(d["x"], d["x"], c0=float(d["c0"]), dt=float(d["dt"]), pml_width=float(d["pml_width"]),
                             pml_scale=float(d["pml_scale"]), impl="fused")
        ctx.set_source_shape(d["source_shape"], float(d["freq"]))
        ts = d["tspan"]
        mode = os.environ.get("DBG_MODE", "")
        if mode == "static":
            ctx.set_design((d["pos0"], d["r0"], d["c_0"]), (d["pos0"], d["r0"], d["c_0"]), ts[0], ts[-1])
        elif mode == "nocyl":
            ctx.set_design(None, None, ts[0], ts[-1])
        else:
            ctx.set_design((d["pos0"], d["r0"], d["c_0"]), (d["pos1"], d["r1"], d["c_1"]), ts[0], ts[-1])
        if os.environ.get("DBG_ZERO"):
            u0 = np.array(d["u0"]); u0[:, :, [3, 4, 5, 9, 10, 11]] = 0
        else:
            u0 = d["u0"]
        ctx.set_state(u0)
        ctx.integrate(ts[:n + 1])
        out[str(n)] = ctx.get_state()
        print(sys.argv[1], n, ctx.timing()["resident"])
        ctx.close()
    np.savez(sys.argv[1], **out)
else:
    for name, res in (("/tmp/res.npz", "1"), ("/tmp/one.npz", "0")):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, WAVES_AMD_FUSED_RESIDENT=res), check=True)
    a, b = np.load("/tmp/res.npz"), np.load("/tmp/one.npz")
    for n in ("2", "3", "4", "6"):
        bad = a[n] != b[n]
        print("steps", n, "mismatches", int(bad.sum()))
        if bad.any():
            ii, jj, ff = np.nonzero(bad)
            for f in np.unique(ff):
                m = bad[:, :, f]
                print("   field", f, "count", int(m.sum()), "i", np.unique(np.nonzero(m)[0]), "j", np.unique(np.nonzero(m)[1]))
            for k in range(min(6, len(ii))):
                print(f"    i={ii[k]} j={jj[k]} f={ff[k]} resident {a[n][ii[k], jj[k], ff[k]]!r} single {b[n][ii[k], jj[k], ff[k]]!r}")
            break
