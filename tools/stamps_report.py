"""Summarise WAVES_AMD_STAMPS output: per-field-set phase durations (shader-clock ticks) and the launch timeline."""
import sys
import numpy as np
rows = []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        continue
    a, b = line.split("|")
    rows.append([int(v) for v in a.split()] + [int(v) for v in b.split()])
r = np.array(rows, dtype=np.int64)
pos, slot, x0, y0, ox, oy, var, cyl = r[:, :8].T
aux, edge = var & 15, var >> 4
t = r[:, 8:18]
xcc = r[:, 18] & 0xF
real_end = r[:, 20]
real_start = r[:, 22] if r.shape[1] > 22 else real_end
names = ["load+pub1", "comp1", "pub2", "comp2", "pub3", "comp3", "pub4", "comp4", "store"]
r0 = real_start.min()
print(f"tiles {len(r)}; realtime: starts spread over {(real_start.max()-r0)*10} ns, last end at {(real_end.max()-r0)*10} ns")
for v, nm in ((0, "NONE"), (1, "PX"), (2, "PY"), (3, "ALL")):
    m = aux == v
    if not m.any():
        continue
    d = np.diff(t[m], axis=1)
    life = t[m, 9] - t[m, 0]
    print(f"{nm}: n={m.sum()} (edge {int((edge[m] != 0).sum())}) lifetime mean {life.mean():.0f} max {life.max()} ticks | "
          f"start mean {(real_start[m]-r0).mean()*10:.0f} ns max {(real_start[m]-r0).max()*10} ns | end max {(real_end[m]-r0).max()*10} ns")
    print("   phases mean:", " ".join(f"{n}={x:.0f}" for n, x in zip(names, d.mean(axis=0))))
print("tiles per XCC:", np.bincount(xcc, minlength=8))
print("start (ns) by launch position deciles:", [int((real_start[pos.argsort()][k]-r0)*10) for k in np.linspace(0, len(r)-1, 11).astype(int)])
cyl_m = cyl != 0
if cyl_m.any():
    print("tiles with cylinders:", cyl_m.sum(), "lifetime mean", (t[cyl_m, 9] - t[cyl_m, 0]).mean())
