"""Summarise WAVES_AMD_STAMPS output: per-variant phase durations (shader-clock cycles) and the kernel's critical path."""
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
t = r[:, 8:18]
xcc = r[:, 18] & 0xF
hwid = r[:, 19]
real = r[:, 20]
t0 = t[:, 0].min()
names = ["load+pub1", "comp1", "pub2", "comp2", "pub3", "comp3", "pub4", "comp4", "store"]
print(f"tiles {len(r)}  kernel span (first start -> last end): {(t[:, 9].max() - t0)} cycles; realtime span {(real.max()-real.min())*10} ns")
for v, nm in ((0, "FAST"), (1, "MID"), (2, "GEN")):
    m = var == v
    if not m.any():
        continue
    d = np.diff(t[m], axis=1)
    life = t[m, 9] - t[m, 0]
    start = t[m, 0] - t0
    print(f"{nm}: n={m.sum()} lifetime mean {life.mean():.0f} max {life.max()} | start mean {start.mean():.0f} max {start.max()} | end max {(t[m,9]-t0).max()}")
    print("   phases mean:", " ".join(f"{n}={x:.0f}" for n, x in zip(names, d.mean(axis=0))))
print("tiles per XCC:", np.bincount(xcc, minlength=8))
print("start times percentiles (cycles):", np.percentile(t[:, 0] - t0, [0, 25, 50, 75, 90, 100]).astype(int))
print("end times percentiles (cycles):", np.percentile(t[:, 9] - t0, [0, 25, 50, 75, 90, 100]).astype(int))
cyl_m = cyl != 0
if cyl_m.any():
    print("tiles with cylinders:", cyl_m.sum(), "lifetime mean", (t[cyl_m, 9] - t[cyl_m, 0]).mean())
# per-XCC clocks are not synchronised: report spans per XCC too
for k in range(8):
    m = xcc == k
    if m.any():
        print(f"  xcc {k}: n={m.sum()} span {(t[m, 9].max() - t[m, 0].min())} cycles, variants {np.bincount(var[m], minlength=3)}")
