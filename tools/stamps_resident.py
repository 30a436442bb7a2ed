"""Report of the phase stamps k_steps_resident records for the middle step of a call (WAVES_AMD_STAMPS=<file>)."""
import sys
import numpy as np

rows = []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        continue
    head, tail = line.split("|")
    h = [int(v) for v in head.split()]
    t = [int(v) for v in tail.split()]
    rows.append(h + t)
a = np.array(rows, dtype=np.int64)
T = a[:, 8:8 + 6].astype(np.float64) * 10.0  # s_memrealtime ticks of 10 ns -> ns
polls = a[:, 8 + 6]
aux = a[:, 6] & 15
names = {0: "NONE", 1: "PX", 2: "PY", 3: "ALL"}
t0 = T[:, 0].min()
print(f"tiles {len(a)}; step start spread {T[:,0].max()-t0:.0f} ns; last refresh end {T[:,5].max()-t0:.0f} ns")
for k in range(4):
    m = aux == k
    if not m.any():
        continue
    d = np.diff(T[m], axis=1)
    print(f"{names[k]:5s} n={m.sum():3d} mean ns: compute={d[:,0].mean():.0f} store={d[:,1].mean():.0f} "
          f"ack+barrier={d[:,2].mean():.0f} wait={d[:,3].mean():.0f} (max {d[:,3].max():.0f}) refresh={d[:,4].mean():.0f} "
          f"polls mean {polls[m].mean():.1f} max {polls[m].max()}  step total {(T[m,5]-T[m,0]).mean():.0f}")
d = np.diff(T, axis=1)
order = np.argsort(d[:, 3])
print("tiles with the shortest waits (the ones the others wait for):")
for i in order[:12]:
    print(f"  slot {a[i,1]:3d} x0={a[i,2]:3d} y0={a[i,3]:3d} {names[aux[i]]:4s} cyl={a[i,7]:3d} compute={d[i,0]:.0f} store={d[i,1]:.0f} "
          f"ack={d[i,2]:.0f} wait={d[i,3]:.0f} refresh={d[i,4]:.0f} polls={polls[i]} total={T[i,5]-T[i,0]:.0f}")
print("wait percentiles ns:", np.percentile(d[:, 3], [0, 10, 50, 90, 100]).round())
print("compute percentiles ns:", np.percentile(d[:, 0], [0, 10, 50, 90, 100]).round())
print("refresh percentiles ns:", np.percentile(d[:, 4], [0, 10, 50, 90, 100]).round())

X = a[:, 8:8 + 12].astype(np.float64) * 10.0
cyl = a[:, 7]
print("inside the step (ns): init+pub1+barrier | speed | stage1 | stages2-3 | stage4, by tile class")
for nm, m in (("NONE cyl=0", (aux == 0) & (cyl == 0)), ("NONE cyl>0", (aux == 0) & (cyl > 0)), ("PX", aux == 1), ("PY", aux == 2), ("ALL", aux == 3)):
    if m.any():
        print(f"  {nm:11s} n={m.sum():3d} {np.mean(X[m,7]-X[m,0]):6.0f} | {np.mean(X[m,8]-X[m,7]):6.0f} | {np.mean(X[m,9]-X[m,8]):6.0f} | "
              f"{np.mean(X[m,11]-X[m,9]):6.0f} | {np.mean(X[m,1]-X[m,11]):6.0f}")

tot = T[:, 5] - T[:, 0]
print("step total percentiles ns:", np.percentile(tot, [0, 10, 50, 90, 99, 100]).round())
print("slowest tiles by step total:")
for i in np.argsort(-tot)[:14]:
    print(f"  pos {a[i,0]:3d} slot {a[i,1]:3d} x0={a[i,2]:3d} y0={a[i,3]:3d} ox={a[i,4]} oy={a[i,5]} {names[aux[i]]:4s} edge={a[i,6]>>4} cyl={a[i,7]:3d} "
          f"init+pub={X[i,7]-X[i,0]:.0f} speed={X[i,8]-X[i,7]:.0f} st1={X[i,9]-X[i,8]:.0f} st23={X[i,11]-X[i,9]:.0f} st4={X[i,1]-X[i,11]:.0f} "
          f"store={d[i,1]:.0f} ack={d[i,2]:.0f} wait={d[i,3]:.0f} total={tot[i]:.0f}")
