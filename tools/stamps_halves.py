"""Report of the phase stamps k_steps_resident records for the middle step of a call (WAVES_AMD_STAMPS=<file>), for the
round-2 kernel that integrates the two wave sets of a step one after the other.  Stamps are ISSUE times (s_memrealtime,
10 ns ticks, no wait for outstanding memory operations):
  0 step start | 1 after the total set's publish-1 barrier | 2 after the wave-speed evaluation | 3 after stage 1 |
  4 after stage 3 | 5 after stage 4 | 6 after the incident set's halo poll | 7 after the total set's border stores |
  11 after the incident set's publish-1 barrier | 8 after its stage 4 | 9 after outputs + energy terms |
  12 after the total set's halo poll | 10 after the incident set's border stores (end of the step)"""
import sys
import numpy as np

rows = []
for line in open(sys.argv[1]):
    if line.startswith("#"):
        continue
    head, tail = line.split("|")
    rows.append([int(v) for v in head.split()] + [int(v) for v in tail.split()])
a = np.array(rows, dtype=np.int64)
X = a[:, 8:8 + 15].astype(np.float64) * 10.0
aux = a[:, 6] & 15
cyl = a[:, 7]
seq = [0, 1, 2, 3, 4, 5, 6, 7, 11, 8, 9, 12, 10]
lab = ["pub1T+bar", "speed", "stage1T", "st2-3T", "st4T", "pollI", "storeT", "pub1I+bar", "st1-4I", "out+energy", "pollT", "storeI"]
print("tiles", len(a), " step start spread %.0f ns" % (X[:, 0].max() - X[:, 0].min()))
print("%-12s %4s " % ("class", "n") + " ".join("%10s" % l for l in lab) + "      total")
for nm, m in (("NONE cyl=0", (aux == 0) & (cyl == 0)), ("NONE cyl>0", (aux == 0) & (cyl != 0)), ("PX", aux == 1), ("PY", aux == 2), ("ALL", aux == 3)):
    if not m.any():
        continue
    d = [np.mean(X[m, seq[i + 1]] - X[m, seq[i]]) for i in range(len(seq) - 1)]
    print("%-12s %4d " % (nm, m.sum()) + " ".join("%10.0f" % v for v in d) + " %10.0f" % np.mean(X[m, 10] - X[m, 0]))
tot = X[:, 10] - X[:, 0]
print("step total percentiles ns:", np.percentile(tot, [0, 10, 50, 90, 100]).round())
for k, nm in ((6, "pollI"), (12, "pollT")):
    prev = seq[seq.index(k) - 1]
    print(nm, "percentiles ns:", np.percentile(X[:, k] - X[:, prev], [0, 10, 50, 90, 100]).round())
