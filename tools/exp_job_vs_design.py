"""Which designs make a slow action?  Per action of the headline loop: the job time next to what the launch order of that call looked like
(WAVES_AMD_PLAN_DUMP: tiles by cylinder count, who has a CU alone, who are the partners of the tiles with cylinders).  Diagnostic, GPU box."""
import os
import sys
import tempfile

import numpy as np

dump = os.path.join(tempfile.gettempdir(), "wv_plan_dump.txt")
os.environ["WAVES_AMD_PLAN_DUMP"] = dump
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import waves_jl_amd as w  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    ds = w.build_triple_ring_design_space()
    env, policy = bench.make_env(w, w.TwoDim(15.0, 700), ds, 0, "fused", 2.0, 10 ** 6, 5)
    w.rollout_pipelined(env, policy, 5)
    env.ctx.call_times_ms()
    plans = []
    for k in range(n):
        env.step_begin(policy(env))
        rows = np.array([[int(v) for v in line.split()] for line in open(dump) if not line.startswith("#")])
        plans.append(rows)
        if k > 0:
            env.step_end()
    env.step_end()
    t = np.asarray(env.ctx.call_times_ms()) * 1e3
    C = 256
    print("job_us | tiles with 1, 2, 3, 4+ cylinders | alone: corners, cyl tiles, others | paired cyl tiles: max cyl count, with a PML partner | heaviest paired: cyl of the 5 first pairs")
    for k, (rows, tt) in enumerate(zip(plans, t)):
        pos, x0, y0, ox, oy, aux, edge, cyl, slot = rows.T
        nt = len(rows)
        pairs = nt - C
        alone = rows[pairs:C]
        heavy = rows[:pairs]
        light = rows[C:]
        cnt = [int((cyl == 1).sum()), int((cyl == 2).sum()), int((cyl == 3).sum()), int((cyl >= 4).sum())]
        al = [int((alone[:, 5] == 3).sum()), int(((alone[:, 7] > 0) & (alone[:, 5] != 3)).sum())]
        al.append(len(alone) - sum(al))
        hc = heavy[:, 7] > 0
        lpml = light[:, 5] != 0
        lcyl = light[:, 7] > 0
        print(f"{tt:7.1f} | {cnt} | {al} | paired cyl tiles {int(hc.sum())}, max count {int(heavy[:, 7].max())}, light partners that are PML {int(lpml.sum())} / have cylinders {int(lcyl.sum())} | "
              f"{heavy[:5, 7].tolist()} aux {heavy[:5, 5].tolist()}")
    print("correlation of job time with: tiles with cylinders", np.corrcoef(t, [int((p[:, 7] > 0).sum()) for p in plans])[0, 1].round(2),
          "| light partners with cylinders", np.corrcoef(t, [int((p[C:, 7] > 0).sum()) for p in plans])[0, 1].round(2),
          "| light partners in the PML", np.corrcoef(t, [int((p[C:, 5] != 0).sum()) for p in plans])[0, 1].round(2),
          "| max count among paired", np.corrcoef(t, [int(p[:len(p) - C, 7].max()) for p in plans])[0, 1].round(2))


if __name__ == "__main__":
    main()
