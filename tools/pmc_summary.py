"""Summarise the rocprofv3 output of tools/collect_profiles.sh: per-kernel duration statistics, every dispatch of the
integrator kernel, and its PMC counters with the HBM-side byte counts corrected as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950 (FETCH_SIZE x2, WRITE_SIZE exact: DESIGN.md section 6).

Round 3: k_steps_resident serves a whole run of actions as ONE dispatch (a job per action).  The dispatch that served the
bench's timed region is the longest one of the run (PROF_STEPS actions); its duration / PROF_STEPS is what bench.py reports
as avg_kernel_us, and its counters / PROF_STEPS are the per-action figures."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
STEPS = int(os.environ.get("PROF_STEPS", "20"))
KERNELS = ("k_steps_resident", "k_step_fused", "k_stage")


def find(sub, pat):
    f = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return f[0] if f else None


out = {"kernel_trace": {}, "counters": {}, "actions_in_timed_dispatch": STEPS}
f = find("kt", "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if any(k in name for k in KERNELS) or float(r["Percentage"]) > 1.0:
            out["kernel_trace"][name[:80]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                              "max_ns": float(r["MaxNs"]), "percent": float(r["Percentage"])}
f = find("kt", "*kernel_trace.csv")
if f:
    disp = []
    for r in csv.DictReader(open(f)):
        if "k_steps_resident" in r["Kernel_Name"]:
            disp.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    if disp:
        out["resident_dispatches_us"] = [round(d, 1) for d in disp]
        out["timed_dispatch"] = {"us": round(max(disp), 1), "us_per_action": round(max(disp) / STEPS, 2),
                                 "note": "the longest k_steps_resident dispatch = the launch that served bench.py's timed region"}
main_kernel = None
for sub in ("pmc1", "pmc2", "pmc3"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    per = {}   # counter -> dispatch -> value
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = next((k for k in KERNELS if k in name), None)
        if k is None:
            continue
        main_kernel = main_kernel or k
        if k != main_kernel:
            continue
        per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        if main_kernel == "k_steps_resident":
            # the timed dispatch is the one that served the most actions.  Work counters scale with the actions served, so it is
            # the dispatch with the largest value; counters that do not (SQ_WAVES: the grid) are the same for every dispatch.
            ids = sorted(d, key=lambda x: int(x))
            big = max(ids, key=lambda x: d[x])
            scales = c not in ("SQ_WAVES",)
            vals = sorted(d[x] for x in ids)
            if len(ids) >= STEPS:
                # (under --pmc the profiler serialises the process enough for a waiting launch to reach its idle limit between two
                # actions: nearly every action then is a dispatch of its own -- the median dispatch IS one action)
                per_action, how = vals[len(vals) // 2], "median dispatch (one action per dispatch in this pass)"
            else:
                per_action, how = (d[big] / STEPS if scales else d[big]), "largest dispatch / actions of the timed region"
            out["counters"][c] = {"timed_dispatch": d[big], "per_action": per_action, "per_action_is": how, "dispatches": len(ids),
                                  "all_dispatches": [d[x] for x in ids]}
        else:
            out["counters"][c] = {"mean_per_launch": sum(d.values()) / max(1, len(d)), "launches": len(d)}
out["kernel"] = main_kernel
cnt = out["counters"]
key = "per_action" if main_kernel == "k_steps_resident" else "mean_per_launch"
if "FETCH_SIZE" in cnt and "WRITE_SIZE" in cnt:
    # FETCH_SIZE / WRITE_SIZE are in kilobytes (1 KB = 1024 B) on this stack; gfx950 correction: FETCH_SIZE under-counts by 2x
    # (calibrated in profiles/r01/fused_432_pmc_summary.json on 1 GiB streams), WRITE_SIZE is exact.
    rd = cnt["FETCH_SIZE"][key] * 1024.0 * 2.0
    wr = cnt["WRITE_SIZE"][key] * 1024.0
    out["hbm_side_bytes_" + ("per_action" if key == "per_action" else "per_launch")] = {
        "read": rd, "write": wr, "total": rd + wr,
        "note": "FETCH_SIZE [KB] x2 (gfx950 correction), WRITE_SIZE [KB] exact; Infinity-Cache hits included"}
if "TCC_HIT_sum" in cnt and "TCC_MISS_sum" in cnt:
    h, m = cnt["TCC_HIT_sum"][key], cnt["TCC_MISS_sum"][key]
    out["l2_hit_rate"] = h / max(1.0, h + m)
print(json.dumps(out, indent=1))
