"""Summarise the rocprofv3 output of tools/collect_profiles.sh: per-kernel duration statistics and the PMC counters of
the integrator kernel, per launch, with the HBM-side byte counts corrected as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950 (FETCH_SIZE in 32-B units x2 correction -> bytes = value * 64 ... see DESIGN.md section 6)."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
KERNELS = ("k_steps_resident", "k_step_fused", "k_stage")


def find(sub, pat):
    f = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return f[0] if f else None


out = {"kernel_trace": {}, "counters": {}}
f = find("kt", "*kernel_stats.csv")
if f:
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if any(k in name for k in KERNELS) or float(r["Percentage"]) > 1.0:
            out["kernel_trace"][name[:80]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                                              "max_ns": float(r["MaxNs"]), "percent": float(r["Percentage"])}
main_kernel = None
for sub in ("pmc1", "pmc2", "pmc3"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc, n = {}, {}
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        k = next((k for k in KERNELS if k in name), None)
        if k is None:
            continue
        main_kernel = main_kernel or k
        if k != main_kernel:
            continue
        c = r["Counter_Name"]
        acc[c] = acc.get(c, 0.0) + float(r["Counter_Value"])
        n[c] = n.get(c, set())
        n[c].add(r["Dispatch_Id"])
    for c in acc:
        out["counters"][c] = {"mean_per_launch": acc[c] / max(1, len(n[c])), "launches": len(n[c])}
out["kernel"] = main_kernel
cnt = out["counters"]
if "FETCH_SIZE" in cnt and "WRITE_SIZE" in cnt:
    # rocprofv3 reports both in KiB-like units of 1024 B?  No: FETCH_SIZE / WRITE_SIZE are in kilobytes (1 KB = 1024 B) on
    # this stack; gfx950 correction: FETCH_SIZE under-counts by 2x (calibrated in profiles/r01/fused_432_pmc_summary.json on
    # 1 GiB streams), WRITE_SIZE is exact.
    rd = cnt["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * 2.0
    wr = cnt["WRITE_SIZE"]["mean_per_launch"] * 1024.0
    out["hbm_side_bytes_per_launch"] = {"read": rd, "write": wr, "total": rd + wr,
                                        "note": "FETCH_SIZE [KB] x2 (gfx950 correction), WRITE_SIZE [KB] exact; Infinity-Cache hits included"}
if "TCC_HIT_sum" in cnt and "TCC_MISS_sum" in cnt:
    h, m = cnt["TCC_HIT_sum"]["mean_per_launch"], cnt["TCC_MISS_sum"]["mean_per_launch"]
    out["l2_hit_rate"] = h / max(1.0, h + m)
print(json.dumps(out, indent=1))
