#!/bin/bash
# A/B timing of library variants on the GPU box:  bash tools/ab.sh name1 name2 ...   ("base" = the product library)
# Prints per variant the bench line's value, kernel us and fractions.  Extra bench args via AB_ARGS.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out/ab
for v in "$@"; do
  lib=$R/waves.jl_amd/csrc/libwaves_amd_$v.so
  [ "$v" = base ] && lib=$R/waves.jl_amd/csrc/libwaves_amd.so
  WAVES_AMD_LIB=$lib timeout -k 10 300 python3 $R/bench.py --cpu-steps 0 --batch-envs 0 --steps ${AB_STEPS:-20} --warmup 3 $AB_ARGS > $R/gpurun_out/ab/$v.json 2> $R/gpurun_out/ab/$v.err
  rc=$?
  python3 - "$v" "$R/gpurun_out/ab/$v.json" $rc <<'PY'
import json, sys
v, p, rc = sys.argv[1:4]
try:
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{v:16s} rc={rc} value={d['value']:9.1f} ms/step={d['ms_per_step']:.4f} kernel_us={r['avg_kernel_us']:9.2f} frac={r['frac']:.4f} whole={r['whole_job_frac']:.4f} {r['kernel']}")
except Exception as e:
    print(f"{v:16s} rc={rc} FAILED {e}")
PY
done
