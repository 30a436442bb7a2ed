#!/bin/bash
# Collect the rocprofv3 evidence for the bench.py roofline line (run on the GPU box through gpurun):
#   bash tools/collect_profiles.sh [extra bench.py args]
# Writes raw CSVs under gpurun_out/prof/ and a summary JSON (tools/pmc_summary.py) next to them.
# Round 3: the resident kernel serves the whole timed region as ONE dispatch (a job per action), so every pass runs the same
# 20 timed actions; the summary picks that dispatch (the longest k_steps_resident dispatch of the run) and divides by the actions.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-prof}
STEPS=${PROF_STEPS:-20}
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --cpu-steps 0 --batch-envs 0 --side-configs 0 --steps $STEPS --warmup 5 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $BENCH > "$OUT/kt.log" 2>&1
echo "kernel-trace rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc1" -o p -- $BENCH > "$OUT/pmc1.log" 2>&1
echo "pmc1 rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc2" -o p -- $BENCH > "$OUT/pmc2.log" 2>&1
echo "pmc2 rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc3" -o p -- $BENCH > "$OUT/pmc3.log" 2>&1
echo "pmc3 rc=$?"
cd "$R" && PROF_STEPS=$STEPS python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.json" && cat "$OUT/summary.json" | head -80
