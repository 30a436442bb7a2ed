#!/bin/bash
# The fast-host stress over a range of grid sizes (GPU box): self-consistency (pauses + short idle limit against undisturbed) and a
# cross check against the single-step kernels, fresh seeds per size.   tools/stress/sweep.sh [first_seed 7000] [seeds per size 12]
set -u -o pipefail
first=${1:-7000}
per=${2:-12}
bad=0
for n in 128 200 256 320 450 500 550 600 660 700 720 800; do
    for cross in 0 1; do
        out=$(timeout -k 10 240 tools/stress/stress_host $n $((first + n)) $per 160 0 $cross 0 2>&1 | tail -1)
        rc=$?
        echo "grid $n cross $cross: $out (rc $rc)"
        [ $rc -ne 0 ] && bad=1 && break 2
    done
done
[ $bad -eq 0 ] && echo "SWEEP PASS" || echo "SWEEP FAILED"
exit $bad
