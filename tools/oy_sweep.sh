# tile-height experiments on the GPU box: bash tools/oy_sweep.sh   (prints grid, autotile, OY caps, Mcell/s, kernel us, fractions)
run() {
  WAVES_AMD_FUSED_AUTOTILE=$2 WAVES_AMD_FUSED_OY=$3 python3 bench.py --grid $1 --cpu-steps 0 --batch-envs 0 --side-configs 0 --steps ${4:-30} --warmup 3 > gpurun_out/oy.json 2>gpurun_out/oy.err
  python3 - "$1 auto=$2 oy=$3" <<'PY'
import json,sys
d=json.loads(open('gpurun_out/oy.json').read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['value'], r['avg_kernel_us'], r['frac'], r['whole_job_frac'], r['kernel'])
PY
}
for g in 256 384 500 600; do run $g 0 0,0,0,0; run $g 1 0,0,0,0; done
run 700 0 24,16,16,8; run 700 1 0,0,0,0
