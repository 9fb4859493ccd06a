# Forward-sweep occupancy experiment: bench line of one config under different LDS footprints / instances per wave.
# usage: scripts/gpu_occupancy_experiment.sh <config>
cd $GRAFT_REPO_ROOT
c=${1:-2}
run() {
  echo "== $*"
  env "$@" python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(d['ms_per_step'], d['roofline']['kernel_ms'])
"
}
run X=0
run ALTRO_HIP_FWD_PER_WAVE=2
run ALTRO_HIP_FWD_PER_WAVE=1
run ALTRO_X_PAD_LDS=20000
run ALTRO_X_PAD_LDS=20000 ALTRO_HIP_FWD_PER_WAVE=1
