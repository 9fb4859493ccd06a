# Chains of sweeps: bench lines for different chain counts / hand-over points: scripts/gpu_chains.sh
cd $GRAFT_REPO_ROOT
run() {  # config, env...
  c=$1; shift
  env "$@" python bench.py --config $c --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$*', 'config', $c, d['ms_per_step'], d['config']['sweeps'], d['roofline']['kernel_wall_ms'])
"
}
for rep in 1 2; do
run 4 ALTRO_HIP_CHAINS=1
run 4 ALTRO_HIP_CHAINS=2
run 4 ALTRO_HIP_CHAINS=4
run 1 ALTRO_HIP_CHAINS=1
run 1 ALTRO_HIP_CHAINS=2
run 1 ALTRO_HIP_CHAINS=4
run 2 ALTRO_HIP_CHAINS=3
run 2 ALTRO_HIP_CHAINS=4
run 2 ALTRO_HIP_CHAINS=4 ALTRO_HIP_PERSIST_AT=384
run 2 ALTRO_HIP_CHAINS=4 ALTRO_HIP_PERSIST_AT=192
run 3 ALTRO_HIP_CHAINS=3
run 3 ALTRO_HIP_CHAINS=4
run 3 ALTRO_HIP_CHAINS=4 ALTRO_HIP_PERSIST_AT=384
done
