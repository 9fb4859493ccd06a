# Round 5: A/B of variant builds (scripts/build_variant.sh) against the default library on one box: ms per step, configs 2 3 4 1
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
X=$GRAFT_REPO_ROOT/altro-cpp_amd/csrc
for rep in 1 2; do
  for c in ${CONFIGS:-2 3}; do
    for lib in $X/libaltro_hip.so $@; do
      ALTRO_HIP_LIB=$lib timeout 200 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward --no-pipeline2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$lib'.split('/')[-1], 'config', $c, 'ms', d['ms_per_step'], d['roofline']['kernel_ms'])
"
    done
  done
done
