# The start of a solve as one launch (k_begin_solve) against the seven stream operations (ALTRO_HIP_BEGIN_SOLVE=split):
# the bench's latency block (batch 1 / 8 / 64, cold and warm) for both, alternating, then the kernel trace of a batch of one.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
  for mode in merged split; do
    ALTRO_HIP_BEGIN_SOLVE=$mode python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-fast-forward --no-pipeline2 --no-device-loop 2>/dev/null | tail -1 |
      python -c "
import json, sys
d = json.loads(sys.stdin.readline())
l = d['latency']
print('$mode', 'ms_per_step', d['ms_per_step'], {p: {b: (v['cold_ms'], v['warm_ms']) for b, v in l[p].items()} for p in ('kTurn90', 'kThreeObstacles')})
" | tee -a gpurun_out/latency_ab.log
  done
done
bash scripts/gpu_latency_trace.sh 1 | head -16
