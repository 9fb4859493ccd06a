# Round 5: bench.py's headline with and without the shadow columns of the segments, alternating on one box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for rep in 1 2 3; do for seg in 0 1; do
  ALTRO_HIP_SEGMENTS=$seg python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward --no-pipeline2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('segments=$seg', d['ms_per_step'], d['value'], d['roofline']['avg_launch_us'], d['roofline']['kernel_wall_ms'])"
done; done
