# round 3, third visit: the new GPU tests (group, RCCL with one rank, per-knot steps), then the bench line's CPU leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_group.py tests/test_rccl_world1_gpu.py tests/test_knot_times_gpu.py -q -m gpu 2>&1 | grep -v "^E  " | tail -40 | tee gpurun_out/pytest_new.log
python bench.py --no-other-configs --no-latency 2>&1 | tail -1 > gpurun_out/bench_cpu.log
python - <<'PY'
import json
l = json.loads(open('gpurun_out/bench_cpu.log').read())
print(json.dumps(l["cpu_baseline"], indent=1))
print(l["value"], l["ms_per_step"])
PY
