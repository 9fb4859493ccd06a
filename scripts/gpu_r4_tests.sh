# Round 4: the whole -m gpu suite (no -x: every failure listed) + smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  " | tail -60 | tee gpurun_out/r4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r4_smoke.log
for c in 1 2; do
  timeout 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward 2>/dev/null | tail -1 > gpurun_out/r4_quick_c$c.json
  python -c "
import json; d=json.load(open('gpurun_out/r4_quick_c$c.json')); print('config $c', d['ms_per_step'], d['value'], 'cpu cores', d['config'].get('host_cpu_cores_per_rank'))"
done
