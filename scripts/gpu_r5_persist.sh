# Round 5: hand-over point to the persistent kernel, re-measured with twin workgroups (ALTRO_HIP_PERSIST_AT), configs 2 and 3
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in 2 3; do
  for pa in 256 384 512 768 1024; do
    ALTRO_HIP_PERSIST_AT=$pa timeout 200 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward --no-pipeline2 2>/dev/null | tail -1 > gpurun_out/r5_pa_${pa}_c$c.json
    python -c "
import json; d=json.load(open('gpurun_out/r5_pa_${pa}_c$c.json')); print('config $c persist_at $pa ms', d['ms_per_step'], 'value', d['value'], 'fused', d['roofline']['kernel_wall_ms'])"
  done
done
