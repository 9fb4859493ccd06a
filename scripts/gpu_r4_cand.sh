# Round 4: the candidate-slot layout of k_forward2 -- bit-identity test, then ms per step of configs 2, 3, 4 for several `front`
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_fused_gpu.py -q -m gpu -k "candidate" 2>&1 | tail -15 | tee gpurun_out/r4_cand_test.log
for f in 19 12 8 6 4; do
  for c in 2 3 4; do
    ALTRO_HIP_CAND_FRONT=$f timeout 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward 2>/dev/null | tail -1 > gpurun_out/r4_cand_${f}_c$c.json
    python -c "
import json; d=json.load(open('gpurun_out/r4_cand_${f}_c$c.json')); r=d['roofline']; print('front $f config $c ms', d['ms_per_step'], 'value', d['value'], 'kernel_wall_ms', r['kernel_wall_ms'], 'kernel_ms', r['kernel_ms'])"
  done
done 2>&1 | tee gpurun_out/r4_cand_sweep.log
