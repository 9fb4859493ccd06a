# round 3: forward pass reading its rollout inputs from global memory (3 waves per SIMD) vs staged in LDS, config 2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for src in default lds global; do
    if [ $src = default ]; then unset ALTRO_HIP_FWD_SRC; else export ALTRO_HIP_FWD_SRC=$src; fi
    for c in ${@:-2}; do
    timeout 300 python bench.py --config $c --no-cpu-baseline --no-other-configs --no-latency 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; print('fwd_src', '$src', 'config', $c, d['ms_per_step'], d['value'], 'kernel_ms', r['kernel_ms'], 'wall', r['kernel_wall_ms'])
" | tee -a gpurun_out/fwd_src.log
    done
done
done
