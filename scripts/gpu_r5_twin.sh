# Round 5: twin workgroups of the persistent kernel -- the bit-identity tests, then A/B bench lines (ALTRO_HIP_TWIN=0 | 1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py -q -x -s 2>&1 | grep -v "^E  " | tail -30 | tee gpurun_out/r5_twin_tests.log
for tw in 1 0; do
  for c in 2 3; do
    ALTRO_HIP_TWIN=$tw timeout 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward --no-pipeline2 2>gpurun_out/r5_twin_${tw}_c$c.err | tail -1 > gpurun_out/r5_twin_${tw}_c$c.json
  done
done
python - <<'PY'
import json
for tw in (1, 0):
    for c in (2, 3):
        try:
            d = json.load(open(f"gpurun_out/r5_twin_{tw}_c{c}.json"))
            r = d["roofline"]
            print("twin", tw, "config", c, "ms_per_step", d["ms_per_step"], "value", d["value"], "solved", d["config"]["solved_fraction"],
                  "dominant", r.get("kernel"), r.get("avg_launch_us"), "tail_iteration_us", r.get("tail_iteration_us"))
        except Exception as e:
            print(tw, c, "failed", e, open(f"gpurun_out/r5_twin_{tw}_c{c}.err").read()[-500:])
PY
