# Round 5, first GPU visit: the whole -m gpu suite (no -x: every failure is seen), the parity ledger, smoke, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r05_parity_errors.json
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  " | tail -40 | tee gpurun_out/r5_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r5_smoke.log
timeout 900 python bench.py --steps 10 --warmup 2 2>gpurun_out/r5_bench.err | tail -1 > gpurun_out/r5_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5_bench.json"))
print("headline", d["value"], d["ms_per_step"], "tail_iteration_us", d["roofline"].get("tail_iteration_us"),
      "ff", d.get("fast_forward"), "\npipeline_2", d.get("pipeline_2"))
print({k: (v["ms_per_step"], v["value"]) for k, v in (d.get("other_configs") or {}).items()})
print(json.dumps(d.get("latency"))[:600])
PY
tail -5 gpurun_out/r5_bench.err
