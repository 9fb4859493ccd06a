# Round 4, first GPU visit: the whole -m gpu suite, smoke, the default bench line, and the host-wait A/B
# (ALTRO_HIP_HOST_WAIT=spin against the default back-off: ms per step and CPU cores burnt per rank).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^E  " | tail -25 | tee gpurun_out/r4_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r4_smoke.log
timeout 600 python bench.py --steps 10 --warmup 2 2>gpurun_out/r4_bench.err | tail -1 > gpurun_out/r4_bench.json
for w in backoff spin; do
  for c in 2 3; do
    ALTRO_HIP_HOST_WAIT=$w timeout 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward 2>/dev/null | tail -1 > gpurun_out/r4_wait_${w}_c$c.json
  done
done
python - <<'PY'
import json
for w in ("backoff", "spin"):
    for c in (2, 3):
        try:
            d = json.load(open(f"gpurun_out/r4_wait_{w}_c{c}.json"))
            print(w, "config", c, "ms_per_step", d["ms_per_step"], "host_cpu_cores", d["config"].get("host_cpu_cores_per_rank"))
        except Exception as e:
            print(w, c, "failed", e)
d = json.load(open("gpurun_out/r4_bench.json"))
print("headline", d["value"], d["ms_per_step"], "tail_iteration_us", d["roofline"].get("tail_iteration_us"),
      "copy", d["roofline"].get("peak_measured_copy"), "ff", d.get("fast_forward"))
print({k: (v["ms_per_step"], v["value"]) for k, v in (d.get("other_configs") or {}).items()})
print(json.dumps(d.get("latency"))[:800])
PY
