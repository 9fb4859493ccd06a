# Round 4: rocprofv3 passes of the four GPU configs on the final build + the default bench line + 2 ranks on one GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 2 3 4 1; do
  timeout 900 bash scripts/gpu_profile.sh $c > gpurun_out/r4_profile_c$c.log 2>&1
  tail -2 gpurun_out/r4_profile_c$c.log
done
timeout 600 python bench.py 2>gpurun_out/r4_bench_default.err | tail -1 > gpurun_out/r4_bench_default.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench_default.json"))
print("default line:", d["value"], d["ms_per_step"], "cpu", d["cpu_baseline"]["value"], "ff", d["fast_forward"] and d["fast_forward"]["ms_per_step"],
      "copy", d["roofline"].get("peak_measured_copy"), "host cores", d["config"]["host_cpu_cores_per_rank"])
print({k: (v["ms_per_step"], v["value"], v["solved_fraction"]) for k, v in d["other_configs"].items()})
PY
