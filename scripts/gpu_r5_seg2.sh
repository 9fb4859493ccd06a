# Round 5, segments of rejection streaks, second look: timelines of config 3's chains of sweeps with and without segments
# (ALTRO_HIP_SWEEP_LOG), then ms per solve for split policies.  Needs the segments build (branch segments2).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python - > gpurun_out/r5_seg2.log 2>&1 <<'PY'
import os, subprocess, sys
CHILD = r'''
import importlib, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
name = sys.argv[1]
s = P.batch_three_obstacles(make, batch=4096, dtype=A.F32) if name == "c3" else P.batch_turn90(make, batch=4096, seed=P.SEED_BASE + 3)
log = os.environ.pop("WANT_SWEEP_LOG", None)
rows = []
for rep in range(4):
    if log and rep == 3:
        os.environ["ALTRO_HIP_SWEEP_LOG"] = "1"; s.set_options(profiler_enable=1)
    s.reset_trajectory(); t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0; tm = s.get_timing()
    rows.append((round(1e3 * dt, 2), round(tm["fused_ms"], 2), tm["sweep_launches"], tm["fused_sweeps"], tm["twin_handovers"]))
st = s.get_stats()
print(name, {k: os.environ.get(k) for k in ("ALTRO_HIP_SEGMENTS", "ALTRO_HIP_SEG_BELOW", "ALTRO_HIP_SEG_ABOVE", "ALTRO_HIP_SEG_PARTS")},
      "(ms, fused ms, sweep launches, fused sweeps, handovers)", rows[1:],
      "solved", int((st["status"] == 0).sum()), "its", int(st["iterations_total"].sum()), flush=True)
'''
def run(name, env, log=False):
    e = dict(os.environ, **env)
    if log: e["WANT_SWEEP_LOG"] = "1"
    r = subprocess.run([sys.executable, "-c", CHILD, name], env=e, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr[-800:], flush=True)
    if log:
        lines = [l for l in r.stderr.splitlines() if l.startswith("SWEEPLOG") and ("chain 0" in l or "persistent" in l)]
        print("\n".join(lines), flush=True)
run("c3", {"ALTRO_HIP_SEGMENTS": "0"}, log=True)
run("c3", {}, log=True)
for env in ({"ALTRO_HIP_SEG_BELOW": "75"}, {"ALTRO_HIP_SEG_BELOW": "100"}, {"ALTRO_HIP_SEG_PARTS": "3"}, {"ALTRO_HIP_SEG_PARTS": "6"},
            {"ALTRO_HIP_SEG_ABOVE": "1536"}, {"ALTRO_HIP_SEG_ABOVE": "512"}):
    run("c3", env)
run("c2", {"ALTRO_HIP_SEGMENTS": "0"})
run("c2", {})
PY
tail -150 gpurun_out/r5_seg2.log
