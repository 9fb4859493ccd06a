# Round 5: segments of rejection streaks -- ms per solve of configs 3 / 2 for the split threshold and the number of parts
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
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
s.set_options(profiler_enable=1)
rows = []
for rep in range(4):
    s.reset_trajectory(); t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0; tm = s.get_timing()
    rows.append((round(1e3 * dt, 2), round(tm["fused_ms"], 2), tm["sweep_launches"], tm["fused_sweeps"], tm["twin_handovers"]))
st = s.get_stats()
print(name, {k: os.environ.get(k) for k in ("ALTRO_HIP_SEGMENTS", "ALTRO_HIP_SEG_BELOW", "ALTRO_HIP_SEG_PARTS")}, "(ms, fused ms, sweep launches, fused sweeps, handovers)", rows[1:],
      "solved", int((st["status"] == 0).sum()), "its", int(st["iterations_total"].sum()), flush=True)
'''
for name in ("c3", "c2"):
    for env in ({"ALTRO_HIP_SEGMENTS": "0"}, {"ALTRO_HIP_SEG_BELOW": "25"}, {"ALTRO_HIP_SEG_BELOW": "50"}, {"ALTRO_HIP_SEG_BELOW": "100"},
                {"ALTRO_HIP_SEG_BELOW": "50", "ALTRO_HIP_SEG_PARTS": "2"}, {"ALTRO_HIP_SEG_BELOW": "35", "ALTRO_HIP_SEG_PARTS": "3"}):
        if name == "c2" and env.get("ALTRO_HIP_SEG_BELOW") in ("25", "35"):
            continue
        r = subprocess.run([sys.executable, "-c", CHILD, name], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        print(r.stdout.strip() or r.stderr[-500:], flush=True)
PY
ALTRO_HIP_TWIN_DEBUG=1 ALTRO_HIP_SEG_BELOW=50 timeout 120 python - 2>&1 <<'PY' | grep -v "slot" | tail -8
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
s = P.batch_three_obstacles(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=4096, dtype=A.F32)
s.solve(); s.reset_trajectory(); s.solve()
PY
