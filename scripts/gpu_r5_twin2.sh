# Round 5: twin workgroups -- where the tail goes: persistent launch duration (HIP events) with / without twins, lag sweep
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import importlib, os, sys, subprocess
CHILD = r'''
import importlib, os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
s = P.batch_turn90(make, batch=4096, seed=P.SEED_BASE + 3)
s.set_options(profiler_enable=1)
rows = []
for rep in range(5):
    s.reset_trajectory(); s.solve(); tm = s.get_timing()
    rows.append((round(tm["total_ms"], 2), round(tm["fused_ms"], 2), round(tm["total_ms"] - tm["fused_ms"], 2), tm["twin_handovers"], tm["sweeps"], tm["fused_sweeps"]))
print(os.environ.get("ALTRO_HIP_TWIN", "1"), os.environ.get("ALTRO_HIP_TWIN_LAG", "-"), "(total, fused, rest, handovers, sweeps, fused_sweeps)", rows[1:], flush=True)
'''
for env in ({"ALTRO_HIP_TWIN": "0"}, {}, {"ALTRO_HIP_TWIN_LAG": "0"}, {"ALTRO_HIP_TWIN_LAG": "6"}, {"ALTRO_HIP_TWIN_LAG": "10"}, {"ALTRO_HIP_TWIN_LAG": "16"}):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr[-400:])
PY
