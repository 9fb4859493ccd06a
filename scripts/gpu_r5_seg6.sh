# Round 5, segments: timeline + persistent-launch summary of config 3 under one policy (environment of the caller)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ALTRO_HIP_SWEEP_LOG=1 ALTRO_HIP_TWIN_DEBUG=1 timeout 120 python - 2>&1 <<'PY' | grep -v "^ *slot" | grep -v "chain [123]" | tail -70
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
s = P.batch_three_obstacles(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=4096, dtype=A.F32)
s.solve(); s.reset_trajectory(); s.set_options(profiler_enable=1); s.solve()
PY
