# Round 5: timeline of config 2's chains of sweeps and of its persistent launch (ALTRO_HIP_SWEEP_LOG, all sweeps of chain 0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ALTRO_HIP_SWEEP_LOG=all timeout 120 python - 2>&1 <<'PY' | grep -v "chain [123]" | tail -40
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
s = P.batch_turn90(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=4096, seed=P.SEED_BASE + 3)
s.solve(); s.reset_trajectory(); s.set_options(profiler_enable=1); s.solve()
print(s.get_timing())
PY
