# Round 5, segments: what the persistent launch of config 3 does with and without segments (ALTRO_HIP_TWIN_DEBUG summary)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for seg in 0 1; do
echo "== ALTRO_HIP_SEGMENTS=$seg $EXTRA"
ALTRO_HIP_SEGMENTS=$seg ALTRO_HIP_TWIN_DEBUG=1 timeout 120 python - 2>&1 <<'PY' | grep -v "^ *slot" | tail -${TAILN:-40}
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
s = P.batch_three_obstacles(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=4096, dtype=A.F32)
s.solve(); s.reset_trajectory(); s.solve()
PY
done
