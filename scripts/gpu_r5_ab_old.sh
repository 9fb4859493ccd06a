# Round 5: the build before the segments (altro-cpp_amd/csrc/_x/libaltro_pre_segments.so, commit 0c36f99) against the shipped one
# (built with: git archive 0c36f99 altro-cpp_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/altro-cpp_amd/csrc libaltro_hip.so;
#  copied to altro-cpp_amd/csrc/_x/, which is not tracked)
# on ONE box: config 2 / 3 medians and the single-instance latencies (boxes differ by up to 8 % in latency-bound launches)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp REPS=16
OLD=$GRAFT_REPO_ROOT/altro-cpp_amd/csrc/_x/libaltro_pre_segments.so
python scripts/probe_seg_policy.py c2 "ALTRO_HIP_LIB=$OLD" "X=1" "ALTRO_HIP_LIB=$OLD" "X=1" 2>&1 | cut -c1-250
REPS=8 python scripts/probe_seg_policy.py c3 "ALTRO_HIP_LIB=$OLD" "X=1" 2>&1 | cut -c1-250
for lib in $OLD ""; do
ALTRO_HIP_LIB=$lib python - <<'PY'
import importlib, os, sys, time, statistics
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for name, fac in (("kTurn90 batch 1", lambda: P.batch_turn90(make, batch=1, seed=P.SEED_BASE + 3)), ("kThreeObstacles batch 1", lambda: P.batch_three_obstacles(make, batch=1, dtype=A.F64))):
    s = fac(); ms = []
    for rep in range(40):
        s.reset_trajectory(); t0 = time.perf_counter(); s.solve(); ms.append(1e3 * (time.perf_counter() - t0))
    print(os.environ.get("ALTRO_HIP_LIB") and "pre-segments" or "shipped     ", name, "cold ms: min %.3f median %.3f" % (min(ms[5:]), statistics.median(ms[5:])), flush=True)
PY
done
