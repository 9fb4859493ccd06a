# Phase stamps of the persistent kernel (debug build of the unicycle fp64 engine with -DALTRO_STAMPS, see the Makefile
# recipe in scripts/README.md): one speculated iteration of workgroup 0, shader-clock cycles, per speculation mode.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export ALTRO_HIP_LIB=$GRAFT_REPO_ROOT/altro-cpp_amd/csrc/_x/libaltro_stamps.so
for mode in ${@:-wave free}; do
  echo "== ALTRO_HIP_SPECULATION=$mode"
  ALTRO_HIP_SPECULATION=$mode python - 2>&1 <<'PY' | grep -v "^  *$" | awk '/STAMPS/ {n++} n <= 2 || !/STAMPS|^  /' | head -40
import importlib, sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
hm = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
for name, fac, kw, B in (("turn90", P.batch_turn90, {}, 64), ("obstacles", P.batch_three_obstacles, {"dtype": A.F64}, 64)):
    s = fac(hm, batch=B, **kw)
    t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0
    print(f"-- {name} B={B}: {1e3 * dt:.3f} ms (first solve), longest chain {s.get_timing()['sweeps']}", flush=True)
    s.close()
PY
done
