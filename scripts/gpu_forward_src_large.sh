# round 6: config 4's forward pass -- the global-source variant for the large models (two workgroups per CU) against kSrcKdg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for src in global kdg global kdg; do
  echo "== ALTRO_HIP_FWD_SRC=$src"
  ALTRO_HIP_FWD_SRC=$src timeout 600 python bench.py --config 4 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('  ms', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'dominant', r['kernel'], 'avg_launch_us', r['avg_launch_us'], 'solved', d['config']['solved_fraction'])"
done
python - <<'PY'
import importlib, os, subprocess, sys, tempfile
import numpy as np
code = r'''
import importlib, sys, numpy as np
sys.path.insert(0, ".")
import __graft_entry__ as g
A = g.load_package()
P = importlib.import_module("altro_cpp_amd.problems")
make = lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d)
out = {}
for name, dt in (("r32", A.F32), ("f64", A.F64)):
    s = P.batch_quadrotor12(make, batch=300, dtype=dt)
    s.solve()
    X, U = s.get_trajectory(); st = s.get_stats()
    out[name + "_X"] = X; out[name + "_U"] = U; out[name + "_lam"] = s.get_duals(); out[name + "_c"] = s.get_constraint_values()
    for f in st.dtype.names: out[name + "_" + f] = st[f]
np.savez(sys.argv[1], **out)
'''
res = {}
with tempfile.TemporaryDirectory() as tmp:
    for src in ("global", "kdg"):
        out = os.path.join(tmp, src + ".npz")
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, ALTRO_HIP_FWD_SRC=src), timeout=600)
        res[src] = dict(np.load(out))
bad = [k for k in res["global"] if not np.array_equal(res["global"][k], res["kdg"][k], equal_nan=True)]
print("quadrotor12, 300 instances, fp32 records + fp64: global-source variant vs kSrcKdg:", "BIT-IDENTICAL" if not bad else "DIFFERENT " + str(bad))
PY
} 2>&1 | tee gpurun_out/forward_src_large.log
