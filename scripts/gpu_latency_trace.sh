# Kernel trace of a batch-of-one solve (the MPC case): which kernels, how long, what lies between them.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P=gpurun_out/latency_trace
rm -rf $P; mkdir -p $P
cat > /tmp/lat1.py <<'PY'
import importlib, sys, time
sys.path.insert(0, ".")
import __graft_entry__ as g
A = g.load_package(); P = importlib.import_module("altro_cpp_amd.problems")
s = P.batch_turn90(lambda n, m, N, b, d: A.BatchSolver(n, m, N, b, d), batch=int(sys.argv[1]))
for i in range(6):
    s.reset_trajectory()
    t0 = time.perf_counter(); s.solve(); dt = time.perf_counter() - t0
    print("solve", i, round(1e3 * dt, 3), "ms", s.get_stats()["iterations_total"].max())
PY
rocprofv3 --kernel-trace --output-format csv -d $P/trace -o lat -- python /tmp/lat1.py ${1:-1} > $P/run.log 2>&1
python - $P <<'PY'
import csv, glob, sys
P = sys.argv[1]
rows = sorted(csv.DictReader(open(glob.glob(f'{P}/trace/**/*kernel_trace.csv', recursive=True)[0])), key=lambda r: int(r['Start_Timestamp']))
last = [i for i, r in enumerate(rows) if 'k_al_init' in r['Kernel_Name'] or 'k_begin_solve' in r['Kernel_Name']][-1]
t0 = int(rows[last - 2]['Start_Timestamp']) if last >= 2 else int(rows[last]['Start_Timestamp'])
prev = None
with open('gpurun_out/latency_trace.txt', 'w') as out:
    for r in rows[max(0, last - 3):]:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        name = r['Kernel_Name'].split('(')[0].replace('void altro_hip::', '')[:60]
        line = f"{s/1e3:9.1f} us  +{(s - prev)/1e3 if prev is not None else 0:7.1f} gap  {(e - s)/1e3:8.1f} us  {name}  grid {r['Grid_Size_X']}"
        print(line); out.write(line + "\n")
        prev = e
PY
tail -8 $P/run.log
rm -rf $P/trace
