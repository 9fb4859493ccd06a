# Per-launch durations of the forward kernels of one solve: scripts/gpu_trace_launches.sh <config> [ENV=VAL ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
c=$1; shift
O=gpurun_out/trace_launches; rm -rf $O; mkdir -p $O
env "$@" rocprofv3 --kernel-trace --output-format csv -d $O/raw -o bench -- python bench.py --config $c --steps 1 --warmup 0 --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    key = "fwd_s1" if "k_forward2" in n and ", 8, true" in n else "fwd20" if "k_forward2" in n else "bwd" if "k_backward" in n else "exp" if "k_expansions" in n else None
    if key: per[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
    half = v[: len(v) // 2]  # first of the two solves (timed step + profiling solve)
    print(k, len(half), [round(x) for x in half[:40]])
PY
rm -rf $O/raw
