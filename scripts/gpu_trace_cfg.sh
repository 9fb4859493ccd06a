# Kernel-trace statistics of one bench configuration: scripts/gpu_trace_cfg.sh <config> <tag> [ENV=VAL ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
c=$1; tag=$2; shift 2
O=gpurun_out/trace_$tag
rm -rf $O; mkdir -p $O
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o bench -- python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O/raw -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
python - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"altro_hip::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    print(f"{name:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {r['Percentage']:>6s}%")
PY
rm -rf $O/raw
