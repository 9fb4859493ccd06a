cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_run.log 2>&1
ls -R gpurun_out/prof | head
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
per = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name']
    short = 'fwd' if 'k_forward' in name else 'bwd' if 'k_backward' in name else 'exp' if 'k_expansions' in name else None
    if short: per[short].append((int(r['Start_Timestamp']), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
for k,v in per.items():
    v.sort()
    d=[x[1] for x in v]
    n=len(d)//2  # bench runs 2 solves (timed + profiled)
    print(k, 'launches', len(d), 'first solve durations us:', [round(x) for x in d[:n][:24]], '...', [round(x) for x in d[:n][-6:]])
PY
