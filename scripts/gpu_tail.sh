cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/tail
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tail -o t -- python scripts/probe_tail.py > gpurun_out/tail_run.log 2>&1
tail -3 gpurun_out/tail_run.log
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/tail/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
per = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name']
    short = 'fwd' if 'k_forward' in name else 'bwd' if 'k_backward' in name else 'exp' if 'k_expansions' in name else None
    if short: per[short].append((int(r['Start_Timestamp']), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
for k,v in per.items():
    v.sort()
    d=[x[1] for x in v]
    n=len(d)//2
    print(k, 'launches', len(d), [round(x) for x in d[n:n+30]], '...', [round(x) for x in d[-8:]])
PY
