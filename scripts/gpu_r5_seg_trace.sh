# Round 5, segments: per-kernel durations of config 2 with and without the shadow columns allocated (kernel trace)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp REPS=8
for seg in 0 1; do
  rm -rf /tmp/tr$seg
  ALTRO_HIP_SEGMENTS=$seg rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$seg -o t -- python scripts/probe_seg_policy.py c2 "X=1" > /tmp/tr$seg.log 2>&1
  tail -2 /tmp/tr$seg.log | cut -c1-200
done
python - <<'PY'
import csv, glob, re
def load(d):
    out = {}
    for f in glob.glob(f'{d}/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r'\(.*', '', r['Name'])[:90]
            out[n] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3)
    return out
a, b = load('/tmp/tr0'), load('/tmp/tr1')
print("kernel | calls | total us without / with | avg us without / with")
for n in sorted(set(a) | set(b), key=lambda n: -(b.get(n, (0, 0, 0))[1])):
    x, y = a.get(n, (0, 0, 0)), b.get(n, (0, 0, 0))
    if max(x[1], y[1]) > 200: print(f"{n:90s} {x[0]:5d}/{y[0]:5d}  {x[1]:10.0f} {y[1]:10.0f}   {x[2]:8.1f} {y[2]:8.1f}")
PY
