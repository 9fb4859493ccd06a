# Kernel trace of one bench step: per-kernel durations and gaps in the tail sweeps.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_run.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev=[]
for r in rows:
    name=r['Kernel_Name']
    short = 'fwd' if 'k_forward' in name else 'bwd' if 'k_backward' in name else 'exp' if 'k_expansions' in name else 'oth:'+name[:30]
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),short))
ev.sort()
seq=[e for e in ev]
first=[i for i,e in enumerate(seq) if e[2]=='exp'][0]
print('kernels', len(seq))
for base in (first, first+3*5, first+3*60):
    for j in range(base, base+7):
        s,e,k=seq[j]
        print(k, 'dur %.1f' % ((e-s)/1e3), 'gap_to_next %.1f' % ((seq[j+1][0]-e)/1e3))
    print('--')
n=3*119
tot=seq[first+n-1][1]-seq[first][0]
print('first solve span ms',tot/1e6, 'kernel sum ms', sum(e-s for s,e,k in seq[first:first+n])/1e6)
PY
