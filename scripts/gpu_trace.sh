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
    short = 'fused' if 'k_sweep_fused' in name else 'fwd' if 'k_forward' in name else 'bwd' if 'k_backward' in name else 'exp' if 'k_expansions' in name else None
    if short: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),short))
ev.sort()
# first solve = up to the first long gap (> 1 ms) after it started
end = len(ev)
for i in range(1, len(ev)):
    if ev[i][0] - ev[i-1][1] > 1_000_000: end = i; break
seq = ev[:end]
def show(lo, hi):
    for j in range(lo, min(hi, len(seq)-1)):
        s,e,k = seq[j]
        print(k, 'dur %.1f' % ((e-s)/1e3), 'gap_to_next %.1f' % ((seq[j+1][0]-e)/1e3))
    print('--')
show(0, 7); show(15, 22)
nf = [i for i,e in enumerate(seq) if e[2]=='fused']
if nf: show(nf[0]-3, nf[0]+4); show(nf[len(nf)//2], nf[len(nf)//2]+4)
import collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for s,e,k in seq: tot[k] += (e-s)/1e3; cnt[k] += 1
print({k: (cnt[k], round(tot[k]/1e3,3)) for k in tot}, 'first solve span ms', (seq[-1][1]-seq[0][0])/1e6)
PY
