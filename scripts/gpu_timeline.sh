# Timeline of the chains of sweeps of one solve (kernel trace): per sweep and chain the three kernel durations, the gaps
# between them and the grid sizes, bucketed by the number of instances the sweep worked on.
#   scripts/gpu_timeline.sh [config]   -> gpurun_out/timeline_c<config>.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
C=${1:-3}
P=gpurun_out/timeline_c$C
rm -rf $P; mkdir -p $P
rocprofv3 --kernel-trace --output-format csv -d $P/trace -o bench -- python bench.py --config $C --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-latency > $P/run.log 2>&1
python - $P $C <<'PY'
import csv, glob, sys, collections
P, C = sys.argv[1], sys.argv[2]
f = glob.glob(f'{P}/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# the last solve: from the last k_al_init on
last = max(i for i, r in enumerate(rows) if 'k_al_init' in r['Kernel_Name'])
rows = rows[last:]
t0 = rows[0]['s']
def kind(n):
    for k in ('k_expansions', 'k_backward', 'k_forward', 'k_sweep_fused'):
        if k in n: return k[2]
    return None
chains = collections.defaultdict(list)
for r in rows:
    k = kind(r['Kernel_Name'])
    if k: chains[r['Stream_Id']].append((k, r['s'] - t0, r['e'] - t0, int(r['Grid_Size_X']), int(r['Workgroup_Size_X'])))
out = open(f'gpurun_out/timeline_c{C}.txt', 'w')
buckets = collections.defaultdict(lambda: collections.defaultdict(list))
for sid, ks in sorted(chains.items()):
    out.write(f'# stream {sid}: {len(ks)} kernels\n')
    i = 0
    sweep = 0
    while i + 2 < len(ks) and ks[i][0] == 'e' and ks[i + 1][0] == 'b' and ks[i + 2][0] == 'f':
        E, B, F = ks[i], ks[i + 1], ks[i + 2]
        nxt = ks[i + 3] if i + 3 < len(ks) else None
        inst_f = F[3] // F[4]    # forward workgroups
        inst_b = B[3] // B[4]
        d = dict(E=(E[2] - E[1]) / 1e3, B=(B[2] - B[1]) / 1e3, F=(F[2] - F[1]) / 1e3, gEB=(B[1] - E[2]) / 1e3, gBF=(F[1] - B[2]) / 1e3,
                 gFE=((nxt[1] - F[2]) / 1e3 if nxt and nxt[0] == 'e' else float('nan')), period=((nxt[1] - E[1]) / 1e3 if nxt and nxt[0] == 'e' else float('nan')))
        if sweep % 5 == 0 or sweep > 140:
            out.write(f"  sweep {sweep:3d} t={E[1]/1e6:7.3f} ms  wgB {inst_b:5d} wgF {inst_f:5d}  E {d['E']:6.1f} gap {d['gEB']:5.1f} B {d['B']:6.1f} gap {d['gBF']:5.1f} F {d['F']:6.1f} gap {d['gFE']:5.1f}  period {d['period']:6.1f}\n")
        b = 1 << max(0, inst_b - 1).bit_length()
        for k, v in d.items():
            if v == v: buckets[b][k].append(v)
        sweep += 1; i += 3
    for k in ks[i:]:
        out.write(f"  tail kernel {k[0]} t={k[1]/1e6:.3f} ms dur {(k[2]-k[1])/1e3:.1f} us grid {k[3]//k[4]}\n")
out.write('# by backward workgroups (<= bucket): mean us of E, B, F, the gaps and the sweep period; n = sweeps\n')
for b in sorted(buckets):
    d = buckets[b]
    out.write(f"  <= {b:5d}: n {len(d['E']):4d}  " + '  '.join(f"{k} {sum(v)/len(v):6.1f}" for k, v in d.items()) + '\n')
out.close()
print(open(f'gpurun_out/timeline_c{C}.txt').read()[-3000:])
PY
rm -rf $P/trace
