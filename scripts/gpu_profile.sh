# Profiles for profiles/: kernel trace stats + HBM traffic counters (separate --pmc passes).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profile
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf gpurun_out/profile/*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profile/trace -o bench -- $CMD > gpurun_out/profile/trace_run.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/profile/fetch -o bench -- $CMD > gpurun_out/profile/fetch_run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/profile/write -o bench -- $CMD > gpurun_out/profile/write_run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/profile/sq -o bench -- $CMD > gpurun_out/profile/sq_run.log 2>&1
python - <<'PY'
import csv, glob, collections, json
def short(name):
    for k, v in (('k_sweep_fused', 'k_sweep_fused'), ('k_forward', 'k_forward'), ('k_backward', 'k_backward'), ('k_expansions', 'k_expansions'), ('k_rollout', 'k_rollout'), ('k_al_init', 'k_al_init'), ('k_solve_setup', 'k_solve_setup'), ('k_pack_results', 'k_pack_results')):
        if k in name: return v
    return name[:40]
out = {}
for tag in ('fetch', 'write', 'sq'):
    f = glob.glob(f'gpurun_out/profile/{tag}/**/*counter_collection.csv', recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = {'launches': len(v), 'sum': sum(v), 'avg': sum(v) / len(v)}
f = glob.glob('gpurun_out/profile/trace/**/*kernel_stats.csv', recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        out.setdefault(short(r['Name']), {})['trace'] = {'calls': int(r['Calls']), 'total_ns': float(r['TotalDurationNs']), 'avg_ns': float(r['AverageNs']), 'pct': float(r['Percentage'])}
json.dump(out, open('gpurun_out/profile/summary.json', 'w'), indent=1)
for k in ('k_sweep_fused', 'k_forward', 'k_backward', 'k_expansions'):
    print(k, json.dumps(out.get(k, {}))[:900])
PY
cp gpurun_out/profile/trace/*kernel_stats.csv gpurun_out/profile/kernel_stats.csv 2>/dev/null
python bench.py --steps 5 --warmup 1 2>&1 | tail -1 > gpurun_out/profile/bench_line.json
ls gpurun_out/profile
