# Profiles for profiles/: kernel-trace stats + HBM traffic / SQ counters (separate --pmc passes) of one bench config.
#   scripts/gpu_profile.sh [config]      (default 2; output under gpurun_out/profile_c<config>/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
C=${1:-2}
P=gpurun_out/profile_c$C
rm -rf $P; mkdir -p $P
# The SAME number of chains of sweeps in every pass: under --pmc the profiler serialises the kernels,
# the engine's side-by-side check of its chain streams fails and it would fall back to ONE chain -- launches four times
# the size of the trace pass's.  ALTRO_HIP_CHAINS forces the count (and skips the check): what a fresh process picks
# (4 for a batch >= 2048, else 1).
case $C in 2|3) export ALTRO_HIP_CHAINS=4;; *) export ALTRO_HIP_CHAINS=1;; esac
CMD="python bench.py --config $C --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward --no-pipeline2 --no-device-loop ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o bench -- $CMD > $P/trace_run.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/fetch -o bench -- $CMD > $P/fetch_run.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/write -o bench -- $CMD > $P/write_run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $P/sq -o bench -- $CMD > $P/sq_run.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $P/mfma -o bench -- $CMD > $P/mfma_run.log 2>&1
# fp64 vector instructions by type (wave-level counts) + the cycles waves spent in VALU instructions: the COMPUTE side of the
# roofline (valu_f64_frac)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $P/f64 -o bench -- $CMD > $P/f64_run.log 2>&1
# L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS), MI355X_MICROARCH.md): do the candidate stores and re-reads stay in L2?
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/tcc -o bench -- $CMD > $P/tcc_run.log 2>&1
python - $P <<'PY'
import csv, glob, collections, json, sys
P = sys.argv[1]
def short(name):
    for k in ('k_sweep_loop', 'k_sweep_fused', 'k_forward2', 'k_forward', 'k_backward_mfma16', 'k_backward_mfma', 'k_backward_coop', 'k_backward', 'k_expansions',
              'k_begin_solve', 'k_rollout', 'k_al_init', 'k_solve_setup', 'k_pack_results', 'k_set_rows', 'k_reset_stats'):
        if k in name: return k
    return name[:40]
out = {}
for tag in ('fetch', 'write', 'sq', 'mfma', 'f64', 'tcc'):
    f = glob.glob(f'{P}/{tag}/**/*counter_collection.csv', recursive=True)
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        for c, v in d.items():
            out.setdefault(k, {})[c] = {'launches': len(v), 'sum': sum(v), 'avg': sum(v) / len(v)}
f = glob.glob(f'{P}/trace/**/*kernel_stats.csv', recursive=True)
if f:
    import shutil
    shutil.copy(f[0], f'{P}/kernel_stats.csv')
    for r in csv.DictReader(open(f[0])):
        out.setdefault(short(r['Name']), {})['trace'] = {'calls': int(r['Calls']), 'total_ns': float(r['TotalDurationNs']), 'avg_ns': float(r['AverageNs']), 'pct': float(r['Percentage'])}
json.dump(out, open(f'{P}/summary.json', 'w'), indent=1)
for k, v in out.items():
    if 'trace' in v and v['trace']['pct'] > 1.0: print(k, json.dumps(v)[:700])
PY
for d in trace fetch write sq mfma f64 tcc; do rm -rf $P/$d; done
unset ALTRO_HIP_CHAINS
if [ "$C" = "2" ]; then
  python bench.py --config $C --steps 5 --warmup 1 2>/dev/null | tail -1 > $P/bench_line.json
else
  python bench.py --config $C --steps 5 --warmup 1 --no-other-configs --no-latency 2>/dev/null | tail -1 > $P/bench_line.json
fi
ls $P
