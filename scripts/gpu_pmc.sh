cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -q -m gpu 2>&1 | grep -v "^E  " | tail -6
python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['sweeps'])"
rm -rf gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc -o bench -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_run.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc/**/*counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r['Kernel_Name']
    short = 'fwd' if 'k_forward' in name else 'bwd' if 'k_backward' in name else 'exp' if 'k_expansions' in name else None
    if not short: continue
    acc[short][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value']), (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
for k, d in acc.items():
    print('==', k)
    for c, v in d.items():
        v.sort()
        print('  %-20s' % c, 'sweep0 %.0f (%.0f us)' % (v[0][1], v[0][2]), 'sweep3 %.0f' % v[3][1], 'sweep60 %.0f (%.0f us)' % (v[60][1], v[60][2]), 'sweep100 %.0f' % v[100][1])
PY
