# round 3: LDS counter calibration probe under the PMC, then the profile passes of every bench config
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
P=gpurun_out/lds_probe
rm -rf $P; mkdir -p $P
( cd scripts/probes && ./lds_conflict_probe ) > $P/times.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $P/pmc -o probe -- scripts/probes/lds_conflict_probe > $P/pmc_run.log 2>&1
python - $P <<'PY'
import csv, glob, collections, sys
P = sys.argv[1]
f = glob.glob(f'{P}/pmc/**/*counter_collection.csv', recursive=True)
acc = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    acc.setdefault((r['Dispatch_Id'], r['Kernel_Name'][:60]), {})[r['Counter_Name']] = float(r['Counter_Value'])
with open(f'{P}/summary.txt', 'w') as out:
    for (d, k), c in acc.items():
        line = f"dispatch {d} {k}: " + ", ".join(f"{n} {v:.0f}" for n, v in sorted(c.items())) + (f" | conflict cycles per LDS instruction {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_INSTS_LDS', 1), 1):.3f}")
        print(line); out.write(line + "\n")
PY
cat $P/times.txt
rm -rf $P/pmc
for c in ${@:-2 3 1 4}; do
  bash scripts/gpu_profile.sh $c > gpurun_out/profile_c$c.log 2>&1
  tail -3 gpurun_out/profile_c$c.log
done
