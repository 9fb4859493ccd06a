cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ldsprobe
cd scripts/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o lds_conflict_probe lds_conflict_probe.hip && cd ../..
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d gpurun_out/ldsprobe -o probe --output-format csv -- scripts/probes/lds_conflict_probe 2>&1 | grep "us per launch" | tee gpurun_out/lds_probe.txt
python - <<'PY' | tee -a gpurun_out/lds_probe.txt
import csv, glob, collections
f = glob.glob("gpurun_out/ldsprobe/**/*counter_collection.csv", recursive=True)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
    rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, name), c in rows.items():
    if d % 2 == 0:
        print("dispatch", d, name, "conflict cycles per LDS instruction %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_INSTS_LDS", 1))))
PY
