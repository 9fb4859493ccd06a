# Diagnostic PMC passes (round 4): where do the waves of the sweep kernels spend their cycles -- issue classes, LDS pipe,
# instruction cache.   scripts/gpu_pmc_diag.sh [config]   -> gpurun_out/pmc_diag_c<config>.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
C=${1:-2}
P=gpurun_out/pmc_diag_c$C
rm -rf $P; mkdir -p $P
case $C in 2|3) export ALTRO_HIP_CHAINS=4;; *) export ALTRO_HIP_CHAINS=1;; esac
CMD="python bench.py --config $C --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $P/a -o bench -- $CMD > $P/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_IFETCH SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $P/b -o bench -- $CMD > $P/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_MISSES --output-format csv -d $P/c -o bench -- $CMD > $P/c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LEVEL_WAVES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $P/d -o bench -- $CMD > $P/d.log 2>&1
python - $P > gpurun_out/pmc_diag_c$C.txt <<'PY'
import csv, glob, collections, sys
P = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for tag in "abcd":
    for f in glob.glob(f"{P}/{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((s for s in ("k_sweep_fused", "k_forward2", "k_backward_mfma16", "k_backward_mfma", "k_expansions") if s in r["Kernel_Name"]), None)
            if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{P}/{tag}/**/*kernel_trace.csv", recursive=True):
        if tag != "a": continue
        for r in csv.DictReader(open(f)):
            k = next((s for s in ("k_sweep_fused", "k_forward2", "k_backward_mfma16", "k_backward_mfma", "k_expansions") if s in r["Kernel_Name"]), None)
            if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, d in acc.items():
    print(k, "launches", len(dur[k]), "avg us (serialised under pmc)", round(sum(dur[k]) / max(1, len(dur[k])), 1))
    for c, v in sorted(d.items()):
        print(f"   {c:32s} avg {sum(v)/len(v):16.1f}  n {len(v)}")
PY
for d in a b c d; do rm -rf $P/$d; done
cat gpurun_out/pmc_diag_c$C.txt | head -80
