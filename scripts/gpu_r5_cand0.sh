# Round 5: winner replay for EVERY accepted trial (ALTRO_HIP_CAND_FRONT=0: only the last live trial owns a
# candidate slot) against the default 6 + 1 slots: ms per solve of configs 2 / 3 / 4, and the HBM traffic of k_forward2 per launch
# (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes, chain count forced as in gpu_profile.sh)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="--steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward --no-pipeline2"
for c in 2 3 4; do for f in 6 2 0; do
  ALTRO_HIP_CAND_FRONT=$f python bench.py --config $c $B 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config $c front $f ms', d['ms_per_step'], 'forward kernels summed ms', d['roofline']['kernel_ms']['forward_pass'])"
done; done
for c in 3 4; do
  case $c in 3) export ALTRO_HIP_CHAINS=4;; *) export ALTRO_HIP_CHAINS=1;; esac
  for f in 6 0; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pm; ALTRO_HIP_CAND_FRONT=$f rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pm -o b -- python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-latency --no-fast-forward --no-pipeline2 > /dev/null 2>&1
      python - $c $f $ctr <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True)[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'k_forward2' in r['Kernel_Name']]
print('config', sys.argv[1], 'front', sys.argv[2], sys.argv[3], 'KiB per launch of k_forward2: %.0f over %d launches' % (sum(v) / len(v), len(v)))
PY
    done
  done
done
