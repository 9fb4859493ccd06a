# Round 5, segments: grid over the split policy on config 3 (median ms of 9 solves per process)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp REPS=10
args=("ALTRO_HIP_SEGMENTS=0")
for below in 60 70; do for cols in 2560 3072; do for pa in 512 768 1024; do for above in 768 1536; do
  args+=("ALTRO_HIP_SEG_BELOW=$below ALTRO_HIP_SEG_COLUMNS=$cols ALTRO_HIP_SEG_PERSIST_AT=$pa ALTRO_HIP_SEG_ABOVE=$above")
done; done; done; done
python scripts/probe_seg_policy.py c3 "${args[@]}" 2>&1 | sed -e 's/(sweep launches, longest chain in the persistent launch, longest workgroup, hand-overs)//' | cut -c1-200
