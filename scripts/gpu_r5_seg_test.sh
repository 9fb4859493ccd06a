# Round 5: the segments' bit-identity test + the fused / handle tests, then the policy as built on configs 3 and 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py -q -m gpu -s -k "segments or twin" 2>&1 | grep -v "^E  " | tail -25
REPS=12 python scripts/probe_seg_policy.py c3 "ALTRO_HIP_SEGMENTS=0" "X=1" "ALTRO_HIP_SEGMENTS=0" "X=1" 2>&1 | cut -c1-290
REPS=20 python scripts/probe_seg_policy.py c2 "ALTRO_HIP_SEGMENTS=0" "X=1" "ALTRO_HIP_SEGMENTS=0" "X=1" 2>&1 | cut -c1-290
