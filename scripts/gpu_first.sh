cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/probe_fp32.py 2>&1 | tee gpurun_out/fp32.log | tail -40
