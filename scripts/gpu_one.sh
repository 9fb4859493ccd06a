cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_gpu.py -q -m gpu -k "config5" 2>&1 | grep -E "^E  |test_parity_gpu.py:[0-9]+|fp32 solved|passed|failed" | cut -c1-300 | head -30
