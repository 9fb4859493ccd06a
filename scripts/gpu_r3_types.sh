#!/bin/bash
# round 3: lists of user cost / constraint classes (tests/test_user_types_gpu.py) + the user-model tests + facade
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_user_types_gpu.py tests/test_user_model_gpu.py tests/test_facade_gpu.py -x -q -m gpu > gpurun_out/types_tests.log 2>&1
echo "exit $?" >> gpurun_out/types_tests.log
tail -15 gpurun_out/types_tests.log
