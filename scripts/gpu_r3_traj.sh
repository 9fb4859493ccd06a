#!/bin/bash
# round 3: the optional second collective (trajectory all-gather): group library, facade driver, RCCL with one rank
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_group.py tests/test_rccl_world1_gpu.py tests/test_abi.py -x -q -m gpu > gpurun_out/traj_tests.log 2>&1
echo "exit $?" >> gpurun_out/traj_tests.log
tail -15 gpurun_out/traj_tests.log
./perf/benchmark_unicycle 2 4096 --gpus 1 2>&1 | tail -4 | tee gpurun_out/traj_driver.log
