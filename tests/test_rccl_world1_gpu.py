"""The RCCL branch of bench.py on ONE GPU: `torch.distributed.run --nproc-per-node 1` launches the
exact multi-GPU code path -- init_process_group("nccl") AFTER the solver created its streams, all_gather_into_tensor on the
packed device tensor -- so RCCL really loads beside the solver's chains of sweeps, and the records it delivers are
compared with altro_get_stats."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("config", [2, 3])
def test_bench_nccl_path_with_one_rank(config):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(29610 + config), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--config", str(config), "--force-dist", "--no-cpu-baseline", "--no-other-configs", "--no-latency"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    dc = line["dist_check"]
    assert dc["backend"] == "nccl" and dc["world_size"] == 1
    assert dc["records_match_get_stats"] and dc["gather_is_separate_buffer"]
    # the optional second collective: all_gather_into_tensor of the whole trajectories, packed on the device
    assert dc["trajectories_match_get_trajectory"] and dc["trajectory_bytes"] == 4096 * (101 * 3 + 100 * 2) * 8
    assert line["value"] > 0 and line["n_gpus"] == 1
    # the chains of sweeps still run side by side with RCCL's streams in the process (created after the solver's)
    assert line["roofline"]["concurrent_chains"] == 4
