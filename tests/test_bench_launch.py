"""`bench.py --gpus N` launches its own ranks.

CPU side (no GPU): `--launch-check` runs the launcher and the rank / shard plumbing over gloo -- started plainly
(bench.py re-executes itself under torch.distributed.run on a free port) and the way the driver starts it (an explicit
torch.distributed.run line with WORLD_SIZE in the environment).  GPU side: two ranks SHARE the one GPU of the box
(`--dist-backend gloo --share-devices`: RCCL refuses two ranks on one device, so the records travel through host
tensors) -- the world-size-2 path of the real bench, every rank solving its block of the global seeded batch, with the
`dist_check` of the line: gathered records == every rank's own altro_get_stats."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]  # ONE line, from rank 0
    return json.loads(lines[0])


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ALTRO_BENCH_SELF_LAUNCHED")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_plain_start_launches_its_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--config", "3", "--steps", "7", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _last_json(r.stdout)
    lc = line["launch_check"]
    assert line["n_gpus"] == 2 and lc["world_size"] == 2 and lc["env_world_size"] == 2
    assert lc["self_launched"] and lc["distinct_processes"] and lc["every_rank_saw_the_same_world"]
    assert lc["ranks"] == [0, 1] and lc["local_ranks"] == [0, 1]
    # rank r owns block r of the 2 x 4096 global batch of configs[3]
    assert lc["blocks"] == [[0, 4096], [4096, 8192]] and lc["blocks_tile_the_global_batch"]
    # the arguments reached the ranks
    assert lc["args"] == {"gpus": 2, "steps": 7, "warmup": 2, "config": 3, "batch": 4096}


def test_driver_style_launch_is_not_relaunched():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29641", BENCH, "--gpus", "2", "--launch-check", "--batch", "100", "--steps", "1", "--warmup", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lc = _last_json(r.stdout)["launch_check"]
    assert not lc["self_launched"] and lc["world_size"] == 2
    assert lc["blocks"] == [[0, 100], [100, 200]]


def test_world_size_mismatch_is_an_error():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], capture_output=True, text=True, timeout=300,
                       env=env, cwd=ROOT)
    assert r.returncode != 0 and "started 1 ranks" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_two_ranks_share_the_gpu():
    """World size 2 on the one GPU of the box: 2 x 512 instances of configs[2] = the first 1024 instances of the seeded
    global batch, which ONE rank with --batch 1024 solves as well: same number of solved instances, same iterations."""
    common = ["--config", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--no-latency",
              "--no-fast-forward"]
    r2 = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--batch", "512", "--dist-backend", "gloo", "--share-devices"] + common,
                        capture_output=True, text=True, timeout=900, env=_clean_env(), cwd=ROOT)
    print(r2.stdout[-2000:], r2.stderr[-3000:])
    assert r2.returncode == 0
    two = _last_json(r2.stdout)
    dc = two["dist_check"]
    assert two["n_gpus"] == 2 and dc["world_size"] == 2 and dc["backend"] == "gloo"
    assert dc["records_match_get_stats"] and dc["gather_is_separate_buffer"] and len(dc["rank_ms_per_step"]) == 2
    assert two["config"]["global_batch"] == 1024 and two["config"]["batch_per_gpu"] == 512
    r1 = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--batch", "1024"] + common, capture_output=True, text=True,
                        timeout=900, env=_clean_env(), cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = _last_json(r1.stdout)
    for key in ("solved_fraction", "mean_iterations", "max_iterations"):
        assert one["config"][key] == two["config"][key], key
