"""CPU: `python bench.py --gpus N` with no launcher in the environment starts its N ranks itself (a
torch.distributed.run child, started before the parent touches torch or HIP) and relays rank 0's JSON line."""
import json
import os
import subprocess
import sys

from util import REPO


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WANN_NO_TORCH"] = "1"
    return env


def test_launcher_command_is_the_drivers_launch_line():
    sys.path.insert(0, REPO)
    import bench
    cmd = bench.launcher_cmd(4, ["--gpus", "4", "--steps", "2"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")


def test_gpus_2_without_launcher_starts_two_ranks():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--launch-check"],
                         capture_output=True, text=True, timeout=300, env=_clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["launch_check"] == 2 and rec["ranks"] == [0, 1] and rec["rows_ok_on_every_rank"] is True and sum(rec["shards"]) == rec["batch"]


def test_the_drivers_8_rank_launch_line_on_the_deep_workload():
    """What the driver's first real SCALE run does -- `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8` -- as a
    dry run on gloo: eight ranks, the deep workload's flags, strong scaling with the cost-balanced cut, one step's sharding and
    all-gather with a stand-in search function; every rank checks every row."""
    sys.path.insert(0, REPO)
    import bench
    argv = ["--gpus", "8", "--workload", "deep", "--scaling", "strong", "--balance", "cost", "--steps", "2", "--warmup", "1", "--launch-check"]
    env = _clean_env()
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run(bench.launcher_cmd(8, argv, bench.free_port()), capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["launch_check"] == 8 and rec["ranks"] == list(range(8)) and rec["workload"] == "deep-10M-like"
    assert rec["rows_ok_on_every_rank"] is True and len(rec["shards"]) == 8 and sum(rec["shards"]) == rec["batch"] and len(set(rec["shards"])) > 1


def test_levels_dealt_step_on_four_ranks():
    """--balance levels: the step deals single doubling levels to the ranks (level_dealt_batch_search); dry run with a stand-in engine
    that has the doubling loop's semantics"""
    sys.path.insert(0, REPO)
    import bench
    argv = ["--gpus", "4", "--scaling", "strong", "--balance", "levels", "--nq", "41", "--launch-check"]
    env = _clean_env()
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run(bench.launcher_cmd(4, argv, bench.free_port()), capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["launch_check"] == 4 and rec["shard_cut"] == "levels" and rec["rows_ok_on_every_rank"] is True


def test_gpus_must_match_world_size():
    env = _clean_env()
    env.update(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--launch-check"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr
