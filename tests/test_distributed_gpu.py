"""GPU: the N > 1 product path on the RCCL backend -- sharded_batch_search / level_dealt_batch_search drive the HIP kernels on
every rank and their all-gathers run over RCCL; rows must equal the single-GPU call.  With ONE device the process group has one
rank (the collective still runs); on a box with more GPUs the same worker runs with 2 .. 8 ranks (it SKIPS LOUDLY on one GPU).
Also: bench.py under the driver's launch line with one rank prints a line with n_gpus = 1."""
import json
import os
import subprocess
import sys

import pytest

from util import REPO

pytestmark = pytest.mark.gpu


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _torchrun(script, *args, ranks=1):
    sys.path.insert(0, REPO)
    import bench
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(bench.free_port()), script] + list(args)


def _run_worker(ranks):
    out = subprocess.run(_torchrun(os.path.join(REPO, "tests", "nccl_worker.py"), str(ranks), ranks=ranks), capture_output=True, text=True,
                         timeout=1500, env=_env())
    assert out.returncode == 0 and f"NCCL_WORKER_OK world={ranks}" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_sharded_and_level_dealt_search_on_nccl_world_1(gpu):
    _run_worker(1)


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_sharded_and_level_dealt_search_on_nccl_multi_rank(gpu, wa, ranks):
    """One process per GPU, `ranks` of them: RCCL sees `ranks` ranks (asserted inside every rank), every rank runs the HIP engine on
    its own replica, rows are compared with the single-GPU call on every rank."""
    have = wa.device_count()
    if have < ranks:
        pytest.skip(f"NOT EXERCISED: the {ranks}-rank RCCL path needs {ranks} GPUs and this box has {have} "
                    "(the multi-rank collectives are covered on gloo by tests/test_distributed_cpu.py only)")
    _run_worker(ranks)


def test_bench_under_the_launcher_with_one_rank(gpu, tmp_path):
    """small shapes; the RCCL all-gather sits inside the timed step"""
    cmd = _torchrun(os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--points", "30000", "--nq", "500",
                    "--fractions", "headline", "--configs", "none", "--no-cpu-baseline", "--cache", str(tmp_path))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "RCCL" in line["config"]["parallelism"]
    assert line["roofline"]["frac"] > 0 and line["scaling"] == "weak"
