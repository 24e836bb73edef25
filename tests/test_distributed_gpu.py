"""GPU: the N > 1 product path on the RCCL backend.  The GPU box has one device, so the process group has ONE rank:
`nccl` is initialised, sharded_batch_search drives the HIP kernels and its all-gather runs over RCCL; rows must equal
the unsharded call.  Also: bench.py under the driver's launch line with one rank prints a line with n_gpus = 1."""
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


def _torchrun(script, *args):
    sys.path.insert(0, REPO)
    import bench
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
            "--master-port", str(bench.free_port()), script] + list(args)


def test_sharded_batch_search_on_nccl_world_1(gpu):
    out = subprocess.run(_torchrun(os.path.join(REPO, "tests", "nccl_world1_worker.py")), capture_output=True, text=True,
                         timeout=900, env=_env())
    assert out.returncode == 0 and "NCCL_WORLD1_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_bench_under_the_launcher_with_one_rank(gpu, tmp_path):
    """small shapes; the RCCL all-gather sits inside the timed step"""
    cmd = _torchrun(os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--points", "30000", "--nq", "500",
                    "--fractions", "headline", "--configs", "none", "--no-cpu-baseline", "--cache", str(tmp_path))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=_env())
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "RCCL" in line["config"]["parallelism"]
    assert line["roofline"]["frac"] > 0 and line["scaling"] == "weak"
