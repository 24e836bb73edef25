"""GPU: the suite's parity core once more in PRODUCTION mode.  tests/conftest.py sets WANN_TEST_HOOKS=1 for the whole session (tests
flip the engine's development switches between batches on one index, wann_tuning.h), so every other GPU test runs an engine that
re-reads its switches before each call.  The shipped configuration -- no hooks, the switches fixed at index creation, development
switches ignored -- is what a user of the reference runs: a child pytest process with WANN_TEST_HOOKS=0 replays the golden vectors
of the real reference, the index-level oracle comparisons and the C-ABI end-to-end test, and one full-size configuration
(configs[1], SIFT-1M-like, rows against the real reference, mid-fraction legs repeated five times)."""
import os
import subprocess
import sys

import pytest

from util import REPO

pytestmark = pytest.mark.gpu


def _child(args, timeout):
    env = dict(os.environ)
    env["WANN_TEST_HOOKS"] = "0"  # conftest's setdefault keeps it; Tuning::on() reads "0" as off
    for name in list(env):
        if name.startswith("WANN_") and name not in ("WANN_TEST_HOOKS", "WANN_DEVICE", "WANN_BENCH_CACHE"):
            del env[name]
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + args, capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=REPO)
    tail = out.stdout[-2500:] + out.stderr[-1500:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout.splitlines()[-1], tail
    return out.stdout


def test_golden_and_oracle_parity_without_test_hooks(gpu):
    out = _child(["tests/test_gpu_parity.py", "-k",
                  "golden_reference_outputs or golden_ratio_fallback or golden_quirks or index_matches_oracle or edge_cases or tiny_shapes "
                  "or c_abi_end_to_end or fenwick_and_three_split or device_resident_call or asynchronous_calls or raw_beam_search_matches_oracle "
                  "or (tie_heavy_data and default)"], 1500)
    assert "skipped" not in out.splitlines()[-1], out[-400:]


def test_sift_1m_full_size_without_test_hooks(gpu):
    _child(["tests/test_gpu_fullsize.py", "-k", "sift_1m_two_wst"], 1500)
