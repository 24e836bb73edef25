"""CPU: the development switches are closed without WANN_TEST_HOOKS=1 (csrc/wann_tuning.h): the record keeps its defaults and the
switches found in the environment are named once on stderr.  Checked through a tiny C++ program built against the header."""
import os
import subprocess
import sys

from util import REPO

SRC = r'''
#include "wann_tuning.h"
int main() {
  wann::Tuning t = wann::Tuning::from_env();
  wann::Tuning u = wann::Tuning::from_env();
  printf("%d %d %d %d %d %d %.1f\n", (int)t.hooks_live, (int)t.spec, (int)t.gemm, (int)t.deep_min_tasks, (int)t.force_general, (int)t.verbose, t.proof_factor);
  return (t.spec == u.spec) ? 0 : 1;
}
'''


def _run(tmp_path, env_extra):
    exe = tmp_path / "tg"
    if not exe.exists():
        (tmp_path / "tg.cpp").write_text(SRC)
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(REPO, "rangefilteredann_amd", "csrc"), str(tmp_path / "tg.cpp"), "-o", str(exe)], check=True)
    env = {k: v for k, v in os.environ.items() if not k.startswith("WANN_")}
    env.update(env_extra)
    p = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert p.returncode == 0
    return p.stdout.split(), p.stderr


def test_lab_switches_need_the_hooks(tmp_path):
    lab = {"WANN_NO_SPEC": "1", "WANN_NO_GEMM": "1", "WANN_DEEP_MIN_TASKS": "12", "WANN_FORCE_GENERAL": "1", "WANN_VERBOSE": "1", "WANN_PROOF_FACTOR": "5"}
    out, err = _run(tmp_path, lab)
    assert out == ["0", "1", "1", "4096", "0", "1", "5.0"]  # defaults, except the two production names
    assert err.count("IGNORED without WANN_TEST_HOOKS=1") == 1  # (two records were made: named once)
    for name in ("WANN_NO_SPEC", "WANN_NO_GEMM", "WANN_DEEP_MIN_TASKS", "WANN_FORCE_GENERAL"):
        assert name in err
    assert "WANN_VERBOSE" not in err and "WANN_PROOF_FACTOR" not in err
    out, err = _run(tmp_path, dict(lab, WANN_TEST_HOOKS="1"))
    assert out == ["1", "0", "0", "12", "1", "1", "5.0"] and err == ""
    out, err = _run(tmp_path, {})
    assert out == ["0", "1", "1", "4096", "0", "0", "3.0"] and err == ""


def test_unknown_names_are_named_once(tmp_path):
    """A WANN_* variable nothing reads (a typo, a switch of another version) is named on stderr -- once per process."""
    out, err = _run(tmp_path, {"WANN_HANDOFF_COMPANION": "1", "WANN_BENCH_CACHE": "/tmp/x", "WANN_VERBOSE": "1"})
    assert err.count("unknown WANN_* variable") == 1 and "WANN_HANDOFF_COMPANION" in err
    assert "WANN_BENCH_CACHE" not in err and "WANN_VERBOSE" not in err
