"""CPU: `pip install .` (setup.py / pyproject.toml -- the counterpart of the reference's CMakeLists.txt:19-33 + setup.py:31-127)
puts the drop-in module `window_ann`, the engine package with both shared objects and include/wann.h into a target directory,
and the installed copy imports (offline: --no-build-isolation --no-deps)."""
import os
import subprocess
import sys

from util import REPO


def test_pip_install_target_imports(tmp_path):
    target = tmp_path / "site"
    r = subprocess.run([sys.executable, "-m", "pip", "install", REPO, "--no-build-isolation", "--no-deps", "--target", str(target), "-q"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    for rel in ("window_ann/__init__.py", "rangefilteredann_amd/libwann.so", "include/wann.h"):
        assert (target / rel).exists(), rel
    code = ("import window_ann, rangefilteredann_amd as r, os; assert os.path.dirname(r.__file__).startswith(%r); "
            "assert r.abi_version() == 5; assert hasattr(window_ann, 'VamanaRangeFilterTreeIndexFloatEuclidian'); print('ok')" % str(target))
    env = dict(os.environ, PYTHONPATH=str(target), WANN_NO_TORCH="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path), env=env, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
