"""CPU: the engine's host-side C++ (index layout + host Vamana builder in wann_build.cpp, the C ABI's host side in
wann_host.cpp / wann_abi.cpp / wann_raw.cpp) built with g++ -fsanitize=address,undefined and run (`make -C rangefilteredann_amd/csrc sanitize`)."""
import os
import subprocess

from util import REPO


def test_host_code_is_clean_under_asan_and_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(REPO, "rangefilteredann_amd", "csrc"), "sanitize"], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "HOST_SANITIZE_OK" in out.stdout and "ABI_SANITIZE_OK" in out.stdout
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
