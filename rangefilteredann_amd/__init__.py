"""rangefilteredann_amd -- MI355X-native (gfx950) window-filtered ANN search engine.

The package holds only what the one accelerated path needs:

  csrc/            HIP kernels (wann_kernels.hip), host driver (wann_host.cpp) + C ABI (wann_abi.cpp, wann_raw.cpp, include/wann.h),
                   host index builder (wann_build.cpp) and the pybind11 shim (window_ann_pybind.cpp)
  libwann.so       built C-ABI library            (make -C rangefilteredann_amd/csrc)
  _window_ann*.so  built pybind11 module that mirrors the reference's `window_ann` surface
  distributed.py   one-process-per-GPU query sharding + RCCL all-gather of the per-shard top-k

There is no CPU fallback: importing works everywhere, constructing an index needs a gfx950 GPU.
"""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")


def build(verbose: bool = False) -> None:
    """Compile libwann.so and the pybind11 module in-tree (hipcc --offload-arch=gfx950)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", _CSRC, "-j4"], stdout=out)


def lib_path() -> str:
    return os.path.join(_HERE, "libwann.so")


# PyTorch-ROCm bundles its own HIP runtime (libamdhip64.so.7).  Two HIP runtimes in one process
# cannot both own the GPU, so when torch is installed it is imported FIRST: libwann.so then binds
# to the runtime torch already loaded (same SONAME) and device pointers / streams are shared.
if not os.environ.get("WANN_NO_TORCH"):
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass

try:
    from . import _window_ann  # noqa: F401
except ImportError as e:  # pragma: no cover - build missing
    raise ImportError(
        "rangefilteredann_amd: the native extension is not built (%s). Run "
        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C rangefilteredann_amd/csrc`." % e
    ) from e

from ._window_ann import *  # noqa: F401,F403,E402
from ._window_ann import QueryParams, BuildParams, device_count, abi_version  # noqa: F401,E402
