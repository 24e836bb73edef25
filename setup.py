"""`pip install .` for the MI355X engine -- the counterpart of the reference's CMakeLists.txt:19-33 / setup.py:31-127, which build
the pybind11 module `window_ann`: here `make -C rangefilteredann_amd/csrc` (hipcc --offload-arch=gfx950 for the kernels, g++ for
the pybind11 shim) produces libwann.so (the C ABI, include/wann.h) and `_window_ann`, and the install carries the packages
`rangefilteredann_amd` (engine + both shared objects) and `window_ann` (the drop-in module name `experiments/wrapper.py:1`
imports).  No network needed: pip install . --no-build-isolation --no-deps."""
import os
import subprocess
import sysconfig

from setuptools import Distribution, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


class BuildNative(build_py):
    def run(self):
        subprocess.check_call(["make", "-C", os.path.join(HERE, "rangefilteredann_amd", "csrc"), "-j" + str(min(8, os.cpu_count() or 1))])
        super().run()


class BinaryDistribution(Distribution):  # (the wheel holds platform binaries)
    def has_ext_modules(self):
        return True


ext = sysconfig.get_config_var("EXT_SUFFIX")
setup(
    name="rangefilteredann-amd",
    version="0.3.0",
    description="MI355X-native window-filtered ANN search behind the RangeFilteredANN `window_ann` Python surface",
    packages=["rangefilteredann_amd", "window_ann"],
    package_data={"rangefilteredann_amd": ["libwann.so", "_window_ann" + ext]},
    data_files=[("include", ["include/wann.h"])],
    cmdclass={"build_py": BuildNative},
    distclass=BinaryDistribution,
    python_requires=">=3.8",
    zip_safe=False,
)
