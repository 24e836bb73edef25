"""CPU: the C-ABI library loads and exports every symbol include/wann.h declares; without a GPU
every compute entry point fails loudly (there is no CPU fallback in the product)."""
import ctypes
import os
import re

import numpy as np
import pytest

from util import REPO


def declared_functions():
    src = open(os.path.join(REPO, "include", "wann.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wann_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(wa):
    import rangefilteredann_amd
    lib = ctypes.CDLL(rangefilteredann_amd.lib_path())
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"libwann.so does not export {n}"
    assert lib.wann_abi_version() == 5


def test_python_surface_matches_reference_names(wa):
    # python_bindings/python_bindings.cpp:111-157,204-213 + variant names :67-86
    for agn in ("FloatEuclidian", "FloatMips", "UInt8Euclidian", "UInt8Mips", "Int8Euclidian", "Int8Mips"):
        for cls in ("PrefilterIndex", "RangeFilterTreeIndex", "PostfilterVamanaIndex", "VamanaRangeFilterTreeIndex",
                    "SuperOptimizedPostfilterTreeIndex"):
            assert hasattr(wa, cls + agn)
    qp = wa.QueryParams(k=10, beam_width=40, cut=1.35, limit=10_000_000, degree_limit=10_000, final_beam_multiply=1,
                        postfiltering_max_beam=10000, min_query_to_bucket_ratio=None, verbose=False)
    bp = wa.BuildParams(max_degree=64, limit=500, alpha=1.0, cache_path="index_cache/x/")
    assert qp is not None and bp is not None
    assert wa.defaults.GRAPH_DEGREE == 64


def test_harness_resolves_every_class_name(wa):
    """experiments/wrapper.py:233-268 builds class names from (prefix, metric, dtype); its 'Uint8' spelling must resolve too"""
    from rangefilteredann_amd import harness
    for ctor in (harness.prefilter_index_constructor, harness.postfilter_vamana_constructor,
                 harness.vamana_range_filter_tree_constructor, harness.super_optimized_postfilter_tree_constructor):
        for metric in ("Euclidian", "mips"):
            for dtype in ("float", "uint8", "int8"):
                assert ctor(metric, dtype) is not None
    assert wa.PrefilterIndexUint8Euclidian is wa.PrefilterIndexUInt8Euclidian


def test_no_gpu_means_loud_failure(wa):
    if wa.device_count() > 0:
        pytest.skip("a GPU is present")
    X = np.zeros((16, 8), dtype=np.float32)
    lab = np.arange(16, dtype=np.float32)
    for cls in ("VamanaRangeFilterTreeIndexFloatEuclidian", "PrefilterIndexFloatMips", "SuperOptimizedPostfilterTreeIndexFloatMips"):
        with pytest.raises(RuntimeError, match="no usable gfx950 device"):
            getattr(wa, cls)(X, lab)
    with pytest.raises(RuntimeError, match="no usable gfx950 device"):
        wa.raw_beam_search(0, X, np.zeros((16, 5), dtype=np.int32), 0, X[:2], np.arange(2), 4)


def test_byte_variants_accept_any_dimension(wa):
    """uint8 / int8 classes (python_bindings.cpp:234-237) keep their points as bytes and accumulate in int32 like the
    reference (euclidian_point.h:44-60, mips_point.h:44-58): no bound on the dimension; without a device only the
    missing device stops the construction."""
    lab = np.arange(16, dtype=np.float32)
    if wa.device_count() == 0:
        for cls, arr in (("VamanaRangeFilterTreeIndexUInt8Euclidian", np.zeros((16, 300), dtype=np.uint8)),
                         ("PrefilterIndexInt8Euclidian", np.zeros((16, 2048), dtype=np.int8)),
                         ("PrefilterIndexInt8Mips", np.zeros((16, 1025), dtype=np.int8)),
                         ("SuperOptimizedPostfilterTreeIndexUInt8Mips", np.zeros((16, 8), dtype=np.uint8))):
            with pytest.raises(RuntimeError, match="no usable gfx950 device"):
                getattr(wa, cls)(arr, lab)


def test_engine_kernels_use_no_scratch(wa, tmp_path):
    """None of the engine's own gfx950 kernels may need a private (scratch) segment: a queue whose scratch
    has to grow between launches was the trigger of GPU memory faults next to other HIP users (torch)
    in the same process, and scratch traffic is slow anyway.  Reads the code objects inside libwann.so."""
    import shutil
    import subprocess
    import rangefilteredann_amd
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(f"{llvm}/clang-offload-bundler") and shutil.which("objcopy")):
        pytest.skip("ROCm llvm tools not present")
    fat = tmp_path / "fat.bin"
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", rangefilteredann_amd.lib_path(), str(fat)])
    blob = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert len(starts) >= 3  # search, build and gemm kernel files
    seen = {}
    for i, s in enumerate(starts):
        part = tmp_path / f"bundle{i}.bin"
        part.write_bytes(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = tmp_path / f"co{i}.elf"
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        notes = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", str(co)], text=True)
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name and name.startswith("_ZN4wann"):
                seen[name] = int(m.group(1))
    assert len(seen) >= 15, seen
    bad = {k: v for k, v in seen.items() if v != 0}
    assert not bad, f"kernels with a scratch segment: {bad}"


def test_four_wave_search_kernels_fit_two_waves_per_simd(wa):
    """k_search<., 0> runs four waves per workgroup, two workgroups per CU, and a workgroup of k_search<., 1> (a search wave
    + three helper waves) shares its CU with one of those: a wave of either may use at most 256 registers (VGPRs + AGPRs)
    -- for every element type."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "kernel_resources.py"),
                          "k_search"], capture_output=True, text=True, timeout=300)
    if out.returncode != 0 or not out.stdout.strip():
        pytest.skip("ROCm llvm tools not present")
    lean = [l for l in out.stdout.splitlines() if "ELi0EEEv" in l or "ELi1EEEv" in l]
    assert len(lean) == 12  # (2 kernels x 2 metrics x 3 element types)
    for l in lean:
        assert int(l.split("vgpr+agpr")[1].split()[0]) <= 256, l


def test_gather_layout_is_the_contiguous_balanced_cut():
    """wann_gather_layout (no GPU needed): the shard arithmetic of the multi-device calls equals distributed.shard_bounds, the
    capacity is the longest shard, and bad arguments are refused"""
    import ctypes as C
    import rangefilteredann_amd
    from rangefilteredann_amd.distributed import shard_bounds, shard_capacity
    lib = C.CDLL(rangefilteredann_amd.lib_path())
    lib.wann_gather_layout.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    for nq in (0, 1, 7, 8, 333, 10000, 10001):
        for world in (1, 2, 3, 8):
            for s_ in range(world):
                lo, cnt, cap = C.c_int64(-1), C.c_int64(-1), C.c_int64(-1)
                assert lib.wann_gather_layout(nq, world, s_, C.byref(lo), C.byref(cnt), C.byref(cap)) == 0
                a, b = shard_bounds(nq, world, s_)
                assert (lo.value, lo.value + cnt.value) == (a, b) and cap.value == shard_capacity(nq, world)
    assert lib.wann_gather_layout(10, 0, 0, None, None, None) != 0 and lib.wann_gather_layout(10, 2, 2, None, None, None) != 0
