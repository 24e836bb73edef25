"""Dev tool: VGPR / SGPR / scratch / LDS of every gfx950 kernel inside libwann.so (reads the code objects' notes)."""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("WANN_LIB", os.path.join(REPO, "rangefilteredann_amd", "libwann.so"))
llvm = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
blob = open(fat, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
for i, s in enumerate(starts):
    part = os.path.join(tmp, f"b{i}.bin"); open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
    co = os.path.join(tmp, f"co{i}.o")
    subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", f"--output={co}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], stderr=subprocess.DEVNULL)
    notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if len(sys.argv) > 1 and sys.argv[1] not in name: continue
        m = re.match(r":?\s*(\d+)", blk)
        agpr = m.group(1) if m else "?"
        print(f"{name[:60]:60s} vgpr+agpr {g('vgpr_count'):>4} (agpr {agpr:>3}) sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} "
              f"spill {g('vgpr_spill_count'):>3} sgpr_spill {g('sgpr_spill_count'):>4}")
