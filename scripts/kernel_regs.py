"""VGPR / spill / scratch figures of the persistent kernels, from the code-object metadata of the built object:
    python scripts/kernel_regs.py [pattern ...]     (default: sample_ stream_step)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
pats = sys.argv[1:] or ["sample_", "stream_step"]
obj = os.path.join(ROOT, "after_amd", "lib", "denoiser.o")
with tempfile.TemporaryDirectory() as d:
    import glob
    import shutil
    shutil.copy(obj, os.path.join(d, "den.o"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", os.path.join(d, "den.o")], check=True, capture_output=True, cwd=d)
    co = [f for f in glob.glob(os.path.join(d, "*gfx950*"))][0]  # (the extracted code object lands next to the input)
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for b in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", b) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = name.replace("after::(anonymous namespace)::", "").replace("void ", "")
    if any(p in name for p in pats):
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s}  spilled vgprs {g('vgpr_spill_count'):>4s}  scratch {g('private_segment_fixed_size'):>5s} B  "
              f"sgpr {g('sgpr_count'):>4s}  lds {g('group_segment_fixed_size'):>6s}")
