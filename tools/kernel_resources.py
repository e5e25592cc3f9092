#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch use of a built libgswt_hip*.so, from the code object's notes.
usage: tools/kernel_resources.py [lib] [regex] [target, default gfx950; e.g. gfx950:xnack-]"""
import re
import subprocess
import sys
import tempfile

lib = sys.argv[1] if len(sys.argv) > 1 else "gswt_renderer_amd/lib/libgswt_hip.so"
filt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
ARCH = sys.argv[3] if len(sys.argv) > 3 else "gfx950"
LLVM = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as t:
    co, fb = t + "/co", t + "/fatbin"
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, lib, t + "/lib_copy"], check=True, capture_output=True)   # an output file: without one objcopy rewrites its input in place
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--" + ARCH,
                    "--input=" + fb, "--output=" + co], check=True, capture_output=True)
    txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
rows = []
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void gswt::", "")
    if filt.search(name):
        rows.append((name, g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
for r in sorted(rows):
    print(f"{r[1]:>4} vgpr {r[2]:>4} sgpr {r[3]:>6} lds {r[4]:>5} scratch  {r[0]}")
