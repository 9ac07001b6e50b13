#!/usr/bin/env python3
"""VGPR / spill / scratch / SGPR counts of the gfx950 kernels inside a libksa build (code-object notes).
    python tools/kernel_resources.py [path/to/libksa.so] [name filter ...]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin/"
so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "prgs-sdr-kspecanal_amd", "libksa.so")
filters = [a for a in sys.argv[1:] if not a.endswith(".so")]
tmp = tempfile.mkdtemp()
fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "co")
subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so], check=True)
lst = subprocess.run([LLVM + "clang-offload-bundler", "--list", "--type=o", "--input=" + fat], capture_output=True, text=True).stdout.split()
tgt = [t for t in lst if "gfx950" in t][0]
subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=" + tgt, "--output=" + co], check=True)
notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for k in notes.split(".args:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", k)
    if not name:
        continue
    g = lambda key: (re.search(r"\.%s:\s+(\d+)" % key, k) or [None, "?"])[1]
    d = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    if filters and not any(f in d for f in filters):
        continue
    print("%-72s vgpr %3s spill %2s sgpr %3s scratch %3s lds %s" % (d[:72], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"),
                                                                  g("private_segment_fixed_size"), g("group_segment_fixed_size")))
