#!/usr/bin/env python3
"""Every scratch (register-spill) access of every kernel of a HIP source, from the compiler's own assembly (hipcc -S): the
instruction, whether its basic block lies inside a loop (the assembly printer's "in Loop: Header=... Depth=..." block
comments) and the kernel's totals.  usage: scratch_sites.py <file.hip> [extra hipcc flags ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flags = sys.argv[2:]
out = tempfile.mktemp(suffix=".s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                       "-I" + os.path.join(ROOT, "go-muse_amd", "csrc"), "-S", "--cuda-device-only", src, "-o", out] + flags)
kernel, depth, sites = None, 0, {}
for ln in open(out):
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        kernel = subprocess.check_output(["c++filt", m.group(1)], text=True).strip()
        kernel = re.sub(r"\(.*$", "", kernel.replace("void ", "").replace("muse::", ""))
        depth = 0
        sites.setdefault(kernel, [])
        continue
    if re.match(r"^\.LBB\d+_\d+:", ln):
        m = re.search(r"Depth=(\d+)", ln)
        depth = int(m.group(1)) if m else 0
        nxt = None
        continue
    m = re.search(r"Depth=(\d+)", ln) if ln.lstrip().startswith(";") and "Loop" in ln else None
    if m:
        depth = max(depth, int(m.group(1)))
    if kernel and "scratch_" in ln and not ln.lstrip().startswith(";"):
        sites[kernel].append((depth, " ".join(ln.split(";")[0].split())))
for k, v in sites.items():
    if not v:
        continue
    inside = sum(1 for d, _ in v if d > 0)
    print("%s: %d scratch instructions, %d inside a loop" % (k, len(v), inside))
    for d, ins in v:
        print("    %-7s %s" % ("loop" if d else "outside", ins))
os.unlink(out)
