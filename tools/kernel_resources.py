#!/usr/bin/env python3
"""tools/kernel_resources.py <file.hip> [more.hip ...] [-- extra hipcc flags]

Compiles each source for gfx950 with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
VGPRs, AGPRs, scratch bytes per lane (spills), SGPRs, LDS bytes, occupancy (waves per SIMD).  Used to check that a
kernel change did not start spilling (docs/HISTORY.md 4.1: a scratch reload queues on the same in-order vmcnt as the HBM stream).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "go-muse_amd", "csrc")


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
            if len(out) == len(names):
                return out
        except Exception:
            pass
    return names


def resources(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + list(extra)
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in err.splitlines():
        m = re.search(r"remark: +Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["demangled"] = re.sub(r"\(muse::FusedParams.*", "", n).replace("void ", "").replace("muse::", "")
    return rows


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        k = args.index("--")
        args, extra = args[:k], args[k + 1:]
    for src in args:
        if not os.path.exists(src):
            src = os.path.join(CSRC, src)
        print("# %s" % os.path.relpath(src, ROOT))
        print("%-64s %5s %5s %8s %6s %6s %5s %7s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "vspill", "sspill", "SGPR", "LDS", "occ"))
        for r in resources(src, extra):
            print("%-64s %5s %5s %8s %6s %6s %5s %7s %4s" % (r["demangled"][:64], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"),
                                                            r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("TotalSGPRs"), r.get("LDS Size"), r.get("Occupancy")))


if __name__ == "__main__":
    main()
