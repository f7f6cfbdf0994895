"""Builds go-muse_amd/lib/libmuse_hip.so (gfx950 only) with hipcc.

The library is built IN-TREE so it travels with the repo snapshot to the GPU
box; it is git-ignored.  hipcc cross-compiles gfx950 code objects without a
GPU present.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmuse_hip.so")
SOURCES = ["xcorr_kernels.hip", "xcorr_r16_fold.hip", "xcorr_r16_occ4.hip", "xcorr_stockham.hip", "xcorr_two_sided.hip", "xcorr_small.hip", "xcorr_long.hip", "xcorr_real.hip", "xcorr_huge.hip", "xcorr_r16_screen.hip", "xcorr_screen_stk.hip", "reduce_kernels.hip", "diag_kernels.hip", "capi_context.hip", "capi_group.hip", "capi_batch.hip", "capi_run.hip", "capi_rows.hip", "capi_screen.hip", "capi_many.hip", "capi_huge.hip", "capi_xcorr.hip"]
HEADERS = [os.path.join(CSRC, "xcorr_kernels.h"), os.path.join(CSRC, "xcorr_huge.h"), os.path.join(CSRC, "fft_device.h"), os.path.join(CSRC, "r16_device.h"), os.path.join(CSRC, "fold_device.h"), os.path.join(CSRC, "foldk_device.h"), os.path.join(CSRC, "long_device.h"), os.path.join(CSRC, "stk_device.h"), os.path.join(CSRC, "small_device.h"), os.path.join(CSRC, "two_device.h"), os.path.join(CSRC, "capi_internal.h"), os.path.join(ROOT, "include", "muse_hip.h"), os.path.join(ROOT, "include", "muse_hip_test.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


# per-source extra flags: the fp32 screening kernel must keep scalar fp32 ops (2-cycle
# issue); SLP packing into v_pk_*_f32 costs v_mov shuffles and issues no faster
PER_FILE_FLAGS = {"xcorr_r16_screen.hip": ["-fno-slp-vectorize"], "xcorr_screen_stk.hip": ["-fno-slp-vectorize"]}


def build(force=False, verbose=False, extra_flags=()):
    if not force and not stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + list(extra_flags)
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        cmd = base + PER_FILE_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("-") and a != "--force"]
    print(build(force=True, verbose=True, extra_flags=flags))


HOST_TEST = os.path.join(LIBDIR, "muse_host_test")


def build_host_test(force=False):
    """g++ build of the C++ host mirror's test program (links libmuse_hip.so)."""
    src = os.path.join(PKG, "host", "muse_host_test.cpp")
    hdr = os.path.join(PKG, "host", "muse.hpp")
    build()
    if (not force and os.path.exists(HOST_TEST)
            and os.path.getmtime(HOST_TEST) >= max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(LIB))):
        return HOST_TEST
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "host"), src, "-o", HOST_TEST,
                           "-L" + LIBDIR, "-lmuse_hip", "-pthread", "-Wl,-rpath," + LIBDIR])
    return HOST_TEST


REF_BENCH = os.path.join(LIBDIR, "muse_ref_bench")


def build_ref_bench(force=False):
    """g++ build of the reference's own benchmark shapes over the C++ host mirror (host/muse_ref_bench.cpp; bench.py runs it)."""
    src = os.path.join(PKG, "host", "muse_ref_bench.cpp")
    hdr = os.path.join(PKG, "host", "muse.hpp")
    build()
    if (not force and os.path.exists(REF_BENCH)
            and os.path.getmtime(REF_BENCH) >= max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(LIB))):
        return REF_BENCH
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "host"), src, "-o", REF_BENCH,
                           "-L" + LIBDIR, "-lmuse_hip", "-pthread", "-Wl,-rpath," + LIBDIR])
    return REF_BENCH


COLD_PHASES = os.path.join(LIBDIR, "muse_cold_phases")


def build_cold_phases(force=False):
    """g++ build of the cold-path phase breakdown (host/muse_cold_phases.cpp; profiles/r06_cold_path.txt)."""
    src = os.path.join(PKG, "host", "muse_cold_phases.cpp")
    hdr = os.path.join(PKG, "host", "muse.hpp")
    build()
    if (not force and os.path.exists(COLD_PHASES)
            and os.path.getmtime(COLD_PHASES) >= max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(LIB))):
        return COLD_PHASES
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "host"), src, "-o", COLD_PHASES,
                           "-L" + LIBDIR, "-lmuse_hip", "-pthread", "-Wl,-rpath," + LIBDIR])
    return COLD_PHASES
