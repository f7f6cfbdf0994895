"""Host->HBM ingestion rate of muse_group_append: one call per Series (Group.Add pattern,
pinned double-buffered staging) versus one slab call. PCIe-inclusive; never bench.py's value."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
muse = importlib.import_module("go-muse_amd")


def main():
    M, N = 65536, 4096
    rows = np.random.default_rng(0).standard_normal((M, N))
    eng = muse.Engine()
    for label, step in (("per-series", 1), ("slab-256", 256), ("slab-all", M)):
        dg = muse.DeviceGroup(eng, N, capacity=M)
        t0 = time.perf_counter()
        if step == 1:
            for i in range(M):
                dg.append(rows[i])
        else:
            for i in range(0, M, step):
                dg.append(rows[i:i + step])
        dg.read(M - 1, 1)                      # flushes staging and waits for the stream
        dt = time.perf_counter() - t0
        print(f"{label:11s} {M} x {N} f64: {dt*1e3:8.1f} ms  {M*N*8/dt/1e9:6.2f} GB/s  {M/dt:9.0f} series/s", flush=True)
        del dg


if __name__ == "__main__":
    main()
