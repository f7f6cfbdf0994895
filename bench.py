#!/usr/bin/env python3
"""bench.py -- go-muse XCorr/Batch.Run hot path on MI355X.

One "step" = one Batch.Run over the resident Group with EVERY series scored by the
float64 kernel (the reference's arithmetic, xcorr.go:160-197): fused z-norm + FFT
xcorr + argmax over every series, group max, filter, top-N, (N > 1: RCCL gather of
per-shard top-N records + merge).  The library's opt-in filter-and-refine Run (fp32
screening pass + fp64 re-evaluation) is timed next to it and reported as the extra
object `filter_and_refine_run`, never as `value`.  Workload (BASELINE.json configs[2]):
1 reference x 1 000 000 series per GPU, N = 4096 float64, synthetic rect+noise
generated on the device and resident in HBM before the timed region.

Prints ONE JSON line (rank 0).  `roofline.achieved` is algorithmic bytes
(8*N + 16 per series) / the fused kernel's average launch duration,
measured with HIP events on the stream the kernel runs on.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBPS = 6290.0       # MI355X_MICROARCH.md: best measured copy


def csrc_sha():
    """fingerprint of the kernel sources: counters collected from other sources are not attached"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "go-muse_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and not f.startswith("capi_"):   # (the host side of the library launches kernels, it holds none)
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


class Counters:
    """profiles/r*_counters.json (tools/profile.sh + tools/profile_summary.py): per kernel instantiation, the rocprofv3 PMC
    means of one launch on the default bench workload.  Looked up by kernel name AND rows AND length; a miss is reported in
    the object that asked (never silently a number from another workload), a file collected from other kernel sources too."""

    def __init__(self):
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))
        self.path, self.data, self.note = (files[-1] if files else None), {}, None
        if not self.path:
            self.note = "no profiles/r*_counters.json"
            return
        try:
            j = json.load(open(self.path))
        except Exception as e:
            self.note = "unreadable %s: %s" % (os.path.basename(self.path), e)
            return
        self.meta = {k: j.get(k) for k in ("collected_at_commit", "csrc_sha", "method")}
        if j.get("csrc_sha") != csrc_sha():
            self.note = "%s was collected from other kernel sources (csrc_sha %s, now %s): regenerate with tools/profile.sh" % (
                os.path.basename(self.path), j.get("csrc_sha"), csrc_sha())
            return
        self.data = {k["kernel"]: k for k in j.get("kernels", [])}

    def lookup(self, kernel, rows, length):
        """-> (record or None, source dict)"""
        src = {"file": os.path.relpath(self.path, ROOT) if self.path else None}
        if self.note:
            src["note"] = self.note
            return None, src
        src.update(self.meta)
        k = self.data.get(kernel)
        if k is None:
            src["note"] = "no counters for kernel %s" % kernel
            return None, src
        if k.get("rows") != rows or k.get("length") != length:
            src["note"] = "counters for %s are for %s x %s, this run is %d x %d" % (kernel, k.get("rows"), k.get("length"), rows, length)
            return None, src
        return k, src

    def attach(self, obj, kernel, rows, length, algorithmic_bytes):
        """adds `traffic` (HBM bytes per launch from FETCH_SIZE / WRITE_SIZE), its ratio to the algorithmic bytes and the source"""
        k, src = self.lookup(kernel, rows, length)
        obj["traffic"] = k.get("hbm_bytes_per_launch") if k else None
        if k and algorithmic_bytes:
            obj["traffic_over_algorithmic"] = k["hbm_bytes_per_launch"] / algorithmic_bytes
        obj["traffic_source"] = src
        return k


def clock_stats(mhz):
    """the probe runs from just before the first timed launch to just behind the last one: every window is a sample of the
    clock the chip held over the timed region (tools/clock_trace.py: 2.4 GHz idle, down to ~1.3 GHz within 5 ms of the first
    launch of the fp64 kernel, ~1.8 GHz from 30 ms on, short boosts in the gaps between launches)"""
    import numpy as np
    mhz = np.asarray(mhz, dtype=float)
    mhz = mhz[mhz > 0]
    if not len(mhz):
        return None
    return {"median_mhz": float(np.median(mhz)), "mean_mhz": float(mhz.mean()), "p10_mhz": float(np.percentile(mhz, 10)),
            "p90_mhz": float(np.percentile(mhz, 90)), "windows": int(len(mhz)),
            "method": "one probe wave resident beside the timed launches: delta s_memtime / delta s_memrealtime x 100 MHz per 0.5 ms "
                      "window (muse_test_clock_probe_*), all windows of the timed region"}


def cpu_baseline(dg, ref, N):
    """CPU port of the reference algorithm timed on this box's host cores on a bounded sample of the same rows:
    oracle/muse_cpu_fast.c (radix-4 Stockham real FFT, -O3 -march=native, one plan per thread) -- not the checker
    (oracle/muse_oracle.c, radix-2), which is what the parity tests use and what this port is itself checked against."""
    import numpy as np
    from oracle import oracle_py
    oracle_py.build()
    oracle_py.build_fast(force=True)            # -march=native: compiled on the host it is timed on
    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    probe = dg.read(0, min(2048, dg.M))
    oracle_py.fast_batch_scores(ref, probe[:64], nthreads=1)          # build + load
    t0 = time.perf_counter()
    oracle_py.fast_batch_scores(ref, probe, nthreads=threads)
    dt = max(time.perf_counter() - t0, 1e-6)
    rate = len(probe) / dt
    S = int(min(dg.M, max(len(probe), min(400_000, rate * 12.0))))   # ~12 s of work
    S -= S % 2
    rows = dg.read(0, S)
    t0 = time.perf_counter()
    lag, mv = oracle_py.fast_batch_scores(ref, rows, nthreads=threads)
    dt = time.perf_counter() - t0
    # SURVEY 8d also asks for the 1-thread figure: a short sample is enough (linear in rows)
    S1 = int(min(S, max(256, min(8000, rate / threads * 2.0))))
    S1 -= S1 % 2
    t0 = time.perf_counter()
    oracle_py.fast_batch_scores(ref, rows[:S1], nthreads=1)
    dt1 = time.perf_counter() - t0
    # the checker's own rate, for the record (it is what round 1 reported as the baseline)
    S2 = min(S1, 2000)
    t0 = time.perf_counter()
    olag, omv, _ = oracle_py.batch_scores(ref, rows[:S2], nthreads=1, want_gap=False)
    dt2 = time.perf_counter() - t0
    agree = bool(np.allclose(mv[:S2], omv, rtol=1e-9, atol=1e-12))
    return {"value": S / dt, "unit": "series-pairs/s", "cores": threads, "kind": "port",
            "single_thread": {"value": S1 / dt1, "unit": "series-pairs/s", "cores": 1, "sample": "first %d rows" % S1},
            "checker_single_thread": {"value": S2 / dt2, "unit": "series-pairs/s", "cores": 1,
                                      "note": "oracle/muse_oracle.c (radix-2, -O2): the parity checker, not the baseline",
                                      "port_agrees_with_checker": agree},
            "sample": "first %d rows of the same 1Mx%d synthetic matrix (D2H copy), %d pthreads, C port of go-muse "
                      "xCorrWithX with a radix-4 Stockham real FFT, -O3 -march=native (not Go/gonum)" % (S, N, threads)}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU) BEFORE this
    process makes any GPU call and exit with the worst return code.  (Never re-exec a process that has
    initialised the GPU.)  stdout of this process carries exactly what a launcher run would: rank 0's one
    JSON line; whatever else the ranks print goes to stderr, prefixed with the rank."""
    import subprocess
    import threading
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT", "0")) or (29500 + (os.getpid() % 2000))
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                    "MUSE_BENCH_CHILD": "1"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE, text=True))

    def relay(r, pr):
        for ln in pr.stdout:
            if r == 0 and ln.startswith("{"):
                sys.stdout.write(ln)
                sys.stdout.flush()
            else:
                sys.stderr.write("[rank %d] %s" % (r, ln))
    th = [threading.Thread(target=relay, args=(r, pr)) for r, pr in enumerate(procs)]
    for x in th:
        x.start()
    rc = 0
    for r, pr in enumerate(procs):
        c = pr.wait()
        if c != 0:
            print("bench.py: rank %d exited with code %d" % (r, c), file=sys.stderr, flush=True)
            rc = rc or c
    for x in th:
        x.join()
    sys.exit(rc)


def workload_name(M, N, n_gpus, max_lag, top_n):
    """BASELINE.json's configs by shape: [2] = 1 ref x 1 M series on one GPU, [3] = 8 M series sharded over 8 GPUs (1 M rows each)"""
    tail = "N=%d float64 rect+noise, Run(nil), MaxLag=%d TopN=%d" % (N, max_lag, top_n)
    if n_gpus == 1:
        tag = "configs[2]" if (M, N) == (1_000_000, 4096) else "configs[2]-shaped"
        return "%s: 1 ref x %d series on 1 GPU, %s" % (tag, M, tail)
    tag = "configs[3]" if (M * n_gpus, N, n_gpus) == (8_000_000, 4096, 8) else "configs[3]-shaped"
    return "%s: 1 ref x %d series, Group sharded by rows over %d GPUs (%d rows each, weak scaling), RCCL all_gather of per-shard top-N records, %s" % (
        tag, M * n_gpus, n_gpus, M, tail)


# BASELINE configs[4]: "N in {512 ... 65536} zero-padded to next pow2" -- SURVEY 8d's six lengths plus 20000 (-> 32768, the one FFT
# length the six leave out: every per-length kernel of the library is on the line)
CONFIG5_LENGTHS = (512, 1000, 4096, 5000, 16384, 20000, 65536)


def config5_leg(pkg, eng, args, rank, n_gpus, use_dist, tdev, M, N, counters):
    """BASELINE configs[4] as ONE workload, at any number of ranks: a mixed-length Group is seven (ref, Group) pairs, one per
    length (group.go:45-51 and muse_batch.go:24-28 allow one length per Group), each sharded by rows over the ranks (~4 GB per
    GPU; the headline's length: as many rows per GPU as the headline), label groups of 50 series ("graph") INTERLEAVED over the
    rows -- graph g = global rows g, g + G, g + 2G, ... -- so that every label group has members on every rank; every Batch
    Run(["graph"]) into ONE shared Results (results.go:55-72), one Fetch at the end.  Timed per length: the whole Run as the
    public API pays it (dist.ShardedBatch.Run: fused kernel + per-group maxima + G x 25 B copy-back + exchange of the ranks'
    per-group records + merge + the host feed of the shared heap, one Score per label group in group order up to 65 536 groups)
    between two barriers, the maximum over the ranks; and, by HIP events on each rank's own stream, the fused kernel alone
    (the SLOWEST rank's -> share of the HBM roofline on 8 N + 16 bytes per series).  ALL ranks call this (it holds collectives);
    every rank returns the objects, rank 0 prints them.  Parity of exactly this flow: tests/test_dist_gloo.py
    (test_grouped_run_with_straddling_label_groups_gloo) and tests/test_gpu_parity.py
    (test_config5_mixed_lengths_one_shared_results, test_sharded_batch_run_on_one_rank_equals_batch_run)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    D = pkg.dist
    per_len, shared, carry = [], pkg.NewResults(args.max_lag, args.top_n, 0.0, 0), []
    run_s_total, rows_total = 0.0, 0

    def fence():
        eng.synchronize()
        if use_dist:
            dist.barrier()

    for Nl in CONFIG5_LENGTHS:
        try:
            # (the leg of the headline's own length runs on as many rows as the headline: every launch of that kernel
            # instantiation in this process then has ONE workload, and its rocprofv3 averages mean what they say)
            rows_l = M if Nl == N else max(2048, min(400_000, (1 << 32) // (8 * Nl)))
            if args.config5_rows:
                rows_l = min(rows_l, args.config5_rows)
            rows_all = rows_l * n_gpus
            # (without the planted exact copies of the reference: the shared top-N is then a field of distinct scores from
            # all the lengths, not twenty 1.0s from the first)
            dgl, refl = pkg.DeviceGroup.synthetic(eng, rows_l, Nl, seed=0x6D757365, global_first=rank * rows_l, copies=False)
            dbl = pkg.DeviceBatch(eng, dgl, refl)
            Gl = max(1, rows_all // 50)
            gidl = ((rank * rows_l + np.arange(rows_l, dtype=np.int64)) % Gl).astype(np.int32)
            sb = D.ShardedBatch(dbl, rank * rows_l, shared,
                                lambda i, g, Nl=Nl: pkg.NewLabels({"len": str(Nl), "graph": "g%d" % g, "row": str(i)}), device=tdev)
            sb.Run(gidl, Gl)
            fence()
            eng.kernel_time()
            eng.kernel_timing(True)
            reps = 3
            t1 = time.perf_counter()
            for r_ in range(reps):
                shared.Fetch()                                          # every repeat starts from what the EARLIER lengths left
                for sc_ in carry:
                    shared.Update(sc_)
                sb.Run(gidl, Gl)
            fence()
            dtl = (time.perf_counter() - t1) / reps
            eng.kernel_timing(False)
            kl_ms, kl_cnt = eng.kernel_time()
            mine = {"run_s": dtl, "kernel_ms_avg": kl_ms / max(kl_cnt, 1)}
            allr = [mine]
            if use_dist:
                allr = [None] * n_gpus
                dist.all_gather_object(allr, mine)
            dtl = max(x["run_s"] for x in allr)
            kms = [x["kernel_ms_avg"] for x in allr]
            kl_s = max(kms) * 1e-3
            carry = shared.Fetch()[0]                                   # (Fetch drains, results.go:75-87: put it back)
            for sc_ in carry:
                shared.Update(sc_)
            exact = Gl <= pkg.muse.EXACT_FEED_MAX_GROUPS
            e_ = {"N": Nl, "fft_len": dbl.n, "rows": rows_l, "rows_total": rows_all, "length": Nl, "label_groups": Gl,
                  "exchange": ("all_gather of G x 25 B per rank + one Score per label group through Results.Update (the reference's feed)"
                               if exact else "all_to_all of G/W-group slices + per-slice top-N + gather of top_n x 24 B"),
                  "kernel": eng.kernel_name(dbl), "run_ms": dtl * 1e3, "kernel_ms_avg": kl_s * 1e3,
                  "kernel_ms_per_rank": {"min": min(kms), "max": max(kms)},
                  "series_per_s": rows_all / dtl, "series_per_s_kernel": rows_all / kl_s if kl_s > 0 else None,
                  "roofline_frac": rows_l * (8.0 * Nl + 16.0) / kl_s / 1e9 / HBM_PEAK_GBPS if kl_s > 0 else None}
            counters.attach(e_, eng.kernel_name(dbl), rows_l, Nl, rows_l * (8.0 * Nl + 16.0))
            per_len.append(e_)
            run_s_total += dtl
            rows_total += rows_all
            dbl.close()
            dgl.close()
        except Exception as e:
            if use_dist:
                raise                                                   # (a rank that drops out of a collective hangs the others)
            per_len.append({"N": Nl, "error": str(e)})
    top, mean_abs = shared.Fetch()
    tag = "configs[4]" if n_gpus == 8 else ("configs[4], single-GPU half" if n_gpus == 1 else "configs[4]-shaped")
    mixed = {"value": rows_total / run_s_total if run_s_total > 0 else None, "unit": "series/s", "series_total": rows_total,
             "n_gpus": n_gpus, "ms_total": run_s_total * 1e3, "lengths": [e_["N"] for e_ in per_len if "error" not in e_],
             "shared_results": {"top_n": args.top_n, "fetched": len(top), "mean_abs_score": mean_abs,
                                "lengths_in_top_n": sorted({s_.Labels.labels["len"] for s_ in top}, key=int)},
             "note": "%s: seven (ref, Group) pairs, every Group sharded by rows over %d rank(s) with its 50-series label groups interleaved "
                     "over ALL ranks, Run([\"graph\"]) each through dist.ShardedBatch (muse_batch_run_groups + exchange + muse_merge_group_winners "
                     "+ host feed) into ONE Results, one Fetch; per-length Run and kernel times in config5_lengths" % (tag, n_gpus)}
    return per_len, mixed


def reference_bench_shapes(pkg):
    """The reference's own published benchmark shapes (README.md:98-108): go-muse_amd/host/muse_ref_bench.cpp
    restates the loop bodies of muse_test.go:144-215, muse_batch_test.go:104-162 and xcorr_test.go:322-348 over the C++ host
    mirror and is run here as a CHILD process (its own context on the same GPU, while this process is idle); beside each the
    README's laptop figure and this box's CPU port (oracle/muse_cpu_fast.c, `kind: port`) timed on the same shape."""
    import subprocess
    import numpy as np
    exe = pkg.build.build_ref_bench()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        return {"error": "muse_ref_bench exited with %d: %s" % (r.returncode, r.stderr[-400:])}
    obj = json.loads(r.stdout)
    # the CPU port on the same shapes (not Go / gonum; the README's numbers are a 2018 laptop's)
    from oracle import oracle_py
    oracle_py.build_fast()
    rng = np.random.default_rng(20200419)
    threads = max(1, min(os.cpu_count() or 1, 64))

    def cpu_ns(ref, rows, nthreads, min_s=0.2):
        oracle_py.fast_batch_scores(ref, rows, nthreads=nthreads)
        reps, t0 = 0, time.perf_counter()
        while True:
            oracle_py.fast_batch_scores(ref, rows, nthreads=nthreads)
            reps += 1
            el = time.perf_counter() - t0
            if el >= min_s and reps >= 3:
                return el / reps * 1e9
    ref480, rows480 = 0.1 * rng.standard_normal(480), 0.1 * rng.standard_normal((5000, 480))
    for name in ("BenchmarkMuseRunLarge", "BenchmarkMuseBatchRunLarge"):
        obj[name]["cpu_port_ns_per_op"] = {"all_cores": cpu_ns(ref480, rows480, threads), "cores": threads, "one_thread": cpu_ns(ref480, rows480, 1)}
    x, y = rng.uniform(size=16385), rng.uniform(size=(1, 16385))
    obj["BenchmarkXCorrWithX"]["cpu_port_ns_per_op"] = {"one_thread": cpu_ns(x, y, 1), "cores": 1}
    ref8 = np.array([0.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4])
    rows8 = np.array([[0.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4], [0.2, 0.1, 0.2, 0.1, 0.2, 0.1, 0.2, 0.1], [0.0, 0.0, 0.0, 0.0, 0.2, 0.4, 0.5, 0.8],
                      [0.2, 0.1, 0.2, 0.1, 0.2, 0.1, 0.22, 0.1], [0.0, 0.0, 0.0, 0.0, -0.2, -0.4, 0.0, -0.8], [0.0, 0.0, 0.0, -0.2, -0.4, -0.6, 1.0, 0.0]])
    for name in ("BenchmarkMuseRun", "BenchmarkMuseBatchRun"):
        obj[name]["cpu_port_ns_per_op"] = {"one_thread": cpu_ns(ref8, rows8, 1, 0.05), "cores": 1, "note": "the six xCorrWithX alone (no Results, no labels)"}
    obj["note"] = ("ns_per_op: this engine through the C++ host mirror, comparison data resident in HBM where the reference's loop keeps it in RAM "
                   "(Muse.Run uploads its group in every call by design: muse_batch_run_rows); ns_per_op_cold: upload included; "
                   "readme_ns_per_op: README.md:98-108 (2018 MacBook Air, 1.6 GHz i5, 4 logical CPUs; Go + gonum); cpu_port_ns_per_op: "
                   "oracle/muse_cpu_fast.c on this box's host cores.  The 8-sample shapes are launch latency, not throughput: a GPU call "
                   "cannot be cheaper than one kernel launch and one completion wait")
    return obj


def in_process_shards(pkg, args, eng, ref, N, devs=None):
    """SURVEY 8e inside ONE process (what NewBatch over a device list does -- the path a Go / C++ caller of the C ABI takes:
    INTEGRATION.md): the same logical group cut into one contiguous row range per context, every shard scored at the same time
    from its own host thread, per-shard top-N records merged on the host.  With one GPU visible the contexts share it (device 0
    listed twice): that exercises the path and prices its host side; it is NOT a scaling number."""
    import threading
    import numpy as np
    ndev = pkg.device_count()
    sel = args.in_process_devices
    if devs is None:
        if sel == "all":
            devs = list(range(ndev))
        elif sel == "auto":
            devs = list(range(ndev)) if ndev > 1 else [0, 0]
        else:
            devs = list(range(min(int(sel), ndev)))
            if len(devs) < 2:
                devs = [0, 0]
    per = args.in_process_rows
    engs = [pkg.Engine(d) for d in devs]
    # (without the planted exact copies of the reference: among exactly tied scores the ORDER Fetch returns depends on
    # the heap's history, results.go:55-87, and a pre-selected merge has another history than one long feed)
    parts = [pkg.DeviceGroup.synthetic(e, per, N, seed=0x6D757365, global_first=k * per, copies=False) for k, e in enumerate(engs)]
    if ref is None:
        ref = parts[0][1]
    sb = [pkg.DeviceBatch(e, g, ref) for e, (g, _) in zip(engs, parts)]
    whole, _ = pkg.DeviceGroup.synthetic(eng, per * len(devs), N, seed=0x6D757365, copies=False)
    wb = pkg.DeviceBatch(eng, whole, ref)

    def sharded():
        recs = [None] * len(sb)

        def work(k):
            recs[k] = sb[k].run_shard(None, 0, k * per, args.max_lag, args.top_n, 0.0, 0, True)
        th = [threading.Thread(target=work, args=(k,)) for k in range(len(sb))]
        for x in th:
            x.start()
        for x in th:
            x.join()
        return pkg.merge_records(np.concatenate(recs), args.top_n)

    a = sharded()
    b = wb.run(None, 0, args.max_lag, args.top_n, 0.0, 0, True)
    same = a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist() and np.array_equal(a[2], b[2])
    reps = 5
    t1 = time.perf_counter()
    for _ in range(reps):
        sharded()
    dsh = (time.perf_counter() - t1) / reps
    t1 = time.perf_counter()
    for _ in range(reps):
        wb.run(None, 0, args.max_lag, args.top_n, 0.0, 0, True)
    dwh = (time.perf_counter() - t1) / reps
    out = {
        "devices": devs, "distinct_devices": len({e.pci_bus_id() for e in engs}), "pci_bus_ids": [e.pci_bus_id() for e in engs],
        "rows_per_shard": per, "length": N,
        "ms_per_run_sharded": dsh * 1e3, "ms_per_run_one_context_same_rows": dwh * 1e3,
        "value": per * len(devs) / dsh, "unit": "series-pairs/s", "records_identical_to_one_context": bool(same),
        "note": "one process, one muse_ctx + host thread per listed device, muse_batch_run_shard + muse_merge_records"
                + ("" if len(set(devs)) > 1 else "; ONE GPU: the contexts share it -- a check of the path and of its host-side cost, not a scaling number")}
    for x in sb:
        x.close()
    wb.close()
    whole.close()
    for g, _ in parts:
        g.close()
    return out


def in_process_child_main(args):
    """`bench.py --in-process-child --gpus N`: the in_process_shards object over devices 0 .. N-1 in a process of its own (what
    rank 0 of an N-rank run starts once the ranks have finished), printed as one JSON object."""
    pkg = importlib.import_module("go-muse_amd")
    eng = pkg.Engine(0)
    eng.set_screening(False)
    try:
        ndev = pkg.device_count()   # (a rehearsal on a box with fewer GPUs than ranks lists its devices again: the note says so)
        out = in_process_shards(pkg, args, eng, None, args.length, devs=[d % ndev for d in range(args.gpus)])
    except Exception as e:
        out = {"error": str(e)}
    print(json.dumps(out), flush=True)


def run_in_process_child(args):
    """rank 0 of an N-rank run, after the process group is gone and its own rows are freed: the one-process design over the
    same N devices, so that the node the driver measures `dist.py` on also measures what a Go / C++ caller gets"""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "MUSE_BENCH_CHILD") and not k.startswith("TORCHELASTIC") and not k.startswith("NCCL_ASYNC")}
    cmd = [sys.executable, os.path.abspath(__file__), "--in-process-child", "--gpus", str(args.gpus), "--length", str(args.length),
           "--in-process-rows", str(args.in_process_rows), "--top-n", str(args.top_n), "--max-lag", str(args.max_lag)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        for ln in reversed(r.stdout.splitlines()):
            if ln.startswith("{"):
                out = json.loads(ln)
                out["note"] = out.get("note", "") + "; a child process started by rank 0 after the ranks' timed region (the ranks' rows freed first)"
                return out
        return {"error": "child exited with code %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
    except Exception as e:
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000, help="series per GPU")
    ap.add_argument("--length", type=int, default=4096)
    ap.add_argument("--top-n", type=int, default=20)
    ap.add_argument("--max-lag", type=int, default=15)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--clock-probe", action="store_true",
                    help="keep the one-wave shader-clock probe resident INSIDE the timed region (default: the clock that prices "
                         "roofline.co_bounds is measured in a separate, untimed pass of the same steps behind it)")
    ap.add_argument("--in-process-devices", default="auto",
                    help="devices of the in_process_shards extra object (one process, one context + host thread per device, "
                         "SURVEY 8e / INTEGRATION.md): 'all' = every visible GPU, a number k = the first k, 'auto' = all when "
                         "more than one is visible, else device 0 twice (a check of the path, not a scaling number)")
    ap.add_argument("--in-process-rows", type=int, default=200_000, help="rows per shard of in_process_shards")
    ap.add_argument("--with-in-process-child", action="store_true",
                    help="run the in_process_shards child of an N-rank run even with --no-extras / in a rehearsal (tests)")
    ap.add_argument("--in-process-child", action="store_true",
                    help="(internal) print the in_process_shards object over devices 0 .. --gpus - 1 and exit: what rank 0 of an "
                         "N-rank run starts as a child once the ranks are done")
    ap.add_argument("--many-refs", type=int, default=8,
                    help="also time muse_batch_run_many with this many references (N=1 only; 0 = skip); "
                         "reported as an extra object, never as `value`")
    ap.add_argument("--skip-extra", action="append", default=[],
                    help="leave one extra object out (tools/profile.sh: in_process_shards re-launches the headline kernel on other "
                         "row counts, which would blur that kernel's per-launch counter means)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra objects (filter_and_refine_run, f32_storage_group, many_references, config5_lengths)")
    ap.add_argument("--with-config5", action="store_true", help="run the config5 leg even with --no-extras")
    ap.add_argument("--config5-rows", type=int, default=0, help="cap the rows per GPU and length of the config5 leg (tests)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the RCCL gather path even with one rank (rehearsal on a 1-GPU box)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="rehearsal of the N-rank flow on a 1-GPU box: every rank uses GPU 0 and the gather runs over gloo "
                         "(RCCL refuses two ranks on one device); the line it prints is not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.in_process_child:
        in_process_child_main(args)
        return

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        spawn_ranks(args)                                # does not return
    world = int(env_world) if env_world is not None else 1
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or args.force_dist
    ndev = torch.cuda.device_count()                     # (does not initialise the GPU on this image)
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if ndev <= local_rank:
        sys.exit("bench.py: rank %d needs GPU %d but only %d visible: this engine has no CPU path" % (rank, local_rank, ndev))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=max(world, 1))
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=max(world, 1),
                                    device_id=torch.device("cuda", local_rank))
    n_gpus = max(world, 1)

    pkg = importlib.import_module("go-muse_amd")
    # the library ships prebuilt in the tree; rebuild only if a source is newer, and then by ONE rank
    if use_dist:
        if local_rank == 0 and pkg.build.stale():
            pkg.build.build()
        dist.barrier()
    elif pkg.build.stale():
        pkg.build.build()
    eng = pkg.Engine(local_rank)                      # raises without a gfx950 GPU: no fallback
    eng.set_screening(False)                          # headline: every series through the float64 kernel
    dev_name, cus, hbm = eng.device_info()
    M, N = args.rows, args.length
    dg, ref = pkg.DeviceGroup.synthetic(eng, M, N, seed=0x6D757365, global_first=rank * M)
    db = pkg.DeviceBatch(eng, dg, ref)
    tdev = None if args.rehearse_on_one_gpu else torch.device("cuda", local_rank)   # (gloo gathers host tensors)

    def step():
        if use_dist:
            return pkg.dist.run_sharded(db, rank * M, None, 0, args.max_lag, args.top_n, 0.0, 0, True, device=tdev)
        return db.run(None, 0, args.max_lag, args.top_n, 0.0, 0, True)

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def probed(fn, est_total_ms, after=lambda: None):
        """runs fn() with the one-wave shader-clock probe resident beside it, then after() (the caller's closing fence and clock
        reading) BEFORE the probe's windows are collected; -> (clock_stats or None, note, the time fn() began).
        The probe is a measurement hook: whatever goes wrong with it, the benchmark goes on without a clock (sclk = None)."""
        try:
            eng.clock_probe_start(0.5, min(50000.0, est_total_ms * 1.5 + 30.0))   # returns once the probe is resident
        except Exception as e:
            t_begin = time.perf_counter()
            fn()
            after()
            return None, "clock probe did not start: %s" % e, t_begin
        t_begin = time.perf_counter()
        try:
            fn()
        finally:
            try:
                eng.clock_probe_stop()                      # (a host flag: a device-wide synchronize must not wait for the probe)
            except Exception:
                pass
        after()
        try:
            eng.synchronize()
            raw = eng.clock_probe_read()
            if os.environ.get("MUSE_BENCH_DUMP_CLOCK"):
                np.savetxt(os.environ["MUSE_BENCH_DUMP_CLOCK"], raw, fmt="%.0f")
            return clock_stats(raw), None, t_begin
        except Exception as e:
            return None, "clock probe could not be read: %s" % e, t_begin

    est_ms = 15.0 * max(1.0, M * N / 4.096e9)
    for i in range(args.warmup):
        t0 = time.perf_counter()
        step()
        if i == 0:
            fence()
            est_ms = (time.perf_counter() - t0) * 1e3
    fence()
    eng.kernel_time()                                   # drop warm-up events
    eng.redo_time()
    eng.kernel_timing(True)
    out = [None]

    def timed_steps():
        for _ in range(args.steps):
            out[0] = step()

    # ---- the timed region: exactly `steps` Runs between two fences.  By default NOTHING else runs on the GPU in it; with
    # --clock-probe the one-wave probe kernel sits on its own stream beside the Runs (said on the line: clock_probe_in_timed_region)
    sclk = sclk_note = None
    t_end = [None]

    def close_region():
        fence()
        t_end[0] = time.perf_counter()
    t0 = time.perf_counter()
    if args.clock_probe:
        sclk, sclk_note, t0 = probed(timed_steps, est_ms * (args.steps + 2), close_region)   # (t0: behind the probe's start-up)
    else:
        timed_steps()
        close_region()
    dt = t_end[0] - t0
    eng.kernel_timing(False)
    out = out[0]
    k_ms, k_cnt = eng.kernel_time()
    r_ms, r_cnt = eng.redo_time()
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=tdev if tdev is not None else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # every rank reports its own GPU: kernel time by HIP events on its stream, device name and PCI bus id -- the line's proof
    # that N ranks ran on N distinct devices, and what makes a slow GPU visible
    mine = {"rank": rank, "local_rank": local_rank, "device": dev_name, "pci_bus_id": eng.pci_bus_id(), "rows": M,
            "kernel_ms_avg": k_ms / max(k_cnt, 1), "launches_timed": k_cnt, "redo_ms_avg": r_ms / max(r_cnt, 1)}
    per_rank = [mine]
    if use_dist:
        per_rank = [None] * n_gpus
        dist.all_gather_object(per_rank, mine)
    # the clock that prices the co-bounds (rank 0): by default from an UNTIMED repeat of the same steps behind the timed region
    if rank == 0 and not args.clock_probe and not args.no_extras:
        reps = max(3, min(args.steps, 10))
        sclk, sclk_note, _ = probed(lambda: [db.run(None, 0, args.max_lag, args.top_n, 0.0, 0, True) for _ in range(reps)], est_ms * (reps + 2))
        eng.synchronize()

    screened, refined_pairs = db.last_run_info()
    assert not screened, "the headline Run must not take the fp32 filter-and-refine path"
    # BASELINE configs[4] (mixed lengths, label-grouped, one shared Results): every rank takes part, rank 0 reports
    c5 = None
    if (not args.no_extras or args.with_config5) and "config5" not in args.skip_extra:
        c5 = config5_leg(pkg, eng, args, rank, n_gpus, use_dist, tdev, M, N, Counters())
    if rank == 0:
        total_pairs = float(M) * n_gpus * args.steps
        value = total_pairs / dt
        kms = [r["kernel_ms_avg"] for r in per_rank]
        k_avg_s = max(kms) * 1e-3                       # the SLOWEST rank's kernel: roofline.frac is per GPU and conservative
        bytes_per_launch = float(M) * (8 * N + 16)
        achieved = bytes_per_launch / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        kname = eng.kernel_name(db) if hasattr(eng, "kernel_name") else "xcorr_fused"
        counters = Counters()
        line = {
            "metric": "series-pairs XCorr/sec at N=4096, 1M-series batch; achieved HBM GB/s",
            "value": value, "unit": "series-pairs/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rehearsal": bool(args.rehearse_on_one_gpu),
            "config": {"workload": workload_name(M, N, n_gpus, args.max_lag, args.top_n),
                       "rows_per_gpu": M, "rows_total": M * n_gpus, "length": N, "fft_len": db.n, "sharding": "rows x %d" % n_gpus,
                       "device": dev_name},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": kname, "kernel_ms_avg": k_avg_s * 1e3, "launches_timed": min(r["launches_timed"] for r in per_rank),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "per_gpu": True,
                         "per_rank": {"kernel_ms_avg": {"min": min(kms), "max": max(kms), "mean": sum(kms) / len(kms)},
                                      "frac": {"min": bytes_per_launch / (max(kms) * 1e-3) / 1e9 / HBM_PEAK_GBPS if max(kms) > 0 else None,
                                               "max": bytes_per_launch / (min(kms) * 1e-3) / 1e9 / HBM_PEAK_GBPS if min(kms) > 0 else None},
                                      "note": "HIP events on each rank's own stream around the fused launch alone; frac / achieved / "
                                              "kernel_ms_avg above are the SLOWEST rank's"},
                         "redo_ms_avg": max(r["redo_ms_avg"] for r in per_rank),
                         "redo_note": "the launch behind the fused one that redoes listed pairs (NaN / Inf series, sigma spread): outside "
                                      "kernel_ms_avg, inside ms_per_step"},
            "ranks": per_rank,
            "distinct_devices": len({r["pci_bus_id"] for r in per_rank}),
            "clock_probe_in_timed_region": bool(args.clock_probe),
            "achieved_hbm_gbps_whole_step": value / n_gpus * (8 * N + 16) / 1e9,
            "top_score": float(out[2][0]) if len(out[2]) else None,
        }
        # HBM bytes of the dominant kernel from the PMC counters (separate rocprofv3 passes, FETCH_SIZE doubled on gfx950:
        # tools/profile.sh) and the ceilings the kernel shares the chip with: its fp64
        # arithmetic at one vector instruction per SIMD per 4 cycles and the LDS array, both at the clock measured in THIS run
        rl = line["roofline"]
        krec = counters.attach(rl, kname, M, N, bytes_per_launch)
        co = {"hbm_measured_copy": {"peak": HBM_COPY_GBPS, "unit": "GB/s", "frac": achieved / HBM_COPY_GBPS},
              "sclk_in_kernel": sclk,
              "sclk_measured_in": "the timed region (--clock-probe)" if args.clock_probe else
                                  "an untimed repeat of the same Runs on rank 0 behind the timed region (the probe wave is not resident while `value` is timed)"}
        if sclk_note:
            co["sclk_note"] = sclk_note
        if krec and sclk and k_avg_s > 0:
            hz = sclk["median_mhz"] * 1e6
            simds = cus * 4
            valu_ms = krec["SQ_INSTS_VALU"] * 4.0 / simds / hz * 1e3
            lds_ms = krec["SQ_LDS_IDX_ACTIVE"] / cus / hz * 1e3
            co["fp64_valu_issue"] = {"vector_insts_per_launch": krec["SQ_INSTS_VALU"], "floor_ms": valu_ms,
                                     "kernel_over_floor": k_avg_s * 1e3 / valu_ms, "frac": valu_ms / (k_avg_s * 1e3),
                                     "note": "SQ_INSTS_VALU x 4 cycles / (%d CUs x 4 SIMDs) / measured clock: the time the kernel's "
                                             "vector instructions (~1 380 of ~1 610 per wave and pair are v_*_f64) need at full issue" % cus}
            co["lds_array"] = {"array_cycles_per_launch": krec["SQ_LDS_IDX_ACTIVE"], "bank_conflict_cycles": krec.get("SQ_LDS_BANK_CONFLICT"),
                               "floor_ms": lds_ms, "frac": lds_ms / (k_avg_s * 1e3),
                               "note": "SQ_LDS_IDX_ACTIVE / CUs / measured clock: the time the LDS arrays are busy"}
        elif not krec:
            co["note"] = "fp64-VALU / LDS ceilings need the kernel's counters: " + str(rl["traffic_source"].get("note"))
        rl["co_bounds"] = co
        extras = n_gpus == 1 and not use_dist and not args.no_extras
        if extras and db.n >= 512:
            # opt-in filter-and-refine Run (muse_ctx_set_screening(ctx, 1)): fp32 screening pass over every row + fp64
            # re-evaluation of the rows that can reach the top-N; same records.  An extra object, never `value`.
            eng.set_screening(True)
            step()
            eng.synchronize()
            eng.kernel_time()
            eng.kernel_timing(True)
            reps = max(3, min(args.steps, 10))
            t1 = time.perf_counter()
            for _ in range(reps):
                step()
            eng.synchronize()
            dts = (time.perf_counter() - t1) / reps
            eng.kernel_timing(False)
            ks_ms, ks_cnt = eng.kernel_time()
            scr, refined = db.last_run_info()
            eng.set_screening(False)
            ks_s = ks_ms / max(ks_cnt, 1) * 1e-3
            line["filter_and_refine_run"] = {
                "value": float(M) / dts, "unit": "series-pairs/s", "ms_per_step": dts * 1e3, "dtype": "f32+f64",
                "took_the_path": bool(scr), "pairs_re_evaluated_in_fp64": refined,
                "kernel": "xcorr_screen_pass", "kernel_ms_avg": ks_s * 1e3,
                "roofline_frac": bytes_per_launch / ks_s / 1e9 / HBM_PEAK_GBPS if ks_s > 0 else None,
                "note": "opt-in (muse_ctx_set_screening): fp32 transforms screen, fp64 re-evaluates what can reach the top-N"}
        if extras and db.n == 4096:
            # SURVEY 8f-3, opt-in float32-STORAGE group (muse_group_create_f32): the same workload rounded to float32 in
            # HBM, widened exactly on consumption, float64 arithmetic.  Its own denominator: 4 N + 16 bytes per series.
            try:
                dg32, ref32 = pkg.DeviceGroup.synthetic(eng, M, N, seed=0x6D757365, f32=True)
                db32 = pkg.DeviceBatch(eng, dg32, ref32)
                db32.run(None, 0, args.max_lag, args.top_n, 0.0, 0, True)
                eng.synchronize()
                eng.kernel_time()
                eng.kernel_timing(True)
                reps = max(3, min(args.steps, 10))
                t1 = time.perf_counter()
                for _ in range(reps):
                    db32.run(None, 0, args.max_lag, args.top_n, 0.0, 0, True)
                eng.synchronize()
                dt32 = (time.perf_counter() - t1) / reps
                eng.kernel_timing(False)
                k32_ms, k32_cnt = eng.kernel_time()
                k32_s = k32_ms / max(k32_cnt, 1) * 1e-3
                b32 = float(M) * (4 * N + 16)
                line["f32_storage_group"] = {
                    "value": float(M) / dt32, "unit": "series-pairs/s", "ms_per_step": dt32 * 1e3, "dtype": "f64 arithmetic on f32-stored rows",
                    "kernel": eng.kernel_name(db32), "rows": M, "length": N,
                    "kernel_ms_avg": k32_s * 1e3, "algorithmic_bytes_per_launch": b32,
                    "roofline_frac": b32 / k32_s / 1e9 / HBM_PEAK_GBPS if k32_s > 0 else None,
                    "note": "opt-in muse_group_create_f32: half the HBM bytes; inputs rounded to float32 (scores are the "
                            "reference's on the rounded rows, not within 1e-6 of the float64 inputs' in general)"}
                counters.attach(line["f32_storage_group"], eng.kernel_name(db32), M, N, b32)
                db32.close()
                dg32.close()
            except Exception as e:            # (e.g. not enough HBM next to the float64 group)
                line["f32_storage_group"] = {"error": str(e)}
        if extras and args.many_refs > 1 and db.n == 4096 and N == 4096:
            # R references against the same resident group in one pass over the rows
            R = args.many_refs
            refs = [ref] + [dg.read(997 * r + 1, 1)[0] for r in range(1, R)]
            bs = [db] + [pkg.DeviceBatch(eng, dg, x) for x in refs[1:]]
            pkg.run_many(bs, None, 0, args.max_lag, args.top_n, 0.0, 0, True)
            eng.synchronize()
            eng.kernel_time()
            eng.kernel_timing(True)
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                pkg.run_many(bs, None, 0, args.max_lag, args.top_n, 0.0, 0, True)
            eng.synchronize()
            dtm = (time.perf_counter() - t1) / reps
            eng.kernel_timing(False)
            km_ms, km_cnt = eng.kernel_time()
            km_s = km_ms / max(km_cnt, 1) * 1e-3
            bm = float(M) * (8 * N + 16 * R)                      # every row read once, one result per (row, reference)
            mk = "xcorr_fused_n4096_fold_multi<false, false>"
            line["many_references"] = {"references": R, "value": R * float(M) / dtm, "unit": "series-pairs/s",
                                       "ms_per_run": dtm * 1e3, "dtype": "f64", "kernel": mk, "rows": M, "length": N,
                                       "kernel_ms_avg": km_s * 1e3, "algorithmic_bytes_per_launch": bm,
                                       "roofline_frac": bm / km_s / 1e9 / HBM_PEAK_GBPS if km_s > 0 else None,
                                       "vs_single_reference_passes": (R * float(M) / dtm) / value if value > 0 else None,
                                       "note": "muse_batch_run_many: one pass over the rows for all references; the pairs' spectra stay "
                                               "in registers, so the pass is bound by its fp64 arithmetic ((1 + R) / 2R of R single passes), "
                                               "not by HBM"}
            counters.attach(line["many_references"], mk, M, N, bm)
            for b in bs[1:]:
                b.close()
            del bs
        if extras and N == 4096:
            # the batched two-sided xCorr (xcorr.go:102-153), 200 000 independent (x, y) pairs resident in HBM;
            # algorithmic bytes per pair: both rows in, (lag, value, nil flag) out
            try:
                P = min(200_000, M)
                gx, _ = pkg.DeviceGroup.synthetic(eng, P, N, seed=0x78636F72)
                gy, _ = pkg.DeviceGroup.synthetic(eng, P, N, seed=0x6D757365)
                pkg.xcorr_groups(gx, gy, N, True)
                eng.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    pkg.xcorr_groups(gx, gy, N, True)
                eng.synchronize()
                dtx = (time.perf_counter() - t1) / 3
                # the kernel time is taken over a SUSTAINED burst (the measurement hook makes one call launch the kernel 150 times back
                # to back, same results): one call costs the host tens of milliseconds around its 4.5 ms kernel, so calls in a loop
                # leave the GPU idle and every launch runs at the 2.4 GHz boost clock (profiles/r06_two_sided_clock.txt)
                eng.xcorr_repeat(150)   # (0.7 s: the clock settles within the first ~60 ms of a burst)
                pkg.xcorr_groups(gx, gy, N, True)
                eng.synchronize()
                eng.kernel_time()
                eng.kernel_timing(True)
                pkg.xcorr_groups(gx, gy, N, True)
                eng.synchronize()
                eng.kernel_timing(False)
                eng.xcorr_repeat(1)
                kx_ms, kx_cnt = eng.kernel_time()
                kx_s = kx_ms / max(kx_cnt, 1) * 1e-3
                bx = float(P) * (16 * N + 16)
                xk = "xcorr_two_sided_fold<false>"
                line["two_sided_xcorr"] = {"value": P / kx_s if kx_s > 0 else None, "unit": "xCorr pairs/s (kernel)", "pairs": P, "rows": P, "length": N,
                                           "normalize": True, "dtype": "f64", "kernel": xk, "kernel_ms_avg": kx_s * 1e3, "launches_timed": kx_cnt,
                                           "burst": "150 launches back to back (0.6 s) behind one such burst; after several seconds of bursts the part settles at "
                                                    "1.94 GHz and 4.89 ms per launch = 0.335 (profiles/r06_two_sided_clock.txt)",
                                           "ms_per_call_with_copy_back": dtx * 1e3, "algorithmic_bytes_per_launch": bx,
                                           "roofline_frac": bx / kx_s / 1e9 / HBM_PEAK_GBPS if kx_s > 0 else None,
                                           "note": "muse_xcorr_groups: z = (x read backwards) + i y, one forward transform, cc = Im FFT(Z^2) / 2n on the "
                                                   "xCorrWithX kernel's transforms (no spectrum table, no mirrored element); statistics first, "
                                                   "one extra workgroup barrier per pair"}
                counters.attach(line["two_sided_xcorr"], xk, P, N, bx)
                gx.close()
                gy.close()
            except Exception as e:
                line["two_sided_xcorr"] = {"error": str(e)}
        if extras and "in_process_shards" not in args.skip_extra:
            try:
                line["in_process_shards"] = in_process_shards(pkg, args, eng, ref, N)
            except Exception as e:
                line["in_process_shards"] = {"error": str(e)}
        if extras and "long_series" not in args.skip_extra:
            # series longer than 65 536 samples (the reference has no length limit: xcorr.go:19-24, muse_batch.go:33-37): FFT lengths
            # 2^17 and 2^20 through the chip-wide four-step kernels of xcorr_huge.hip, 1 GB groups, HIP events around the whole pass
            out = []
            for NL in (131072, 1048576):
                try:
                    ML = max(8, (1 << 30) // (8 * NL))
                    gl, refl = pkg.DeviceGroup.synthetic(eng, ML, NL)
                    bl = pkg.DeviceBatch(eng, gl, refl)
                    for _ in range(2):
                        bl.score()
                    eng.synchronize()
                    eng.kernel_time()
                    eng.kernel_timing(True)
                    for _ in range(4):
                        bl.score()
                    eng.synchronize()
                    eng.kernel_timing(False)
                    kms, kcnt = eng.kernel_time()
                    ts = kms / max(kcnt, 1) * 1e-3
                    out.append({"length": NL, "fft_len": bl.n, "rows": ML, "pass_ms": ts * 1e3, "value": ML / ts, "unit": "series-pairs/s",
                                "ps_per_sample": ts / (ML * float(NL)) * 1e12, "roofline_frac": ML * (8.0 * NL + 16) / ts / 1e9 / HBM_PEAK_GBPS,
                                "kernels": "huge_stats, huge_norm, huge_sweep1, huge_rows, huge_sweep2, huge_final per batch of pairs"})
                    bl.close()
                    gl.close()
                except Exception as e:
                    out.append({"length": NL, "error": str(e)})
            line["long_series"] = out
        if c5 is not None:
            line["config5_lengths"], line["config5_mixed_run"] = c5
        if extras and "reference_bench_shapes" not in args.skip_extra:
            try:
                eng.synchronize()
                line["reference_bench_shapes"] = reference_bench_shapes(pkg)
            except Exception as e:
                line["reference_bench_shapes"] = {"error": str(e)}
        if not args.no_cpu_baseline:
            # (at every N, on rank 0's host cores over a sample of rank 0's shard: the other ranks wait at the closing barrier)
            line["cpu_baseline"] = cpu_baseline(dg, ref, N)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if n_gpus > 1 and (args.with_in_process_child or (not args.no_extras and not args.rehearse_on_one_gpu
                                                          and "in_process_shards" not in args.skip_extra and pkg.device_count() >= n_gpus)):
            # the design a Go / C++ caller of the C ABI uses (one process, one context + host thread per device, host merge) on
            # the SAME node, after the ranks are done: the ranks have passed the closing barrier and left the process group (they
            # exit; nothing of theirs is running), this rank's rows go back first, and the measurement runs in a child process
            db.close()
            dg.close()
            eng.trim()
            line["in_process_shards"] = run_in_process_child(args)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
