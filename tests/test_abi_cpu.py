"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950,
loads, exports every symbol include/muse_hip.h declares, fails loudly without a
GPU, and the host-only pieces (merge, label bookkeeping, Results heap) behave
like the reference."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

from _load import ROOT, pkg


@pytest.fixture(scope="module")
def muse():
    m = pkg()
    m.build.build()
    return m


def test_header_symbols_all_exported(muse):
    declared = set()
    for h, least in (("muse_hip.h", 25), ("muse_hip_test.h", 3)):   # the drop-in boundary; the test / measurement hooks
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names = set(re.findall(r"\b(muse_[a-z0-9_]+)\s*\(", hdr))
        assert len(names) >= least, h
        declared |= names
    lib = ctypes.CDLL(muse.build.LIB)
    for name in sorted(declared):
        assert hasattr(lib, name), "libmuse_hip.so does not export %s" % name
    assert declared == set(muse.binding.SIGNATURES), "binding.py and the headers disagree"
    assert muse.binding.load().muse_abi_version() == 5
    # nothing is exported that no header declares
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", muse.build.LIB], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T muse_" in l}
    assert exported == declared, exported ^ declared


def _split_top_level(args):
    """splits an argument list at its top-level commas (nested parentheses, brackets and braces stay together)"""
    parts, depth, cur = [], 0, ""
    for ch in args:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return parts


def _call_args(text, start):
    """text[start] is the '(' of a call: -> the text between it and its matching ')'"""
    depth = 0
    for k in range(start, len(text)):
        if text[k] == "(":
            depth += 1
        elif text[k] == ")":
            depth -= 1
            if depth == 0:
                return text[start + 1:k]
    raise AssertionError("unbalanced call at %d" % start)


def test_go_shim_calls_match_the_header():
    """go-muse_amd/go/muse_hip.go is written blind (no Go toolchain in this image): every C.muse_*( call it makes must name a
    function include/muse_hip.h declares -- the product header, not the test hooks -- with the declared number of arguments,
    the C enum constants it uses must exist, and the record struct it reads must have the fields the header gives it."""
    hdr = open(os.path.join(ROOT, "include", "muse_hip.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char \*)\s*(muse_[a-z0-9_]+)\s*\(", hdr_nc):
        params = _call_args(hdr_nc, m.end() - 1).strip()
        declared[m.group(1)] = 0 if params in ("", "void") else len(_split_top_level(params))
    assert len(declared) >= 40
    go = open(os.path.join(ROOT, "go-muse_amd", "go", "muse_hip.go")).read()
    go_nc = re.sub(r"//[^\n]*", "", go)
    cgo_preamble = go[:go.index('import "C"')]
    assert '#include "muse_hip.h"' in cgo_preamble and "muse_hip_test.h" not in go
    calls = list(re.finditer(r"\bC\.(muse_[a-z0-9_]+)\(", go_nc))
    assert len(calls) >= 40
    used = set()
    for m in calls:
        name = m.group(1)
        assert name in declared, "muse_hip.go calls C.%s, which include/muse_hip.h does not declare" % name
        nargs = len(_split_top_level(_call_args(go_nc, m.end() - 1)))
        assert nargs == declared[name], "C.%s called with %d arguments, the header declares %d" % (name, nargs, declared[name])
        used.add(name)
    # the calls a drop-in Batch / Muse / sharded Run cannot do without
    for need in ("muse_ctx_create", "muse_group_create", "muse_group_stage", "muse_group_commit", "muse_batch_create",
                 "muse_batch_create_like", "muse_batch_run",
                 "muse_batch_run_shard", "muse_batch_run_groups", "muse_merge_records", "muse_merge_group_records",
                 "muse_batch_run_rows", "muse_last_error", "muse_device_count"):
        assert need in used, need
    # constants and types taken from the header by name
    for const in set(re.findall(r"\bC\.(MUSE_[A-Z0-9_]+)\b", go_nc)):
        assert re.search(r"\b%s\b" % const, hdr_nc), "muse_hip.go uses C.%s, which the header does not define" % const
    for typ in set(re.findall(r"\bC\.(muse_[a-z_]+)\b(?!\()", go_nc)) - set(declared):
        assert re.search(r"\b%s\b" % typ, hdr_nc), "muse_hip.go uses the type C.%s, which the header does not define" % typ
    rec = re.search(r"typedef struct muse_record \{(.*?)\} muse_record;", hdr_nc, flags=re.S)
    assert rec
    fields = set(re.findall(r"\b(\w+);", rec.group(1)))
    for f in set(re.findall(r"\brecs?\[[^\]]*\]\.(\w+)\b", go_nc)):
        assert f in fields, "muse_hip.go reads muse_record.%s; the header has %s" % (f, sorted(fields))
    # the reference's language level (go.mod:3, go 1.13): nothing of Go 1.17+ (unsafe.Slice / unsafe.Add), no generics, no 1.21 builtins
    for later in (r"\bunsafe\.(Slice|Add|String|SliceData)\(", r"\batomic\.(Int32|Int64|Uint32|Uint64|Bool|Pointer)\b", r"\bfunc \w+\[",
                  r"\bany\b", r"(?<![.\w])(min|max|clear)\("):
        assert not re.search(later, go_nc), "muse_hip.go needs a newer Go than the reference's go.mod: %s" % later


def test_record_layout(muse):
    assert ctypes.sizeof(muse.binding.MuseRecord) == 24 == muse.binding.RECORD_DTYPE.itemsize


def test_next_pow2_host(muse, golden):                  # xcorr_test.go:20-38
    for c in golden["next_pow2"]:
        assert muse.next_pow2(c["val"]) == c["expected"]
    assert muse.next_pow2(4096) == 4096 and muse.next_pow2(480) == 512


def test_no_gpu_fails_loudly(muse):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(muse.MuseError) as e:
        muse.Engine(0)
    assert e.value.status == muse.binding.MUSE_ERR_NO_DEVICE
    with pytest.raises(muse.MuseError) as e:          # the device set of a sharded Batch starts here: no device, no count
        muse.device_count()
    assert e.value.status == muse.binding.MUSE_ERR_NO_DEVICE
    # the host entry points that need no device still answer (merges of shard records run on the host)
    s, l, v, mean = muse.merge_records(np.zeros(0, dtype=muse.binding.RECORD_DTYPE), 5)
    assert len(s) == 0 and math.isnan(mean)
    x = np.zeros((2, 512))
    import ctypes
    lag, mv, nil = np.zeros(2, dtype=np.int32), np.zeros(2), np.zeros(2, dtype=np.int32)
    rc = muse.binding.load().muse_xcorr_batch(None, muse.binding.dptr(x), muse.binding.dptr(x), 2, 512, 512, 512, 1,
                                              muse.binding.i32ptr(lag), muse.binding.dptr(mv), muse.binding.i32ptr(nil), None)
    assert rc == muse.binding.MUSE_ERR_INVALID        # NULL context: an error, never a CPU computation


def test_product_never_imports_oracle():
    """The shipped package must not reference the oracle (test infrastructure)."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "go-muse_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".go")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"oracle_py|muse_oracle|libmuse_oracle|from oracle|import oracle", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_merge_records_matches_oracle_results(muse, oracle):
    """muse_merge_records == Results.Update/Fetch (results.go:55-87) on the
    union of shard candidates, including exact ties and the Go heap order."""
    rng = np.random.default_rng(3)
    M = 500
    mv = np.round(rng.uniform(-1.2, 1.2, M), 2)          # many exact ties
    lag = rng.integers(-20, 21, M).astype(np.int32)
    for top_n in (1, 7, 20, 600):
        for sign in (0, 1, -1):
            oi, ol, osc, omean = oracle.results(lag, mv, None, 0, False, 15, top_n, 0.25, sign)
            v = np.clip(mv, -1, 1)
            ok = (np.abs(lag) <= 15) & (np.abs(v) >= 0.25)
            if sign:
                ok &= (np.sign(v) == sign)
            rec = np.zeros(int(ok.sum()), dtype=muse.binding.RECORD_DTYPE)
            rec["series"] = np.nonzero(ok)[0]
            rec["group"] = rec["series"]
            rec["score"] = v[ok]
            rec["lag"] = lag[ok]
            rec = rec[rng.permutation(len(rec))]            # arrival order must not matter
            s, l, sc, mean = muse.merge_records(rec, top_n)
            assert s.tolist() == oi.tolist() and l.tolist() == ol.tolist()
            assert sc.tolist() == osc.tolist()
            assert (math.isnan(mean) and math.isnan(omean)) or mean == omean


def test_labels_series_group(muse):
    """labels_test.go / series_test.go / group_test.go behaviours on the path."""
    l = muse.NewLabels({"host": "h1", "graph": "g1"})
    assert l.Keys() == ["graph", "host"] and l.Len() == 2
    assert l.ID(l.Keys()) == "graph:g1,host:h1" and l.ID(["host"]) == "host:h1"
    assert l.Get("nope") == ("", False)
    s = muse.NewSeries([1, 2, 3], None)
    assert s.Labels().Keys() == [muse.DefaultLabel] and s.Length() == 3
    g = muse.NewGroup("targets")
    a = muse.NewSeries([1, 2, 3], muse.NewLabels({"graph": "a", "host": "1"}))
    b = muse.NewSeries([1, 2, 4], muse.NewLabels({"graph": "a", "host": "2"}))
    c = muse.NewSeries([1, 2, 5], muse.NewLabels({"graph": "b", "host": "1"}))
    g.Add(a, b, c)
    assert g.Length() == 3
    with pytest.raises(ValueError):
        g.Add(a)                                        # duplicate uid, group.go:38-41
    with pytest.raises(ValueError):
        g.Add(muse.NewSeries([1, 2], muse.NewLabels({"graph": "z"})))  # group.go:45-51
    lvs = g.indexLabelValues(["graph"])
    assert [x.ID(None) for x in lvs] == ["graph:a", "graph:b"]
    assert [x.UID() for x in g.FilterByLabelValues(lvs[0])] == [a.UID(), b.UID()]
    assert len(g.indexLabelValues(None)) == 3
    # the cached partition is keyed by the registry's VERSION, not its size: the registry is a public dict -- a series
    # removed and another added at the same count, or one replaced under its uid, must be seen (group.go:80-81 rebuilds every Run)
    assert [x.ID(None) for x in g.indexLabelValues(["graph"])] == ["graph:a", "graph:b"]
    del g.registry[c.UID()]
    d = muse.NewSeries([9, 9, 1], muse.NewLabels({"graph": "c", "host": "1"}))
    g.Add(d)
    assert len(g.registry) == 3
    assert [x.ID(None) for x in g.indexLabelValues(["graph"])] == ["graph:a", "graph:c"]
    assert g._group_ids().tolist() == [0, 0, 1]
    v0 = g.registry.version
    g.registry[b.UID()] = muse.NewSeries([7, 7, 8], muse.NewLabels({"graph": "a", "host": "2"}))   # replaced in place
    assert g.registry.version == v0 + 1 and not g._appended_only(v0, 3)
    g.Add(muse.NewSeries([0, 1, 0], muse.NewLabels({"graph": "c", "host": "2"})))
    assert g._appended_only(g.registry.version - 1, 3)


def test_results_heap_matches_oracle(muse, oracle):
    rng = np.random.default_rng(11)
    mv = np.round(rng.uniform(-1, 1, 200), 1)
    lag = rng.integers(-12, 13, 200).astype(np.int32)
    r = muse.NewResults(10, 9, 0.2, muse.SignFilter_ANY)
    for i in range(200):
        r.Update(muse.Score(muse.NewLabels({"i": str(i)}), int(lag[i]), float(mv[i])))
    got, mean = r.Fetch()
    oi, ol, osc, omean = oracle.results(lag, mv, None, 0, False, 10, 9, 0.2, 0)
    assert [int(s.Labels.Get("i")[0]) for s in got] == oi.tolist()
    assert [s.PercentScore for s in got] == osc.tolist() and mean == omean
    assert r.Fetch()[0] == [] and math.isnan(r.Fetch()[1])   # Fetch drains (results.go:75-87)


def test_cpp_host_mirror_builds_and_fails_loudly_without_gpu(muse):
    """go-muse_amd/host/muse.hpp (the compiled-language host layer over the C ABI)
    compiles with g++, links libmuse_hip.so, and its test program refuses to run
    without a GPU instead of falling back to anything."""
    import subprocess
    import torch
    exe = muse.build.build_host_test()
    assert os.path.exists(exe)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no HIP device" in r.stdout


def test_screen_bound_constants(muse):
    """docs/screen_error_bound.md: the library's bound on the fp32 screening pass's error is at least the sum of the
    per-stage constants of the standard fp32 FFT error analysis (re-derived here), for every FFT length."""
    import ctypes as C
    u = 2.0 ** -24
    stage = u + 4 * math.sqrt(2) * u              # eta per radix-2 stage with rounded butterfly constants
    prod4 = (4 * u + 3 * 2 * math.sqrt(2) * u) + 2 * math.sqrt(2) * u   # scaling by a product of <= 4 rounded entries
    single = u + 2 * math.sqrt(2) * u             # scaling by one rounded table entry
    for n in (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        t = int(math.log2(n))
        if n == 4096:
            scal = prod4 + single                 # pass 1: products, pass 2: LDS table
        elif n <= 2048:
            scal = 2 * prod4                      # Stockham, three passes
        elif n == 8192:
            scal = 3 * prod4                      # Stockham, four passes
        else:
            scal = single + prod4 + single        # four-step twiddle + a 4096-point row
        c_needed = (2 * (t * stage + scal) + single) / u
        for xmax in (0.05, 1.0, 7.5):
            Es = C.c_double(0)
            muse.binding.check(muse.binding.load().muse_test_screen_bound(n, xmax, C.byref(Es)))
            norm_z = 2.0 * math.sqrt(2.0 * n)     # ||z||_2 < sqrt 2 * 2 sqrt(N-1): scl * sigma in [1, 2)
            needed = c_needed * u * norm_z * xmax + 36 * u * math.sqrt(n / (n - 1.0)) + 1e-6 * (1 if n > 0 else 0)
            assert Es.value >= needed, (n, xmax, Es.value, needed)
            assert Es.value <= 2.0 * needed       # ... and not wastefully above it


def test_lane_relabelling_removes_the_modelled_lds_bank_conflicts():
    """tools/lds_bank_sim.py models the read / write lane groups of ds_read_b128 / ds_write_b128 (MI355X_MICROARCH.md, LDS) on the
    half-round transposes of xcorr_small.hip: with column = lane every read costs 8 LDS cycles and every transpose-A write 16; the
    kernel's lane -> column maps (column_of_lane<LOGN>, mirrored in the tool) bring them to the conflict-free 4 and 8, and the maps
    are bijections.  (Measured: SQ_LDS_BANK_CONFLICT 1.02e8 -> 0 at n = 1024, profiles/r02_small_final_counters.txt.)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lds_bank_sim", os.path.join(root, "tools", "lds_bank_sim.py"))
    sim = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sim)
    for logn in (10, 11, 13, 14):
        W = sim.shape(logn)[4]
        assert sorted(sim.column_of_lane(logn, l) for l in range(W)) == list(range(W))
        rd0, wa0, _ = sim.cycles(logn, identity=True)
        rd, wa, wbs = sim.cycles(logn)
        assert (rd0, wa0) == (8, 16.0) and (rd, wa) == (4, 8.0)
        assert wbs[0] <= 8.0
    # n = 512: column bit 4 must stay lane bit 4 (register trade of transpose B), so only the transpose-A stores are fixed
    assert sorted(sim.column_of_lane(9, l) for l in range(32)) == list(range(32))
    assert all(sim.column_of_lane(9, l + 16) == sim.column_of_lane(9, l) + 16 for l in range(16))
    assert sim.cycles(9, identity=True)[:2] == (8, 16.0) and sim.cycles(9)[:2] == (8, 8.0)
    # the kernel source carries the same maps: the XOR terms of column bit 3 and the parity of column bit 4
    src = open(os.path.join(root, "go-muse_amd", "csrc", "small_device.h")).read()   # (column_of_lane: shared by xcorr_small.hip and xcorr_real.hip)
    assert "((l >> 4) ^ (l >> 3) ^ (l >> 2)) & 1" in src and "(LOGN == 10 || LOGN == 14) ? (l >> 1) : l" in src
    assert "(l & ~8) | ((((l >> 3) ^ (l >> 2)) & 1) << 3)" in src


def test_merge_group_records_semantics(muse):
    """muse_merge_group_records on hand-made shard records (host only): the first shard with a member decides the NaN rule,
    ties keep the earlier shard, filters apply to the merged maximum, empty groups are skipped"""
    G, W = 5, 3
    rec = np.zeros((W, G), dtype=muse.binding.RECORD_DTYPE)
    rec["series"] = -1
    state = np.zeros((W, G), dtype=np.uint8)

    def put(s, g, series, score, lag, st=1):
        rec[s, g] = (series, score, lag, g)
        state[s, g] = st
    put(0, 0, 3, 0.5, 1)
    put(1, 0, 40, 0.9, 2)                # group 0: later shard strictly greater -> wins
    put(2, 0, 90, 0.9, 3)                # ... tie with shard 1: the earlier one stays
    state[0, 1] = 2                      # group 1: its first member (shard 0) scores NaN, no numeric member there
    put(1, 1, 41, 0.99, 0)               # ... a later shard's number must not replace it
    put(1, 2, 42, 0.7, 30)               # group 2: only on shard 1; lag filtered below
    put(0, 3, 5, 0.2, 0)
    put(2, 3, 91, 0.3, 0, st=2)          # group 3: a LATER shard whose own first member is NaN but has a number: counts
    # group 4: no member anywhere
    s, l, v, mean = muse.merge_group_records(rec, state, 10, 10, 0.0, 0)
    assert list(s) == [40, 91] and list(l) == [2, 0] and list(v) == [0.9, 0.3]
    assert mean == (0.3 + 0.9) / 2
    s, l, v, mean = muse.merge_group_records(rec, state, 40, 1, 0.0, 0)
    assert list(s) == [40]
    s, l, v, mean = muse.merge_group_records(rec, state, 40, 10, 0.5, 0)
    assert list(s) == [40, 42]
    # the per-group part alone (muse_merge_group_winners): what Batch.Run feeds through Results.Update, one Score per label group
    win, st = muse.merge_group_winners(rec, state)
    assert list(st) == [1, 2, 1, 1, 0]                       # winner / the group's score is NaN / winner / winner / no member
    assert [int(win[g]["series"]) for g in (0, 2, 3)] == [40, 42, 91]
    assert [float(win[g]["score"]) for g in (0, 2, 3)] == [0.9, 0.7, 0.3] and [int(win[g]["lag"]) for g in (0, 2, 3)] == [2, 30, 0]
    assert list(win["group"]) == [0, 1, 2, 3, 4]
    # feeding those winners through the mirror's Results in group order is exactly what the filtering merge returns
    for max_lag, top, thr in ((10, 10, 0.0), (40, 1, 0.0), (40, 10, 0.5)):
        r = muse.NewResults(max_lag, top, thr, muse.SignFilter_ANY)
        for g in range(G):
            if st[g] == 1:
                r.Update(muse.Score(muse.NewLabels({"g": str(g)}), int(win[g]["lag"]), float(win[g]["score"])))
        got, mean_fed = r.Fetch()
        s, l, v, mean = muse.merge_group_records(rec, state, max_lag, top, thr, 0)
        assert [x.PercentScore for x in got] == list(v) and [x.Lag for x in got] == list(l) and mean_fed == mean
    # one shard: the records pass through, the NaN-first state becomes "the group's score is NaN"
    w1, s1 = muse.merge_group_winners(rec[:1], state[:1])
    assert list(s1) == [1, 2, 0, 1, 0] and int(w1[0]["series"]) == 3 and int(w1[3]["series"]) == 5


def test_bench_names_the_workload_it_runs():
    """bench.py's config.workload: BASELINE configs[2] on one GPU, configs[3] when 8 x 1 M rows of 4096 samples run, and a -shaped
    label for every other size -- an N-rank line must not call itself configs[2]"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.workload_name(1_000_000, 4096, 1, 15, 20).startswith("configs[2]: 1 ref x 1000000 series on 1 GPU")
    assert bench.workload_name(20_001, 4096, 1, 15, 20).startswith("configs[2]-shaped")
    w8 = bench.workload_name(1_000_000, 4096, 8, 15, 20)
    assert w8.startswith("configs[3]: 1 ref x 8000000 series") and "over 8 GPUs" in w8 and "RCCL" in w8
    assert bench.workload_name(1_000_000, 4096, 4, 15, 20).startswith("configs[3]-shaped: 1 ref x 4000000 series")
    assert bench.workload_name(1_000_000, 512, 8, 15, 20).startswith("configs[3]-shaped")
    # the fingerprint bench.py refuses stale counters by covers the kernel sources, not the host side of the library
    sha = bench.csrc_sha()
    assert len(sha) == 16 and sha == bench.csrc_sha()
