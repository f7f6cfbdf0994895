// muse_ref_bench.cpp -- the reference's own `go test -bench` shapes (README.md:98-108), restated against the C++ host mirror
// (muse.hpp) over libmuse_hip.so: the loop bodies follow muse_test.go:144-215 (BenchmarkMuseRun, BenchmarkMuseRunLarge),
// muse_batch_test.go:104-162 (BenchmarkMuseBatchRun, BenchmarkMuseBatchRunLarge) and xcorr_test.go:310-348 (BenchmarkXCorr,
// BenchmarkXCorrWithX); inputs are uniform / Gaussian noise from a fixed generator (the reference draws them from Go's
// math/rand and siggen.Noise: same shapes and distributions, other streams).  Prints ONE JSON object on stdout: per
// benchmark the time per op here ("warm": the comparison data already resident in HBM where the reference's loop body
// keeps it in RAM; "cold": upload included) next to the README's figure (2018 MacBook Air, 1.6 GHz i5, 4 logical CPUs).
// bench.py runs it as a child process and puts the object on the bench line (`reference_bench_shapes`).
// Needs a gfx950 GPU (there is no CPU fallback).
#include <chrono>
#include <cstdio>
#include <cmath>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "muse.hpp"

using namespace muse;
using Clock = std::chrono::steady_clock;

static std::mt19937_64 rng(20200419);
static std::vector<double> noise(double amp, int n) // siggen.Noise(amp, n): Gaussian, sigma = amp
{
    std::normal_distribution<double> d(0.0, amp);
    std::vector<double> v((size_t)n);
    for (auto &x : v)
        x = d(rng);
    return v;
}
static std::vector<double> uniform(int n) // rand.Float64()
{
    std::uniform_real_distribution<double> d(0.0, 1.0);
    std::vector<double> v((size_t)n);
    for (auto &x : v)
        x = d(rng);
    return v;
}

// ns per op of f(): repeated until min_ms of wall time have passed (at least min_reps times), after one untimed call
template <class F> static double ns_per_op(F &&f, double min_ms = 300.0, int min_reps = 5, int *reps_out = nullptr)
{
    f();
    int reps = 0;
    const auto t0 = Clock::now();
    double el = 0.0;
    do {
        f();
        reps++;
        el = std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
    } while (el < min_ms || reps < min_reps);
    if (reps_out)
        *reps_out = reps;
    return el * 1e6 / reps;
}

static std::vector<std::vector<SeriesPtr>> small_comp() // muse_test.go:151-170
{
    auto S = [](std::vector<double> y, const char *g, const char *h) { return NewSeries(std::move(y), NewLabels({{"graph", g}, {"host", h}})); };
    return {{S({0.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4}, "graph1", "host1"), S({0.2, 0.1, 0.2, 0.1, 0.2, 0.1, 0.2, 0.1}, "graph1", "host2")},
            {S({0.0, 0.0, 0.0, 0.0, 0.2, 0.4, 0.5, 0.8}, "graph2", "host1")},
            {S({0.2, 0.1, 0.2, 0.1, 0.2, 0.1, 0.22, 0.1}, "graph3", "host1")},
            {S({0.0, 0.0, 0.0, 0.0, -0.2, -0.4, 0.0, -0.8}, "graph4", "host1")},
            {S({0.0, 0.0, 0.0, -0.2, -0.4, -0.6, 1.0, 0.0}, "graph5", "host1")}};
}

int main(int argc, char **argv)
{
    try {
        if (argc > 1) // (experiments: parked packing threads beside the caller; default 7)
            detail::Workers::MaxExtra = atoi(argv[1]);
        if (argc > 4) // (bytes gathered before a Group's first hand-over to the uploader)
            Group::StreamFirstBytes = (size_t)atol(argv[4]);
        if (argc > 3) // (0: rows go into the pinned window with memcpy instead of streaming stores)
            detail::StreamingStores = atoi(argv[3]) != 0;
        if (argc > 2) // (how long a packing thread spins for the next job before parking, us)
            detail::Workers::SpinMicros = atoi(argv[2]);
        auto eng = Engine::Default();
        std::string out = "{";
        char buf[512];
        auto emit = [&](const char *name, const char *body) {
            if (out.size() > 1)
                out += ", ";
            out += "\"";
            out += name;
            out += "\": {";
            out += body;
            out += "}";
        };
        const auto ref8 = NewSeries({0.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4}, NewLabels({{"graph", "graph1"}}));

        { // BenchmarkMuseRun, muse_test.go:144-180: five Muse.Run calls over six series of 8 samples per op
            auto comp = small_comp();
            auto g = New(ref8, NewResults(10, 20, 0, SignFilter_ANY));
            int reps = 0;
            const double ns = ns_per_op([&] { for (auto &c : comp) g->Run(c); }, 300.0, 5, &reps);
            snprintf(buf, sizeof(buf), "\"ns_per_op\": %.0f, \"reps\": %d, \"runs_per_op\": 5, \"series_per_op\": 6, \"N\": 8, \"readme_ns_per_op\": 5019, "
                                       "\"source\": \"muse_test.go:144-180, README.md:101\"", ns, reps);
            emit("BenchmarkMuseRun", buf);
        }
        { // BenchmarkMuseRunLarge, muse_test.go:182-215: 100 graphs x 50 hosts x 480 samples, one goroutine per graph per op
            const int n = 480, numGraphs = 100, numHosts = 50;
            auto ref = NewSeries(noise(0.1, n), nullptr);
            std::vector<std::vector<SeriesPtr>> comp((size_t)numGraphs);
            for (int i = 0; i < numGraphs; i++)
                for (int j = 0; j < numHosts; j++)
                    comp[(size_t)i].push_back(NewSeries(noise(0.1, n), NewLabels({{"graph", "graph" + std::to_string(i)}, {"host", "host" + std::to_string(j)}})));
            auto g = New(ref, NewResults(10, 20, 0, SignFilter_ANY));
            int reps = 0;
            const double ns_seq = ns_per_op([&] { for (auto &c : comp) g->Run(c); }, 300.0, 5, &reps);
            const int T = (int)std::max(2u, std::min(16u, std::thread::hardware_concurrency()));
            const double ns_par = ns_per_op([&] {
                std::vector<std::thread> th;
                for (int w = 0; w < T; w++)
                    th.emplace_back([&, w] { for (int i = w; i < numGraphs; i += T) g->Run(comp[(size_t)i]); });
                for (auto &t : th)
                    t.join();
            });
            snprintf(buf, sizeof(buf), "\"ns_per_op\": %.0f, \"ns_per_op_one_caller\": %.0f, \"caller_threads\": %d, \"reps\": %d, \"runs_per_op\": 100, "
                                       "\"series_per_op\": 5000, \"N\": 480, \"pairs_per_s\": %.0f, \"readme_ns_per_op\": 128044546, "
                                       "\"source\": \"muse_test.go:182-215, README.md:102\"",
                     std::min(ns_par, ns_seq), ns_seq, T, reps, 5000.0 / (std::min(ns_par, ns_seq) * 1e-9));
            emit("BenchmarkMuseRunLarge", buf);
        }
        { // BenchmarkMuseBatchRun, muse_batch_test.go:104-132: one Run(["graph"]) over six series of 8 samples per op
            auto grp = NewGroup("targets");
            for (auto &c : small_comp())
                grp->Add(c);
            auto g = NewBatch(ref8, grp, NewResults(10, 20, 0, SignFilter_ANY), 1);
            int reps = 0;
            const double ns = ns_per_op([&] { g->Run({"graph"}); }, 300.0, 5, &reps);
            snprintf(buf, sizeof(buf), "\"ns_per_op\": %.0f, \"reps\": %d, \"series_per_op\": 6, \"N\": 8, \"readme_ns_per_op\": null, "
                                       "\"source\": \"muse_batch_test.go:104-132 (not in the README table)\"", ns, reps);
            emit("BenchmarkMuseBatchRun", buf);
        }
        { // BenchmarkMuseBatchRunLarge, muse_batch_test.go:134-162: one Run(["graph"]) over 100 x 50 series of 480 samples per op
            const int n = 480;
            auto ref = NewSeries(noise(0.1, n), nullptr);
            std::vector<SeriesPtr> all;
            for (int i = 0; i < 100; i++)
                for (int j = 0; j < 50; j++)
                    all.push_back(NewSeries(noise(0.1, n), NewLabels({{"graph", "graph" + std::to_string(i)}, {"host", "host" + std::to_string(j)}})));
            auto grp = NewGroup("targets");
            for (auto &s : all)
                grp->Add({s}); // (one Add per Series, as the benchmark does)
            auto g = NewBatch(ref, grp, NewResults(10, 20, 0, SignFilter_ANY), 100);
            int reps = 0;
            const double ns = ns_per_op([&] { g->Run({"graph"}); }, 300.0, 5, &reps);
            const double ns_cold = ns_per_op([&] { // Group built and uploaded, reference transformed, one Run: everything but the Series
                auto g2 = NewGroup("targets");
                for (auto &s : all)
                    g2->Add({s});
                NewBatch(ref, g2, NewResults(10, 20, 0, SignFilter_ANY), 100)->Run({"graph"});
            }, 300.0, 3);
            snprintf(buf, sizeof(buf), "\"ns_per_op\": %.0f, \"ns_per_op_cold\": %.0f, \"reps\": %d, \"series_per_op\": 5000, \"N\": 480, \"pairs_per_s\": %.0f, "
                                       "\"readme_ns_per_op\": null, \"source\": \"muse_batch_test.go:134-162 (not in the README table)\"",
                     ns, ns_cold, reps, 5000.0 / (ns * 1e-9));
            emit("BenchmarkMuseBatchRunLarge", buf);
        }
        { // BenchmarkXCorrWithX, xcorr_test.go:330-348: one pair, 16 385 samples -> n = 32 768, the reference's spectrum precomputed
            const int n = 16385;
            auto x = uniform(n), y = uniform(n);
            auto ref = NewSeries(x, NewLabels({{"s", "x"}}));
            auto ys = NewSeries(y, NewLabels({{"s", "y"}}));
            auto m = New(ref, NewResults(32768, 1, 0, SignFilter_ANY)); // New computes X once, as the benchmark's set-up does
            int reps = 0;
            const double ns_cold = ns_per_op([&] { m->Run({ys}); }, 300.0, 5, &reps); // y crosses PCIe in every op
            auto grp = NewGroup("y");
            grp->Add({ys});
            auto b = NewBatch(ref, grp, NewResults(32768, 1, 0, SignFilter_ANY), 1);
            const double ns_warm = ns_per_op([&] { b->Run({}); }); // y resident in HBM
            snprintf(buf, sizeof(buf), "\"ns_per_op\": %.0f, \"ns_per_op_cold\": %.0f, \"reps\": %d, \"N\": 16385, \"fft_len\": 32768, \"readme_ns_per_op\": 4910405, "
                                       "\"source\": \"xcorr_test.go:330-348, README.md:108\"", ns_warm, ns_cold, reps);
            emit("BenchmarkXCorrWithX", buf);
        }
        { // BenchmarkXCorr, xcorr_test.go:322-328: one pair, both series transformed, not normalised
            const int n = 16385;
            auto x = uniform(n), y = uniform(n);
            int reps = 0;
            const double ns_cold = ns_per_op([&] { eng->XCorrBatch(x, y, 1, n, n, 32768, false, false); }, 300.0, 5, &reps);
            muse_group *gx = nullptr, *gy = nullptr;
            check(muse_group_upload(eng->handle(), x.data(), 1, n, n, &gx));
            check(muse_group_upload(eng->handle(), y.data(), 1, n, n, &gy));
            int32_t lag = 0, nil = 0;
            double mv = 0.0;
            const double ns_warm = ns_per_op([&] { check(muse_xcorr_groups(gx, gy, 32768, 0, &lag, &mv, &nil, nullptr)); });
            muse_group_free(gx);
            muse_group_free(gy);
            snprintf(buf, sizeof(buf), "\"ns_per_op\": %.0f, \"ns_per_op_cold\": %.0f, \"reps\": %d, \"N\": 16385, \"fft_len\": 32768, \"readme_ns_per_op\": 8228987, "
                                       "\"source\": \"xcorr_test.go:322-328, README.md:107\"", ns_warm, ns_cold, reps);
            emit("BenchmarkXCorr", buf);
        }
        out += "}";
        printf("%s\n", out.c_str());
    } catch (const Error &e) {
        fprintf(stderr, "muse::Error %d: %s\n", e.status, e.what());
        return 2;
    }
    return 0;
}
