// muse_host_test.cpp -- the reference's driver tests, restated against the C++
// host mirror (muse.hpp) over libmuse_hip.so.  Tables are the reference's own
// (muse_batch_test.go:9-102, muse_test.go:41-142); compareScores follows
// muse_test.go:11-39 (order, exact lag, score within 1e-3, labels).
// Exit code 0 = all passed.  Needs a gfx950 GPU (there is no CPU fallback).
#include <cstdio>
#include <cmath>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "muse.hpp"

using namespace muse;
static int failures = 0;
#define EXPECT(cond, ...)                                                                       \
    do {                                                                                        \
        if (!(cond)) {                                                                          \
            failures++;                                                                         \
            printf("FAIL %s:%d: ", __FILE__, __LINE__);                                         \
            printf(__VA_ARGS__);                                                                \
            printf("\n");                                                                       \
        }                                                                                       \
    } while (0)

struct Expect {
    LabelMap labels;
    std::vector<int> lags; // more than one: exact tie in exact arithmetic
    double score;
};

static void compareScores(const Scores &got, const std::vector<Expect> &exp, const char *name)
{
    EXPECT(got.size() == exp.size(), "%s: got %zu scores, expected %zu", name, got.size(), exp.size());
    for (size_t i = 0; i < got.size() && i < exp.size(); i++) {
        bool lag_ok = false;
        for (int l : exp[i].lags)
            lag_ok |= (l == got[i].Lag);
        EXPECT(lag_ok, "%s[%zu]: lag %d", name, i, got[i].Lag);
        EXPECT(std::fabs(got[i].PercentScore - exp[i].score) <= 1e-3, "%s[%zu]: score %.4f vs %.3f", name, i,
               got[i].PercentScore, exp[i].score);
        EXPECT(got[i].Labels->Map() == exp[i].labels, "%s[%zu]: labels %s", name, i, got[i].Labels->ID().c_str());
    }
}

static const std::vector<double> REF12 = {0, 0, 0, 0, 1, 2, 3, 3, 2, 1, 0, 0};

static void TestBatchRunSimple() // muse_batch_test.go:9-44
{
    auto ref = NewSeries(REF12, NewLabels({{"graph", "graph1"}}));
    auto g = NewGroup("targets");
    g->Add({NewSeries({0, 0, 0, 0, 2, 4, 6, 6, 4, 2, 0, 0}, NewLabels({{"graph", "perfectMatch"}})),
            NewSeries({0, 0, 0, 0, 2, 4, 6, 4, 2, 0, 0, 0}, NewLabels({{"graph", "slightlyLower"}})),
            NewSeries({0, 0, 0, 2, 4, 2, 0, 0, 0, 0, 0, 0}, NewLabels({{"graph", "evenLower"}})),
            NewSeries({0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 2, 0}, NewLabels({{"graph", "evenLowerShiftedAhead"}})),
            NewSeries(std::vector<double>(12, 3.0), NewLabels({{"graph", "zeros"}}))});
    auto b = NewBatch(ref, g, NewResults(10, 20, 0, SignFilter_ANY), 10);
    b->Run({"graph"});
    compareScores(b->Results_->Fetch().first,
                  {{{{"graph", "perfectMatch"}}, {0}, 1.000},
                   {{{"graph", "slightlyLower"}}, {0}, 0.929},
                   {{{"graph", "evenLowerShiftedAhead"}}, {-3, -2}, 0.754},
                   {{{"graph", "evenLower"}}, {2}, 0.733},
                   {{{"graph", "zeros"}}, {0}, 0.0}},
                  "TestBatchRunSimple");
}

static void TestBatchRunMultiDimensional() // muse_batch_test.go:46-82
{
    auto ref = NewSeries({0.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4}, NewLabels({{"graph", "graph1"}}));
    auto L = [](const char *g, const char *h) { return NewLabels({{"graph", g}, {"host", h}}); };
    auto grp = NewGroup("targets");
    grp->Add({NewSeries({0.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4}, L("graph1", "host1")),
              NewSeries({0.2, 0.1, 0.2, 0.1, 0.2, 0.1, 0.2, 0.1}, L("graph1", "host2")),
              NewSeries({0.0, 0.0, 0.0, 0.0, 0.2, 0.4, 0.4, 0.8}, L("graph2", "host1")),
              NewSeries({0.2, 0.1, 0.2, 0.1, 0.2, 0.1, 0.22, 0.1}, L("graph3", "host1")),
              NewSeries({0.0, 0.0, 0.0, 0.0, -0.2, -0.4, 0.0, -0.8}, L("graph4", "host1")),
              NewSeries({0.0, 0.0, 0.0, -0.2, -0.4, -0.6, 1.0, 0.0}, L("graph5", "host1"))});
    auto m = NewBatch(ref, grp, NewResults(10, 20, 0, SignFilter_ANY), 10);
    m->Run({"graph"});
    compareScores(m->Results_->Fetch().first,
                  {{{{"graph", "graph1"}, {"host", "host1"}}, {0}, 1.000},
                   {{{"graph", "graph2"}, {"host", "host1"}}, {0}, 0.976},
                   {{{"graph", "graph4"}, {"host", "host1"}}, {0}, 0.759},
                   {{{"graph", "graph5"}, {"host", "host1"}}, {2}, 0.719},
                   {{{"graph", "graph3"}, {"host", "host1"}}, {1}, 0.248}},
                  "TestBatchRunMultiDimensional");
}

static void TestBatchRunWithLargerGroup() // muse_batch_test.go:83-102
{
    auto ref = NewSeries({0, 1, 2, 3, 3, 2, 1, 0}, NewLabels({{"graph", "graph1"}}));
    auto g = NewGroup("targets");
    std::vector<double> longer(20, 0.0);
    g->Add({NewSeries(longer, NewLabels({{"graph", "longer"}}))});
    bool threw = false;
    try {
        NewBatch(ref, g, NewResults(10, 20, 0, SignFilter_ANY), 1);
    } catch (const Error &e) {
        threw = e.status == MUSE_ERR_LENGTH;
    }
    EXPECT(threw, "TestBatchRunWithLargerGroup: expected a length-mismatch error");
    threw = false;
    try { // muse_batch.go:39-41
        NewBatch(NewSeries({2, 2, 2, 2}), NewGroup("e"), NewResults(10, 20, 0, SignFilter_ANY), 1);
    } catch (const Error &e) {
        threw = e.status == MUSE_ERR_ZERO_STD;
    }
    EXPECT(threw, "NewBatch: expected Invalid input query on sigma == 0");
}

static std::vector<std::pair<std::vector<double>, const char *>> museComp()
{
    return {{{0, 0, 0, 0, 2, 4, 6, 6, 4, 2, 0, 0}, "perfectMatch"},
            {{0, 0, 0, 0, 2, 4, 6, 4, 2, 0, 0, 0}, "slightlyLower"},
            {{0, 0, 0, 2, 4, 2, 0, 0, 0, 0, 0, 0}, "evenLower"},
            {{0, 0, 0, 0, 0, 0, 0, 0, -2, -3, -2, 0}, "evenLowerShiftedAhead"},
            {std::vector<double>(12, 3.0), "zeros"}};
}

static void TestRunSimple() // muse_test.go:41-73
{
    auto g = New(NewSeries(REF12, NewLabels({{"graph", "graph1"}})), NewResults(10, 20, 0, SignFilter_ANY));
    for (auto &c : museComp())
        g->Run({NewSeries(c.first, NewLabels({{"graph", c.second}}))});
    compareScores(g->Results_->Fetch().first,
                  {{{{"graph", "perfectMatch"}}, {0}, 1.000},
                   {{{"graph", "slightlyLower"}}, {0}, 0.929},
                   {{{"graph", "evenLowerShiftedAhead"}}, {-3, -2}, -0.754},
                   {{{"graph", "evenLower"}}, {2}, 0.733},
                   {{{"graph", "zeros"}}, {0}, 0.0}},
                  "TestRunSimple");
}

static void TestRunSimpleSignFilter() // muse_test.go:75-104 (first pass) + NEG on fresh inputs
{
    auto g = New(NewSeries(REF12, NewLabels({{"graph", "graph1"}})), NewResults(10, 20, 0, SignFilter_POS));
    for (auto &c : museComp())
        g->Run({NewSeries(c.first, NewLabels({{"graph", c.second}}))});
    compareScores(g->Results_->Fetch().first,
                  {{{{"graph", "perfectMatch"}}, {0}, 1.000},
                   {{{"graph", "slightlyLower"}}, {0}, 0.929},
                   {{{"graph", "evenLower"}}, {2}, 0.733}},
                  "TestRunSimpleSignFilter/POS");
    g = New(NewSeries(REF12, NewLabels({{"graph", "graph1"}})), NewResults(10, 20, 0, SignFilter_NEG));
    for (auto &c : museComp())
        g->Run({NewSeries(c.first, NewLabels({{"graph", c.second}}))});
    compareScores(g->Results_->Fetch().first, {{{{"graph", "evenLowerShiftedAhead"}}, {-3, -2}, -0.754}},
                  "TestRunSimpleSignFilter/NEG");
}

static void TestRunNoInput() // muse_test.go:122-142
{
    auto g = New(NewSeries(REF12, NewLabels({{"graph", "graph1"}})), NewResults(10, 20, 0, SignFilter_ANY));
    g->Run({});
    auto r = g->Results_->Fetch();
    EXPECT(r.first.empty() && std::isnan(r.second), "TestRunNoInput");
    bool threw = false;
    try { // muse.go:68-70
        g->Run({NewSeries({1, 2, 3})});
    } catch (const Error &e) {
        threw = e.status == MUSE_ERR_LENGTH;
    }
    EXPECT(threw, "Muse.Run: expected a length error");
}

// many references against one group in one pass (muse_batch_run_many) == one Run per reference
static void TestRunManyEqualsRuns()
{
    const int N = 4096, M = 37, R = 3;
    auto mk = [&](int seed, int shift) {
        std::vector<double> v(N);
        unsigned long long h = 0x9E3779B97F4A7C15ull * (unsigned long long)(seed + 1);
        for (int i = 0; i < N; i++) {
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            v[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5 + ((i + shift) % N >= 2000 && (i + shift) % N < 2040 ? 3.0 : 0.0);
        }
        return v;
    };
    auto g = NewGroup("targets");
    for (int i = 0; i < M; i++)
        g->Add({NewSeries(mk(100 + i, 7 * i), NewLabels({{"graph", "g" + std::to_string(i / 3)}, {"host", "h" + std::to_string(i)}}))});
    std::vector<std::shared_ptr<Batch>> many, single;
    for (int r = 0; r < R; r++) {
        auto ref = NewSeries(mk(r, 11 * r), NewLabels({{"graph", "ref"}}));
        many.push_back(NewBatch(ref, g, NewResults(4096, 8, 0, SignFilter_ANY), 4));
        single.push_back(NewBatch(ref, g, NewResults(4096, 8, 0, SignFilter_ANY), 4));
    }
    Batch::RunMany(many, {"graph"});
    for (int r = 0; r < R; r++) {
        single[r]->Run({"graph"});
        auto a = many[r]->Results_->Fetch(), b = single[r]->Results_->Fetch();
        EXPECT(a.first.size() == b.first.size() && a.first.size() == 8, "RunMany[%d]: %zu vs %zu scores", r, a.first.size(), b.first.size());
        for (size_t i = 0; i < a.first.size() && i < b.first.size(); i++) {
            EXPECT(a.first[i].Lag == b.first[i].Lag && a.first[i].Labels->ID() == b.first[i].Labels->ID(), "RunMany[%d][%zu]: lag/labels", r, i);
            EXPECT(std::fabs(a.first[i].PercentScore - b.first[i].PercentScore) <= 1e-12, "RunMany[%d][%zu]: score", r, i);
        }
        EXPECT(std::fabs(a.second - b.second) <= 1e-12, "RunMany[%d]: mean", r);
    }
}

// muse_test.go:203-214: one *Muse driven from many goroutines (here: std::thread)
static void TestRunConcurrentCallers()
{
    const int N = 512, G = 16;
    auto mk = [&](int seed, int shift) {
        std::vector<double> v(N);
        unsigned long long h = 0x9E3779B97F4A7C15ull * (unsigned long long)(seed + 1);
        for (int i = 0; i < N; i++) {
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            v[i] = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5 + ((i + shift) % N >= 200 && (i + shift) % N < 230 ? 2.0 : 0.0);
        }
        return v;
    };
    auto ref = NewSeries(mk(0, 0), NewLabels({{"graph", "ref"}}));
    std::vector<std::vector<SeriesPtr>> groups(G);
    for (int g = 0; g < G; g++)
        for (int k = 0; k < 4; k++)
            groups[g].push_back(NewSeries(mk(100 * g + k + 1, 5 * g + k), NewLabels({{"graph", "g" + std::to_string(g)}, {"host", "h" + std::to_string(k)}})));
    auto seq = New(ref, NewResults(N, 8, 0, SignFilter_ANY));
    for (auto &gr : groups)
        seq->Run(gr);
    auto par = New(ref, NewResults(N, 8, 0, SignFilter_ANY));
    std::vector<std::thread> th;
    for (int w = 0; w < 4; w++)
        th.emplace_back([&, w] {
            for (int g = w; g < G; g += 4)
                par->Run(groups[g]);
        });
    for (auto &t : th)
        t.join();
    auto a = seq->Results_->Fetch(), b = par->Results_->Fetch();
    EXPECT(a.first.size() == b.first.size(), "concurrent Run: %zu vs %zu scores", a.first.size(), b.first.size());
    for (size_t i = 0; i < a.first.size() && i < b.first.size(); i++) {
        EXPECT(a.first[i].Lag == b.first[i].Lag && a.first[i].Labels->ID() == b.first[i].Labels->ID(), "concurrent Run[%zu]: lag/labels", i);
        EXPECT(std::fabs(a.first[i].PercentScore - b.first[i].PercentScore) <= 1e-12, "concurrent Run[%zu]: score", i);
    }
}

// First use from many threads at once: the process-wide engine (Engine::Default(), the Go shim's sync.Once getEngine) is
// created inside the first New / NewBatch -- here eight threads race for it (the pattern of muse_test.go:203-214 with a
// cold package).  Must run before any other test touches the engine.
static void TestFirstUseConcurrent()
{
    const int N = 256, T = 8;
    std::vector<double> base(N);
    for (int i = 0; i < N; i++)
        base[i] = std::sin(0.05 * i) + (i >= 100 && i < 120 ? 2.0 : 0.0) + 1e-3 * ((i * 2654435761u) % 97);
    std::vector<Scores> out(T);
    std::vector<int> status(T, 0);
    std::vector<std::string> what(T);
    std::vector<std::thread> th;
    for (int w = 0; w < T; w++)
        th.emplace_back([&, w] {
            try {
                auto ref = NewSeries(base, NewLabels({{"graph", "ref"}}));
                std::vector<double> y(N);
                for (int i = 0; i < N; i++)
                    y[i] = base[(i + 3) % N] * 2.0 + 1.0;
                auto m = New(ref, NewResults(N, 4, 0, SignFilter_ANY)); // <- first use of the engine, T threads at once
                m->Run({NewSeries(y, NewLabels({{"graph", "g"}, {"host", "h" + std::to_string(w)}}))});
                out[w] = m->Results_->Fetch().first;
            } catch (const Error &e) { // (an exception must not leave a thread: reported by the main thread below)
                status[w] = e.status;
                what[w] = e.what();
            }
        });
    for (auto &t : th)
        t.join();
    for (int w = 0; w < T; w++)
        if (status[w])
            throw Error(status[w], what[w]);
    for (int w = 0; w < T; w++) {
        EXPECT(out[w].size() == 1, "first use, thread %d: %zu scores", w, out[w].size());
        if (out[w].size() == 1 && out[0].size() == 1) {
            EXPECT(out[w][0].Lag == out[0][0].Lag, "first use, thread %d: lag %d vs %d", w, out[w][0].Lag, out[0][0].Lag);
            EXPECT(std::fabs(out[w][0].PercentScore - out[0][0].PercentScore) <= 1e-12, "first use, thread %d: score", w);
            EXPECT(out[w][0].PercentScore > 0.99, "first use, thread %d: score %g", w, out[w][0].PercentScore);
        }
    }
}


// ---- the Comparison group sharded over a list of devices behind the same NewBatch / Run (muse_batch.go:99-130).
// On a one-GPU box the list names device 0 several times: distinct contexts, distinct shards, the same code path.
static void sameScores(const std::pair<Scores, double> &a, const std::pair<Scores, double> &b, const char *name)
{
    EXPECT(a.first.size() == b.first.size(), "%s: %zu vs %zu scores", name, a.first.size(), b.first.size());
    for (size_t i = 0; i < a.first.size() && i < b.first.size(); i++) {
        EXPECT(a.first[i].Lag == b.first[i].Lag && a.first[i].Labels->ID() == b.first[i].Labels->ID(), "%s[%zu]: lag %d vs %d, labels %s vs %s", name,
               i, a.first[i].Lag, b.first[i].Lag, a.first[i].Labels->ID().c_str(), b.first[i].Labels->ID().c_str());
        EXPECT(a.first[i].PercentScore == b.first[i].PercentScore, "%s[%zu]: score %.17g vs %.17g", name, i, a.first[i].PercentScore,
               b.first[i].PercentScore);
    }
    EXPECT(a.second == b.second || (std::isnan(a.second) && std::isnan(b.second)), "%s: mean %.17g vs %.17g", name, a.second, b.second);
}

static void TestBatchRunShardedTables() // the reference's two Batch tables, three shards
{
    auto engines = Engine::List({0, 0, 0});
    {
        auto ref = NewSeries(REF12, NewLabels({{"graph", "graph1"}}));
        auto g = NewGroup("targets");
        g->Add({NewSeries({0, 0, 0, 0, 2, 4, 6, 6, 4, 2, 0, 0}, NewLabels({{"graph", "perfectMatch"}})),
                NewSeries({0, 0, 0, 0, 2, 4, 6, 4, 2, 0, 0, 0}, NewLabels({{"graph", "slightlyLower"}})),
                NewSeries({0, 0, 0, 2, 4, 2, 0, 0, 0, 0, 0, 0}, NewLabels({{"graph", "evenLower"}})),
                NewSeries({0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 2, 0}, NewLabels({{"graph", "evenLowerShiftedAhead"}})),
                NewSeries(std::vector<double>(12, 3.0), NewLabels({{"graph", "zeros"}}))});
        auto b = NewBatch(ref, g, NewResults(10, 20, 0, SignFilter_ANY), 10, engines);
        b->Run({"graph"});
        compareScores(b->Results_->Fetch().first,
                      {{{{"graph", "perfectMatch"}}, {0}, 1.000},
                       {{{"graph", "slightlyLower"}}, {0}, 0.929},
                       {{{"graph", "evenLowerShiftedAhead"}}, {-3, -2}, 0.754},
                       {{{"graph", "evenLower"}}, {2}, 0.733},
                       {{{"graph", "zeros"}}, {0}, 0.0}},
                      "TestBatchRunSimple over 3 shards");
    }
    {
        auto ref = NewSeries({0.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4}, NewLabels({{"graph", "graph1"}}));
        auto L = [](const char *g, const char *h) { return NewLabels({{"graph", g}, {"host", h}}); };
        auto grp = NewGroup("targets");
        // (graph1's two hosts sit in different shards of two rows each: the straddling-group merge)
        grp->Add({NewSeries({0.2, 0.1, 0.2, 0.1, 0.2, 0.1, 0.2, 0.1}, L("graph1", "host2")),
                  NewSeries({0.0, 0.0, 0.0, 0.0, 0.2, 0.4, 0.4, 0.8}, L("graph2", "host1")),
                  NewSeries({0.2, 0.1, 0.2, 0.1, 0.2, 0.1, 0.22, 0.1}, L("graph3", "host1")),
                  NewSeries({0.0, 0.0, 0.0, 0.0, 0.1, 0.2, 0.3, 0.4}, L("graph1", "host1")),
                  NewSeries({0.0, 0.0, 0.0, 0.0, -0.2, -0.4, 0.0, -0.8}, L("graph4", "host1")),
                  NewSeries({0.0, 0.0, 0.0, -0.2, -0.4, -0.6, 1.0, 0.0}, L("graph5", "host1"))});
        auto m = NewBatch(ref, grp, NewResults(10, 20, 0, SignFilter_ANY), 10, engines);
        m->Run({"graph"});
        compareScores(m->Results_->Fetch().first,
                      {{{{"graph", "graph1"}, {"host", "host1"}}, {0}, 1.000},
                       {{{"graph", "graph2"}, {"host", "host1"}}, {0}, 0.976},
                       {{{"graph", "graph4"}, {"host", "host1"}}, {0}, 0.759},
                       {{{"graph", "graph5"}, {"host", "host1"}}, {2}, 0.719},
                       {{{"graph", "graph3"}, {"host", "host1"}}, {1}, 0.248}},
                      "TestBatchRunMultiDimensional over 3 shards (straddling group)");
    }
}

// 1 003 series of 480 samples, 37 graphs x 28 hosts interleaved (every graph straddles every shard), a NaN series first in
// one graph and a constant one elsewhere: Run(nil), Run(["graph"]), Run(["host"]) and filtered Runs over 2, 3 and
// (devices visible) shards must return exactly what the one-device Run returns
static void TestBatchRunShardedEqualsUnsharded()
{
    const int N = 480, M = 1003;
    auto mk = [&](int seed, int shift, double amp) {
        std::vector<double> v(N);
        unsigned long long h = 0x9E3779B97F4A7C15ull * (unsigned long long)(seed + 1);
        for (int i = 0; i < N; i++) {
            h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
            v[i] = 0.2 * ((double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5) + ((i + shift) % N >= 200 && (i + shift) % N < 215 ? amp : 0.0);
        }
        return v;
    };
    auto ref = NewSeries(mk(0, 0, 1.5), NewLabels({{"graph", "ref"}}));
    auto build = [&]() {
        auto g = NewGroup("targets");
        for (int i = 0; i < M; i++) {
            auto v = mk(1 + i, (i * 7) % 60 - 30, (i % 3 == 0 ? -1.0 : 1.0) * (0.5 + (i % 11)));
            if (i == 5)
                v[100] = std::nan(""); // the FIRST member of graph g5: its group's score is NaN whatever follows (muse_batch.go:87)
            if (i == 8 + 37 * 20)
                v[7] = std::nan(""); // a LATER member of graph g8, in another shard: skipped, the group keeps its maximum
            if (i == 11)
                std::fill(v.begin(), v.end(), 2.5); // sigma == 0: score 0
            g->Add({NewSeries(v, NewLabels({{"graph", "g" + std::to_string(i % 37)}, {"host", "h" + std::to_string(i / 37)}}))});
        }
        return g;
    };
    std::vector<std::vector<int>> device_lists = {{0, 0}, {0, 0, 0}, {0, 0, 0, 0, 0}};
    struct Case { std::vector<std::string> by; int maxLag, topN; double thr; SignFilter sf; };
    const std::vector<Case> cases = {{{}, N, 20, 0.0, SignFilter_ANY},          {{"graph"}, N, 20, 0.0, SignFilter_ANY},
                                     {{"host"}, N, 7, 0.0, SignFilter_ANY},     {{"graph"}, 12, 10, 0.3, SignFilter_POS},
                                     {{"graph", "host"}, 25, 2000, 0.0, SignFilter_ANY}};
    for (auto &devs : device_lists) {
        auto engines = Engine::List(devs);
        auto g1 = build(), gs = build();
        for (size_t c = 0; c < cases.size(); c++) {
            auto one = NewBatch(ref, g1, NewResults(cases[c].maxLag, cases[c].topN, cases[c].thr, cases[c].sf), 8);
            auto sh = NewBatch(ref, gs, NewResults(cases[c].maxLag, cases[c].topN, cases[c].thr, cases[c].sf), 8, engines);
            one->Run(cases[c].by);
            sh->Run(cases[c].by);
            char name[96];
            snprintf(name, sizeof(name), "sharded x%zu, case %zu", devs.size(), c);
            auto a = one->Results_->Fetch(), b = sh->Results_->Fetch();
            EXPECT(!a.first.empty(), "%s: empty result", name);
            sameScores(a, b, name);
        }
        // series added after the cut extend the last shard
        auto extra = mk(9999, 3, 4.0);
        g1->Add({NewSeries(extra, NewLabels({{"graph", "g3"}, {"host", "late"}}))});
        gs->Add({NewSeries(extra, NewLabels({{"graph", "g3"}, {"host", "late"}}))});
        auto one = NewBatch(ref, g1, NewResults(N, 5, 0, SignFilter_ANY), 8);
        auto sh = NewBatch(ref, gs, NewResults(N, 5, 0, SignFilter_ANY), 8, engines);
        one->Run({"graph"});
        sh->Run({"graph"});
        sameScores(one->Results_->Fetch(), sh->Results_->Fetch(), "sharded, series added after the cut");
    }
}

// every device the process can see, one shard each (on a one-GPU box this is the single-engine path through the list API)
static void TestBatchRunAllVisibleDevices()
{
    int ndev = 0;
    for (; ndev < 64; ndev++) {
        muse_ctx *c = nullptr;
        if (muse_ctx_create(ndev, &c) != MUSE_OK)
            break;
        muse_ctx_destroy(c);
    }
    EXPECT(ndev >= 1, "no device");
    std::vector<int> devs;
    for (int d = 0; d < ndev; d++)
        devs.push_back(d);
    auto engines = Engine::List(devs);
    const int N = 4096, M = 300;
    auto g = NewGroup("targets");
    std::vector<double> refv(N);
    for (int i = 0; i < N; i++)
        refv[i] = std::sin(0.01 * i) + (i >= 2000 && i < 2010 ? 1.5 : 0.0);
    for (int i = 0; i < M; i++) {
        std::vector<double> v(N);
        for (int k = 0; k < N; k++)
            v[k] = refv[(k + 3 * i) % N] * (1.0 + 0.01 * i) + 1e-3 * ((k * 2654435761u + i) % 1013);
        g->Add({NewSeries(v, NewLabels({{"graph", "g" + std::to_string(i % 10)}, {"host", "h" + std::to_string(i)}}))});
    }
    auto ref = NewSeries(refv, NewLabels({{"graph", "ref"}}));
    auto one = NewBatch(ref, g, NewResults(N, 10, 0, SignFilter_ANY), 8);
    auto all = NewBatch(ref, g, NewResults(N, 10, 0, SignFilter_ANY), 8, engines);
    one->Run({"graph"});
    all->Run({"graph"});
    sameScores(one->Results_->Fetch(), all->Results_->Fetch(), "all visible devices");
    printf("sharded Run over %d visible device(s)\n", ndev);
}

static void TestXCorrBatchEqualsSinglePairs() // xcorr_test.go:86-202 exercises xCorr pair by pair; the batch entry must agree
{
    const int M = 37, lenx = 3000, leny = 4096, n = 4096;
    std::vector<double> X((size_t)M * lenx), Y((size_t)M * leny);
    for (int i = 0; i < M; i++) {
        for (int k = 0; k < lenx; k++)
            X[(size_t)i * lenx + k] = std::sin(0.013 * k * (1 + i % 5)) * (1.0 + i) + ((k * 2654435761u + i) % 997) * 1e-3;
        for (int k = 0; k < leny; k++)
            Y[(size_t)i * leny + k] = std::cos(0.011 * k + i) - 0.25 * ((k * 40503u + 7 * i) % 1009) * 1e-3;
    }
    for (int k = 0; k < lenx; k++)
        X[(size_t)5 * lenx + k] = 2.5; // sigma(x) == 0: (nil, 0, 0) when normalized
    auto eng = Engine::Default();
    for (int normalize = 0; normalize < 2; normalize++) {
        const auto r = eng->XCorrBatch(X, Y, M, lenx, leny, n, normalize != 0, true);
        EXPECT(r.n == 4096 && (int)r.lag.size() == M, "XCorrBatch sizes");
        for (int i = 0; i < M; i++) {
            std::vector<double> cc((size_t)n);
            int32_t lag = 0, nil = 0;
            double mv = 0.0;
            check(muse_xcorr(eng->handle(), X.data() + (size_t)i * lenx, lenx, Y.data() + (size_t)i * leny, leny, n, normalize, cc.data(),
                             &lag, &mv, &nil));
            EXPECT(nil == r.nil[i], "XCorrBatch nil flag");
            if (nil)
                continue;
            EXPECT(lag == r.lag[i], "XCorrBatch lag");
            EXPECT(std::fabs(mv - r.mv[i]) <= 1e-9 * std::fabs(mv) + 1e-12, "XCorrBatch value");
            double worst = 0.0, scale = 0.0;
            for (int k = 0; k < n; k++) {
                worst = std::max(worst, std::fabs(cc[k] - r.cc[(size_t)i * n + k]));
                scale = std::max(scale, std::fabs(cc[k]));
            }
            EXPECT(worst <= 1e-9 * scale + 1e-12, "XCorrBatch cc");
        }
        EXPECT(normalize == 0 || r.nil[5] == 1, "XCorrBatch: constant x is nil when normalized");
    }
}

// Rows streamed towards the device while the Group is still being built (Group::StreamOnAdd: background hand-overs of >= 4 MB, the
// group growing past its first capacity while copies are in flight, series added after a Run, a second Run) must give the Results
// of the same Group uploaded in one piece at Run.
static void TestStreamedGroupEqualsUploadedAtRun()
{
    std::mt19937_64 rng(77);
    std::normal_distribution<double> nd(0.0, 1.0);
    const int N = 700, M = 9000; // 50 MB: a dozen hand-overs, three growths
    std::vector<double> refv((size_t)N);
    for (auto &x : refv)
        x = nd(rng);
    auto ref = NewSeries(refv, NewLabels({{"graph", "ref"}}));
    std::vector<SeriesPtr> all;
    for (int i = 0; i < M; i++) {
        std::vector<double> y((size_t)N);
        for (auto &x : y)
            x = nd(rng);
        if (i % 7 == 0)
            for (int t = 0; t < N; t++)
                y[(size_t)t] += 1.5 * refv[(size_t)((t + 3 + i % 5) % N)];
        all.push_back(NewSeries(std::move(y), NewLabels({{"graph", "g" + std::to_string(i % 300)}, {"host", "h" + std::to_string(i)}})));
    }
    std::pair<Scores, double> out[2], out2[2];
    for (int streamed = 0; streamed < 2; streamed++) {
        Group::StreamOnAdd = streamed != 0;
        auto g = NewGroup("targets");
        for (int i = 0; i < M - 500; i++)
            g->Add({all[(size_t)i]});
        auto b = NewBatch(ref, g, NewResults(40, 25, 0, SignFilter_ANY), 4);
        b->Run({"graph"});
        out[streamed] = b->Results_->Fetch();
        for (int i = M - 500; i < M; i++) // series added after a Run extend the resident rows
            g->Add({all[(size_t)i]});
        b->Run({"graph"});
        out2[streamed] = b->Results_->Fetch();
    }
    Group::StreamOnAdd = true;
    EXPECT(out[0].first.size() == 25 && out2[0].first.size() == 25, "streamed group: 25 scores");
    sameScores(out[0], out[1], "streamed group vs uploaded at Run");
    sameScores(out2[0], out2[1], "streamed group vs uploaded at Run, after more Adds");
}

// detail::copy_row (streaming stores into the pinned window where alignment and length allow, memcpy otherwise) moves every
// length and alignment bit for bit, and leaves the bytes around the destination alone.
static void TestCopyRow()
{
    std::mt19937_64 rng(11);
    std::vector<double> src(2048 + 8), dst(2048 + 16);
    for (auto &x : src)
        x = (double)rng() / 3.0;
    bool ok = true;
    for (int streaming = 0; streaming < 2; streaming++) {
        detail::StreamingStores = streaming != 0;
        for (size_t n : {(size_t)1, (size_t)3, (size_t)4, (size_t)5, (size_t)8, (size_t)12, (size_t)15, (size_t)16, (size_t)17, (size_t)20, (size_t)36, (size_t)480, (size_t)700, (size_t)2048})
            for (size_t doff = 0; doff < 5; doff++)
                for (size_t soff = 0; soff < 3; soff++) {
                    std::fill(dst.begin(), dst.end(), -7.0);
                    detail::copy_row(dst.data() + 4 + doff, src.data() + soff, n);
                    detail::copy_rows_done();
                    ok = ok && memcmp(dst.data() + 4 + doff, src.data() + soff, n * sizeof(double)) == 0;
                    for (size_t i = 0; i < 4 + doff; i++)
                        ok = ok && dst[i] == -7.0;
                    for (size_t i = 4 + doff + n; i < dst.size(); i++)
                        ok = ok && dst[i] == -7.0;
                }
    }
    detail::StreamingStores = true;
    EXPECT(ok, "copy_row: every length and alignment, nothing outside the row");
}

// Series longer than 65 536 samples through the mirror (the reference has no length limit: xcorr.go:19-24, muse_batch.go:33-37):
// planted shifted copies of the reference come back with score ~1 and the planted lag, a constant series with score 0, through
// Batch.Run with label groups and through Muse.Run.
static void TestLongSeries()
{
    std::mt19937_64 rng(5);
    std::normal_distribution<double> nd(0.0, 1.0);
    const int N = 100000;
    std::vector<double> refv((size_t)N);
    for (auto &x : refv)
        x = nd(rng);
    auto ref = NewSeries(refv, NewLabels({{"graph", "ref"}}));
    auto shifted = [&](int by, double gain, double noise) {
        std::vector<double> y((size_t)N);
        for (int t = 0; t < N; t++)
            y[(size_t)t] = gain * refv[(size_t)(((t - by) % N + N) % N)] + noise * nd(rng);
        return y;
    };
    auto g = NewGroup("targets");
    g->Add({NewSeries(shifted(0, 3.0, 0.0), NewLabels({{"graph", "copy"}, {"host", "a"}})),
            NewSeries(shifted(7, -2.0, 0.1), NewLabels({{"graph", "late"}, {"host", "a"}})),
            NewSeries(shifted(-5, 1.0, 0.1), NewLabels({{"graph", "early"}, {"host", "a"}})),
            NewSeries(std::vector<double>((size_t)N, 4.0), NewLabels({{"graph", "flat"}, {"host", "a"}})),
            NewSeries(shifted(3, 1.0, 3.0), NewLabels({{"graph", "late"}, {"host", "b"}}))});
    auto b = NewBatch(ref, g, NewResults(10, 20, 0, SignFilter_ANY), 4);
    EXPECT(b->n == 131072, "long series: n = %d", b->n);
    b->Run({"graph"});
    auto got = b->Results_->Fetch().first;
    EXPECT(got.size() == 4, "long series: %zu label groups", got.size());
    if (got.size() == 4) {
        EXPECT(std::fabs(got[0].PercentScore - 1.0) < 1e-9 && got[0].Lag == 0, "long series: copy %.6f lag %d", got[0].PercentScore, got[0].Lag);
        // (a series that lags the reference by k samples has lag -k: xcorr.go:103)
        EXPECT(got[1].PercentScore > 0.98 && std::abs(got[1].Lag) <= 7 && got[1].Lag != 0, "long series: second %.4f lag %d", got[1].PercentScore, got[1].Lag);
        EXPECT(got[3].PercentScore == 0.0 && got[3].Lag == 0, "long series: flat %.4f", got[3].PercentScore);
    }
    auto m = New(ref, NewResults(10, 5, 0, SignFilter_ANY));
    m->Run({g->series()[1], g->series()[4]});
    auto one = m->Results_->Fetch().first;
    EXPECT(one.size() == 1 && one[0].PercentScore < -0.98 && one[0].Lag == -7, "long series Muse.Run: %zu scores, %.4f lag %d", one.size(),
           one.empty() ? 0.0 : one[0].PercentScore, one.empty() ? 0 : one[0].Lag);
}

int main()
{
    try {
        TestFirstUseConcurrent();
        TestBatchRunSimple();
        TestBatchRunMultiDimensional();
        TestBatchRunWithLargerGroup();
        TestRunSimple();
        TestRunSimpleSignFilter();
        TestRunNoInput();
        TestRunManyEqualsRuns();
        TestRunConcurrentCallers();
        TestBatchRunShardedTables();
        TestBatchRunShardedEqualsUnsharded();
        TestBatchRunAllVisibleDevices();
        // the same Batch tests through the pre-selecting paths a Run over more than EXACT_FEED_MAX_GROUPS label groups takes
        // (muse_batch_run / _run_shard + muse_merge_records, _run_groups + muse_merge_group_records)
        Batch::EXACT_FEED_MAX_GROUPS = 0;
        TestBatchRunSimple();
        TestBatchRunMultiDimensional();
        TestBatchRunShardedTables();
        TestBatchRunShardedEqualsUnsharded();
        TestBatchRunAllVisibleDevices();
        Batch::EXACT_FEED_MAX_GROUPS = 65536;
        TestXCorrBatchEqualsSinglePairs();
        TestStreamedGroupEqualsUploadedAtRun();
        TestCopyRow();
        TestLongSeries();
    } catch (const Error &e) {
        printf("muse::Error %d: %s\n", e.status, e.what());
        return 2;
    }
    printf(failures ? "%d FAILURES\n" : "all host tests passed\n", failures);
    return failures ? 1 : 0;
}
