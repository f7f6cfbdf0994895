// muse_cold_phases.cpp -- where a COLD BenchmarkMuseBatchRunLarge op goes (muse_batch_test.go:134-162 with the Group
// build, the upload and NewBatch inside the timed region: muse_ref_bench.cpp's `ns_per_op_cold`).  The op is replayed
// step by step against the C ABI with a wall clock around every step; the same steps through the host mirror
// (muse.hpp) are timed as a whole next to it.  Prints a table (profiles/r06_cold_path.txt).  Needs a gfx950 GPU.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "muse.hpp"

using namespace muse;
using Clock = std::chrono::steady_clock;

static double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }

struct Phases {
    std::vector<std::pair<std::string, double>> rows;
    Clock::time_point t;
    void start() { t = Clock::now(); }
    void lap(const char *name)
    {
        const double us = us_since(t);
        for (auto &r : rows)
            if (r.first == name) {
                r.second += us;
                t = Clock::now();
                return;
            }
        rows.emplace_back(name, us);
        t = Clock::now();
    }
};

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    const int numGraphs = argc > 2 ? atoi(argv[2]) : 100, numHosts = argc > 3 ? atoi(argv[3]) : 50, n = argc > 4 ? atoi(argv[4]) : 480;
    try {
        std::mt19937_64 rng(20200419);
        std::normal_distribution<double> d(0.0, 0.1);
        auto noise = [&](int k) {
            std::vector<double> v((size_t)k);
            for (auto &x : v)
                x = d(rng);
            return v;
        };
        auto eng = Engine::Default();
        muse_ctx *ctx = eng->handle();
        auto ref = NewSeries(noise(n), nullptr);
        std::vector<SeriesPtr> all;
        for (int i = 0; i < numGraphs; i++)
            for (int j = 0; j < numHosts; j++)
                all.push_back(NewSeries(noise(n), NewLabels({{"graph", "graph" + std::to_string(i)}, {"host", "host" + std::to_string(j)}})));
        const int64_t M = (int64_t)all.size();

        // ---- the whole op through the mirror
        auto whole = [&] {
            auto g2 = NewGroup("targets");
            for (auto &s : all)
                g2->Add({s});
            NewBatch(ref, g2, NewResults(10, 20, 0, SignFilter_ANY), 100)->Run({"graph"});
        };
        whole();
        whole();
        double whole_us = 0.0, whole_min = 1e30;
        for (int r = 0; r < reps; r++) {
            const auto t0 = Clock::now();
            whole();
            const double us = us_since(t0);
            whole_us += us;
            whole_min = std::min(whole_min, us);
        }

        // ---- the same op, step by step against the C ABI
        Phases ph;
        for (int r = -2; r < reps; r++) {
            if (r == 0)
                ph.rows.clear();
            ph.start();
            auto g2 = NewGroup("targets");
            for (auto &s : all)
                g2->Add({s});
            ph.lap("host: NewGroup + Add x M (label ids, registry)");
            std::vector<int32_t> gid;
            std::vector<LabelsPtr> lvs;
            muse_group *probe = nullptr;
            muse_batch *tmpl = nullptr;
            check(muse_group_create(ctx, 0, n, &probe));
            check(muse_batch_create(ctx, probe, ref->Values().data(), n, &tmpl));
            ph.lap("NewBatch: probe group + muse_batch_create (reference validated and transformed once)");
            const std::function<void()> side = [&] { lvs = g2->indexLabelValues({"graph"}, &gid); };
            muse_group *dg = g2->device(eng, &side);
            ph.lap("Group.device: muse_group_create + rows packed into pinned windows by the host threads + commits (H2D beside the packing); indexLabelValues on the calling thread meanwhile");
            muse_batch *b = nullptr;
            check(muse_batch_create_like(tmpl, dg, &b));
            ph.lap("muse_batch_create_like (shares the spectrum)");
            const int32_t G = (int32_t)lvs.size();
            std::vector<muse_record> recs((size_t)G), win((size_t)G);
            std::vector<uint8_t> state((size_t)G), st((size_t)G);
            check(muse_batch_run_groups(b, gid.data(), G, 0, 1, recs.data(), state.data()));
            ph.lap("muse_batch_run_groups (first Run: workspaces, kernel, group max, records back)");
            check(muse_merge_group_winners(recs.data(), state.data(), 1, G, win.data(), st.data()));
            auto res = NewResults(10, 20, 0, SignFilter_ANY);
            for (int32_t g = 0; g < G; g++)
                if (st[(size_t)g] == 1)
                    res->Update(Score{all[(size_t)win[(size_t)g].series]->Labels(), win[(size_t)g].lag, win[(size_t)g].score});
            ph.lap("host: merge + Results.Update x G");
            check(muse_batch_run_groups(b, gid.data(), G, 0, 1, recs.data(), state.data()));
            ph.lap("(a second muse_batch_run_groups: the warm Run, for scale)");
            muse_batch_free(b);
            muse_batch_free(tmpl);
            muse_group_free(probe);
            ph.lap("muse_batch_free x 2 + probe group");
            g2.reset();
            ph.lap("host: Group destructor (muse_group_free inside)");
        }
        printf("cold BenchmarkMuseBatchRunLarge: %d graphs x %d hosts x %d samples (%.1f MB), %d reps\n", numGraphs, numHosts, n,
               (double)M * n * 8 / 1e6, reps);
        printf("whole op through the host mirror: mean %.1f us, min %.1f us (%d packing threads)\n", whole_us / reps, whole_min, detail::Workers::get().width());
        double sum = 0.0;
        for (auto &r : ph.rows)
            if (r.first[0] != '(')
                sum += r.second / reps;
        for (auto &r : ph.rows)
            printf("  %9.1f us  %5.1f %%  %s\n", r.second / reps, r.first[0] == '(' ? 0.0 : 100.0 * r.second / reps / sum, r.first.c_str());
        printf("  %9.1f us  sum of the steps (without the parenthesised one)\n", sum);
    } catch (const Error &e) {
        fprintf(stderr, "muse::Error %d: %s\n", e.status, e.what());
        return 2;
    }
    return 0;
}
