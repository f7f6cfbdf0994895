// muse.hpp -- C++17 host-side mirror of go-muse's exported API for the
// Batch.Run / Muse.Run path, layered on the C ABI (include/muse_hip.h).
//
// The reference is compiled Go and no Go toolchain exists in the build image,
// so this header is the compiled-language host layer a Go maintainer's cgo shim
// (go-muse_amd/go/) would mirror one to one.  Names, argument
// meaning and error behaviour follow /root/reference:
//   Labels / NewLabels          labels.go:12-73
//   Series / NewSeries          series.go:8-42
//   Group / NewGroup            group.go:7-104
//   Score, Results / NewResults scores.go:11-15, results.go:11-87
//   Batch / NewBatch, Run       muse_batch.go:13-130
//   Muse / New, Run             muse.go:15-92
// Go's (value, error) returns become exceptions of type muse::Error carrying the
// muse_status.  All arithmetic runs on the GPU; this layer is label bookkeeping.
// Unlike the reference (xcorr.go:86,93) caller data is never mutated.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "muse_hip.h"

namespace muse {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &m) : std::runtime_error(m), status(s) {}
};
inline void check(int status)
{
    if (status != MUSE_OK)
        throw Error(status, muse_last_error());
}

constexpr const char *DefaultLabel = "uid"; // labels.go:7
using LabelMap = std::map<std::string, std::string>;

// ------------------------------------------------------------- labels.go
class Labels {
public:
    explicit Labels(LabelMap m) : labels_(std::move(m))
    {
        for (auto &kv : labels_)
            keys_.push_back(kv.first); // std::map iterates in sorted key order (labels.go:29)
    }
    int Len() const { return (int)labels_.size(); }
    const std::vector<std::string> &Keys() const { return keys_; }
    bool Get(const std::string &key, std::string *value) const // labels.go:44-49
    {
        auto it = labels_.find(key);
        if (it == labels_.end())
            return false;
        if (value)
            *value = it->second;
        return true;
    }
    std::string ID(std::vector<std::string> labels = {}) const // labels.go:54-73
    {
        if (labels.empty())
            labels = keys_;
        else
            std::sort(labels.begin(), labels.end());
        std::string out;
        for (auto &l : labels) {
            auto it = labels_.find(l);
            if (it != labels_.end()) {
                if (!out.empty())
                    out += ",";
                out += l + ":" + it->second;
            }
        }
        return out;
    }
    const LabelMap &Map() const { return labels_; }

private:
    LabelMap labels_;
    std::vector<std::string> keys_;
};
using LabelsPtr = std::shared_ptr<const Labels>;
inline LabelsPtr NewLabels(LabelMap m) { return std::make_shared<Labels>(std::move(m)); }

// ------------------------------------------------------------- series.go
class Series {
public:
    Series(std::vector<double> y, LabelsPtr labels) : y_(std::move(y)), labels_(std::move(labels))
    {
        if (!labels_ || labels_->Len() == 0) { // series.go:16-18: default uid label
            static std::mt19937_64 rng{std::random_device{}()};
            char buf[40];
            snprintf(buf, sizeof(buf), "%016llx%016llx", (unsigned long long)rng(), (unsigned long long)rng());
            labels_ = NewLabels({{DefaultLabel, buf}});
        }
    }
    int Length() const { return (int)y_.size(); }
    const std::vector<double> &Values() const { return y_; }
    LabelsPtr Labels() const { return labels_; }
    std::string UID() const { return labels_->ID(); } // series.go:40-42

private:
    std::vector<double> y_;
    LabelsPtr labels_;
};
using SeriesPtr = std::shared_ptr<Series>;
inline SeriesPtr NewSeries(std::vector<double> y, LabelsPtr labels = nullptr)
{
    return std::make_shared<Series>(std::move(y), std::move(labels));
}

// ------------------------------------------------------ engine (muse_ctx)
class Engine {
public:
    explicit Engine(int device = 0) { check(muse_ctx_create(device, &ctx_)); }
    ~Engine() { muse_ctx_destroy(ctx_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    muse_ctx *handle() const { return ctx_; }
    static std::shared_ptr<Engine> Default()
    {
        static std::shared_ptr<Engine> e = std::make_shared<Engine>(0);
        return e;
    }

private:
    muse_ctx *ctx_ = nullptr;
};

// -------------------------------------------------------------- group.go
class Group {
public:
    explicit Group(std::string name) : Name(std::move(name)) {}
    ~Group()
    {
        if (dev_)
            muse_group_free(dev_);
    }
    std::string Name;
    int Length() const { return n_; }
    // group.go:31-56; errors come back as muse::Error(MUSE_ERR_INVALID / MUSE_ERR_LENGTH)
    void Add(const std::vector<SeriesPtr> &series)
    {
        for (auto &s : series) {
            if (s->Labels()->Keys().empty())
                throw Error(MUSE_ERR_INVALID, "Invalid Series with no labels");
            const std::string uid = s->UID();
            if (registry_.count(uid))
                throw Error(MUSE_ERR_INVALID,
                            "Series with label:values, " + uid + ", already exists within group, " + Name);
            if (order_.empty())
                n_ = s->Length();
            else if (s->Length() != n_)
                throw Error(MUSE_ERR_LENGTH, "Timeseries has length " + std::to_string(s->Length()) +
                                                 ", but current group has length " + std::to_string(n_));
            registry_[uid] = order_.size();
            order_.push_back(s);
        }
    }
    std::vector<SeriesPtr> FilterByLabelValues(const Labels &labels) const // group.go:60-71
    {
        std::vector<SeriesPtr> out;
        auto it = index_.find(labels.ID());
        if (it != index_.end())
            for (size_t i : it->second)
                out.push_back(order_[i]);
        return out;
    }
    // group.go:76-104.  Partition order = first appearance in insertion order (Go
    // iterates a map here, so its order is unspecified).  group_id_out[i] receives
    // the partition index of series i.
    std::vector<LabelsPtr> indexLabelValues(std::vector<std::string> groupByLabels,
                                            std::vector<int32_t> *group_id_out = nullptr)
    {
        std::vector<LabelsPtr> distinct;
        index_.clear();
        std::unordered_map<std::string, int32_t> gid;
        if (group_id_out)
            group_id_out->assign(order_.size(), 0);
        for (size_t i = 0; i < order_.size(); i++) {
            auto &s = order_[i];
            std::string guid;
            if (!groupByLabels.empty()) {
                guid = s->Labels()->ID(groupByLabels);
            } else {
                guid = s->UID();
                groupByLabels = s->Labels()->Keys(); // group.go:88 (first series' keys)
            }
            auto it = gid.find(guid);
            if (it == gid.end()) {
                LabelMap lv;
                for (auto &name : groupByLabels) {
                    std::string v;
                    if (s->Labels()->Get(name, &v))
                        lv[name] = v;
                }
                distinct.push_back(NewLabels(lv));
                it = gid.emplace(guid, (int32_t)gid.size()).first;
            }
            index_[guid].push_back(i);
            if (group_id_out)
                (*group_id_out)[i] = it->second;
        }
        return distinct;
    }
    const std::vector<SeriesPtr> &series() const { return order_; }

    // device residency: rows are uploaded once and appended to (muse_group_append)
    muse_group *device(const std::shared_ptr<Engine> &eng)
    {
        if (!dev_ || eng_ != eng) {
            if (dev_)
                muse_group_free(dev_);
            dev_ = nullptr;
            eng_ = eng;
            uploaded_ = 0;
            check(muse_group_create(eng->handle(), (int64_t)order_.size(), n_ > 0 ? n_ : 1, &dev_));
        }
        for (; uploaded_ < order_.size(); uploaded_++)
            check(muse_group_append(dev_, order_[uploaded_]->Values().data(), 1, n_));
        return dev_;
    }

private:
    int n_ = 0;
    std::vector<SeriesPtr> order_;
    std::unordered_map<std::string, size_t> registry_;
    std::unordered_map<std::string, std::vector<size_t>> index_;
    std::shared_ptr<Engine> eng_;
    muse_group *dev_ = nullptr;
    size_t uploaded_ = 0;
};
using GroupPtr = std::shared_ptr<Group>;
inline GroupPtr NewGroup(std::string name) { return std::make_shared<Group>(std::move(name)); }

// -------------------------------------------------- scores.go / results.go
struct Score {
    LabelsPtr Labels;
    int Lag = 0;
    double PercentScore = 0.0;
};
using Scores = std::vector<Score>;

enum SignFilter { SignFilter_NEG = -1, SignFilter_ANY = 0, SignFilter_POS = 1 }; // results.go:22-26

class Results {
public:
    Results(int maxLag, int topN, double threshold, SignFilter sf)
        : MaxLag(maxLag), TopN(topN), Threshold(threshold), Filter(sf)
    {
    }
    int MaxLag, TopN;
    double Threshold;
    SignFilter Filter;

    bool passed(const Score &s) const // results.go:46-52
    {
        return std::fabs((double)s.Lag) <= (double)MaxLag && std::fabs(s.PercentScore) >= Threshold &&
               (Filter == SignFilter_ANY || (s.PercentScore > 0 && Filter == SignFilter_POS) ||
                (s.PercentScore < 0 && Filter == SignFilter_NEG));
    }
    void Update(const Score &s) // results.go:55-72 (Go container/heap on |score|)
    {
        if (!s.Labels)
            return;
        std::lock_guard<std::mutex> lock(mu_); // results.go:12,60: Muse.Run is called from many goroutines
        if (!passed(s))
            return;
        if ((int)h_.size() == TopN) {
            if (TopN > 0 && std::fabs(s.PercentScore) > std::fabs(h_[0].PercentScore)) {
                pop();
                push(s);
            }
        } else {
            push(s);
        }
    }
    // results.go:75-87: descending |score| + mean |score| (NaN when empty); drains the heap
    std::pair<Scores, double> Fetch()
    {
        std::lock_guard<std::mutex> lock(mu_);
        const size_t num = h_.size();
        Scores out(num);
        double sum = 0.0;
        for (size_t i = num; i-- > 0;) {
            out[i] = pop();
            sum += std::fabs(out[i].PercentScore);
        }
        return {out, num ? sum / (double)num : std::numeric_limits<double>::quiet_NaN()};
    }

private:
    std::mutex mu_;
    std::vector<Score> h_;
    bool less(size_t i, size_t j) const { return std::fabs(h_[i].PercentScore) < std::fabs(h_[j].PercentScore); }
    void up(size_t j)
    {
        while (j > 0) {
            size_t i = (j - 1) / 2;
            if (!less(j, i))
                break;
            std::swap(h_[i], h_[j]);
            j = i;
        }
    }
    void down(size_t i, size_t n)
    {
        for (;;) {
            size_t j1 = 2 * i + 1;
            if (j1 >= n)
                break;
            size_t j = j1;
            if (j1 + 1 < n && less(j1 + 1, j1))
                j = j1 + 1;
            if (!less(j, i))
                break;
            std::swap(h_[i], h_[j]);
            i = j;
        }
    }
    void push(const Score &s)
    {
        h_.push_back(s);
        up(h_.size() - 1);
    }
    Score pop()
    {
        size_t n = h_.size() - 1;
        std::swap(h_[0], h_[n]);
        down(0, n);
        Score s = h_.back();
        h_.pop_back();
        return s;
    }
};
using ResultsPtr = std::shared_ptr<Results>;
inline ResultsPtr NewResults(int maxLag, int topN, double threshold, SignFilter sf)
{
    return std::make_shared<Results>(maxLag, topN, threshold, sf);
}

// ---------------------------------------------------------- muse_batch.go
class Batch {
public:
    // NewBatch (muse_batch.go:23-52): length check against every series of the group,
    // reference spectrum computed right away ("Invalid input query" on sigma == 0).
    Batch(SeriesPtr ref, GroupPtr comp, ResultsPtr results, int cc, std::shared_ptr<Engine> eng = Engine::Default())
        : Comparison(std::move(comp)), Results_(std::move(results)), Concurrency(cc < 1 ? 1 : cc), eng_(std::move(eng)),
          ref_(ref->Values())
    {
        for (auto &s : Comparison->series())
            if (s->Length() != ref->Length())
                throw Error(MUSE_ERR_LENGTH, s->UID() + " from comparison group series does not have the same "
                                                        "length as the reference");
        n = (int)muse_next_pow2((double)ref->Length());
        muse_group *probe = nullptr; // validates the reference even when the group is empty
        check(muse_group_create(eng_->handle(), 0, ref->Length() > 0 ? ref->Length() : 1, &probe));
        muse_batch *b = nullptr;
        int rc = muse_batch_create(eng_->handle(), probe, ref_.data(), (int32_t)ref_.size(), &b);
        std::string msg = rc ? muse_last_error() : "";
        muse_batch_free(b);
        muse_group_free(probe);
        if (rc)
            throw Error(rc, msg);
    }
    ~Batch() { muse_batch_free(batch_); }
    int n = 0;
    GroupPtr Comparison;
    ResultsPtr Results_;
    int Concurrency; // kept for API compatibility: the GPU is the fan-out

    // Run (muse_batch.go:99-130).  Always "returns nil": errors can only be device failures.
    void Run(const std::vector<std::string> &groupByLabels)
    {
        std::vector<int32_t> gid;
        auto lvs = Comparison->indexLabelValues(groupByLabels, &gid);
        if (lvs.empty())
            return;
        muse_group *dg = Comparison->device(eng_);
        ensure(dg);
        const int cap = std::max(Results_->TopN, 1);
        std::vector<int64_t> idx(cap);
        std::vector<int32_t> lag(cap);
        std::vector<double> score(cap);
        int32_t cnt = 0;
        double mean = 0;
        check(muse_batch_run(batch_, gid.data(), (int32_t)lvs.size(), Results_->MaxLag, Results_->TopN,
                             Results_->Threshold, (int32_t)Results_->Filter, 1, idx.data(), lag.data(), score.data(),
                             &cnt, &mean));
        // feed Results in group order, as the ordered drain does (muse_batch.go:124-128)
        std::vector<int> order(cnt);
        for (int i = 0; i < cnt; i++)
            order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gid[idx[a]] < gid[idx[b]]; });
        for (int k : order)
            Results_->Update(Score{Comparison->series()[idx[k]]->Labels(), lag[k], score[k]});
    }

    // The README use case (README.md:10-13) runs many references against one Group: the same as
    // calling Run on every batch, but the resident rows are read and transformed once for all of
    // them (muse_batch_run_many).  The batches must share the Comparison group and the Results settings
    // that select (MaxLag, TopN, Threshold, Filter are taken from each batch's own Results only if
    // they all agree; otherwise the batches are run one by one).
    static void RunMany(const std::vector<std::shared_ptr<Batch>> &batches, const std::vector<std::string> &groupByLabels)
    {
        if (batches.empty())
            return;
        bool same = true;
        for (auto &b : batches) {
            same &= b->Comparison == batches[0]->Comparison && b->eng_ == batches[0]->eng_;
            same &= b->Results_->MaxLag == batches[0]->Results_->MaxLag && b->Results_->TopN == batches[0]->Results_->TopN;
            same &= b->Results_->Threshold == batches[0]->Results_->Threshold && b->Results_->Filter == batches[0]->Results_->Filter;
        }
        if (!same) {
            for (auto &b : batches)
                b->Run(groupByLabels);
            return;
        }
        Batch &b0 = *batches[0];
        std::vector<int32_t> gid;
        auto lvs = b0.Comparison->indexLabelValues(groupByLabels, &gid);
        if (lvs.empty())
            return;
        muse_group *dg = b0.Comparison->device(b0.eng_);
        std::vector<muse_batch *> hs;
        for (auto &b : batches) {
            b->ensure(dg);
            hs.push_back(b->batch_);
        }
        const int R = (int)hs.size(), top = b0.Results_->TopN, cap = std::max(top, 1);
        std::vector<int64_t> idx((size_t)R * cap);
        std::vector<int32_t> lag((size_t)R * cap), cnt((size_t)R);
        std::vector<double> score((size_t)R * cap), mean((size_t)R);
        check(muse_batch_run_many(hs.data(), R, gid.data(), (int32_t)lvs.size(), b0.Results_->MaxLag, top,
                                  b0.Results_->Threshold, (int32_t)b0.Results_->Filter, 1, idx.data(), lag.data(),
                                  score.data(), cnt.data(), mean.data()));
        for (int r = 0; r < R; r++) {
            const size_t o = (size_t)r * std::max(top, 0);
            std::vector<int> order(cnt[r]);
            for (int i = 0; i < cnt[r]; i++)
                order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gid[idx[o + a]] < gid[idx[o + b]]; });
            for (int k : order)
                batches[r]->Results_->Update(Score{b0.Comparison->series()[idx[o + k]]->Labels(), lag[o + k], score[o + k]});
        }
    }

private:
    void ensure(muse_group *dg)
    {
        if (!batch_ || batch_group_ != dg) {
            muse_batch_free(batch_);
            batch_ = nullptr;
            check(muse_batch_create(eng_->handle(), dg, ref_.data(), (int32_t)ref_.size(), &batch_));
            batch_group_ = dg;
        }
    }
    std::shared_ptr<Engine> eng_;
    std::vector<double> ref_;
    muse_batch *batch_ = nullptr;
    muse_group *batch_group_ = nullptr;
};
inline std::shared_ptr<Batch> NewBatch(SeriesPtr ref, GroupPtr comp, ResultsPtr results, int cc)
{
    return std::make_shared<Batch>(std::move(ref), std::move(comp), std::move(results), cc);
}

// ----------------------------------------------------------------- muse.go
class Muse {
public:
    Muse(SeriesPtr ref, ResultsPtr results, std::shared_ptr<Engine> eng = Engine::Default())
        : Results_(std::move(results)), eng_(std::move(eng)), ref_(ref->Values())
    {
        if (ref->Length() < 1) // muse.go:24-26
            throw Error(MUSE_ERR_EMPTY, "Reference series length must be greater than zero");
        refN_ = ref->Length();
        // the reference spectrum is computed once (muse.go:29-39); every Run shares it (muse_batch_create_like)
        check(muse_group_create(eng_->handle(), 0, refN_, &probe_));
        int rc = muse_batch_create(eng_->handle(), probe_, ref_.data(), refN_, &template_);
        if (rc) {
            std::string msg = muse_last_error();
            muse_group_free(probe_);
            probe_ = nullptr;
            throw Error(rc, msg);
        }
    }
    ~Muse()
    {
        muse_batch_free(template_);
        muse_group_free(probe_);
    }
    Muse(const Muse &) = delete;
    Muse &operator=(const Muse &) = delete;
    ResultsPtr Results_;
    void Run(const std::vector<SeriesPtr> &compGraphs) // muse.go:46-92
    {
        if (compGraphs.empty())
            return;
        std::vector<double> rows;
        for (auto &s : compGraphs) {
            if (s->Length() != refN_) // muse.go:68-70
                throw Error(MUSE_ERR_LENGTH, "Encountered a comparison graph with differing length than the reference");
            rows.insert(rows.end(), s->Values().begin(), s->Values().end());
        }
        muse_group *g = nullptr;
        check(muse_group_upload(eng_->handle(), rows.data(), (int64_t)compGraphs.size(), refN_, refN_, &g));
        muse_batch *b = nullptr;
        int rc = muse_batch_create_like(template_, g, &b);
        std::vector<int32_t> gid(compGraphs.size(), 0);
        int64_t idx = 0;
        int32_t lag = 0, cnt = 0;
        double score = 0, mean = 0;
        if (!rc)
            rc = muse_batch_run(b, gid.data(), 1, Results_->MaxLag, 1, Results_->Threshold, (int32_t)Results_->Filter,
                                0 /* signed scores: muse.go:72-76 */, &idx, &lag, &score, &cnt, &mean);
        std::string msg = rc ? muse_last_error() : "";
        muse_batch_free(b);
        muse_group_free(g);
        if (rc)
            throw Error(rc, msg);
        if (cnt == 1)
            Results_->Update(Score{compGraphs[idx]->Labels(), lag, score});
    }

private:
    std::shared_ptr<Engine> eng_;
    std::vector<double> ref_;
    int refN_ = 0;
    muse_group *probe_ = nullptr;    // empty group the template batch is bound to
    muse_batch *template_ = nullptr; // owns the reference spectrum
};
inline std::shared_ptr<Muse> New(SeriesPtr ref, ResultsPtr results)
{
    return std::make_shared<Muse>(std::move(ref), std::move(results));
}

} // namespace muse
