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
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <limits>
#include <map>
#include <memory>
#include <memory_resource>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

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
        uid_ = labels_->ID(); // (labels are immutable: formed once, not on every Group.Add / indexLabelValues)
    }
    int Length() const { return (int)y_.size(); }
    const std::vector<double> &Values() const { return y_; }
    const LabelsPtr &Labels() const { return labels_; }
    const std::string &UID() const { return uid_; } // series.go:40-42

private:
    std::vector<double> y_;
    LabelsPtr labels_;
    std::string uid_;
};
using SeriesPtr = std::shared_ptr<Series>;
inline SeriesPtr NewSeries(std::vector<double> y, LabelsPtr labels = nullptr)
{
    return std::make_shared<Series>(std::move(y), std::move(labels));
}

// ------------------------------------------------ rows into the pinned upload window
// The window is written once and read by the DMA engine, never by this CPU: streaming stores skip the read-for-ownership an
// ordinary copy pays per destination line and leave the caches to the Series.  Rows whose window address or length is not a
// multiple of 32 bytes, CPUs without AVX2, and `detail::StreamingStores = false` take memcpy.
namespace detail {
inline bool StreamingStores = true;
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void copy_row_streaming(double *dst, const double *src, size_t n)
{
    size_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m256d a = _mm256_loadu_pd(src + i), b = _mm256_loadu_pd(src + i + 4), c = _mm256_loadu_pd(src + i + 8),
                      d = _mm256_loadu_pd(src + i + 12);
        _mm256_stream_pd(dst + i, a);
        _mm256_stream_pd(dst + i + 4, b);
        _mm256_stream_pd(dst + i + 8, c);
        _mm256_stream_pd(dst + i + 12, d);
    }
    for (; i + 4 <= n; i += 4)
        _mm256_stream_pd(dst + i, _mm256_loadu_pd(src + i));
}
#endif
inline void copy_row(double *dst, const double *src, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (StreamingStores && avx2 && n % 4 == 0 && ((uintptr_t)dst & 31u) == 0) {
        copy_row_streaming(dst, src, n);
        return;
    }
#endif
    memcpy(dst, src, n * sizeof(double));
}
inline void copy_rows_done()
{
#if defined(__x86_64__)
    _mm_sfence(); // streaming stores are weakly ordered: before the commit tells the device to read them
#endif
}
} // namespace detail

// ------------------------------------------------ a few parked host threads
// Packing a Group's rows into the pinned upload window (muse_group_stage) is a memory copy of the whole Group: one
// thread moves ~10 GB/s where PCIe takes 57.  The reference fans its work out over goroutines (muse_batch.go:99-130);
// here a handful of parked threads take the pieces of a window.  run(n, fn) calls fn(0) .. fn(n - 1), the caller
// taking part, and returns when all have finished; one job at a time.
namespace detail {
class Workers {
public:
    static Workers &get()
    {
        static Workers w;
        return w;
    }
    // parked threads beside the caller (read once, when the first job starts the pool): set it before anything uploads
    static inline int MaxExtra = 7;
    // how long a thread that has just finished a job keeps looking for the next one before it parks (0: parks at once)
    static inline int SpinMicros = 1000;
    int width() const { return (int)threads_.size() + 1; }
    void run(int n, const std::function<void(int)> &fn)
    {
        if (n <= 1 || threads_.empty()) {
            for (int i = 0; i < n; i++)
                fn(i);
            return;
        }
        std::lock_guard<std::mutex> one(job_mu_);
        {
            std::lock_guard<std::mutex> lock(mu_);
            fn_ = &fn;
            total_ = n;
            next_.store(1);
            pending_ = n - 1;
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        fn(0);
        for (;;) { // the caller keeps taking tasks, then waits for the ones still running
            const int i = next_.fetch_add(1);
            if (i >= n)
                break;
            fn(i);
            std::lock_guard<std::mutex> lock(mu_);
            pending_--;
        }
        std::unique_lock<std::mutex> lock(mu_);
        done_cv_.wait(lock, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }
    ~Workers()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_)
            t.join();
    }

private:
    Workers()
    {
        const unsigned hw = std::thread::hardware_concurrency();
        const int extra = (int)std::min((unsigned)std::max(MaxExtra, 0), hw > 1 ? hw / 2 : 0u);
        for (int i = 0; i < extra; i++)
            threads_.emplace_back([this] { loop(); });
    }
    void loop()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
            if (SpinMicros > 0 && seen != 0) { // the windows of a streamed Group follow each other within ~0.1 ms: stay awake that long
                lock.unlock();
                const auto t0 = std::chrono::steady_clock::now();
                while (generation_.load(std::memory_order_acquire) == seen &&
                       std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(SpinMicros))
                    __builtin_ia32_pause();
                lock.lock();
            }
            cv_.wait(lock, [&] { return stop_ || generation_.load() != seen; });
            if (stop_)
                return;
            seen = generation_.load();
            for (;;) {
                const std::function<void(int)> *fn = fn_;
                if (!fn)
                    break;
                const int i = next_.fetch_add(1);
                if (i >= total_)
                    break;
                lock.unlock();
                (*fn)(i);
                lock.lock();
                if (--pending_ == 0)
                    done_cv_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mu_, job_mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(int)> *fn_ = nullptr;
    std::atomic<int> next_{0};
    int total_ = 0, pending_ = 0;
    std::atomic<uint64_t> generation_{0};
    bool stop_ = false;
};
// One parked thread that runs upload jobs in order (Group.Add streams the rows it has been given towards the device while
// the caller goes on adding: the job packs with the Workers above and commits; the caller meets it again in Group::device).
class Uploader {
public:
    static Uploader &get()
    {
        static Uploader u;
        return u;
    }
    std::future<void> submit(std::function<void()> fn)
    {
        std::packaged_task<void()> task(std::move(fn));
        std::future<void> f = task.get_future();
        {
            std::lock_guard<std::mutex> lock(mu_);
            q_.push_back(std::move(task));
        }
        cv_.notify_one();
        return f;
    }
    ~Uploader()
    {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }

private:
    Uploader() : th_([this] { loop(); }) {}
    void loop()
    {
        std::unique_lock<std::mutex> lock(mu_);
        for (;;) {
            cv_.wait(lock, [&] { return stop_ || !q_.empty(); });
            if (q_.empty())
                return;
            std::packaged_task<void()> task = std::move(q_.front());
            q_.pop_front();
            lock.unlock();
            task();
            lock.lock();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::packaged_task<void()>> q_;
    bool stop_ = false;
    std::thread th_; // (declared last: the thread starts with every other member built)
};
} // namespace detail

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
        default_created().store(true);
        return e;
    }
    // the default engine if some caller has already made it (never creates a context: Group.Add uses it to stream rows ahead)
    static std::shared_ptr<Engine> DefaultIfCreated() { return default_created().load() ? Default() : nullptr; }
    // One engine (context, stream, tables) per listed device: the device set a Batch shards its Comparison group over
    // (the goroutine fan-out of muse_batch.go:99-130 becomes one host thread per device).  A device may be
    // listed more than once (several contexts on one GPU: what the tests on a one-GPU box do).
    static std::vector<std::shared_ptr<Engine>> List(const std::vector<int> &devices)
    {
        std::vector<std::shared_ptr<Engine>> out;
        for (int d : devices)
            out.push_back(std::make_shared<Engine>(d));
        return out;
    }

    // xCorr (xcorr.go:102-153) for M independent (x, y) pairs in one launch: x_rows is M x lenx, y_rows M x leny,
    // dense row-major; n is raised to max(n, lenx, leny).  nil[i] != 0 where the reference returns (nil, 0, 0).
    struct XCorrResult {
        std::vector<int32_t> lag, nil;
        std::vector<double> mv, cc; // cc: M x n when asked for
        int32_t n = 0;
    };
    XCorrResult XCorrBatch(const std::vector<double> &x_rows, const std::vector<double> &y_rows, int64_t M, int32_t lenx,
                           int32_t leny, int32_t n, bool normalize, bool want_cc = false) const
    {
        if (M < 0 || (int64_t)x_rows.size() != M * lenx || (int64_t)y_rows.size() != M * leny)
            throw Error(MUSE_ERR_INVALID, "XCorrBatch: row counts and lengths do not match the data");
        XCorrResult r;
        r.n = std::max(n, std::max(lenx, leny));
        r.lag.assign((size_t)M, 0);
        r.nil.assign((size_t)M, 0);
        r.mv.assign((size_t)M, 0.0);
        if (want_cc)
            r.cc.assign((size_t)M * (size_t)r.n, 0.0);
        check(muse_xcorr_batch(ctx_, x_rows.data(), y_rows.data(), M, lenx, leny, n, normalize ? 1 : 0, r.lag.data(), r.mv.data(),
                               r.nil.data(), want_cc ? r.cc.data() : nullptr));
        return r;
    }

private:
    static std::atomic<bool> &default_created()
    {
        static std::atomic<bool> flag{false};
        return flag;
    }
    muse_ctx *ctx_ = nullptr;
};

// -------------------------------------------------------------- group.go
class Group {
public:
    explicit Group(std::string name) : Name(std::move(name)) {}
    ~Group()
    {
        drain_stream(false);
        if (dev_)
            muse_group_free(dev_);
        free_shards();
    }
    // Rows are streamed towards the default engine's HBM as they are added ("Group.Add stages series into a
    // C-allocated pinned buffer"): whenever STREAM_BYTES of new series have gathered, a background job packs them into a pinned
    // window and commits them, while the caller goes on adding -- by the time a Batch runs, most of the Group is resident.
    // Only if the process already HAS a default engine (Add never creates a GPU context), and only towards it: a Batch on
    // another engine or over a device list uploads on its own, as before.  Off: Group::StreamOnAdd = false.
    static inline bool StreamOnAdd = true;
    static constexpr size_t STREAM_BYTES = (size_t)4 << 20;
    // the first hand-over of a Group could leave earlier; measured, 1 MB is worse than 4 (cold BenchmarkMuseBatchRunLarge 740 -> 791 us)
    static inline size_t StreamFirstBytes = (size_t)4 << 20;
    std::string Name;
    int Length() const { return n_; }
    // group.go:31-56; errors come back as muse::Error(MUSE_ERR_INVALID / MUSE_ERR_LENGTH)
    void Add(const std::vector<SeriesPtr> &series)
    {
        for (auto &s : series)
            Add(s);
    }
    // (one Series: Go's variadic Add(series ...*Series) called with one argument -- `g->Add({s})` binds here without building a vector)
    void Add(const SeriesPtr &s)
    {
        {
            if (s->Labels()->Keys().empty())
                throw Error(MUSE_ERR_INVALID, "Invalid Series with no labels");
            const std::string &uid = s->UID();
            if (order_.empty())
                n_ = s->Length();
            else if (s->Length() != n_ && !registry_.count(uid))
                throw Error(MUSE_ERR_LENGTH, "Timeseries has length " + std::to_string(s->Length()) +
                                                 ", but current group has length " + std::to_string(n_));
            // (the key views the Series' own uid string: the Series lives in order_ as long as the registry does)
            if (!registry_.try_emplace(std::string_view(uid), order_.size()).second)
                throw Error(MUSE_ERR_INVALID,
                            "Series with label:values, " + uid + ", already exists within group, " + Name);
            order_.push_back(s);
            if (StreamOnAdd && (order_.size() - uploaded_) * (size_t)n_ * sizeof(double) >= (uploaded_ ? STREAM_BYTES : StreamFirstBytes))
                maybe_stream();
        }
    }
    std::vector<SeriesPtr> FilterByLabelValues(const Labels &labels) const // group.go:60-71
    {
        std::vector<SeriesPtr> out;
        auto it = index_ids_.find(labels.ID());
        if (it != index_ids_.end())
            for (size_t i : index_members_[(size_t)it->second])
                out.push_back(order_[i]);
        return out;
    }
    // group.go:76-104.  Partition order = first appearance in insertion order (Go
    // iterates a map here, so its order is unspecified).  group_id_out[i] receives
    // the partition index of series i.
    std::vector<LabelsPtr> indexLabelValues(std::vector<std::string> groupByLabels,
                                            std::vector<int32_t> *group_id_out = nullptr)
    {
        // The partition depends on the label names and on the series alone, and series are only ever added (group.go has no
        // removal, and the registry is private here): a repeated Run with the same grouping over the same series (the
        // reference's benchmark loops, muse_batch_test.go:127-131, 157-161) reuses it instead of rebuilding 5 000 label
        // strings per Run.
        std::string key = std::to_string(order_.size());
        for (auto &l : groupByLabels)
            key += '\x1f' + l;
        if (key == index_key_ && !order_.empty()) {
            if (group_id_out)
                *group_id_out = index_gid_;
            return index_distinct_;
        }
        std::vector<LabelsPtr> distinct;
        index_ids_.clear();
        index_members_.clear();
        std::vector<int32_t> gid_of(order_.size(), 0);
        const bool by_uid = groupByLabels.empty();
        if (by_uid && !order_.empty())
            groupByLabels = order_[0]->Labels()->Keys(); // group.go:88 (first series' keys)
        std::sort(groupByLabels.begin(), groupByLabels.end()); // labels.go:60: ID sorts the names it is given
        std::string guid;
        for (size_t i = 0; i < order_.size(); i++) {
            auto &s = order_[i];
            const std::string *id = &s->UID();
            if (!by_uid) { // Labels.ID(groupByLabels), labels.go:54-73, into a reused buffer
                guid.clear();
                const LabelMap &m = s->Labels()->Map();
                for (auto &name : groupByLabels) {
                    auto it = m.find(name);
                    if (it != m.end()) {
                        if (!guid.empty())
                            guid += ',';
                        guid += name;
                        guid += ':';
                        guid += it->second;
                    }
                }
                id = &guid;
            }
            auto it = index_ids_.find(*id);
            if (it == index_ids_.end()) {
                LabelMap lv;
                for (auto &name : groupByLabels) {
                    std::string v;
                    if (s->Labels()->Get(name, &v))
                        lv[name] = v;
                }
                distinct.push_back(NewLabels(lv));
                it = index_ids_.emplace(*id, (int32_t)index_ids_.size()).first;
                index_members_.emplace_back();
            }
            index_members_[(size_t)it->second].push_back(i);
            gid_of[i] = it->second;
        }
        if (group_id_out)
            *group_id_out = gid_of;
        index_key_ = std::move(key);
        index_gid_ = std::move(gid_of);
        index_distinct_ = distinct;
        return distinct;
    }
    const std::vector<SeriesPtr> &series() const { return order_; }

    // One contiguous row range per engine (the shard_bounds rule of go-muse_amd/dist.py: equal shares rounded up to an even
    // row count, the fused kernels pack two series per pass).  Cut once, at the first sharded Run; series added later
    // extend the last shard.
    struct Shard {
        std::shared_ptr<Engine> eng;
        muse_group *dev = nullptr;
        size_t lo = 0, hi = 0, uploaded = 0; // rows [lo, hi) of the group; `uploaded` of them are resident
    };
    std::vector<Shard> &shards(const std::vector<std::shared_ptr<Engine>> &engs)
    {
        bool same = shards_.size() == engs.size();
        for (size_t i = 0; same && i < engs.size(); i++)
            same = shards_[i].eng == engs[i];
        if (!same) {
            free_shards();
            const size_t M = order_.size(), W = engs.size();
            size_t per = (M + W - 1) / W;
            per = (per + 1) / 2 * 2;
            for (size_t r = 0; r < W; r++) {
                Shard sh;
                sh.eng = engs[r];
                sh.lo = std::min(r * per, M);
                sh.hi = std::min(sh.lo + per, M);
                check(muse_group_create(engs[r]->handle(), (int64_t)(sh.hi - sh.lo), n_ > 0 ? n_ : 1, &sh.dev));
                shards_.push_back(sh);
            }
        }
        if (!shards_.empty())
            shards_.back().hi = order_.size(); // series added since the cut
        for (auto &sh : shards_) {
            append_rows(sh.dev, sh.lo + sh.uploaded, sh.hi);
            sh.uploaded = sh.hi - sh.lo;
        }
        return shards_;
    }

    // device residency: rows are uploaded once and appended to (muse_group_append)
    muse_group *device(const std::shared_ptr<Engine> &eng, const std::function<void()> *side = nullptr)
    {
        if (dev_ && eng_ == eng && stream_job_.valid()) {
            // rows are already streaming towards this engine: the rest goes to the background uploader as well (behind the job
            // in flight), the caller's side work runs beside it, then the two meet
            if (uploaded_ < order_.size())
                submit_rest();
            if (side) {
                (*side)();
                side = nullptr;
            }
        }
        drain_stream(true); // (rows handed to the background uploader are committed; its errors surface here)
        if (!dev_ || eng_ != eng) {
            if (dev_)
                muse_group_free(dev_);
            dev_ = nullptr;
            eng_ = eng;
            uploaded_ = 0;
            check(muse_group_create(eng->handle(), (int64_t)order_.size(), n_ > 0 ? n_ : 1, &dev_));
        }
        append_rows(dev_, uploaded_, order_.size(), side);
        uploaded_ = order_.size();
        return dev_;
    }

private:
    // series [first, last) to a device group through the library's pinned windows (muse_group_stage / _commit): every Series is
    // copied ONCE, straight into pinned memory, by a few threads in pieces of ~256 KB; a piece that completes the packed
    // prefix of the window hands that prefix -- its own rows and every finished piece behind them, as ONE copy -- to the
    // copy stream, so the rows cross PCIe while the next pieces are packed and the copies stay few and large.
    // `side` (optional) runs on the calling thread while the other threads pack: host work that does not need the rows
    // (Batch.Run partitions the labels there).
    void append_rows(muse_group *dev, size_t first, size_t last, const std::function<void()> *side = nullptr)
    {
        if (first >= last) {
            if (side)
                (*side)();
            return;
        }
        const std::vector<const double *> rows = row_pointers(first, last);
        append_window_loop(dev, rows.data(), 0, rows.size(), n_, side);
    }
    // the sample pointers of series [first, last): what a packing job reads (the Series stay alive in order_ as long as the Group
    // does, and the Group's destructor waits for its jobs; order_ itself may be growing under Add meanwhile)
    std::vector<const double *> row_pointers(size_t first, size_t last) const
    {
        std::vector<const double *> rows(last - first);
        for (size_t i = first; i < last; i++)
            rows[i - first] = order_[i]->Values().data();
        return rows;
    }
    static void append_window_loop(muse_group *dev, const double *const *rows, size_t first, const size_t last, const int n_,
                                   const std::function<void()> *side)
    {
        const size_t row_bytes = sizeof(double) * (size_t)n_;
        bool side_done = side == nullptr;
        while (first < last) {
            double *win = nullptr;
            int64_t granted = 0;
            check(muse_group_stage(dev, (int64_t)(last - first), &win, &granted));
            // (measured on the MI355X box, tools/probe/upload_probe.cpp: one H2D copy command costs ~13 us whatever its size --
            // 19.2 MB cross in 365 us as one copy, 409 us in 4 MB copies, 1 116 us in 256 KB copies -- while packing wants
            // pieces small enough to spread over the threads: pack in 256 KB pieces, commit in runs of >= 4 MB)
            const size_t piece = std::max<size_t>(1, ((size_t)256 << 10) / row_bytes);
            const int npieces = (int)(((size_t)granted + piece - 1) / piece);
            const int commit_pieces = (int)std::max<size_t>(1, ((size_t)4 << 20) / (piece * row_bytes));
            std::vector<char> done((size_t)npieces, 0);
            int watermark = 0; // pieces [0, watermark) are committed
            int status = MUSE_OK;
            std::string message;
            std::mutex commit_mu;
            const int extra = side_done ? 0 : 1;
            detail::Workers::get().run(npieces + extra, [&](int task) {
                if (task < extra) { // (task 0 runs on the calling thread)
                    (*side)();
                    return;
                }
                const int p = task - extra;
                const size_t lo = (size_t)p * piece, hi = std::min((size_t)granted, lo + piece);
                for (size_t r = lo; r < hi; r++)
                    detail::copy_row(win + r * (size_t)n_, rows[first + r], (size_t)n_);
                detail::copy_rows_done();
                std::lock_guard<std::mutex> lock(commit_mu);
                done[(size_t)p] = 1;
                int w = watermark;
                while (w < npieces && done[(size_t)w])
                    w++;
                if (w > watermark && (w == npieces || w - watermark >= commit_pieces)) { // (every row is committed even after a failure: the window has to close)
                    const size_t clo = (size_t)watermark * piece, chi = std::min((size_t)granted, (size_t)w * piece);
                    const int rc = muse_group_commit(dev, (int64_t)clo, (int64_t)(chi - clo));
                    if (rc && status == MUSE_OK) {
                        status = rc;
                        message = muse_last_error(); // (thread-local: read it on the thread that failed)
                    }
                    watermark = w;
                }
            });
            side_done = true;
            if (status)
                throw Error(status, message);
            first += (size_t)granted;
        }
        if (!side_done)
            (*side)();
    }
    // hand the series added since the last hand-over to the background uploader (one job in flight at a time: while one runs,
    // the next Adds just accumulate)
    void maybe_stream()
    {
        if (!stream_checked_) {
            stream_checked_ = true;
            stream_eng_ = Engine::DefaultIfCreated();
        }
        if (!stream_eng_ || (dev_ && eng_ != stream_eng_) || !shards_.empty())
            return;
        if (stream_job_.valid()) {
            if (stream_job_.wait_for(std::chrono::seconds(0)) != std::future_status::ready)
                return;
            stream_job_.get(); // (rethrows what the job met)
        }
        if (!dev_) {
            eng_ = stream_eng_;
            uploaded_ = 0;
            // (capacity for four hand-overs; growth beyond is asynchronous in the library: muse_group_append's reserve)
            check(muse_group_create(eng_->handle(), (int64_t)(4 * (order_.size() - uploaded_)), n_, &dev_));
        }
        submit_rest();
    }
    // series [uploaded_, size) to the background uploader (jobs run in order: one that is still queued behind another is fine)
    void submit_rest()
    {
        auto part = std::make_shared<std::vector<const double *>>(row_pointers(uploaded_, order_.size()));
        uploaded_ = order_.size();
        muse_group *const dev = dev_;
        const int n = n_;
        if (stream_job_.valid())
            stream_older_.push_back(std::move(stream_job_));
        stream_job_ = detail::Uploader::get().submit([part, dev, n] { append_window_loop(dev, part->data(), 0, part->size(), n, nullptr); });
    }
    void drain_stream(bool rethrow)
    {
        std::exception_ptr first;
        for (auto &f : stream_older_) {
            try {
                f.get();
            } catch (...) {
                if (!first)
                    first = std::current_exception();
            }
        }
        stream_older_.clear();
        if (stream_job_.valid()) {
            try {
                stream_job_.get();
            } catch (...) {
                if (!first)
                    first = std::current_exception();
            }
        }
        if (first && rethrow)
            std::rethrow_exception(first);
    }
    void free_shards()
    {
        for (auto &sh : shards_)
            if (sh.dev)
                muse_group_free(sh.dev);
        shards_.clear();
    }
    int n_ = 0;
    std::vector<SeriesPtr> order_;
    // (nodes from a monotonic arena: no allocation per Add, one release in the destructor)
    std::pmr::monotonic_buffer_resource registry_arena_{(size_t)64 << 10};
    std::pmr::unordered_map<std::string_view, size_t> registry_{&registry_arena_};
    std::unordered_map<std::string, int32_t> index_ids_;     // label-values id -> partition index (indexLabelValues)
    std::vector<std::vector<size_t>> index_members_;          // partition index -> its series, in insertion order
    std::string index_key_;                  // the grouping index_ / index_gid_ / index_distinct_ were built for (series count + label names)
    std::vector<int32_t> index_gid_;
    std::vector<LabelsPtr> index_distinct_;
    std::shared_ptr<Engine> eng_;
    muse_group *dev_ = nullptr;
    size_t uploaded_ = 0;                 // series resident in dev_ or handed to the background uploader
    std::vector<Shard> shards_;
    std::shared_ptr<Engine> stream_eng_;  // the engine Add streams towards (the default engine, if the process had one at the first hand-over)
    bool stream_checked_ = false;
    std::future<void> stream_job_;              // the newest job
    std::vector<std::future<void>> stream_older_; // jobs queued in front of it whose outcome has not been collected yet
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
    // Over several devices: the Comparison group is cut into one contiguous row range per engine, every Run
    // scores the shards at the same time (one host thread per device) and merges their candidates on the host.
    Batch(SeriesPtr ref, GroupPtr comp, ResultsPtr results, int cc, std::vector<std::shared_ptr<Engine>> engines)
        : Batch(ref, comp, results, cc, engines.empty() ? Engine::Default() : engines[0])
    {
        if (engines.size() > 1)
            engines_ = std::move(engines);
    }
    Batch(SeriesPtr ref, GroupPtr comp, ResultsPtr results, int cc, std::shared_ptr<Engine> eng = Engine::Default())
        : Comparison(std::move(comp)), Results_(std::move(results)), Concurrency(cc < 1 ? 1 : cc), eng_(std::move(eng)),
          ref_(ref->Values())
    {
        for (auto &s : Comparison->series())
            if (s->Length() != ref->Length())
                throw Error(MUSE_ERR_LENGTH, s->UID() + " from comparison group series does not have the same "
                                                        "length as the reference");
        n = (int)muse_next_pow2((double)ref->Length());
        // the reference is validated and transformed ONCE, here (muse_batch.go:35-47), against an empty group; the batch over
        // the Comparison group's resident rows shares that spectrum (muse_batch_create_like)
        check(muse_group_create(eng_->handle(), 0, ref->Length() > 0 ? ref->Length() : 1, &probe_));
        int rc = muse_batch_create(eng_->handle(), probe_, ref_.data(), (int32_t)ref_.size(), &template_);
        if (rc) {
            std::string msg = muse_last_error();
            muse_group_free(probe_);
            probe_ = nullptr;
            throw Error(rc, msg);
        }
    }
    ~Batch()
    {
        muse_batch_free(batch_);
        for (auto &sb : shard_batches_)
            muse_batch_free(sb.batch);
        muse_batch_free(template_);
        muse_group_free(probe_);
    }
    Batch(const Batch &) = delete;
    Batch &operator=(const Batch &) = delete;
    int n = 0;
    GroupPtr Comparison;
    ResultsPtr Results_;
    int Concurrency; // kept for API compatibility: the GPU is the fan-out

    // Run (muse_batch.go:99-130).  Always "returns nil": errors can only be device failures.
    void Run(const std::vector<std::string> &groupByLabels)
    {
        std::vector<int32_t> gid;
        std::vector<LabelsPtr> lvs;
        if (engines_.empty()) {
            // rows added since the last Run go up now; the calling thread partitions the labels (group.go:76-104) while the
            // other host threads pack them into the pinned window
            const std::function<void()> side = [&] { lvs = Comparison->indexLabelValues(groupByLabels, &gid); };
            Comparison->device(eng_, &side);
        } else {
            lvs = Comparison->indexLabelValues(groupByLabels, &gid);
        }
        if (lvs.empty())
            return;
        if ((int64_t)lvs.size() <= EXACT_FEED_MAX_GROUPS) {
            feed_groups(gid, (int32_t)lvs.size());
            return;
        }
        // very many label groups (Run(nil) over a million series): the device pre-selects the TopN candidates, 24 B x TopN cross
        // the host; among EXACTLY tied scores the order / the survivor at the boundary may then differ from a full feed
        if (!engines_.empty()) {
            run_sharded(gid, (int32_t)lvs.size());
            return;
        }
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

    // Batch.Run feeds Results one Score per label group (the reference's feed) up to this many groups
    // (settable: the tests also drive the pre-selecting path, which is otherwise only taken above this many groups)
    static inline int64_t EXACT_FEED_MAX_GROUPS = 65536;

private:
    // The reference's own feed (muse_batch.go:124-128): ONE Score per label group, in group order, through Results.Update --
    // the heap's history, and with it the order Fetch returns exactly tied scores in and which of them survives at the TopN
    // boundary, is the reference's (for insertion-ordered groups), also when the Results already holds the Scores of earlier
    // Runs (results.go:55-72) and whether the Group sits on one device or is cut over several: every shard returns its winner
    // per group, unfiltered (muse_batch_run_groups), muse_merge_group_winners takes the per-group maximum.
    void feed_groups(const std::vector<int32_t> &gid, int32_t G)
    {
        size_t W = 1;
        std::vector<muse_record> recs;
        std::vector<uint8_t> state;
        if (engines_.empty()) {
            muse_group *dg = Comparison->device(eng_);
            ensure(dg);
            recs.resize((size_t)G);
            state.resize((size_t)G);
            check(muse_batch_run_groups(batch_, gid.data(), G, 0, 1, recs.data(), state.data()));
        } else {
            auto &shards = Comparison->shards(engines_);
            ensure_shard_batches(shards);
            W = shards.size();
            recs.assign(W * (size_t)G, muse_record{-1, 0.0, 0, 0});
            state.assign(W * (size_t)G, 0);
            std::vector<int> status(W, MUSE_OK);
            std::vector<std::string> message(W);
            std::vector<std::thread> workers;
            for (size_t r = 0; r < W; r++) {
                if (shards[r].lo == shards[r].hi)
                    continue; // an empty shard (fewer rows than devices)
                workers.emplace_back([&, r]() {
                    status[r] = muse_batch_run_groups(shard_batches_[r].batch, gid.data() + shards[r].lo, G, (int64_t)shards[r].lo, 1,
                                                      recs.data() + r * (size_t)G, state.data() + r * (size_t)G);
                    if (status[r])
                        message[r] = muse_last_error(); // (thread-local: read it on the thread that failed)
                });
            }
            for (auto &w : workers)
                w.join();
            for (size_t r = 0; r < W; r++)
                if (status[r])
                    throw Error(status[r], message[r]);
        }
        std::vector<muse_record> win((size_t)G);
        std::vector<uint8_t> st((size_t)G);
        check(muse_merge_group_winners(recs.data(), state.data(), (int32_t)W, G, win.data(), st.data()));
        for (int32_t g = 0; g < G; g++)
            if (st[(size_t)g] == 1) // (0: no member; 2: the group's score is NaN, which never passes Results.passed)
                Results_->Update(Score{Comparison->series()[(size_t)win[(size_t)g].series]->Labels(), win[(size_t)g].lag, win[(size_t)g].score});
    }
    template <typename Shards>
    void ensure_shard_batches(Shards &shards)
    {
        const size_t W = shards.size();
        if (shard_batches_.size() != W)
            shard_batches_.resize(W);
        for (size_t r = 0; r < W; r++) {
            auto &sb = shard_batches_[r];
            if (!sb.batch || sb.group != shards[r].dev) {
                muse_batch_free(sb.batch);
                sb.batch = nullptr;
                check(muse_batch_create(shards[r].eng->handle(), shards[r].dev, ref_.data(), (int32_t)ref_.size(), &sb.batch));
                sb.group = shards[r].dev;
            }
        }
    }
    // The sharded Run over very many label groups.  Label groups that live on ONE shard each (always the case when every series is its own group):
    // each shard returns its top-N candidates (muse_batch_run_shard) and muse_merge_records selects -- 24 B x TopN per
    // device cross the host.  Label groups that straddle shards: each shard returns its winner per group, unfiltered
    // (muse_batch_run_groups), and muse_merge_group_records takes the per-group maximum BEFORE filtering and selecting.
    void run_sharded(const std::vector<int32_t> &gid, int32_t G)
    {
        auto &shards = Comparison->shards(engines_);
        const size_t W = shards.size();
        ensure_shard_batches(shards);
        // does any label group have members on two shards?
        bool straddle = false;
        {
            std::vector<int32_t> owner((size_t)G, -1);
            for (size_t r = 0; r < W && !straddle; r++)
                for (size_t i = shards[r].lo; i < shards[r].hi; i++) {
                    int32_t &o = owner[(size_t)gid[i]];
                    if (o >= 0 && o != (int32_t)r) {
                        straddle = true;
                        break;
                    }
                    o = (int32_t)r;
                }
        }
        const int top = Results_->TopN, cap = std::max(top, 1);
        std::vector<muse_record> recs(straddle ? W * (size_t)G : W * (size_t)cap);
        std::vector<uint8_t> state(straddle ? W * (size_t)G : 0);
        std::vector<int32_t> cnt(W, 0);
        std::vector<int> status(W, MUSE_OK);
        std::vector<std::string> message(W);
        std::vector<std::thread> workers;
        for (size_t r = 0; r < W; r++) {
            if (shards[r].lo == shards[r].hi)
                continue; // an empty shard (fewer rows than devices)
            workers.emplace_back([&, r]() {
                const int32_t *g = gid.data() + shards[r].lo;
                int rc;
                if (straddle)
                    rc = muse_batch_run_groups(shard_batches_[r].batch, g, G, (int64_t)shards[r].lo, 1,
                                               recs.data() + r * (size_t)G, state.data() + r * (size_t)G);
                else
                    rc = muse_batch_run_shard(shard_batches_[r].batch, g, G, (int64_t)shards[r].lo, Results_->MaxLag, top,
                                              Results_->Threshold, (int32_t)Results_->Filter, 1, recs.data() + r * (size_t)cap,
                                              &cnt[r]);
                status[r] = rc;
                if (rc)
                    message[r] = muse_last_error(); // (thread-local: read it on the thread that failed)
            });
        }
        for (auto &w : workers)
            w.join();
        for (size_t r = 0; r < W; r++)
            if (status[r])
                throw Error(status[r], message[r]);
        std::vector<int64_t> idx(cap);
        std::vector<int32_t> lag(cap);
        std::vector<double> score(cap);
        int32_t n_out = 0;
        double mean = 0;
        if (straddle) {
            check(muse_merge_group_records(recs.data(), state.data(), (int32_t)W, G, Results_->MaxLag, top, Results_->Threshold,
                                           (int32_t)Results_->Filter, idx.data(), lag.data(), score.data(), &n_out, &mean));
        } else {
            std::vector<muse_record> all;
            for (size_t r = 0; r < W; r++)
                all.insert(all.end(), recs.begin() + r * (size_t)cap, recs.begin() + r * (size_t)cap + cnt[r]);
            check(muse_merge_records(all.data(), (int64_t)all.size(), top, idx.data(), lag.data(), score.data(), &n_out, &mean));
        }
        std::vector<int> order(n_out);
        for (int i = 0; i < n_out; i++)
            order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gid[idx[a]] < gid[idx[b]]; });
        for (int k : order)
            Results_->Update(Score{Comparison->series()[idx[k]]->Labels(), lag[k], score[k]});
    }
    struct ShardBatch {
        muse_batch *batch = nullptr;
        muse_group *group = nullptr;
    };
    std::vector<std::shared_ptr<Engine>> engines_; // more than one device: the sharded Run
    std::vector<ShardBatch> shard_batches_;
    void ensure(muse_group *dg)
    {
        if (!batch_ || batch_group_ != dg) {
            muse_batch_free(batch_);
            batch_ = nullptr;
            check(muse_batch_create_like(template_, dg, &batch_));
            batch_group_ = dg;
        }
    }
    std::shared_ptr<Engine> eng_;
    std::vector<double> ref_;
    muse_batch *batch_ = nullptr;
    muse_group *batch_group_ = nullptr;
    muse_group *probe_ = nullptr;    // empty group the template batch is bound to
    muse_batch *template_ = nullptr; // owns the reference spectrum
};
inline std::shared_ptr<Batch> NewBatch(SeriesPtr ref, GroupPtr comp, ResultsPtr results, int cc)
{
    return std::make_shared<Batch>(std::move(ref), std::move(comp), std::move(results), cc);
}
// the same over a set of devices (Engine::List): one shard of the Comparison group per device
inline std::shared_ptr<Batch> NewBatch(SeriesPtr ref, GroupPtr comp, ResultsPtr results, int cc,
                                       std::vector<std::shared_ptr<Engine>> engines)
{
    return std::make_shared<Batch>(std::move(ref), std::move(comp), std::move(results), cc, std::move(engines));
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
        // the reference spectrum is computed once (muse.go:29-39); every Run shares it (muse_batch_run_rows)
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
        // one ABI call: the series are gathered straight into the call's pinned buffer (no packed copy here), then upload,
        // fused kernel, group maximum, record back (muse_batch_run_row_ptrs; signed scores: muse.go:72-76).
        // Results.Update applies passed() to the group's Score exactly as the reference does (results.go:55-72).
        const double *small[64];
        std::vector<const double *> many;
        const double **rows = small;
        if (compGraphs.size() > 64) {
            many.resize(compGraphs.size());
            rows = many.data();
        }
        for (size_t i = 0; i < compGraphs.size(); i++) {
            if (compGraphs[i]->Length() != refN_) // muse.go:68-70
                throw Error(MUSE_ERR_LENGTH, "Encountered a comparison graph with differing length than the reference");
            rows[i] = compGraphs[i]->Values().data();
        }
        muse_record win{};
        uint8_t state = 0;
        check(muse_batch_run_row_ptrs(template_, rows, (int64_t)compGraphs.size(), 0, &win, &state));
        if (state == 1 && win.series >= 0)
            Results_->Update(Score{compGraphs[(size_t)win.series]->Labels(), win.lag, win.score});
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
