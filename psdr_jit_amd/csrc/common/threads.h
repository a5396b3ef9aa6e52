// threads.h — the host-side loops of both native libraries (tree build and blob sections in libpsdr_hip.so, Mesh / Sensor / edge
// configuration in _psdr_core) run on plain std::threads sized for what the PROCESS may use.  Not OpenMP: a container sees every core of
// its host (256 logical CPUs on the test box) while its cgroup grants 16; an OpenMP team of 256 spinning threads burns the quota of the
// whole 100 ms period in a few milliseconds and the process is throttled for the rest (measured: a 34 MB blob rewrite 5 -> 90 ms).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#include <unistd.h>
#endif

namespace psdr {

// threads a host-side loop may keep busy: the affinity mask, cut down to the cgroup's CPU quota and to 32; PSDR_HOST_THREADS overrides
inline int host_threads() {
    static const int n = [] {
        if (const char *e = std::getenv("PSDR_HOST_THREADS")) return std::max(1, std::atoi(e));
        int t = (int) std::thread::hardware_concurrency();
        if (t <= 0) t = 1;
#if defined(__linux__)
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) t = std::min(t, std::max(1, CPU_COUNT(&set)));
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {                        // cgroup v2: "<quota> <period>" or "max <period>"
            char q[64]; long long per = 0;
            if (std::fscanf(f, "%63s %lld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) t = std::min(t, std::max(1, (int) (std::atoll(q) / per)));
            std::fclose(f);
        } else {
            long long quota = -1, per = 0;                                                // cgroup v1
            if (FILE *fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(fq, "%lld", &quota) != 1) quota = -1; std::fclose(fq); }
            if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(fp, "%lld", &per) != 1) per = 0; std::fclose(fp); }
            if (quota > 0 && per > 0) t = std::min(t, std::max(1, (int) (quota / per)));
        }
#endif
        return std::min(t, 32);
    }();
    return n;
}

// The loops run on a PERSISTENT team (round 6): creating and joining fifteen threads per loop cost 0.2-0.5 ms each (measured: 210 us at 8 threads, 530 us at 16 for a
// 100 k-element loop whose work takes 20 us), and a moved-vertex Scene.configure() of BASELINE config 5 runs about fifteen such loops - a third of its 11 ms.  The team's
// threads SLEEP on a condition variable between loops (no spinning: an idle spinning team is what exhausted the cgroup's CPU quota under OpenMP, see above).
// A loop issued from inside a loop, or while another host thread's loop holds the team, runs on threads of its own as before; a child of fork() gets a new team.
class Team {
public:
    static Team &instance() { static Team t; return t; }
    // part(p) for p in [0, parts) on the caller + the team; returns false when the team is not available (the caller then uses threads of its own)
    template <typename G> bool run(size_t parts, const G &part) {
        if (in_job()) return false;
        std::unique_lock<std::mutex> own(owner_, std::try_to_lock);
        if (!own.owns_lock()) return false;
        ensure_workers();
        if (workers_.empty()) return false;
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = [&part](size_t p) { part(p); };
            parts_ = parts; next_.store(1); pending_ = workers_.size(); ++generation_;
        }
        cv_work_.notify_all();
        in_job() = true;
        part(0);
        for (size_t p; (p = next_.fetch_add(1)) < parts;) part(p);
        in_job() = false;
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
        return true;
    }
    ~Team() { stop(); }
private:
    static bool &in_job() { static thread_local bool f = false; return f; }
    void stop() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; ++generation_; }
        cv_work_.notify_all();
#if defined(__linux__)
        if (pid_ != getpid()) { for (std::thread &t : workers_) t.detach(); workers_.clear(); return; }      // (a forked child: the parent's threads do not exist here)
#endif
        for (std::thread &t : workers_) if (t.joinable()) t.join();
        workers_.clear();
    }
    void ensure_workers() {
#if defined(__linux__)
        if (!workers_.empty() && pid_ != getpid()) { for (std::thread &t : workers_) t.detach(); workers_.clear(); }
        pid_ = getpid();
#endif
        if (!workers_.empty()) return;
        stop_ = false;
        const int n = host_threads() - 1;
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { work(); });
    }
    void work() {
        in_job() = true;
        size_t seen = 0;
        { std::lock_guard<std::mutex> lk(m_); seen = generation_ - 1; }        // (created under run()'s owner lock, before the job is published)
        for (;;) {
            std::function<void(size_t)> job;
            size_t parts = 0;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_work_.wait(lk, [&] { return generation_ != seen + 0 && (stop_ || job_); });
                if (stop_) return;
                seen = generation_; job = job_; parts = parts_;
            }
            for (size_t p; (p = next_.fetch_add(1)) < parts;) job(p);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::mutex owner_, m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> workers_;
    std::function<void(size_t)> job_;
    size_t parts_ = 0, pending_ = 0, generation_ = 0;
    std::atomic<size_t> next_{0};
    bool stop_ = false;
#if defined(__linux__)
    pid_t pid_ = 0;
#endif
};

// fn(begin, end) over [0, n) in contiguous chunks of at least min_chunk items on up to `threads` threads (0 = host_threads()).
// The chunks are a function of n and the thread count only (whichever thread runs them); loops whose results must not depend on the
// thread count write disjoint outputs per item.
template <typename F> inline void parallel_for(size_t n, size_t min_chunk, F fn, int threads = 0) {
    if (threads <= 0) threads = host_threads();
    const size_t parts = std::max<size_t>(1, std::min<size_t>((size_t) threads, n / std::max<size_t>(1, min_chunk)));
    if (parts <= 1) { fn((size_t) 0, n); return; }
    if (Team::instance().run(parts, [&](size_t p) { fn(n * p / parts, n * (p + 1) / parts); })) return;
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    for (size_t p = 1; p < parts; ++p) th.emplace_back([=] { fn(n * p / parts, n * (p + 1) / parts); });
    fn((size_t) 0, n / parts);
    for (std::thread &t : th) t.join();
}

} // namespace psdr
