// threads.h — the host-side loops of both native libraries (tree build and blob sections in libpsdr_hip.so, Mesh / Sensor / edge
// configuration in _psdr_core) run on plain std::threads sized for what the PROCESS may use.  Not OpenMP: a container sees every core of
// its host (256 logical CPUs on the test box) while its cgroup grants 16; an OpenMP team of 256 spinning threads burns the quota of the
// whole 100 ms period in a few milliseconds and the process is throttled for the rest (measured: a 34 MB blob rewrite 5 -> 90 ms).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
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

// fn(begin, end) over [0, n) in contiguous chunks of at least min_chunk items on up to `threads` threads (0 = host_threads(); the calling
// thread takes the first chunk).  The chunks are a function of n and the thread count only; loops whose results must not depend on the
// thread count write disjoint outputs per item.
template <typename F> inline void parallel_for(size_t n, size_t min_chunk, F fn, int threads = 0) {
    if (threads <= 0) threads = host_threads();
    const size_t parts = std::max<size_t>(1, std::min<size_t>((size_t) threads, n / std::max<size_t>(1, min_chunk)));
    if (parts <= 1) { fn((size_t) 0, n); return; }
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    for (size_t p = 1; p < parts; ++p) th.emplace_back([=] { fn(n * p / parts, n * (p + 1) / parts); });
    fn((size_t) 0, n / parts);
    for (std::thread &t : th) t.join();
}

} // namespace psdr
