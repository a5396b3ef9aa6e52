// CPU check of the host thread team (psdr_jit_amd/csrc/common/threads.h): compiled and run by tests/test_threads_cpu.py.
//   * parallel_for covers [0, n) exactly once whatever the thread count, and the chunk boundaries are a function of n and the thread count only;
//   * loops issued from several host threads at once, and from inside a loop, complete (the team serves one loop, the others run on threads of their own);
//   * a forked child can run loops (it gets a new team);
//   * the team's threads sleep between loops: an idle second costs (almost) no CPU time.
#include "../../psdr_jit_amd/csrc/common/threads.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <numeric>
#include <sys/resource.h>
#include <sys/wait.h>
#include <unistd.h>

static int fail(const char *what) { std::printf("FAIL %s\n", what); return 1; }

static bool covers(size_t n, size_t min_chunk, int threads) {
    std::vector<std::atomic<int>> hit(n);
    for (auto &h : hit) h.store(0);
    psdr::parallel_for(n, min_chunk, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) hit[i].fetch_add(1); }, threads);
    for (size_t i = 0; i < n; ++i) if (hit[i].load() != 1) return false;
    return true;
}

int main() {
    for (int threads : {0, 1, 2, 3, 7, 16, 40})
        for (size_t n : {(size_t) 1, (size_t) 63, (size_t) 4096, (size_t) 100003})
            if (!covers(n, 64, threads)) return fail("coverage");
    // chunk boundaries: a function of n and the thread count
    {
        std::vector<size_t> cuts1, cuts2;
        std::mutex m;
        auto run = [&](std::vector<size_t> &c) { psdr::parallel_for(100003, 64, [&](size_t b, size_t e) { std::lock_guard<std::mutex> lk(m); c.push_back(b); c.push_back(e); }, 5); std::sort(c.begin(), c.end()); };
        run(cuts1); run(cuts2);
        if (cuts1 != cuts2 || cuts1.size() != 10) return fail("chunks");
    }
    // many loops in a row (the team is re-used), sums as a serial loop gives them
    {
        std::vector<double> a(200000);
        std::iota(a.begin(), a.end(), 0.0);
        for (int rep = 0; rep < 2000; ++rep) psdr::parallel_for(a.size(), 4096, [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) a[i] += 1.0; });
        for (size_t i = 0; i < a.size(); ++i) if (a[i] != (double) i + 2000.0) return fail("repeat");
    }
    // concurrent callers and nested loops
    {
        std::atomic<long long> total{0};
        auto work = [&]() {
            for (int rep = 0; rep < 200; ++rep)
                psdr::parallel_for(50000, 1024, [&](size_t b, size_t e) {
                    long long s = 0;
                    for (size_t i = b; i < e; ++i) s += 1;
                    psdr::parallel_for(4096, 512, [&](size_t b2, size_t e2) { total.fetch_add((long long) (e2 - b2)); });      // nested: runs, on threads of its own
                    total.fetch_add(s);
                });
        };
        std::thread t1(work), t2(work), t3(work);
        t1.join(); t2.join(); t3.join();
        // every outer loop adds 50000; every outer CHUNK adds 4096 through its nested loop: at least one chunk, at most host_threads() chunks per loop
        const long long lo = 3LL * 200 * (50000 + 4096), hi = 3LL * 200 * (50000 + 4096LL * psdr::host_threads());
        if (total.load() < lo || total.load() > hi) return fail("concurrent / nested");
    }
    // a forked child
    {
        const pid_t pid = fork();
        if (pid == 0) { _exit(covers(100003, 64, 0) ? 0 : 3); }
        int st = 0;
        if (waitpid(pid, &st, 0) != pid || !WIFEXITED(st) || WEXITSTATUS(st) != 0) return fail("fork");
    }
    // idle: the team sleeps
    {
        rusage r0, r1;
        getrusage(RUSAGE_SELF, &r0);
        std::this_thread::sleep_for(std::chrono::milliseconds(500));
        getrusage(RUSAGE_SELF, &r1);
        const double cpu = (r1.ru_utime.tv_sec - r0.ru_utime.tv_sec) + 1e-6 * (r1.ru_utime.tv_usec - r0.ru_utime.tv_usec) + (r1.ru_stime.tv_sec - r0.ru_stime.tv_sec) + 1e-6 * (r1.ru_stime.tv_usec - r0.ru_stime.tv_usec);
        if (cpu > 0.05) return fail("idle team burns CPU");
    }
    std::printf("OK threads %d\n", psdr::host_threads());
    return 0;
}
