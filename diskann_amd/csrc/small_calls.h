// Small host-pointer search calls of several threads share launches (dann_search_batch, api.hip).  Host code only -- no
// HIP in here: the launch itself is the `run` functor the caller passes, so that tests/test_small_calls_host.py can
// compile this header with g++ and ThreadSanitizer and drive it with a stand-in for the device.
//
// A call of a few queries used to cost four runtime calls (copy in, launch, copy out, wait) on its own stream, and
// sixteen threads making such calls met in the runtime's launch path (42 k calls/s whatever the kernel does).  Now the
// queries are copied into page-locked, device-mapped staging the kernel reads directly, the results are written the
// same way, and calls of several threads that ask for the same (L, beam, k) travel in ONE launch: the first caller to
// find no leader leads -- takes every waiting call that fits, launches, waits, hands out the rows -- the others wait
// for their rows (or for the leadership, if the launch in flight left without them).  The results are those of the
// calls made one by one: a query's search does not depend on what else is in its launch.
//
// Synchronisation.  Calls queue by a compare-and-swap on `head` (sixteen threads arriving together must not put one
// another to sleep on a mutex: the first version had one, 68 k calls/s against 110 k).  The leadership is a flag taken
// by compare-and-swap and given back with a release store; `pending`, `recent` and the batch being cut belong to
// whoever holds it.  A call is owned by its caller (it lives on the caller's stack) until a leader has taken it off
// the queue and again from the moment that leader stores `done` (release) -- the leader's last access to it.
#pragma once
#include <sched.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <string>
#include <thread>

#include "../../include/dann.h"

namespace dann {
constexpr uint32_t kSmallCall = 16;              // queries per call up to which calls are combined
constexpr uint32_t kSmallBatch = 256;            // queries per combined launch
constexpr size_t kSmallStage = (size_t)1 << 20;  // the staging block of a context (h_stage)
constexpr int32_t kSmallCallDeclined = 1;        // (not a status: the caller takes the general path)

// staging bytes of `nq` queries of `qb` bytes with k results each: queries | ids | distances | statistics, 16-byte aligned
inline size_t small_call_bytes(uint32_t nq, size_t qb, uint32_t k) {
    return (((size_t)nq * qb + 15) & ~(size_t)15) + 2 * (((size_t)nq * k * 4 + 15) & ~(size_t)15) +
           (((size_t)nq * sizeof(dann_search_stats) + 15) & ~(size_t)15);
}

struct SmallCall {
    const void* queries = nullptr;
    uint32_t nq = 0, l_value = 0, beam = 0, k = 0;
    uint32_t* out_ids = nullptr;
    float* out_dists = nullptr;
    dann_search_stats* out_stats = nullptr;
    SmallCall* next = nullptr;  // (queue link)
    std::atomic<bool> done{false};
    int32_t rc = DANN_OK;
    std::string text;  // the error text that goes with rc (set_error is thread-local: the waiter repeats it)
};

struct SmallCallQueue {
    std::atomic<SmallCall*> head{nullptr};  // calls not yet seen by a leader, last arrival first
    std::atomic<uint32_t> npending{0};      // calls queued and not yet in a launch
    std::atomic<bool> leader{false};
    std::deque<SmallCall*> pending;         // (leader) calls taken off `head`, in arrival order
    // (leader) calls seen side by side lately: a leader waits a few microseconds for that many before it launches --
    // callers in lockstep come back a moment after their results
    uint32_t recent = 1;
    std::atomic<uint64_t> stats[2] = {{0}, {0}};  // launches, calls served
};

inline void small_relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// Waiting callers poll (a launch takes ~100 us: a futex sleep would double a call's latency) -- but only as many of them as
// this process has processors to spare: under a CPU quota (cgroup cpu.max; 16 of this pool's 256 hardware threads) every
// polling thread beyond it gets the whole process throttled for tens of milliseconds.  The rest nap between polls.
inline uint32_t small_spin_budget() {
    static const uint32_t budget = [] {
        uint32_t cpus = std::thread::hardware_concurrency();
        if (cpus == 0) cpus = 4;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min<uint32_t>(cpus, (uint32_t)std::max(1, CPU_COUNT(&set)));
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // "<quota> <period>" in microseconds, or "max <period>"
            long long q = 0, p = 0;
            if (fscanf(f, "%lld %lld", &q, &p) == 2 && q > 0 && p > 0)
                cpus = std::min<uint32_t>(cpus, (uint32_t)std::max<long long>(1, q / p));
            fclose(f);
        } else if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // (cgroup v1)
            long long q = 0, p = 100000;
            if (fscanf(g, "%lld", &q) != 1) q = 0;
            fclose(g);
            if (FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%lld", &p) != 1) p = 100000;
                fclose(h);
            }
            if (q > 0 && p > 0) cpus = std::min<uint32_t>(cpus, (uint32_t)std::max<long long>(1, q / p));
        }
        return std::min<uint32_t>(cpus > 1 ? cpus - 1 : 1, 32u);  // (one is the leader's)
    }();
    return budget;
}
inline std::atomic<uint32_t>& small_spinners() {
    static std::atomic<uint32_t> n{0};
    return n;
}

// the leader's turn: cut a batch off the waiting calls, run it, mark its calls.
// run(calls, n, total_queries, text) -> status of the launch as a whole (DANN_OK: every call's rc / text were set by it)
template <class Run>
void small_lead(SmallCallQueue& q, size_t qb, Run& run) {
    using clk = std::chrono::steady_clock;
    if (q.recent > 1 && q.npending.load(std::memory_order_acquire) < q.recent) {
        // callers in lockstep: the threads whose results the last launch delivered are on their way back -- a launch that
        // leaves without them makes them wait for the whole of it
        const uint32_t want = q.recent;
        const auto t0 = clk::now();
        while (q.npending.load(std::memory_order_acquire) < want && clk::now() - t0 < std::chrono::microseconds(25)) small_relax();
    }
    {   // the new arrivals, oldest first, behind the calls earlier leaders left
        SmallCall* got = q.head.exchange(nullptr, std::memory_order_acq_rel);
        SmallCall* rev = nullptr;
        while (got) {
            SmallCall* nx = got->next;
            got->next = rev;
            rev = got;
            got = nx;
        }
        for (; rev; rev = rev->next) q.pending.push_back(rev);
    }
    if (q.pending.empty()) return;
    SmallCall* batch[kSmallBatch];
    uint32_t n = 0, total = 0;
    size_t bytes = 0;
    const SmallCall& first = *q.pending.front();
    const uint32_t L = first.l_value, W = first.beam, K = first.k;
    for (auto it = q.pending.begin(); it != q.pending.end();) {
        SmallCall* r = *it;
        const size_t b = small_call_bytes(r->nq, qb, K);
        if (r->l_value == L && r->beam == W && r->k == K && total + r->nq <= kSmallBatch && bytes + b <= kSmallStage) {
            batch[n++] = r;
            total += r->nq;
            bytes += b;
            it = q.pending.erase(it);
        } else {
            ++it;
        }
    }
    q.npending.fetch_sub(n, std::memory_order_acq_rel);
    q.recent = std::max<uint32_t>(n, q.recent > 1 ? q.recent - 1 : 1);
    q.stats[0].fetch_add(1, std::memory_order_relaxed);
    q.stats[1].fetch_add(n, std::memory_order_relaxed);
    int32_t rc;
    std::string text;
    try {
        rc = run(batch, n, total, text);
    } catch (...) {
        rc = DANN_EINTERNAL;
        text = "exception in a combined small search call";
    }
    for (uint32_t c = 0; c < n; ++c) {
        if (rc != DANN_OK) {
            batch[c]->rc = rc;
            batch[c]->text = text;
        }
        batch[c]->done.store(true, std::memory_order_release);  // (the call's owner may be gone the moment this is seen)
    }
}

// one call, from queueing to its results (me.rc / me.text; kSmallCallDeclined: take the general path)
template <class Run>
int32_t small_call(SmallCallQueue& q, SmallCall& me, size_t qb, Run&& run) {
    me.next = q.head.load(std::memory_order_relaxed);
    while (!q.head.compare_exchange_weak(me.next, &me, std::memory_order_release, std::memory_order_relaxed)) {
    }
    q.npending.fetch_add(1, std::memory_order_acq_rel);
    bool spinner = false;
    for (uint32_t spins = 0; !me.done.load(std::memory_order_acquire); ++spins) {
        bool free_ = false;
        if (!q.leader.load(std::memory_order_relaxed) &&
            q.leader.compare_exchange_strong(free_, true, std::memory_order_acquire, std::memory_order_relaxed)) {
            // (a call that is not done is queued: in `head` or in `pending` -- only leaders take calls out)
            if (!me.done.load(std::memory_order_acquire)) small_lead(q, qb, run);
            q.leader.store(false, std::memory_order_release);
            continue;
        }
        // another thread leads: poll; a wait that goes on for long, or one processor too many polling, naps between polls
        if (!spinner && spins < 20000u) {
            if (small_spinners().fetch_add(1, std::memory_order_relaxed) < small_spin_budget()) spinner = true;
            else small_spinners().fetch_sub(1, std::memory_order_relaxed);
        }
        if (spinner && spins < 4096u) {
            small_relax();
        } else if (spinner && spins < 20000u) {
            std::this_thread::yield();
        } else {
            if (spinner) {
                small_spinners().fetch_sub(1, std::memory_order_relaxed);
                spinner = false;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    if (spinner) small_spinners().fetch_sub(1, std::memory_order_relaxed);
    return me.rc;
}
}  // namespace dann
