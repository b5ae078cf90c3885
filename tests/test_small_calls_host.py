"""diskann_amd/csrc/small_calls.h -- the queue and the leadership through which small dann_search_batch calls of several
threads share launches -- is host code with the launch passed in as a functor: compiled here with g++ and ThreadSanitizer
and driven by a stand-in for the device that sleeps for the length of a launch and fills every call's rows from its
query bytes and parameters.  Checked: every call gets exactly its own rows (none lost, none served twice, none mixed up),
a launch only ever holds calls with one (L, beam, k) and at most kSmallBatch queries, launches are shared when threads call
side by side, a failed launch reaches every call in it and none outside, a declined launch sends its calls to the
general path, more threads than the polling budget (the nap path) -- and ThreadSanitizer sees no data race in any of it.
The GPU side: tests/test_gpu_server.py::test_small_host_calls_of_many_threads_share_launches."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include "small_calls.h"
#include <cstring>
#include <random>
#include <vector>
using namespace dann;

static std::atomic<uint64_t> g_bad{0}, g_launches{0}, g_max_calls{0}, g_failed_calls{0}, g_declined_calls{0};
constexpr uint32_t QB = 64;                  // bytes per query
constexpr uint32_t kFailMark = 0xDEADu;      // a query whose first word is this makes its launch fail
constexpr uint32_t kDeclineMark = 0xBEEFu;   // ... makes its launch decline

static uint32_t expect_id(uint32_t qword, uint32_t L, uint32_t W, uint32_t k, uint32_t j) { return qword * 31u + L * 7u + W * 3u + k + j; }

// the stand-in for small_batch_run: checks what a launch may hold, "runs" for `us` microseconds, fills the rows
static int32_t fake_run(SmallCall* const* calls, uint32_t n, uint32_t total, std::string& text, unsigned us) {
    g_launches.fetch_add(1);
    uint64_t mx = g_max_calls.load();
    while (n > mx && !g_max_calls.compare_exchange_weak(mx, n)) {
    }
    uint32_t sum = 0;
    bool fail = false, decline = false;
    for (uint32_t c = 0; c < n; ++c) {
        const SmallCall& r = *calls[c];
        if (r.l_value != calls[0]->l_value || r.beam != calls[0]->beam || r.k != calls[0]->k) g_bad.fetch_add(1);
        if (r.done.load()) g_bad.fetch_add(1);  // a call served twice
        sum += r.nq;
        for (uint32_t i = 0; i < r.nq; ++i) {
            uint32_t w;
            memcpy(&w, (const uint8_t*)r.queries + (size_t)i * QB, 4);
            fail |= w == kFailMark;
            decline |= w == kDeclineMark;
        }
    }
    if (sum != total || total > kSmallBatch || n == 0) g_bad.fetch_add(1);
    std::this_thread::sleep_for(std::chrono::microseconds(us));
    if (decline) return kSmallCallDeclined;
    if (fail) {
        text = "boom";
        return DANN_EHIP;
    }
    for (uint32_t c = 0; c < n; ++c) {
        SmallCall& r = *calls[c];
        for (uint32_t i = 0; i < r.nq; ++i) {
            uint32_t w;
            memcpy(&w, (const uint8_t*)r.queries + (size_t)i * QB, 4);
            for (uint32_t j = 0; j < r.k; ++j) {
                r.out_ids[(size_t)i * r.k + j] = expect_id(w, r.l_value, r.beam, r.k, j);
                r.out_dists[(size_t)i * r.k + j] = (float)j;
            }
            if (r.out_stats) r.out_stats[i].cmps = w;
        }
        r.rc = DANN_OK;
    }
    return DANN_OK;
}

int main(int argc, char** argv) {
    const unsigned threads = argc > 1 ? atoi(argv[1]) : 8, rounds = argc > 2 ? atoi(argv[2]) : 200, us = argc > 3 ? atoi(argv[3]) : 60;
    const unsigned poison = argc > 4 ? atoi(argv[4]) : 0;  // one call in `poison` carries a fail / decline mark
    SmallCallQueue q;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([&, t]() {
            std::mt19937 rng(1234 + t);
            const uint32_t params[3][3] = {{20, 1, 5}, {33, 1, 10}, {64, 2, 10}};
            for (unsigned r = 0; r < rounds; ++r) {
                const uint32_t* p = params[(t + (rng() % 4 == 0 ? 1 : 0)) % 3];
                const uint32_t nq = 1 + rng() % kSmallCall, k = p[2];
                std::vector<uint8_t> qs((size_t)nq * QB);
                std::vector<uint32_t> ids((size_t)nq * k, 0xFFFFFFFFu);
                std::vector<float> d((size_t)nq * k, -1.0f);
                std::vector<dann_search_stats> st(nq);
                int mark = 0;
                if (poison && rng() % poison == 0) mark = 1 + rng() % 2;
                for (uint32_t i = 0; i < nq; ++i) {
                    uint32_t w = (t << 20) | (r << 6) | i;
                    if (mark && i == nq - 1) w = mark == 1 ? kFailMark : kDeclineMark;
                    memcpy(qs.data() + (size_t)i * QB, &w, 4);
                }
                SmallCall me;
                me.queries = qs.data();
                me.nq = nq;
                me.l_value = p[0];
                me.beam = p[1];
                me.k = k;
                me.out_ids = ids.data();
                me.out_dists = d.data();
                me.out_stats = (r & 1) ? st.data() : nullptr;
                const int32_t rc = small_call(q, me, QB, [&](SmallCall* const* calls, uint32_t n, uint32_t total, std::string& text) {
                    return fake_run(calls, n, total, text, us);
                });
                if (rc == DANN_EHIP) {
                    if (me.text != "boom") g_bad.fetch_add(1);
                    g_failed_calls.fetch_add(1);
                    continue;  // (a call that merely shared the launch of a poisoned one fails with it: one launch, one status)
                }
                if (rc == kSmallCallDeclined) {
                    g_declined_calls.fetch_add(1);
                    continue;
                }
                if (rc != DANN_OK) g_bad.fetch_add(1);
                for (uint32_t i = 0; i < nq; ++i) {
                    uint32_t w;
                    memcpy(&w, qs.data() + (size_t)i * QB, 4);
                    for (uint32_t j = 0; j < k; ++j)
                        if (ids[(size_t)i * k + j] != expect_id(w, p[0], p[1], k, j) || d[(size_t)i * k + j] != (float)j) g_bad.fetch_add(1);
                    if (me.out_stats && st[i].cmps != w) g_bad.fetch_add(1);
                }
                if (rng() % 8 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 200));
            }
        });
    for (auto& th : pool) th.join();
    const uint64_t calls = q.stats[1].load(), launches = q.stats[0].load();
    printf("bad=%llu calls=%llu launches=%llu max_calls_per_launch=%llu failed=%llu declined=%llu npending=%u head=%d leader=%d spinners=%u budget=%u\n",
           (unsigned long long)g_bad.load(), (unsigned long long)calls, (unsigned long long)launches,
           (unsigned long long)g_max_calls.load(), (unsigned long long)g_failed_calls.load(),
           (unsigned long long)g_declined_calls.load(), q.npending.load(), q.head.load() != nullptr, (int)q.leader.load(),
           small_spinners().load(), small_spin_budget());
    return g_bad.load() ? 1 : 0;
}
"""


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    d = tmp_path_factory.mktemp("small_calls")
    src = d / "harness.cpp"
    src.write_text(HARNESS)
    exes = {}
    for name, flags in (("tsan", ["-fsanitize=thread", "-O1", "-g"]), ("plain", ["-O2"])):
        exe = d / f"harness_{name}"
        subprocess.run(["g++", "-std=c++17", *flags, "-pthread", "-I", os.path.join(ROOT, "diskann_amd", "csrc"), str(src), "-o",
                        str(exe)], check=True, capture_output=True, text=True)
        exes[name] = str(exe)
    return exes


def _run(exe, *args, timeout=600):
    r = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout)
    fields = dict(kv.split("=") for kv in r.stdout.split()) if r.stdout.strip() else {}
    return r, {k: int(v) for k, v in fields.items()}


@pytest.mark.parametrize("threads,rounds,us,poison", [(8, 150, 60, 0), (24, 60, 80, 0), (6, 150, 40, 9)])
def test_combiner_under_thread_sanitizer(harness, threads, rounds, us, poison):
    r, f = _run(harness["tsan"], threads, rounds, us, poison)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:4000]
    assert r.returncode == 0, (r.stdout, r.stderr[:2000])
    assert f["bad"] == 0 and f["calls"] == threads * rounds and f["launches"] <= f["calls"]
    assert f["npending"] == 0 and f["head"] == 0 and f["leader"] == 0 and f["spinners"] == 0  # nothing left behind
    if poison:
        assert f["failed"] > 0 and f["declined"] > 0
    else:
        assert f["failed"] == 0 and f["declined"] == 0


def test_combiner_shares_launches_and_loses_no_call(harness):
    """uninstrumented build, launches of 100 us: threads in a closed loop end up in the same launches"""
    r, f = _run(harness["plain"], 16, 400, 100, 0)
    assert r.returncode == 0, (r.stdout, r.stderr[:2000])
    assert f["bad"] == 0 and f["calls"] == 16 * 400
    assert f["launches"] < f["calls"] / 2 and f["max_calls_per_launch"] >= 4, f
    assert f["npending"] == 0 and f["head"] == 0 and f["leader"] == 0 and f["spinners"] == 0
    # one thread: one launch per call
    r, f = _run(harness["plain"], 1, 200, 20, 0)
    assert r.returncode == 0 and f["bad"] == 0 and f["launches"] == f["calls"] == 200
