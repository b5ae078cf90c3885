// server.hip -- persistent search server: many host threads, one shared index, no kernel launch per query.
//
// The reference's serving model is N workers (tokio tasks) calling DiskANNIndex::search on one shared index
// (diskann-benchmark-core/src/search/api.rs:399-436, tokio.rs:10-14); each call is one query.  On the GPU a call per
// query would pay a kernel launch and a stream round trip each time and -- at one wavefront per query -- leave the chip
// empty.  The server keeps `workers` wavefronts resident (the PERSIST loop of the beam-search kernel, fed from a ring
// instead of a batch): dann_search_submit() copies the query into a slot of a ring in host-mapped memory and
// publishes it with one store; dann_search_wait() spins on the slot's completion word and copies the result out.
// Nothing on this path takes the index's exclusive lock or calls into the HIP runtime.
//
//   host                                       device (one launch, grid = workers + 1 wavefronts)
//   slot = a free result slot                  wave 0 (dispatcher): polls the submission ring over PCIe, 64 entries
//   ring.query[slot] = query                                        per poll, advances `avail` in device memory
//   seq = next++                               worker: seq' = head++; waits for avail > seq'
//   wait until ring[seq % ring] was taken               takes ring[seq' % ring] (-> slot), acknowledges it,
//   ring[seq % ring] = lap tag | slot  ------>          stages ring.query[slot] into device memory, runs the search
//   ...                                                  result -> ring.result[slot]; done[slot] = seq' + 1
//   spin on done[slot] == seq + 1  <--------
//   copy result out; slot back to the free list
//
// Result slots and ring positions are separate on purpose: a ring position is free again as soon as a worker has taken
// its entry, whatever its caller does with the ticket -- a caller that collects tickets late or out of order (or is
// descheduled with tickets outstanding) cannot stall other callers' submissions.  Only running out of result slots
// (`ring` tickets submitted and not waited for) makes dann_search_submit wait.
//
// The kernel leaves when the host asks (dann_server_stop) or after idle_timeout_us without a submission, so that a
// device-wide synchronisation elsewhere in the process never waits on an idle server; the next submission (or a waiter
// that sees the exit word) relaunches it.  Tickets published but not yet served survive a relaunch (`avail` and the
// ring persist; `head` is reset to `avail`, no worker is running at that point).
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <vector>

#include "dann_device.h"
#include "dann_internal.h"

using namespace dann;

struct dann_server {
    dann_server_config cfg;
    SearchCtx ctx;          // the server's own stream / spill pool
    ServerView sv;          // what the kernel gets: the device's addresses of the host ring
    ServerView hv;          // the same ring through the host's addresses (identical under unified addressing)
    uint8_t* h_block = nullptr;   // one host-mapped allocation: queries | pub | ids | dists | stats | done | ctl
    size_t h_bytes = 0;
    void* d_block = nullptr;      // one device allocation: head | avail | stop | staged queries | out ids | dists | stats
    uint32_t* d_out_ids = nullptr;
    float* d_out_d = nullptr;
    dann_search_stats* d_stats = nullptr;
    alignas(64) std::atomic<uint64_t> next{0};  // next sequence number (position in the submission ring); own cache line
    std::atomic<uint64_t>* slot_owner = nullptr;  // per result slot: the ticket outstanding on it + 1, 0 = free
    // free result slots: kFreeStacks independent Treiber stacks (slot + 1 on top, 0: empty | ABA tag << 32), each head on
    // its own cache line.  A caller thread pops from and pushes to "its" stack (by a hash of the thread) and only walks
    // on to the others when that one is empty: sixteen callers on one stack spent most of their time retrying the CAS
    // (0.8 M instead of 3.9 M queries/s at 1024 tickets in flight).
    static constexpr uint32_t kFreeStacks = 32;
    struct alignas(64) FreeStack {
        std::atomic<uint64_t> head{0};
    };
    FreeStack free_stack[kFreeStacks];
    uint32_t* free_next = nullptr;            //   next slot + 1 of each free slot
    static uint32_t home_stack() {
        static thread_local const uint32_t h =
            (uint32_t)(std::hash<std::thread::id>()(std::this_thread::get_id()) * 0x9E3779B97F4A7C15ull >> 59);
        return h & (kFreeStacks - 1u);
    }
    uint32_t take_slot() {                    // 0xFFFFFFFF if none is free right now
        const uint32_t home = home_stack();
        for (uint32_t i = 0; i < kFreeStacks; ++i) {
            std::atomic<uint64_t>& head = free_stack[(home + i) & (kFreeStacks - 1u)].head;
            uint64_t h = head.load(std::memory_order_acquire);
            while ((uint32_t)h) {
                const uint32_t top = (uint32_t)h;
                const uint64_t nh = (((h >> 32) + 1) << 32) | free_next[top - 1];
                if (head.compare_exchange_weak(h, nh, std::memory_order_acq_rel, std::memory_order_acquire)) return top - 1;
            }
        }
        return 0xFFFFFFFFu;
    }
    void give_slot(uint32_t slot, uint32_t stack) {
        std::atomic<uint64_t>& head = free_stack[stack & (kFreeStacks - 1u)].head;
        uint64_t h = head.load(std::memory_order_acquire);
        for (;;) {
            free_next[slot] = (uint32_t)h;
            const uint64_t nh = (((h >> 32) + 1) << 32) | (slot + 1u);
            if (head.compare_exchange_weak(h, nh, std::memory_order_acq_rel, std::memory_order_acquire)) return;
        }
    }
    std::mutex launch_mu;
    bool launched = false;
    std::atomic<uint64_t> relaunches{0};
    std::atomic<bool> stopping{false};  // dann_server_stop is draining the callers: every wait loop leaves
    // a submit whose ring position was not taken for kWaitLimitSeconds: the resident kernel is dead or the ring is
    // wedged.  The sequence number that submit drew can never be published, and the in-order ring cannot pass it: from
    // here on every submit and wait fails at once (DANN_EHIP) instead of after its own timeout; stop / start recovers.
    std::atomic<bool> poisoned{false};
};

namespace {
// submit / wait / poll / stats hold one for the duration of the call (see dann_index::srv_users)
struct ServerPin {
    dann_index* i;
    dann_server* s;
    explicit ServerPin(dann_index* idx) : i(idx) {
        i->srv_users.add(1);
        s = i->server.load(std::memory_order_seq_cst);
    }
    ~ServerPin() { i->srv_users.add(-1); }
    ServerPin(const ServerPin&) = delete;
    ServerPin& operator=(const ServerPin&) = delete;
};
}  // namespace

namespace {

struct DeviceGuard {
    int prev = -1, cur = -1;
    explicit DeviceGuard(int dev) : cur(dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != cur) (void)hipSetDevice(prev);
    }
};

inline uint32_t* host_u32(const uint32_t* p) { return const_cast<uint32_t*>(p); }

// (re)launch the kernel; caller holds launch_mu and the kernel is not running
int32_t launch_locked(dann_index* idx, dann_server* s) {
    DeviceGuard guard(idx->device);
    // no worker is running: the tickets drawn but never served are exactly [avail, head)
    unsigned long long avail = 0;
    DANN_HIP(hipMemcpyAsync(&avail, s->sv.d_avail, 8, hipMemcpyDeviceToHost, s->ctx.stream));
    DANN_HIP(hipStreamSynchronize(s->ctx.stream));
    DANN_HIP(hipMemcpyAsync(s->sv.d_head, &avail, 8, hipMemcpyHostToDevice, s->ctx.stream));
    DANN_HIP(hipMemsetAsync(s->sv.d_stop, 0, 4, s->ctx.stream));
    DANN_HIP(hipStreamSynchronize(s->ctx.stream));  // `avail` is a local
    __atomic_store_n(&s->hv.h_ctl[0], 0u, __ATOMIC_RELAXED);
    __atomic_store_n(&s->hv.h_ctl[1], 0u, __ATOMIC_RELEASE);
    SearchArgs a;
    a.ix = idx->view();
    a.queries = s->sv.d_q;
    a.nq = s->cfg.workers;
    a.l_value = s->cfg.l_value;
    a.beam_width = 1;
    a.k = s->cfg.k;
    a.ht_entries = auto_visited_entries(idx, s->cfg.l_value, 1);
    a.out_ids = s->d_out_ids;
    a.out_dists = s->d_out_d;
    a.stats = s->d_stats;
    a.srv = s->sv;
    int32_t rc = launch_search_server(idx, s->ctx, a);
    if (rc != DANN_OK) return rc;
    s->launched = true;
    return DANN_OK;
}

// the dispatcher has announced its exit (idle timeout): wait for the kernel to drain and start it again
int32_t relaunch_if_exited(dann_index* idx, dann_server* s) {
    if (__atomic_load_n(&s->hv.h_ctl[1], __ATOMIC_ACQUIRE) == 0u) return DANN_OK;
    std::lock_guard<std::mutex> lk(s->launch_mu);
    if (__atomic_load_n(&s->hv.h_ctl[1], __ATOMIC_ACQUIRE) == 0u) return DANN_OK;  // another caller did it
    if (__atomic_load_n(&s->hv.h_ctl[0], __ATOMIC_ACQUIRE) != 0u) return DANN_OK;  // a stop is in progress
    {
        DeviceGuard guard(idx->device);
        DANN_HIP(hipStreamSynchronize(s->ctx.stream));
    }
    s->relaunches.fetch_add(1, std::memory_order_relaxed);
    return launch_locked(idx, s);
}

}  // namespace

// A mutation of the index is about to run (the caller holds the index exclusively and no ticket is outstanding): the
// resident kernel is asked to leave and waited for.  Its waves read rows and adjacency with plain cached loads, and the
// per-XCD L2s are not coherent within a launch: a wave that stays resident across the mutation could serve a stale line of
// a row another kernel or a copy has rewritten.  The next submit finds the exit word and relaunches -- the kernel
// boundary gives the acquire / invalidate.
namespace dann {
bool server_quiesce(dann_index* idx) {
    dann_server* s = idx->server.load(std::memory_order_seq_cst);
    if (!s) return true;
    // a poisoned server's kernel may be wedged: synchronising with it here -- under the exclusive index lock -- would turn
    // a clean refusal into an unbounded hang.  The mutation is refused; dann_server_stop clears the way.
    if (s->poisoned.load(std::memory_order_acquire)) return false;
    std::lock_guard<std::mutex> lk(s->launch_mu);
    if (!s->launched || __atomic_load_n(&s->hv.h_ctl[1], __ATOMIC_ACQUIRE) != 0u) {
        // never launched, or the dispatcher has already left: wait for the workers of that launch to drain
        if (s->launched) {
            DeviceGuard guard(idx->device);
            (void)hipStreamSynchronize(s->ctx.stream);
        }
        return true;
    }
    DeviceGuard guard(idx->device);
    __atomic_store_n(&s->hv.h_ctl[0], 1u, __ATOMIC_RELEASE);  // the dispatcher polls this word, sets the exit word, leaves
    (void)hipStreamSynchronize(s->ctx.stream);
    __atomic_store_n(&s->hv.h_ctl[0], 0u, __ATOMIC_RELEASE);  // (not a stop: relaunch_if_exited may start it again)
    return true;
}
}  // namespace dann

namespace {
// a wait that lasts this long is a fault (a lost ticket, a dead kernel), not load: report it instead of spinning on
constexpr uint32_t kWaitLimitSeconds = 30;
inline bool wait_expired(uint64_t spins, std::chrono::steady_clock::time_point& began) {
    const auto now = std::chrono::steady_clock::now();
    if (began == std::chrono::steady_clock::time_point{}) {
        began = now;
        return false;
    }
    (void)spins;
    return now - began > std::chrono::seconds(kWaitLimitSeconds);
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

}  // namespace

extern "C" {

int32_t dann_server_start(dann_index* idx, const dann_server_config* cfg) try {
    if (!idx || !cfg) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);  // start / stop are rare and exclude mutations; submit / wait take no lock at all
    if (idx->server.load()) {
        set_error("dann_server_start: a server is already running on this index");
        return DANN_EINVAL;
    }
    if (cfg->l_value == 0 || cfg->k == 0 || cfg->workers == 0 || cfg->workers > 8192) {
        set_error("dann_server_start: l_value, k must be non-zero and 1 <= workers <= 8192");
        return DANN_EINVAL;
    }
    const uint32_t qbytes = idx->cfg.dtype == DT_PQ ? idx->cfg.dim * 4u : idx->layer_bytes;
    if (qbytes % 16u) {
        set_error("dann_server_start: query rows of %u bytes (the server stages 16-byte units)", qbytes);
        return DANN_EUNSUPPORTED;
    }
    if (idx->cfg.dtype == DT_PQ && (!idx->d_pq_pivots || !idx->d_pq_offsets)) {
        set_error("DANN_PQ index has no pivot table: call dann_set_pq_table first");
        return DANN_EINVAL;
    }
    uint32_t ring = cfg->ring ? cfg->ring : 4u * cfg->workers;
    uint32_t shift = 6;
    while ((1u << shift) < ring && shift < 20) ++shift;
    ring = 1u << shift;
    if (ring < 2u * cfg->workers) {
        set_error("dann_server_start: ring of %u entries is too small for %u workers", ring, cfg->workers);
        return DANN_EINVAL;
    }
    DeviceGuard guard(idx->device);
    dann_server* s = new (std::nothrow) dann_server();
    if (!s) return DANN_ENOMEM;
    auto fail = [&](int32_t rc) {
        s->ctx.destroy();
        if (s->h_block) (void)hipHostFree(s->h_block);
        if (s->d_block) (void)hipFree(s->d_block);
        delete[] s->slot_owner;
        delete[] s->free_next;
        delete s;
        return rc;
    };
    s->cfg = *cfg;
    s->cfg.ring = ring;
    if (int32_t rc = s->ctx.init()) return fail(rc);
    const uint32_t k = cfg->k, W = cfg->workers;
    // host ring
    const size_t o_q = 0, o_pub = o_q + (size_t)ring * qbytes, o_ids = o_pub + (size_t)ring * 4,
                 o_d = o_ids + (size_t)ring * k * 4, o_st = o_d + (size_t)ring * k * 4,
                 o_done = o_st + (size_t)ring * sizeof(dann_search_stats), o_ack = o_done + (size_t)ring * 4,
                 o_ctl = o_ack + (size_t)ring * 4;
    s->h_bytes = o_ctl + 64;
    hipError_t e = hipHostMalloc((void**)&s->h_block, s->h_bytes, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) return fail(hip_fail(e, "hipHostMalloc(server ring)"));
    memset(s->h_block, 0, s->h_bytes);
    uint8_t* dev_view = nullptr;  // the device's address of the host block (identical under unified addressing)
    e = hipHostGetDevicePointer((void**)&dev_view, s->h_block, 0);
    if (e != hipSuccess) return fail(hip_fail(e, "hipHostGetDevicePointer"));
    // device block: head | avail | stop | staged queries | ids | dists | stats
    const size_t d_q = 64, d_ids = d_q + (size_t)W * qbytes, d_d = d_ids + (((size_t)W * k * 4 + 15) & ~(size_t)15),
                 d_st = d_d + (((size_t)W * k * 4 + 15) & ~(size_t)15), d_total = d_st + (size_t)W * sizeof(dann_search_stats) + 16;
    e = hipMalloc(&s->d_block, d_total);
    if (e != hipSuccess) return fail(hip_fail(e, "hipMalloc(server)"));
    e = hipMemset(s->d_block, 0, d_total);
    if (e != hipSuccess) return fail(hip_fail(e, "hipMemset(server)"));
    uint8_t* db = reinterpret_cast<uint8_t*>(s->d_block);
    ServerView& sv = s->sv;
    sv.h_queries = dev_view + o_q;
    sv.h_pub = reinterpret_cast<const uint32_t*>(dev_view + o_pub);
    sv.h_res_ids = reinterpret_cast<uint32_t*>(dev_view + o_ids);
    sv.h_res_d = reinterpret_cast<float*>(dev_view + o_d);
    sv.h_res_stats = reinterpret_cast<dann_search_stats*>(dev_view + o_st);
    sv.h_done = reinterpret_cast<uint32_t*>(dev_view + o_done);
    sv.h_ack = reinterpret_cast<uint32_t*>(dev_view + o_ack);
    sv.h_ctl = reinterpret_cast<uint32_t*>(dev_view + o_ctl);
    sv.d_head = reinterpret_cast<unsigned long long*>(db);
    sv.d_avail = reinterpret_cast<unsigned long long*>(db + 8);
    sv.d_stop = reinterpret_cast<uint32_t*>(db + 16);
    sv.d_q = db + d_q;
    sv.ring = ring;
    sv.ring_shift = shift;
    sv.qstride = qbytes;
    sv.qbytes = qbytes;
    sv.workers = W;
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, idx->device) == hipSuccess && khz >= 1000)
        sv.ticks_per_us = (uint32_t)(khz / 1000);
    sv.idle_timeout_us = cfg->idle_timeout_us ? cfg->idle_timeout_us : 100000u;
    // development switch (dann_debug_set; tests: relaunches under load)
    sv.max_resident_us = std::max<uint32_t>(100u, idx->dbg_u32(DANN_DBG_SERVER_MAX_RESIDENT_US, sv.max_resident_us));
    s->d_out_ids = reinterpret_cast<uint32_t*>(db + d_ids);
    s->d_out_d = reinterpret_cast<float*>(db + d_d);
    s->d_stats = reinterpret_cast<dann_search_stats*>(db + d_st);
    s->slot_owner = new (std::nothrow) std::atomic<uint64_t>[ring];
    s->free_next = new (std::nothrow) uint32_t[ring];
    if (!s->slot_owner || !s->free_next) return fail(DANN_ENOMEM);
    for (uint32_t i = 0; i < ring; ++i) s->slot_owner[i].store(0, std::memory_order_relaxed);
    for (uint32_t i = ring; i-- > 0;) s->give_slot(i, i);  // slot i starts on stack i % kFreeStacks, low slots on top
    s->hv = sv;
    s->hv.h_queries = s->h_block + o_q;
    s->hv.h_pub = reinterpret_cast<const uint32_t*>(s->h_block + o_pub);
    s->hv.h_res_ids = reinterpret_cast<uint32_t*>(s->h_block + o_ids);
    s->hv.h_res_d = reinterpret_cast<float*>(s->h_block + o_d);
    s->hv.h_res_stats = reinterpret_cast<dann_search_stats*>(s->h_block + o_st);
    s->hv.h_done = reinterpret_cast<uint32_t*>(s->h_block + o_done);
    s->hv.h_ack = reinterpret_cast<uint32_t*>(s->h_block + o_ack);
    s->hv.h_ctl = reinterpret_cast<uint32_t*>(s->h_block + o_ctl);
    // the first launch happens under the index's exclusive lock (no search is running) and before the server is
    // published: no caller can see a server that failed to start
    int32_t rc;
    {
        std::lock_guard<std::mutex> lk(s->launch_mu);
        rc = launch_locked(idx, s);
    }
    if (rc != DANN_OK) {
        (void)hipStreamSynchronize(s->ctx.stream);
        return fail(rc);
    }
    idx->srv_outstanding.reset();
    idx->server.store(s, std::memory_order_seq_cst);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_server_stop(dann_index* idx) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    // unpublish first: a caller that pins after this sees no server; the ones already inside are waited for below
    dann_server* s = idx->server.exchange(nullptr, std::memory_order_seq_cst);
    if (!s) return DANN_OK;
    DeviceGuard guard(idx->device);
    s->stopping.store(true, std::memory_order_seq_cst);
    {
        std::lock_guard<std::mutex> lk(s->launch_mu);
        __atomic_store_n(&s->hv.h_ctl[0], 1u, __ATOMIC_RELEASE);  // the dispatcher polls this word
        if (s->launched) (void)hipStreamSynchronize(s->ctx.stream);
        s->launched = false;
    }
    // callers still inside submit / wait / poll: their loops see `stopping` and return (DANN_EINVAL: the server is
    // gone); nothing of the server is freed before the last of them has left
    while (idx->srv_users.sum() != 0) std::this_thread::yield();
    idx->srv_outstanding.reset();  // uncollected tickets die with the server
    s->ctx.destroy();
    if (s->h_block) (void)hipHostFree(s->h_block);
    if (s->d_block) (void)hipFree(s->d_block);
    delete[] s->slot_owner;
    delete[] s->free_next;
    delete s;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_search_submit(dann_index* idx, const void* query, uint64_t* ticket) try {
    if (!idx || !query || !ticket) return DANN_EINVAL;
    ServerPin pin(idx);
    dann_server* s = pin.s;
    if (!s) {
        set_error("dann_search_submit: no server on this index (dann_server_start)");
        return DANN_EINVAL;
    }
    auto gone = [&]() {
        set_error("dann_search_submit: the server is being stopped");
        return DANN_EINVAL;
    };
    auto dead = [&]() {
        set_error("dann_search_submit: the server's ring is wedged (an earlier submit timed out): dann_server_stop / "
                  "dann_server_start");
        return DANN_EHIP;
    };
    if (s->poisoned.load(std::memory_order_acquire)) return dead();
    // the "no mutation while tickets are outstanding" rule, this side: count the ticket, then look for a mutation
    idx->srv_outstanding.add(1);
    auto uncount = [&]() { idx->srv_outstanding.add(-1); };
    if (idx->mutating.load(std::memory_order_seq_cst) != 0) {
        uncount();
        set_error("dann_search_submit: the index is being mutated");
        return DANN_EBUSY;
    }
    const ServerView& sv = s->hv;
    std::chrono::steady_clock::time_point began{};
    // a result slot: only `ring` tickets submitted and not yet waited for make this wait
    uint32_t slot;
    for (uint64_t spins = 0; (slot = s->take_slot()) == 0xFFFFFFFFu; ++spins) {
        if (spins < 64) {
            cpu_relax();
        } else {
            std::this_thread::yield();
            if (s->stopping.load(std::memory_order_acquire)) return uncount(), gone();
            if (s->poisoned.load(std::memory_order_acquire)) return uncount(), dead();
            if ((spins & 1023u) == 0 && wait_expired(spins, began)) {
                uncount();
                set_error("dann_search_submit: no result slot was released within %u s (%u tickets outstanding: every ticket "
                          "must be waited for)", kWaitLimitSeconds, sv.ring);
                return DANN_EOVERFLOW;
            }
        }
    }
    memcpy(const_cast<uint8_t*>(sv.h_queries) + (size_t)slot * sv.qstride, query, sv.qbytes);
    // From here to the publication nothing may return without filling the ring position: the ring is consumed in order
    // and a sequence number that never shows up would stall every later ticket of every caller.
    const uint64_t seq = s->next.fetch_add(1, std::memory_order_relaxed);
    const uint64_t t = (seq << 20) | slot;
    s->slot_owner[slot].store(t + 1, std::memory_order_release);
    // the ring position: free once a worker has taken the previous lap's entry -- which depends on nothing but the
    // workers getting through older tickets (a relaunch included: this thread may be the one that has to do it; a
    // relaunch that fails is tried again on the next round -- a transient HIP error must not cost the ring a position)
    const uint32_t pos = (uint32_t)(seq & (sv.ring - 1u));
    const uint32_t prev_tag = (seq >> sv.ring_shift) ? server_lap_tag(seq - sv.ring, sv.ring_shift) : 0u;
    began = {};
    bool abandon = false;  // stop in progress: the position gets an entry without a query
    for (uint64_t spins = 0; __atomic_load_n(sv.h_ack + pos, __ATOMIC_ACQUIRE) != prev_tag; ++spins) {
        if (spins < 64) {
            cpu_relax();
        } else {
            if ((spins & 63u) == 0) (void)relaunch_if_exited(idx, s);
            std::this_thread::yield();
            if (s->stopping.load(std::memory_order_acquire)) {  // no worker will take anything any more: nothing to keep in order
                abandon = true;
                break;
            }
            if (s->poisoned.load(std::memory_order_acquire) || ((spins & 1023u) == 0 && wait_expired(spins, began))) {
                // a resident kernel that takes no entry for 30 s is dead (or an earlier submit already found it so)
                s->poisoned.store(true, std::memory_order_release);
                s->slot_owner[slot].store(0, std::memory_order_release);
                s->give_slot(slot, dann_server::home_stack());
                uncount();
                set_error("dann_search_submit: ring position %u was not taken by a worker within %u s; the server is "
                          "unusable until dann_server_stop / dann_server_start", pos, kWaitLimitSeconds);
                return DANN_EHIP;
            }
        }
    }
    if (abandon) {
        s->slot_owner[slot].store(0, std::memory_order_release);
        s->give_slot(slot, dann_server::home_stack());
        uncount();
        return gone();
    }
    __atomic_store_n(host_u32(sv.h_pub) + pos, (server_lap_tag(seq, sv.ring_shift) << 20) | slot, __ATOMIC_RELEASE);
    *ticket = t;
    // published: the ticket is live whatever happens now -- a kernel that left in the meantime is relaunched here or,
    // if that fails, by dann_search_wait / dann_search_poll
    (void)relaunch_if_exited(idx, s);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_search_poll(dann_index* idx, uint64_t ticket) try {
    if (!idx) return DANN_EINVAL;
    ServerPin pin(idx);
    dann_server* s = pin.s;
    if (!s) return DANN_EINVAL;
    const ServerView& sv = s->hv;
    const uint32_t slot = (uint32_t)(ticket & 0xFFFFFu);
    if (slot >= sv.ring) return DANN_EINVAL;
    if (__atomic_load_n(sv.h_done + slot, __ATOMIC_ACQUIRE) == (uint32_t)(ticket >> 20) + 1u) return 1;
    // not there yet: a caller that only polls must not wait for ever on a kernel that left on its idle timeout between
    // the submit's last look and now
    if (s->poisoned.load(std::memory_order_acquire)) {
        set_error("dann_search_poll: the server's ring is wedged: dann_server_stop / dann_server_start");
        return DANN_EHIP;
    }
    if (int32_t rc = relaunch_if_exited(idx, s)) return rc;
    return 0;
} DANN_CATCH_ALL

int32_t dann_search_wait(dann_index* idx, uint64_t ticket, uint32_t* out_ids, float* out_dists, dann_search_stats* out_stats) try {
    if (!idx || !out_ids || !out_dists) return DANN_EINVAL;
    // (the pin is dropped before the rare re-run through dann_search_batch below: that call takes the index shared and
    // must not hold up a dann_server_stop that holds it exclusively)
    std::vector<uint8_t> q;
    dann_search_stats st;
    uint32_t k, l_value;
    {
    ServerPin pin(idx);
    dann_server* s = pin.s;
    if (!s) {
        set_error("dann_search_wait: no server on this index");
        return DANN_EINVAL;
    }
    const ServerView& sv = s->hv;
    const uint32_t slot = (uint32_t)(ticket & 0xFFFFFu);
    const uint32_t expect = (uint32_t)(ticket >> 20) + 1u;
    if (slot >= sv.ring || s->slot_owner[slot].load(std::memory_order_acquire) != ticket + 1) {
        set_error("dann_search_wait: ticket %llu is not outstanding (already waited for, or never submitted)",
                  (unsigned long long)ticket);
        return DANN_EINVAL;
    }
    std::chrono::steady_clock::time_point began{};
    for (uint64_t spins = 0; __atomic_load_n(sv.h_done + slot, __ATOMIC_ACQUIRE) != expect; ++spins) {
        if (spins < 256) {
            cpu_relax();
        } else {
            if ((spins & 63u) == 0) (void)relaunch_if_exited(idx, s);  // (a failed relaunch is tried again; the limit below ends it)
            std::this_thread::yield();
            if (s->stopping.load(std::memory_order_acquire)) {
                set_error("dann_search_wait: the server is being stopped");
                return DANN_EINVAL;
            }
            if (s->poisoned.load(std::memory_order_acquire) || ((spins & 1023u) == 0 && wait_expired(spins, began))) {
                set_error("dann_search_wait: no answer for ticket %llu within %u s (or the server's ring is wedged): "
                          "dann_server_stop / dann_server_start", (unsigned long long)ticket, kWaitLimitSeconds);
                // the ticket is given up: it no longer counts as outstanding (mutations of the index are not refused on
                // its account).  Its result slot stays retired -- a late answer may still land in it.  A wait that ran into
                // the limit means the resident kernel is not answering: the server is poisoned (every later submit / wait /
                // poll fails at once instead of after its own 30 s, and server_quiesce does not wait for a kernel that may
                // never leave); dann_server_stop / dann_server_start recovers.
                s->poisoned.store(true, std::memory_order_release);
                s->slot_owner[slot].store(0, std::memory_order_release);
                idx->srv_outstanding.add(-1);
                return DANN_EHIP;
            }
        }
    }
    k = s->cfg.k;
    l_value = s->cfg.l_value;
    memcpy(out_ids, sv.h_res_ids + (size_t)slot * k, (size_t)k * 4);
    memcpy(out_dists, sv.h_res_d + (size_t)slot * k, (size_t)k * 4);
    st = sv.h_res_stats[slot];
    if (st.status) q.assign(sv.h_queries + (size_t)slot * sv.qstride, sv.h_queries + (size_t)slot * sv.qstride + sv.qbytes);
    s->slot_owner[slot].store(0, std::memory_order_release);
    s->give_slot(slot, dann_server::home_stack());
    idx->srv_outstanding.add(-1);
    }
    if (st.status) {
        // the resident waves carry a fixed LDS visited table: the rare query that outgrows it (and the spill pool) is
        // re-run through the launch path, which retries with larger tables
        int32_t rc = dann_search_batch(idx, q.data(), 1, l_value, 1, k, out_ids, out_dists, &st);
        if (out_stats) *out_stats = st;
        return rc;
    }
    if (out_stats) *out_stats = st;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_server_stats(dann_index* idx, uint64_t* submitted, uint64_t* relaunches) try {
    if (!idx) return DANN_EINVAL;
    ServerPin pin(idx);
    if (!pin.s) return DANN_EINVAL;
    if (submitted) *submitted = pin.s->next.load(std::memory_order_relaxed);
    if (relaunches) *relaunches = pin.s->relaunches.load(std::memory_order_relaxed);
    return DANN_OK;
} DANN_CATCH_ALL

// ---- measurement harness: `threads` host threads, each issuing single-query calls on the shared index ---------------
// mode 0: dann_search_batch(nq = 1) per call (a kernel launch per query, concurrent on the context pool);
// mode 1: dann_search_submit / dann_search_wait with up to `depth` tickets outstanding per thread (depth 1 = strictly
// synchronous calls).  Thread t serves queries t, t + threads, ... of the set.  Latency of a query = submit -> result in
// the caller's buffer, host clock.
int32_t dann_debug_concurrent_callers(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value, uint32_t k,
                                      uint32_t threads, uint32_t mode, uint32_t depth, uint32_t* out_ids, float* out_dists,
                                      float* out_latency_us, double* out_seconds) try {
    if (!idx || !queries || !out_ids || !out_dists || threads == 0 || nq == 0 || k == 0 || mode > 1) return DANN_EINVAL;
    if (mode == 1) {
        ServerPin pin(idx);
        if (!pin.s || pin.s->cfg.k != k) {
            set_error("mode 1 needs a running server with the same k");
            return DANN_EINVAL;
        }
    }
    if (depth == 0) depth = 1;
    const size_t qb = idx->cfg.dtype == DT_PQ ? (size_t)idx->cfg.dim * 4 : idx->layer_bytes;
    std::atomic<int32_t> status{DANN_OK};
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false};
    using clk = std::chrono::steady_clock;
    auto worker = [&](uint32_t tid) {
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) cpu_relax();
        const uint8_t* qs = reinterpret_cast<const uint8_t*>(queries);
        if (mode == 0) {
            for (uint32_t q = tid; q < nq; q += threads) {
                const auto t0 = clk::now();
                int32_t rc = dann_search_batch(idx, qs + (size_t)q * qb, 1, l_value, 1, k, out_ids + (size_t)q * k,
                                               out_dists + (size_t)q * k, nullptr);
                if (rc != DANN_OK) status.store(rc);
                if (out_latency_us) out_latency_us[q] = std::chrono::duration<float, std::micro>(clk::now() - t0).count();
            }
            return;
        }
        std::vector<uint64_t> tick(depth);
        std::vector<uint32_t> qid(depth);
        std::vector<clk::time_point> t0(depth);
        uint32_t head = 0, tail = 0;  // outstanding tickets of this thread: [tail, head)
        auto collect = [&]() {
            const uint32_t i = tail % depth;
            int32_t rc = dann_search_wait(idx, tick[i], out_ids + (size_t)qid[i] * k, out_dists + (size_t)qid[i] * k, nullptr);
            if (rc != DANN_OK) status.store(rc);
            if (out_latency_us) out_latency_us[qid[i]] = std::chrono::duration<float, std::micro>(clk::now() - t0[i]).count();
            ++tail;
        };
        for (uint32_t q = tid; q < nq; q += threads) {
            if (head - tail == depth) collect();
            const uint32_t i = head % depth;
            qid[i] = q;
            t0[i] = clk::now();
            int32_t rc = dann_search_submit(idx, qs + (size_t)q * qb, &tick[i]);
            if (rc != DANN_OK) {
                status.store(rc);
                break;
            }
            ++head;
        }
        while (tail != head) collect();
    };
    std::vector<std::thread> pool;
    pool.reserve(threads);
    for (uint32_t t = 0; t < threads; ++t) pool.emplace_back(worker, t);
    while (ready.load() != threads) std::this_thread::yield();
    const auto start = clk::now();
    go.store(true, std::memory_order_release);
    for (auto& th : pool) th.join();
    if (out_seconds) *out_seconds = std::chrono::duration<double>(clk::now() - start).count();
    return status.load();
} DANN_CATCH_ALL

}  // extern "C"
