// sharded.hip -- multi-GPU behind the C ABI: communicators (RCCL over xGMI, an in-process form for one process that
// drives several devices, or caller-supplied callbacks), the sharded index build and the sharded search.
//
// Reference seams: DiskANNIndex::multi_insert and its only exchange point (diskann/src/graph/index.rs:815-1030,
// :911-1024) for the build; the query partition of the benchmark runner
// (diskann-benchmark-core/src/search/api.rs:399-436, diskann/src/utils/async_tools.rs:289-365) for the search.
//
// Layout: every rank holds a byte-identical replica of the index in its own HBM.  Search shards with no data-path
// collective (each rank searches its partition of the query block; the k results per query are all-gathered so that
// every rank returns the whole block).  The build splits every multi_insert batch at its exchange point: each rank
// generates the candidates (insert search + RobustPrune) for its partition of the batch positions, the pending
// adjacency rows are all-gathered, every rank applies the same graph update -- with the expensive part of that update
// (the prunes of overflowing back-edge targets) done only by the target's owner (id % world) and a second, small
// all-gather of the rewritten rows.  Both exchanges are ncclAllGather calls on device buffers, issued from here.
#include <dlfcn.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "dann_device.h"
#include "dann_internal.h"

using namespace dann;

namespace dann {
bool build_bootstrap_too_big(dann_index* idx);  // build_kernels.hip
}

// ---- communicators ----------------------------------------------------------------------------------------------------
namespace {

// the handful of RCCL entry points used, bound at run time (librccl is not a link-time dependency of the library: a
// single-GPU host never loads it)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, dann_rccl_unique_id, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;  // why the library is unusable (dlerror() is read once, when the load fails)
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
            const char* e = dlerror();
            if (e) r.why = e;
        }
        if (!r.lib) {
            if (r.why.empty()) r.why = "librccl.so not found";
            return;
        }
        r.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<int (*)(void**, int, dann_rccl_unique_id, int)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(r.lib, "ncclAllGather"));
        r.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
        if (!r.ok) r.why = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
    });
    return r;
}

int32_t rccl_fail(int rc, const char* what) {
    Rccl& r = rccl();
    set_error("RCCL error %d (%s) in %s", rc, r.GetErrorString ? r.GetErrorString(rc) : "?", what);
    return DANN_EHIP;
}

// one process, several devices: the ranks are threads of this process.  all_gather = every rank publishes its send
// buffer, a barrier, every rank copies every block into its own receive buffer (hipMemcpyPeerAsync: device to device
// over xGMI, or a plain device copy when two ranks share a device), a second barrier before the send buffers are reused.
struct LocalGroup {
    uint32_t world = 0;
    std::mutex mu;
    std::condition_variable cv;
    uint32_t arrived = 0;
    uint64_t generation = 0;
    std::vector<const void*> send;
    std::vector<int> device;
    bool failed = false;  // a rank left a collective sequence with an error: the others must not wait for it
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || failed; });
        }
        return !failed;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};

}  // namespace

struct dann_comm {
    uint32_t rank = 0, world = 1;
    int kind = 0;  // 0 callbacks, 1 RCCL, 2 in-process
    dann_comm_ops ops{};
    void* nccl = nullptr;
    int device = 0;
    bool in_sequence = false;  // this rank has entered a collective of the current call (see AbortOnError)
    std::shared_ptr<LocalGroup> group;
    // all-gather of `bytes` bytes per rank on device buffers, complete when it returns
    int32_t all_gather(const void* d_send, void* d_recv, uint64_t bytes, hipStream_t stream) {
        in_sequence = true;
        if (world == 1 && kind != 1) {  // (a world-1 RCCL communicator still goes through ncclAllGather: pre-flight)
            if (d_send != d_recv) DANN_HIP(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, stream));
            DANN_HIP(hipStreamSynchronize(stream));
            return DANN_OK;
        }
        if (kind == 0) {
            int32_t rc = ops.all_gather(ops.ctx, d_send, d_recv, bytes, stream);
            if (rc != DANN_OK) {
                set_error("communicator callback all_gather failed with status %d", rc);
                return rc < 0 ? rc : DANN_EHIP;
            }
            DANN_HIP(hipStreamSynchronize(stream));
            return DANN_OK;
        }
        if (kind == 1) {
            int rc = rccl().AllGather(d_send, d_recv, (size_t)bytes, /*ncclInt8*/ 0, nccl, stream);
            if (rc != 0) return rccl_fail(rc, "ncclAllGather");
            DANN_HIP(hipStreamSynchronize(stream));
            return DANN_OK;
        }
        LocalGroup& g = *group;
        g.send[rank] = d_send;
        if (!g.barrier()) {
            set_error("in-process communicator: another rank failed");
            return DANN_EHIP;
        }
        for (uint32_t r = 0; r < world; ++r) {
            void* dst = reinterpret_cast<uint8_t*>(d_recv) + (size_t)r * bytes;
            if (g.device[r] == device) DANN_HIP(hipMemcpyAsync(dst, g.send[r], bytes, hipMemcpyDeviceToDevice, stream));
            else DANN_HIP(hipMemcpyPeerAsync(dst, device, g.send[r], g.device[r], bytes, stream));
        }
        DANN_HIP(hipStreamSynchronize(stream));
        if (!g.barrier()) {
            set_error("in-process communicator: another rank failed");
            return DANN_EHIP;
        }
        return DANN_OK;
    }
    void abort() {
        if (kind == 2 && group) group->abort();
    }
};

namespace {
// A rank that leaves a collective sequence with an error releases the ranks waiting for it (in-process communicators;
// the group stays failed: its ranks are out of step for good).  The one exception: an error found by the checks of the
// call's own arguments that every rank passes alike by contract (null output pointers, k = 0, a bad growth factor ...),
// before anything rank-specific is looked at -- `past_args` still false -- is the same on every rank and nobody waits
// for anybody: the communicator stays usable.  The same holds for what depends only on the configuration the replicas
// share by contract (a range beyond the capacity, an invalid build configuration, L = 0).  Everything later (a null or
// foreign index on ONE rank, a per-rank dtype mismatch, an unsupported configuration of one device, an exception)
// releases the peers.
struct AbortOnError {
    dann_comm* c;
    int32_t rc = DANN_OK;
    bool past_args = false;
    explicit AbortOnError(dann_comm* comm) : c(comm) { c->in_sequence = false; }
    ~AbortOnError() {
        if (rc < 0 && (c->in_sequence || past_args)) c->abort();
    }
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

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) {
        if (p) (void)hipFree(p);
        p = nullptr;
        return hipMalloc(&p, n ? n : 1);
    }
    template <class T>
    T* as() {
        return reinterpret_cast<T*>(p);
    }
};

// diskann/src/utils/async_tools.rs:289-365: contiguous ranges that differ in length by at most one
void partition(uint32_t nitems, uint32_t ntasks, uint32_t task, uint32_t* lo, uint32_t* hi) {
    const uint32_t k = nitems / ntasks, m = nitems % ntasks;
    if (task >= m) {
        *lo = m * (k + 1) + (task - m) * k;
        *hi = *lo + k;
    } else {
        *lo = task * (k + 1);
        *hi = *lo + k + 1;
    }
}

}  // namespace

extern "C" {

int32_t dann_comm_create_callbacks(const dann_comm_ops* ops, dann_comm** out) try {
    if (!ops || !out || !ops->all_gather || ops->world == 0 || ops->rank >= ops->world) return DANN_EINVAL;
    dann_comm* c = new (std::nothrow) dann_comm();
    if (!c) return DANN_ENOMEM;
    c->rank = ops->rank;
    c->world = ops->world;
    c->kind = 0;
    c->ops = *ops;
    *out = c;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_comm_rccl_unique_id(dann_rccl_unique_id* out) try {
    if (!out) return DANN_EINVAL;
    Rccl& r = rccl();
    if (!r.ok) {
        set_error("librccl could not be loaded (%s)", r.why.c_str());
        return DANN_EUNSUPPORTED;
    }
    int rc = r.GetUniqueId(out);
    if (rc != 0) return rccl_fail(rc, "ncclGetUniqueId");
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_comm_create_rccl(const dann_rccl_unique_id* id, uint32_t rank, uint32_t world, int32_t device, dann_comm** out) try {
    if (!id || !out || world == 0 || rank >= world) return DANN_EINVAL;
    Rccl& r = rccl();
    if (!r.ok) {
        set_error("librccl could not be loaded (%s)", r.why.c_str());
        return DANN_EUNSUPPORTED;
    }
    if (device < 0) DANN_HIP(hipGetDevice(&device));
    DeviceGuard guard(device);
    dann_comm* c = new (std::nothrow) dann_comm();
    if (!c) return DANN_ENOMEM;
    c->rank = rank;
    c->world = world;
    c->kind = 1;
    c->device = device;
    int rc = r.CommInitRank(&c->nccl, (int)world, *id, (int)rank);
    if (rc != 0) {
        delete c;
        return rccl_fail(rc, "ncclCommInitRank");
    }
    *out = c;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_comm_create_local(const int32_t* devices, uint32_t world, dann_comm** out) try {
    if (!devices || !out || world == 0) return DANN_EINVAL;
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    g->send.assign(world, nullptr);
    g->device.assign(devices, devices + world);
    for (uint32_t r = 0; r < world; ++r) out[r] = nullptr;
    for (uint32_t r = 0; r < world; ++r) {
        dann_comm* c = new (std::nothrow) dann_comm();
        if (!c) {
            for (uint32_t q = 0; q < r; ++q) delete out[q];
            return DANN_ENOMEM;
        }
        c->rank = r;
        c->world = world;
        c->kind = 2;
        c->device = devices[r];
        c->group = g;
        out[r] = c;
    }
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_comm_destroy(dann_comm* c) try {
    if (!c) return DANN_OK;
    if (c->kind == 1 && c->nccl) {
        DeviceGuard guard(c->device);
        (void)rccl().CommDestroy(c->nccl);
    }
    delete c;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_comm_rank(const dann_comm* c) { return c ? (int32_t)c->rank : DANN_EINVAL; }
int32_t dann_comm_world(const dann_comm* c) { return c ? (int32_t)c->world : DANN_EINVAL; }

int32_t dann_comm_all_gather_device(dann_comm* c, int32_t device, const void* d_send, void* d_recv, uint64_t bytes) try {
    if (!c || !d_send || !d_recv) return DANN_EINVAL;
    if (device < 0) DANN_HIP(hipGetDevice(&device));
    DeviceGuard guard(device);
    hipStream_t st = nullptr;
    DANN_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    int32_t rc = c->all_gather(d_send, d_recv, bytes, st);
    (void)hipStreamDestroy(st);
    return rc;
} DANN_CATCH_ALL

int32_t dann_memcpy_device(int32_t device, void* dst, const void* src, uint64_t bytes, int32_t kind) try {
    if ((!dst || !src) && bytes) return DANN_EINVAL;
    if (kind < 0 || kind > 2) return DANN_EINVAL;
    if (device < 0) DANN_HIP(hipGetDevice(&device));
    DeviceGuard guard(device);
    const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    DANN_HIP(hipMemcpy(dst, src, bytes, k));
    return DANN_OK;
} DANN_CATCH_ALL

// ---- sharded build ----------------------------------------------------------------------------------------------------
static int32_t build_sharded_impl(dann_index* idx, dann_comm* comm, const dann_build_config* cfg, uint32_t first, uint32_t n,
                                  float growth, uint32_t max_batch, uint64_t* stats, bool* past_args) {
    if (!comm || !cfg) return DANN_EINVAL;
    if (!(growth > 0.0f) || max_batch == 0) return DANN_EINVAL;
    // (the replicas of one sharded build share one dann_config by contract -- dann.h -- so a range beyond the capacity
    // or an invalid build configuration is every rank's error alike: plain argument errors, the communicator stays usable)
    if (idx && (uint64_t)first + n > idx->cfg.capacity) return DANN_EBOUNDS;
    if (idx && (cfg->pruned_degree == 0 || cfg->l_build == 0 || cfg->max_degree < cfg->pruned_degree ||
                cfg->max_degree > idx->cfg.max_degree || !(cfg->alpha >= 1.0f))) {
        set_error("invalid build config (pruned_degree %u, max_degree %u (provider %u), l_build %u, alpha %g)",
                  cfg->pruned_degree, cfg->max_degree, idx->cfg.max_degree, cfg->l_build, (double)cfg->alpha);
        return DANN_EINVAL;
    }
    *past_args = true;  // (from here on an error may be this rank's alone: AbortOnError releases the peers)
    if (!idx) return DANN_EINVAL;
    const uint32_t world = comm->world, rank = comm->rank;
    DeviceGuard guard(idx->device);
    const uint32_t width = cfg->pruned_degree + 1, rw = idx->cfg.max_degree + 2;
    const uint32_t bmax = std::min(max_batch, std::max<uint32_t>(n, 1u));
    uint32_t lo0, hi0;
    partition(bmax, world, 0, &lo0, &hi0);
    const size_t longest_max = std::max<uint32_t>(hi0 - lo0, 1u);
    // buffers sized once for the largest batch (the library writes the rows it reports; nothing is zero-filled per batch)
    DevBuf mine, gathered, pending, rows_mine, rows_all, counts;
    DANN_HIP(mine.alloc(longest_max * width * 4));
    DANN_HIP(gathered.alloc((size_t)world * longest_max * width * 4));
    DANN_HIP(pending.alloc((size_t)bmax * width * 4));
    DANN_HIP(counts.alloc((size_t)(world + 1) * 4));
    size_t rows_cap = 0, rows_all_cap = 0;  // rows (grow-only)
    hipStream_t st = idx->main.stream;
    uint64_t st_rounds = 0, st_bytes = 0, st_rows = 0, st_row_bytes = 0;
    std::vector<uint32_t> slots, h_counts(world);
    uint32_t done = 0, limit = max_batch;
    int32_t batches = 0;
    while (done < n) {
        uint32_t b = (uint32_t)std::ceil((double)(first + done) * (double)growth);
        b = std::max<uint32_t>(1, std::min(b, limit));
        b = std::min(b, n - done);
        slots.resize(b);
        for (uint32_t i = 0; i < b; ++i) slots[i] = first + done + i;
        uint32_t lo, hi, l0, h0;
        partition(b, world, rank, &lo, &hi);
        partition(b, world, 0, &l0, &h0);
        const uint32_t longest = std::max<uint32_t>(h0 - l0, 1u);
        // (1) candidates for this rank's positions
        int32_t rc = dann_insert_batch_candidates(idx, cfg, slots.data(), b, lo, hi, mine.as<uint32_t>());
        if (rc != DANN_OK) return rc;
        // (2) the build's exchange: pending rows of the whole batch
        const uint32_t* d_pending = mine.as<uint32_t>();
        if (world > 1) {
            rc = comm->all_gather(mine.p, gathered.p, (uint64_t)longest * width * 4, st);
            if (rc != DANN_OK) return rc;
            for (uint32_t r = 0; r < world; ++r) {  // shards are padded to `longest` rows: pack them in batch order
                uint32_t a, z;
                partition(b, world, r, &a, &z);
                if (z > a)
                    DANN_HIP(hipMemcpyAsync(pending.as<uint32_t>() + (size_t)a * width,
                                            gathered.as<uint32_t>() + (size_t)r * longest * width, (size_t)(z - a) * width * 4,
                                            hipMemcpyDeviceToDevice, st));
            }
            DANN_HIP(hipStreamSynchronize(st));
            d_pending = pending.as<uint32_t>();
            st_rounds += 1;
            st_bytes += (uint64_t)world * longest * width * 4;
        }
        // (3) graph update; the prunes of overflowing targets only on their owner
        uint32_t cnt = 0;
        if (world > 1) {
            const size_t need = (size_t)b * cfg->pruned_degree + 1;  // at most one row per distinct back-edge target
            if (rows_cap < need) {
                rows_cap = 0;
                DANN_HIP(rows_mine.alloc(need * rw * 4));
                rows_cap = need;
            }
            rc = dann_insert_batch_commit_part(idx, cfg, slots.data(), b, d_pending, rank, world, rows_mine.as<uint32_t>(),
                                               (uint32_t)std::min<size_t>(rows_cap, 0xFFFFFFFFu), &cnt);
        } else {
            rc = dann_insert_batch_commit(idx, cfg, slots.data(), b, d_pending);
        }
        if (rc == DANN_EUNSUPPORTED && build_bootstrap_too_big(idx) && b > 1) {
            // the bootstrap test (index.rs:926-931) ran on identical data on every rank: all of them take this branch
            limit = std::max<uint32_t>(1, b / 2);
            continue;
        }
        if (rc != DANN_OK) return rc;
        if (world > 1) {  // (4) second exchange: the rows each owner rewrote (counts, then rows padded to the longest list)
            DANN_HIP(hipMemcpyAsync(counts.as<uint32_t>() + world, &cnt, 4, hipMemcpyHostToDevice, st));
            DANN_HIP(hipStreamSynchronize(st));
            rc = comm->all_gather(counts.as<uint32_t>() + world, counts.p, 4, st);
            if (rc != DANN_OK) return rc;
            DANN_HIP(hipMemcpyAsync(h_counts.data(), counts.p, (size_t)world * 4, hipMemcpyDeviceToHost, st));
            DANN_HIP(hipStreamSynchronize(st));
            const uint32_t longest_rows = *std::max_element(h_counts.begin(), h_counts.end());
            if (longest_rows) {
                if (rows_cap < longest_rows) {  // this rank rewrote fewer rows than another: the send buffer is read that far
                    DevBuf grown;
                    DANN_HIP(grown.alloc((size_t)longest_rows * rw * 4));
                    if (cnt) DANN_HIP(hipMemcpy(grown.p, rows_mine.p, (size_t)cnt * rw * 4, hipMemcpyDeviceToDevice));
                    std::swap(grown.p, rows_mine.p);
                    rows_cap = longest_rows;
                }
                const size_t need_all = (size_t)world * longest_rows;
                if (rows_all_cap < need_all) {
                    rows_all_cap = 0;
                    DANN_HIP(rows_all.alloc(need_all * rw * 4));
                    rows_all_cap = need_all;
                }
                rc = comm->all_gather(rows_mine.p, rows_all.p, (uint64_t)longest_rows * rw * 4, st);
                if (rc != DANN_OK) return rc;
                for (uint32_t r = 0; r < world; ++r)
                    if (r != rank && h_counts[r]) {
                        rc = dann_apply_neighbor_rows_device(idx, rows_all.as<uint32_t>() + (size_t)r * longest_rows * rw, h_counts[r]);
                        if (rc != DANN_OK) return rc;
                    }
                st_row_bytes += (uint64_t)world * longest_rows * rw * 4;
                for (uint32_t r = 0; r < world; ++r) st_rows += h_counts[r];
            }
        }
        limit = std::min<uint32_t>(max_batch, limit * 2 > limit ? limit * 2 : limit);
        done += b;
        ++batches;
    }
    if (stats) {
        stats[0] = st_rounds;
        stats[1] = st_bytes;
        stats[2] = st_rows;
        stats[3] = st_row_bytes;
    }
    return batches;
}

int32_t dann_build_sharded(dann_index* idx, dann_comm* comm, const dann_build_config* cfg, uint32_t first, uint32_t n,
                           float growth, uint32_t max_batch, uint64_t* stats) try {
    if (!comm) return DANN_EINVAL;
    AbortOnError guard(comm);
    try {
        guard.rc = build_sharded_impl(idx, comm, cfg, first, n, growth, max_batch, stats, &guard.past_args);
    } catch (...) {
        guard.rc = DANN_EINVAL;
        guard.past_args = true;
        throw;
    }
    return guard.rc;
} DANN_CATCH_ALL

// ---- sharded search: every rank searches its partition of the block, the results are all-gathered -------------------
static int32_t search_sharded_impl(dann_index* idx, dann_comm* comm, const void* queries, uint32_t nq, uint32_t l_value,
                                   uint32_t beam_width, uint32_t k, uint32_t* out_ids, float* out_dists, bool* past_args) {
    if (!comm) return DANN_EINVAL;
    if (nq == 0) return DANN_OK;
    if (!queries || !out_ids || !out_dists || k == 0) return DANN_EINVAL;
    if (l_value == 0 || beam_width == 0) {  // (KnnSearchError, knn_search.rs:27-33: the same on every rank)
        set_error("l_value and beam_width must be non-zero (KnnSearchError, knn_search.rs:27-33)");
        return DANN_EINVAL;
    }
    *past_args = true;  // (from here on an error may be this rank's alone: AbortOnError releases the peers)
    if (!idx) return DANN_EINVAL;
    const uint32_t world = comm->world, rank = comm->rank;
    const size_t qb = idx->cfg.dtype == DT_PQ ? (size_t)idx->cfg.dim * 4 : idx->layer_bytes;
    uint32_t lo, hi, l0, h0;
    partition(nq, world, rank, &lo, &hi);
    partition(nq, world, 0, &l0, &h0);
    const uint32_t longest = h0 - l0;
    // this rank's rows: [ids | distance bits] per query, padded to the longest partition
    std::vector<uint32_t> mine((size_t)longest * 2 * k, 0xFFFFFFFFu);
    if (hi > lo) {
        std::vector<uint32_t> ids((size_t)(hi - lo) * k);
        std::vector<float> dd((size_t)(hi - lo) * k);
        int32_t rc = dann_search_batch(idx, reinterpret_cast<const uint8_t*>(queries) + (size_t)lo * qb, hi - lo, l_value,
                                       beam_width, k, ids.data(), dd.data(), nullptr);
        if (rc != DANN_OK) return rc;
        for (uint32_t i = 0; i < hi - lo; ++i) {
            memcpy(&mine[(size_t)i * 2 * k], &ids[(size_t)i * k], (size_t)k * 4);
            memcpy(&mine[(size_t)i * 2 * k + k], &dd[(size_t)i * k], (size_t)k * 4);
        }
    }
    if (world == 1) {
        for (uint32_t i = 0; i < nq; ++i) {
            memcpy(out_ids + (size_t)i * k, &mine[(size_t)i * 2 * k], (size_t)k * 4);
            memcpy(out_dists + (size_t)i * k, &mine[(size_t)i * 2 * k + k], (size_t)k * 4);
        }
        return DANN_OK;
    }
    DeviceGuard guard(idx->device);
    const size_t shard_b = (size_t)longest * 2 * k * 4;
    DevBuf send, recv;
    DANN_HIP(send.alloc(shard_b));
    DANN_HIP(recv.alloc(shard_b * world));
    hipStream_t st = nullptr;
    DANN_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamDestroy(s); }
    } sg{st};
    DANN_HIP(hipMemcpyAsync(send.p, mine.data(), shard_b, hipMemcpyHostToDevice, st));
    DANN_HIP(hipStreamSynchronize(st));
    int32_t rc = comm->all_gather(send.p, recv.p, shard_b, st);
    if (rc != DANN_OK) return rc;
    std::vector<uint32_t> all((size_t)world * longest * 2 * k);
    DANN_HIP(hipMemcpyAsync(all.data(), recv.p, shard_b * world, hipMemcpyDeviceToHost, st));
    DANN_HIP(hipStreamSynchronize(st));
    for (uint32_t r = 0; r < world; ++r) {
        uint32_t a, z;
        partition(nq, world, r, &a, &z);
        for (uint32_t i = a; i < z; ++i) {
            const uint32_t* row = &all[((size_t)r * longest + (i - a)) * 2 * k];
            memcpy(out_ids + (size_t)i * k, row, (size_t)k * 4);
            memcpy(out_dists + (size_t)i * k, row + k, (size_t)k * 4);
        }
    }
    return DANN_OK;
}

int32_t dann_search_sharded(dann_index* idx, dann_comm* comm, const void* queries, uint32_t nq, uint32_t l_value,
                            uint32_t beam_width, uint32_t k, uint32_t* out_ids, float* out_dists) try {
    if (!comm) return DANN_EINVAL;
    AbortOnError guard(comm);
    try {
        guard.rc = search_sharded_impl(idx, comm, queries, nq, l_value, beam_width, k, out_ids, out_dists, &guard.past_args);
    } catch (...) {
        guard.rc = DANN_EINVAL;
        guard.past_args = true;
        throw;
    }
    return guard.rc;
} DANN_CATCH_ALL

// ---- one process, several devices: dann_multi ------------------------------------------------------------------------
}  // extern "C"

struct dann_multi {
    std::vector<dann_index*> replica;
    std::vector<dann_comm*> comm;
    std::vector<int32_t> device;
};

namespace {
// run f(rank) on one host thread per replica (each thread makes its device current through the entry points it calls)
template <class F>
int32_t for_each_rank(uint32_t world, F&& f) {
    std::vector<int32_t> rc(world, DANN_OK);
    std::vector<std::string> msg(world);
    std::vector<std::thread> th;
    th.reserve(world);
    for (uint32_t r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            rc[r] = f(r);
            if (rc[r] < 0) {
                char buf[512];
                dann_last_error(buf, sizeof(buf));
                msg[r] = buf;
            }
        });
    for (auto& t : th) t.join();
    for (uint32_t r = 0; r < world; ++r)
        if (rc[r] < 0) {
            set_error("rank %u: %s", r, msg[r].c_str());
            return rc[r];
        }
    return rc[0];
}
}  // namespace

extern "C" {

int32_t dann_multi_create(const dann_config* cfg, const void* start_rows, uint64_t start_len, const int32_t* devices,
                          uint32_t ndev, dann_multi** out) try {
    if (!cfg || !devices || !out || ndev == 0) return DANN_EINVAL;
    *out = nullptr;
    std::unique_ptr<dann_multi> m(new dann_multi());
    m->device.assign(devices, devices + ndev);
    m->replica.assign(ndev, nullptr);
    m->comm.assign(ndev, nullptr);
    auto cleanup = [&]() {
        for (dann_index* r : m->replica) (void)dann_index_destroy(r);
        for (dann_comm* c : m->comm) (void)dann_comm_destroy(c);
    };
    for (uint32_t r = 0; r < ndev; ++r) {
        dann_config c = *cfg;
        c.device = devices[r];
        int32_t rc = dann_index_create(&c, start_rows, start_len, &m->replica[r]);
        if (rc != DANN_OK) {
            cleanup();
            return rc;
        }
    }
    int32_t rc = dann_comm_create_local(devices, ndev, m->comm.data());
    if (rc != DANN_OK) {
        cleanup();
        return rc;
    }
    *out = m.release();
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_multi_destroy(dann_multi* m) try {
    if (!m) return DANN_OK;
    for (dann_index* r : m->replica) (void)dann_index_destroy(r);
    for (dann_comm* c : m->comm) (void)dann_comm_destroy(c);
    delete m;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_multi_size(const dann_multi* m) { return m ? (int32_t)m->replica.size() : DANN_EINVAL; }

dann_index* dann_multi_replica(dann_multi* m, uint32_t i) { return (m && i < m->replica.size()) ? m->replica[i] : nullptr; }

int32_t dann_multi_set_elements(dann_multi* m, uint32_t first_slot, uint32_t n, const void* rows, uint64_t len) try {
    if (!m) return DANN_EINVAL;
    return for_each_rank((uint32_t)m->replica.size(),
                         [&](uint32_t r) { return dann_set_elements(m->replica[r], first_slot, n, rows, len); });
} DANN_CATCH_ALL

int32_t dann_multi_build(dann_multi* m, const dann_build_config* cfg, uint32_t first, uint32_t n, float growth,
                         uint32_t max_batch, uint64_t* stats) try {
    if (!m) return DANN_EINVAL;
    return for_each_rank((uint32_t)m->replica.size(), [&](uint32_t r) {
        return dann_build_sharded(m->replica[r], m->comm[r], cfg, first, n, growth, max_batch, r == 0 ? stats : nullptr);
    });
} DANN_CATCH_ALL

int32_t dann_multi_search_batch(dann_multi* m, const void* queries, uint32_t nq, uint32_t l_value, uint32_t beam_width,
                                uint32_t k, uint32_t* out_ids, float* out_dists, dann_search_stats* out_stats) try {
    if (!m) return DANN_EINVAL;
    if (nq == 0) return DANN_OK;
    if (!queries || !out_ids || !out_dists) return DANN_EINVAL;
    const uint32_t world = (uint32_t)m->replica.size();
    const dann_index* i0 = m->replica[0];
    const size_t qb = i0->cfg.dtype == DT_PQ ? (size_t)i0->cfg.dim * 4 : i0->layer_bytes;
    // the query block is partitioned over the replicas (async_tools.rs:289-365); every slice lands in place
    return for_each_rank(world, [&](uint32_t r) -> int32_t {
        uint32_t lo, hi;
        partition(nq, world, r, &lo, &hi);
        if (hi == lo) return DANN_OK;
        return dann_search_batch(m->replica[r], reinterpret_cast<const uint8_t*>(queries) + (size_t)lo * qb, hi - lo, l_value,
                                 beam_width, k, out_ids + (size_t)lo * k, out_dists + (size_t)lo * k,
                                 out_stats ? out_stats + lo : nullptr);
    });
} DANN_CATCH_ALL

}  // extern "C"
