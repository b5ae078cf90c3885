// api.hip -- the C ABI of include/dann.h: index lifetime, HBM layout, host<->device
// staging and kernel dispatch.  No CPU fallback exists anywhere in this library: every
// distance, search and prune result comes from a HIP kernel, and a missing/failed device
// surfaces as DANN_EHIP.
#include <sched.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <time.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dann_device.h"
#include "dann_internal.h"

namespace dann {

static thread_local char g_err[512] = {0};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int32_t hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return DANN_EHIP;
}

static uint32_t elem_size(int32_t dtype) { return dtype == DT_F32 ? 4u : dtype == DT_F16 ? 2u : 1u; }
static uint32_t layer_bytes_of(int32_t dtype, uint32_t dim) { return dim * elem_size(dtype) + (dtype == DT_SQ8 ? 4u : 0u); }
static bool valid_dtype(int32_t d) { return d >= 0 && d <= 5; }
static bool valid_metric(int32_t m) { return m >= 0 && m <= 3; }

// temporary device buffer with RAII
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 1); }
    template <class T>
    T* as() {
        return reinterpret_cast<T*>(p);
    }
};

struct DeviceGuard {
    int prev = -1, cur = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) : cur(dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != cur) (void)hipSetDevice(prev);  // hipSetDevice costs ~0.5 ms on ROCm 7.2
    }
};

// time one launch on the index stream with HIP events (the stream the kernel runs on)
// a block of a context's arena, carved into 256-byte aligned pieces
struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return o;
    }
};
struct ArenaPtr {
    void* p;
    ArenaPtr(void* base, size_t off) : p(reinterpret_cast<uint8_t*>(base) + off) {}
    ArenaPtr() : p(nullptr) {}
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

template <class F>
static int32_t timed(dann_index* idx, int which, F&& f) {
    DANN_HIP(hipEventRecord(idx->main.ev0, idx->main.stream));
    int32_t rc = f();
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipEventRecord(idx->main.ev1, idx->main.stream));
    DANN_HIP(hipEventSynchronize(idx->main.ev1));
    float ms = 0.f;
    DANN_HIP(hipEventElapsedTime(&ms, idx->main.ev0, idx->main.ev1));
    std::lock_guard<std::mutex> lk(idx->stat_mu);
    idx->clocks[which].total_ms += ms;
    idx->clocks[which].launches += 1;
    return DANN_OK;
}

int32_t SearchCtx::init() {
    DANN_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    DANN_HIP(hipEventCreate(&ev0));
    DANN_HIP(hipEventCreate(&ev1));
    DANN_HIP(hipHostMalloc((void**)&h_flag, 64, hipHostMallocMapped));
    *h_flag = 0;
    return DANN_OK;
}

void SearchCtx::destroy() {
    if (stream) (void)hipStreamSynchronize(stream);
    if (copy_stream) (void)hipStreamSynchronize(copy_stream);
    if (d_fail) (void)hipFree(d_fail);
    if (h_flag) (void)hipHostFree(h_flag);
    if (d_spill) (void)hipFree(d_spill);
    for (void* p : stage)
        if (p) (void)hipFree(p);
    if (h_stage) (void)hipHostFree(h_stage);
    for (hipEvent_t e : chunk_ev)
        if (e) (void)hipEventDestroy(e);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (stream) (void)hipStreamDestroy(stream);
    *this = SearchCtx();
}

CtxLease::CtxLease(dann_index* i, bool try_only) : idx(i) {
    std::unique_lock<std::mutex> lk(idx->ctx_mu);
    for (;;) {
        if (!idx->ctx_free.empty()) {
            ctx = idx->ctx_free.back();
            idx->ctx_free.pop_back();
            return;
        }
        if (idx->ctx_created < kMaxSearchCtx) {
            ++idx->ctx_created;
            lk.unlock();
            SearchCtx* c = new (std::nothrow) SearchCtx();
            status = c ? c->init() : DANN_ENOMEM;
            if (status != DANN_OK) {
                if (c) {
                    c->destroy();
                    delete c;
                }
                lk.lock();
                --idx->ctx_created;
                idx->ctx_cv.notify_one();
                return;
            }
            ctx = c;
            return;
        }
        if (try_only) {  // (a caller that can do without: no context is free and none may be created)
            status = DANN_EBUSY;
            return;
        }
        idx->ctx_cv.wait(lk);
    }
}

CtxLease::~CtxLease() {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lk(idx->ctx_mu);
        idx->ctx_free.push_back(ctx);
    }
    idx->ctx_cv.notify_one();
}

uint32_t auto_visited_entries(const dann_index* idx, uint32_t, uint32_t) {
    if (idx->visited_bits >= 64) return std::min<uint32_t>((idx->visited_bits + 63u) / 64u * 64u, 32768u);
    if (idx->visited_bits) return 1u << idx->visited_bits;
    return 0;  // sized per launch by search_with_retry (search_kernels.hip)
}

}  // namespace dann

using namespace dann;

dann::IndexView dann_index::view() const {
    IndexView v;
    v.rows = d_rows;
    v.adj = d_adj;
    v.row_stride = cfg.row_stride;
    v.adj_stride = cfg.max_degree + 1;
    v.dim = cfg.dim;
    v.capacity = cfg.capacity;
    v.nslots = nslots;
    v.max_degree = cfg.max_degree;
    v.nstart = cfg.num_start_points;
    v.dtype = cfg.dtype;
    v.metric = cfg.metric;
    v.layer_bytes = layer_bytes;
    {   // (1/255)^2 * scale^2 in f32, in the reference's order (vectors.rs:236-241, quantizer.rs:316-320)
        const float ibs = 1.0f / 255.0f;
        const float bit_scale = ibs * ibs;
        const float scale_sq = cfg.sq_scale * cfg.sq_scale;
        v.sq_k = bit_scale * scale_sq;
        v.sq_shift_norm_sq = cfg.sq_shift_norm_sq;
    }
    v.pq_pivots = d_pq_pivots;
    v.pq_offsets = d_pq_offsets;
    v.pq_chunks = cfg.pq_chunks;
    v.pq_pack = pq_pack_valid ? d_pq_pack : nullptr;
    v.pq_pack_stride = pq_pack_stride;
    v.pq_pack_codes = pq_pack_codes;
    v.tag_off = cfg.inline_tags ? layer_bytes : 0u;
    return v;
}

extern "C" {

int32_t dann_last_error(char* buf, uint64_t len) try {
    size_t n = strlen(g_err);
    if (buf && len) {
        size_t c = n < len - 1 ? n : len - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return (int32_t)n;
} DANN_CATCH_ALL

int32_t dann_layer_bytes(int32_t dtype, uint32_t dim) try {
    if (!valid_dtype(dtype)) {
        set_error("bad dtype %d", dtype);
        return DANN_EINVAL;
    }
    return (int32_t)layer_bytes_of(dtype, dim);
} DANN_CATCH_ALL

int32_t dann_inmem2_row_stride(int32_t dtype, uint32_t dim) try {
    int32_t b = dann_layer_bytes(dtype, dim);
    if (b < 0) return b;
    return (b + 1 + 31) / 32 * 32;
} DANN_CATCH_ALL

int32_t dann_index_create(const dann_config* cfg, const void* start_rows, uint64_t start_len, dann_index** out) try {
    if (!cfg || !out) {
        set_error("null argument");
        return DANN_EINVAL;
    }
    *out = nullptr;
    if (!valid_dtype(cfg->dtype) || !valid_metric(cfg->metric) || cfg->dim == 0 || cfg->max_degree == 0) {
        set_error("invalid config (dtype %d metric %d dim %u max_degree %u)", cfg->dtype, cfg->metric, cfg->dim,
                  cfg->max_degree);
        return DANN_EINVAL;
    }
    if (cfg->dtype == DT_PQ && (cfg->pq_chunks == 0 || cfg->pq_chunks > 128 || cfg->pq_chunks > cfg->dim)) {
        set_error("DANN_PQ needs 1 <= pq_chunks <= min(dim, 128)");
        return DANN_EINVAL;
    }
    const uint32_t lb = cfg->dtype == DT_PQ ? cfg->pq_chunks : layer_bytes_of(cfg->dtype, cfg->dim);
    {
        int op;
        bool norm;
        if (!resolve_metric(cfg->dtype, cfg->metric, &op, &norm)) {
            set_error("metric %d is not defined for dtype %d", cfg->metric, cfg->dtype);
            return DANN_EUNSUPPORTED;
        }
        if (cfg->dtype == DT_SQ8 && !(cfg->sq_scale > 0.0f)) {
            set_error("DANN_SQ8 needs sq_scale > 0");
            return DANN_EINVAL;
        }
    }
    if ((uint64_t)cfg->capacity + cfg->num_start_points >= 0x7FFFFFFFull) {
        set_error("capacity + start points must be below 2^31 - 1");
        return DANN_EINVAL;
    }
    if (start_len != (uint64_t)lb * cfg->num_start_points || (cfg->num_start_points && !start_rows)) {
        set_error("start rows: expected %llu bytes, got %llu", (unsigned long long)lb * cfg->num_start_points,
                  (unsigned long long)start_len);
        return DANN_ELENGTH;
    }
    dann_index* idx = new (std::nothrow) dann_index();
    if (!idx) return DANN_ENOMEM;
    idx->cfg = *cfg;
    idx->layer_bytes = lb;
    for (auto& d : idx->dbg) d.store(__builtin_nan(""), std::memory_order_relaxed);
    if (idx->cfg.row_stride == 0) idx->cfg.row_stride = (lb + 15u) & ~15u;
    if (idx->cfg.row_stride < lb || (idx->cfg.row_stride & 15u)) {
        set_error("row_stride %u must be >= %u and a multiple of 16", idx->cfg.row_stride, lb);
        delete idx;
        return DANN_EINVAL;
    }
    if (idx->cfg.inline_tags && idx->cfg.row_stride <= lb) {
        set_error("inline_tags needs row_stride > %u payload bytes (the tag byte follows the payload, store.rs:133-158)", lb);
        delete idx;
        return DANN_EINVAL;
    }
    idx->nslots = cfg->capacity + cfg->num_start_points;
    int dev = cfg->device;
    if (dev < 0) {
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) {
            delete idx;
            return hip_fail(e, "hipGetDevice");
        }
    }
    idx->device = dev;
    idx->cfg.device = dev;
    DeviceGuard guard(dev);
    auto fail = [&](hipError_t e, const char* what) {
        int32_t rc = hip_fail(e, what);
        dann_index_destroy(idx);
        return rc;
    };
    if (!guard.ok) return fail(hipErrorInvalidDevice, "hipSetDevice");
    hipError_t e;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            idx->num_cus = (uint32_t)cus;
    }
    if (int32_t irc = idx->main.init()) {
        dann_index_destroy(idx);
        return irc;
    }
    const size_t rows_bytes = (size_t)idx->nslots * idx->cfg.row_stride + 256;
    const size_t adj_bytes = (size_t)idx->nslots * (cfg->max_degree + 1) * 4;
    if ((e = hipMalloc((void**)&idx->d_rows, rows_bytes)) != hipSuccess) return fail(e, "hipMalloc(rows)");
    if ((e = hipMalloc((void**)&idx->d_adj, adj_bytes)) != hipSuccess) return fail(e, "hipMalloc(adjacency)");
    if ((e = hipMemsetAsync(idx->d_rows, 0, rows_bytes, idx->main.stream)) != hipSuccess) return fail(e, "hipMemset");
    if ((e = hipMemsetAsync(idx->d_adj, 0, adj_bytes, idx->main.stream)) != hipSuccess) return fail(e, "hipMemset");
    if (cfg->num_start_points) {
        e = hipMemcpy2DAsync(idx->d_rows + (size_t)cfg->capacity * idx->cfg.row_stride, idx->cfg.row_stride, start_rows,
                             lb, lb, cfg->num_start_points, hipMemcpyHostToDevice, idx->main.stream);
        if (e != hipSuccess) return fail(e, "hipMemcpy2D(start rows)");
    }
    if (idx->cfg.inline_tags) {  // dynamic slots AVAILABLE (0, the memset above), start points FROZEN (store.rs:766-772)
        idx->h_tags.assign(idx->nslots, 0);
        if (cfg->num_start_points) {
            std::fill(idx->h_tags.begin() + cfg->capacity, idx->h_tags.end(), (uint8_t)255);
            e = hipMemset2DAsync(idx->d_rows + (size_t)cfg->capacity * idx->cfg.row_stride + lb, idx->cfg.row_stride, 255, 1,
                                 cfg->num_start_points, idx->main.stream);
            if (e != hipSuccess) return fail(e, "hipMemset2D(start tags)");
        }
    }
    if ((e = hipStreamSynchronize(idx->main.stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
    *out = idx;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_index_destroy(dann_index* idx) try {
    if (!idx) return DANN_OK;
    DeviceGuard guard(idx->device);
    if (idx->server) (void)dann_server_stop(idx);
    idx->main.destroy();
    for (SearchCtx* c : idx->ctx_free) {
        c->destroy();
        delete c;
    }
    idx->ctx_free.clear();
    if (idx->d_rows) (void)hipFree(idx->d_rows);
    if (idx->d_adj) (void)hipFree(idx->d_adj);
    if (idx->d_pq_pivots) (void)hipFree(idx->d_pq_pivots);
    if (idx->d_pq_offsets) (void)hipFree(idx->d_pq_offsets);
    if (idx->d_pq_pack) (void)hipFree(idx->d_pq_pack);
    if (idx->build_scratch && idx->build_scratch_free) idx->build_scratch_free(idx->build_scratch);
    delete idx;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_index_max_degree(const dann_index* idx) { return idx ? (int32_t)idx->cfg.max_degree : DANN_EINVAL; }

int32_t dann_index_get_config(const dann_index* idx, dann_config* out) try {
    if (!idx || !out) return DANN_EINVAL;
    *out = idx->cfg;
    return DANN_OK;
} DANN_CATCH_ALL

#define CHECK_IDX(idx)                                  \
    if (!(idx)) {                                       \
        set_error("null index");                        \
        return DANN_EINVAL;                             \
    }                                                   \
    ::dann::ExclusiveGuard _lock(idx); \
    DeviceGuard _guard((idx)->device)

int32_t dann_set_elements(dann_index* idx, uint32_t first_slot, uint32_t n, const void* rows, uint64_t len) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (n == 0) return DANN_OK;
    if (!rows) return DANN_EINVAL;
    if (len != (uint64_t)n * idx->layer_bytes) {
        set_error("raw byte slice of length %llu does not match expected length %llu", (unsigned long long)len,
                  (unsigned long long)n * idx->layer_bytes);
        return DANN_ELENGTH;
    }
    if ((uint64_t)first_slot + n > idx->cfg.capacity) {
        set_error("slot range [%u, %llu) exceeds capacity %u", first_slot, (unsigned long long)first_slot + n,
                  idx->cfg.capacity);
        return DANN_EBOUNDS;
    }
    DANN_HIP(hipMemcpy2DAsync(idx->d_rows + (size_t)first_slot * idx->cfg.row_stride, idx->cfg.row_stride, rows,
                              idx->layer_bytes, idx->layer_bytes, n, hipMemcpyHostToDevice, idx->main.stream));
    if (idx->cfg.inline_tags) {  // Slot::publish (store.rs:776-782)
        DANN_HIP(hipMemset2DAsync(idx->d_rows + (size_t)first_slot * idx->cfg.row_stride + idx->layer_bytes,
                                  idx->cfg.row_stride, 254, 1, n, idx->main.stream));
        std::fill(idx->h_tags.begin() + first_slot, idx->h_tags.begin() + first_slot + n, (uint8_t)254);
    }
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_elements_device(dann_index* idx, uint32_t first_slot, uint32_t n, const void* d_rows, uint64_t src_stride) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (n == 0) return DANN_OK;
    if (!d_rows) return DANN_EINVAL;
    if (src_stride < idx->layer_bytes) {
        set_error("source stride %llu is shorter than the layer's %u bytes", (unsigned long long)src_stride, idx->layer_bytes);
        return DANN_ELENGTH;
    }
    if ((uint64_t)first_slot + n > idx->cfg.capacity) {
        set_error("slot range [%u, %llu) exceeds capacity %u", first_slot, (unsigned long long)first_slot + n,
                  idx->cfg.capacity);
        return DANN_EBOUNDS;
    }
    DANN_HIP(hipMemcpy2DAsync(idx->d_rows + (size_t)first_slot * idx->cfg.row_stride, idx->cfg.row_stride, d_rows, src_stride,
                              idx->layer_bytes, n, hipMemcpyDeviceToDevice, idx->main.stream));
    if (idx->cfg.inline_tags) {  // Slot::publish (store.rs:776-782)
        DANN_HIP(hipMemset2DAsync(idx->d_rows + (size_t)first_slot * idx->cfg.row_stride + idx->layer_bytes,
                                  idx->cfg.row_stride, 254, 1, n, idx->main.stream));
        std::fill(idx->h_tags.begin() + first_slot, idx->h_tags.begin() + first_slot + n, (uint8_t)254);
    }
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_index_device_pointers(const dann_index* idx, const void** d_rows, const uint32_t** d_adjacency) try {
    if (!idx) return DANN_EINVAL;
    if (d_rows) *d_rows = idx->d_rows;
    if (d_adjacency) *d_adjacency = idx->d_adj;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_tags(dann_index* idx, uint32_t first_slot, uint32_t n, const uint8_t* tags) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (n == 0) return DANN_OK;
    if (!tags) return DANN_EINVAL;
    if (!idx->cfg.inline_tags) {
        set_error("dann_set_tags: the index was created without inline_tags");
        return DANN_EUNSUPPORTED;
    }
    if ((uint64_t)first_slot + n > idx->nslots) return DANN_EBOUNDS;
    // start points are FROZEN for the lifetime of the store (store.rs:766-772); a search that cannot read one fails
    // ("could not retrieve start point", provider.rs:408-431) -- refuse the state instead of producing it
    for (uint32_t i = 0; i < n; ++i)
        if (first_slot + i >= idx->cfg.capacity && tags[i] < 254) {
            set_error("dann_set_tags: start point slot %u must stay readable (tag >= 254), got %u", first_slot + i, tags[i]);
            return DANN_EINVAL;
        }
    DANN_HIP(hipMemcpy2DAsync(idx->d_rows + (size_t)first_slot * idx->cfg.row_stride + idx->layer_bytes,
                              idx->cfg.row_stride, tags, 1, 1, n, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    memcpy(idx->h_tags.data() + first_slot, tags, n);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_get_tags(const dann_index* idx, uint32_t first_slot, uint32_t n, uint8_t* tags) try {
    if (!idx || (n && !tags)) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    if ((uint64_t)first_slot + n > idx->nslots) return DANN_EBOUNDS;
    if (!idx->cfg.inline_tags) memset(tags, 254, n);  // a store without tags: every slot readable
    else memcpy(tags, idx->h_tags.data() + first_slot, n);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_element(dann_index* idx, uint32_t slot, const void* bytes, uint64_t len) try {
    return dann_set_elements(idx, slot, 1, bytes, len);
} DANN_CATCH_ALL

int32_t dann_get_element(const dann_index* idx, uint32_t slot, void* bytes, uint64_t len) try {
    CHECK_IDX(idx);
    if (!bytes) return DANN_EINVAL;
    if (len != idx->layer_bytes) {
        set_error("expected slice of length %u - instead got %llu", idx->layer_bytes, (unsigned long long)len);
        return DANN_ELENGTH;
    }
    if (slot >= idx->nslots) return DANN_EBOUNDS;
    DANN_HIP(hipMemcpyAsync(bytes, idx->d_rows + (size_t)slot * idx->cfg.row_stride, len, hipMemcpyDeviceToHost,
                            idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_upload_store(dann_index* idx, const void* base, uint64_t stride, uint32_t nrows) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (!base) return DANN_EINVAL;
    if (nrows > idx->nslots) return DANN_EBOUNDS;
    if (stride < idx->layer_bytes) return DANN_ELENGTH;
    // with inline_tags the tag byte that follows each payload travels with it (needs stride > payload)
    const bool tags = idx->cfg.inline_tags != 0;
    if (tags && stride <= idx->layer_bytes) {
        set_error("dann_upload_store: inline_tags needs a source stride > %u payload bytes", idx->layer_bytes);
        return DANN_ELENGTH;
    }
    if (tags) {  // as dann_set_tags: a start point must stay readable
        const uint8_t* b = reinterpret_cast<const uint8_t*>(base) + idx->layer_bytes;
        for (uint32_t i = idx->cfg.capacity; i < nrows; ++i)
            if (b[(size_t)i * stride] < 254) {
                set_error("dann_upload_store: start point slot %u must be readable (tag >= 254), got %u", i, b[(size_t)i * stride]);
                return DANN_EINVAL;
            }
    }
    DANN_HIP(hipMemcpy2DAsync(idx->d_rows, idx->cfg.row_stride, base, stride, idx->layer_bytes + (tags ? 1u : 0u), nrows,
                              hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    if (tags) {
        const uint8_t* b = reinterpret_cast<const uint8_t*>(base) + idx->layer_bytes;
        for (uint32_t i = 0; i < nrows; ++i) idx->h_tags[i] = b[(size_t)i * stride];
    }
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_pq_table(dann_index* idx, const float* pivots, const uint32_t* chunk_offsets) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (!pivots || !chunk_offsets) return DANN_EINVAL;
    if (idx->cfg.dtype != DT_PQ) {
        set_error("dann_set_pq_table: the index is not DANN_PQ");
        return DANN_EINVAL;
    }
    const uint32_t nc = idx->cfg.pq_chunks, dim = idx->cfg.dim;
    if (chunk_offsets[0] != 0 || chunk_offsets[nc] != dim) {
        set_error("chunk offsets must start at 0 and end at dim");
        return DANN_EINVAL;
    }
    for (uint32_t c = 0; c < nc; ++c)
        if (chunk_offsets[c + 1] <= chunk_offsets[c]) {
            set_error("chunk offsets must be strictly increasing");
            return DANN_EINVAL;
        }
    if (!idx->d_pq_pivots) DANN_HIP(hipMalloc((void**)&idx->d_pq_pivots, (size_t)256 * dim * 4));
    if (!idx->d_pq_offsets) DANN_HIP(hipMalloc((void**)&idx->d_pq_offsets, (size_t)(nc + 1) * 4));
    DANN_HIP(hipMemcpyAsync(idx->d_pq_pivots, pivots, (size_t)256 * dim * 4, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(idx->d_pq_offsets, chunk_offsets, (size_t)(nc + 1) * 4, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

// ---- packed search layout of PQ indexes ----------------------------------------------------------------------------
int32_t dann_pq_pack_neighbors(dann_index* idx) try {
    CHECK_IDX(idx);
    // (only pq_search_kernel reads the packed rows: an index it can never serve -- pq_lut_shape / plain_mode -- would pay
    // 1.3 KB per node for nothing)
    const uint32_t code_words = (idx->cfg.pq_chunks + 15u) / 16u;  // 16-byte words of a code row
    if (idx->cfg.dtype != DT_PQ || idx->cfg.pq_chunks > 64u || idx->cfg.inline_tags || idx->cfg.max_degree > 64u ||
        idx->cfg.num_start_points > 64u || idx->cfg.row_stride < 16u * code_words || idx->cfg.row_stride % 16u) {
        set_error("dann_pq_pack_neighbors: a DANN_PQ index of at most 64 chunks, degree <= 64, at most 64 start points, "
                  "without inline tags, rows at a 16-byte stride");
        return DANN_EUNSUPPORTED;
    }
    // the rebuild rewrites (and may free) what a concurrent dann_search_submit's relaunch reads through idx->view():
    // it is a mutation -- refused while tickets are outstanding, submits bounce while it runs, the resident kernel leaves
    DANN_MUTATION(idx);
    const uint32_t R = idx->cfg.max_degree;
    const uint32_t codes_off = ((R + 1u) * 4u + 15u) & ~15u;
    const uint32_t stride = (codes_off + 16u * code_words * R + 63u) & ~63u;
    const size_t bytes = (size_t)idx->nslots * stride;
    if (idx->pq_pack_bytes < bytes) {
        if (idx->d_pq_pack) (void)hipFree(idx->d_pq_pack);
        idx->d_pq_pack = nullptr;
        idx->pq_pack_bytes = 0;
        idx->pq_pack_valid = false;
        DANN_HIP(hipMalloc((void**)&idx->d_pq_pack, bytes));
        idx->pq_pack_bytes = bytes;
    }
    idx->pq_pack_stride = stride;
    idx->pq_pack_codes = codes_off;
    idx->pq_pack_valid = false;
    int32_t rc = launch_pq_pack(idx->view(), idx->d_pq_pack, stride, codes_off, idx->main.stream);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    idx->pq_pack_valid = true;
    return DANN_OK;
} DANN_CATCH_ALL

// ---- external ids ------------------------------------------------------------------------
int32_t dann_set_external_ids(dann_index* idx, uint32_t first_slot, uint32_t n, const uint64_t* ext_ids) try {
    if (!idx) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    if (n == 0) return DANN_OK;
    if (!ext_ids) return DANN_EINVAL;
    if ((uint64_t)first_slot + n > idx->cfg.capacity) return DANN_EBOUNDS;
    if (idx->ext_ids.empty()) idx->ext_ids.assign(idx->cfg.capacity, ~0ull);
    memcpy(idx->ext_ids.data() + first_slot, ext_ids, (size_t)n * 8);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_to_external(const dann_index* idx, const uint32_t* slot_ids, uint64_t n, uint64_t* out_ext) try {
    if (!idx || (n && (!slot_ids || !out_ext))) return DANN_EINVAL;
    ::dann::ExclusiveGuard lock(idx);
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t s = slot_ids[i];
        if (s >= idx->cfg.capacity) out_ext[i] = ~0ull;  // start points / padding have no mapping
        else out_ext[i] = idx->ext_ids.empty() ? (uint64_t)s : idx->ext_ids[s];
    }
    return DANN_OK;
} DANN_CATCH_ALL

// ---- adjacency ---------------------------------------------------------------------------
int32_t dann_get_neighbors(const dann_index* idx, uint32_t slot, uint32_t* out, uint32_t cap, uint32_t* out_len) try {
    CHECK_IDX(idx);
    if (!out_len) return DANN_EINVAL;
    if (slot >= idx->nslots) {
        set_error("adjacency list %u is out of bounds (%u entries)", slot, idx->nslots);
        return DANN_EBOUNDS;
    }
    std::vector<uint32_t> row(idx->cfg.max_degree + 1);
    DANN_HIP(hipMemcpyAsync(row.data(), idx->d_adj + (size_t)slot * row.size(), row.size() * 4, hipMemcpyDeviceToHost,
                            idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    uint32_t len = std::min(row[0], idx->cfg.max_degree);
    *out_len = len;
    if (len > cap || (len && !out)) return DANN_ETOOLONG;
    if (len) memcpy(out, row.data() + 1, (size_t)len * 4);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_neighbors(dann_index* idx, uint32_t slot, const uint32_t* ids, uint32_t n) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (slot >= idx->nslots) return DANN_EBOUNDS;
    if (n > idx->cfg.max_degree) {
        set_error("adjacency list of length %u exceeds max degree %u", n, idx->cfg.max_degree);
        return DANN_ETOOLONG;
    }
    if (n && !ids) return DANN_EINVAL;
    std::vector<uint32_t> row(n + 1);
    row[0] = n;
    if (n) memcpy(row.data() + 1, ids, (size_t)n * 4);
    DANN_HIP(hipMemcpyAsync(idx->d_adj + (size_t)slot * (idx->cfg.max_degree + 1), row.data(), row.size() * 4,
                            hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_append_neighbors(dann_index* idx, uint32_t slot, const uint32_t* ids, uint32_t n) try {
    CHECK_IDX(idx);  // recursive mutex: the get / set calls below re-enter it
    DANN_MUTATION(idx);
    if (n && !ids) return DANN_EINVAL;
    if (slot >= idx->nslots) return DANN_EBOUNDS;
    std::vector<uint32_t> cur(idx->cfg.max_degree);
    uint32_t len = 0;
    int32_t rc = dann_get_neighbors(idx, slot, cur.data(), idx->cfg.max_degree, &len);
    if (rc != DANN_OK) return rc;
    uint32_t slack = idx->cfg.max_degree - len;  // clamp, provider.rs:804-816
    uint32_t take = std::min(n, slack);
    for (uint32_t i = 0; i < take; ++i) cur[len + i] = ids[i];
    return dann_set_neighbors(idx, slot, cur.data(), len + take);
} DANN_CATCH_ALL

int32_t dann_set_neighbors_bulk(dann_index* idx, const uint32_t* slots, uint32_t n, const uint32_t* lists) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (n == 0) return DANN_OK;
    if (!slots || !lists) return DANN_EINVAL;
    const uint32_t w = idx->cfg.max_degree + 1;
    for (uint32_t i = 0; i < n; ++i) {
        if (slots[i] >= idx->nslots) return DANN_EBOUNDS;
        if (lists[(size_t)i * w] > idx->cfg.max_degree) return DANN_ETOOLONG;
    }
    for (uint32_t i = 0; i < n; ++i) {
        DANN_HIP(hipMemcpyAsync(idx->d_adj + (size_t)slots[i] * w, lists + (size_t)i * w, (size_t)w * 4,
                                hipMemcpyHostToDevice, idx->main.stream));
    }
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_upload_graph(dann_index* idx, const uint32_t* adj, uint64_t nrows) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (!adj) return DANN_EINVAL;
    if (nrows > idx->nslots) return DANN_EBOUNDS;
    DANN_HIP(hipMemcpyAsync(idx->d_adj, adj, (size_t)nrows * (idx->cfg.max_degree + 1) * 4, hipMemcpyHostToDevice,
                            idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_download_graph(const dann_index* idx, uint32_t* adj, uint64_t nrows) try {
    CHECK_IDX(idx);
    if (!adj) return DANN_EINVAL;
    if (nrows > idx->nslots) return DANN_EBOUNDS;
    DANN_HIP(hipMemcpyAsync(adj, idx->d_adj, (size_t)nrows * (idx->cfg.max_degree + 1) * 4, hipMemcpyDeviceToHost,
                            idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

// ---- distances ---------------------------------------------------------------------------
int32_t dann_distance(const dann_index* idx, const void* x, uint64_t xlen, const void* y, uint64_t ylen, float* out) try {
    CHECK_IDX(idx);
    if (!x || !y || !out) return DANN_EINVAL;
    if (xlen != idx->layer_bytes || ylen != idx->layer_bytes) {
        set_error("expected slices of length %u - instead got %llu and %llu", idx->layer_bytes,
                  (unsigned long long)xlen, (unsigned long long)ylen);
        return DANN_ELENGTH;
    }
    const size_t stride = (idx->layer_bytes + 15u) & ~15u;
    DevBuf buf;
    DANN_HIP(buf.alloc(2 * stride + 16));
    uint8_t* d = buf.as<uint8_t>();
    DANN_HIP(hipMemcpyAsync(d, x, xlen, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(d + stride, y, ylen, hipMemcpyHostToDevice, idx->main.stream));
    float* d_out = reinterpret_cast<float*>(d + 2 * stride);
    int32_t rc = launch_distance_raw(idx->view(), d, d + stride, stride, 1, d_out, idx->main.stream);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(out, d_out, 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_distance_pairs(const dann_index* idx, const uint32_t* a, const uint32_t* b, uint32_t n, float* out) try {
    CHECK_IDX(idx);
    if (n == 0) return DANN_OK;
    if (!a || !b || !out) return DANN_EINVAL;
    for (uint32_t i = 0; i < n; ++i)
        if (a[i] >= idx->nslots || b[i] >= idx->nslots) return DANN_EBOUNDS;
    DevBuf buf;
    DANN_HIP(buf.alloc((size_t)n * 12));
    uint32_t* da = buf.as<uint32_t>();
    uint32_t* db = da + n;
    float* dout = reinterpret_cast<float*>(db + n);
    DANN_HIP(hipMemcpyAsync(da, a, (size_t)n * 4, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(db, b, (size_t)n * 4, hipMemcpyHostToDevice, idx->main.stream));
    int32_t rc = launch_distance_pairs(idx->view(), da, db, n, dout, idx->main.stream);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_query_create(const dann_index* idx, const void* query, uint64_t len, dann_query** out) try {
    CHECK_IDX(idx);
    if (!query || !out) return DANN_EINVAL;
    *out = nullptr;
    if (len != idx->layer_bytes) {  // Full::check_dim (full.rs:86-99)
        set_error("query of %llu bytes does not match the layer's %u bytes", (unsigned long long)len, idx->layer_bytes);
        return DANN_ELENGTH;
    }
    dann_query* q = new (std::nothrow) dann_query();
    if (!q) return DANN_ENOMEM;
    q->idx = idx;
    hipError_t e = hipMalloc(&q->d_query, (len + 15) & ~15ull);
    if (e != hipSuccess) {
        delete q;
        return hip_fail(e, "hipMalloc(query)");
    }
    e = hipMemcpyAsync(q->d_query, query, len, hipMemcpyHostToDevice, idx->main.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(idx->main.stream);
    if (e != hipSuccess) {
        (void)hipFree(q->d_query);
        delete q;
        return hip_fail(e, "hipMemcpy(query)");
    }
    *out = q;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_query_destroy(dann_query* q) try {
    if (!q) return DANN_OK;
    DeviceGuard guard(q->idx->device);
    if (q->d_query) (void)hipFree(q->d_query);
    delete q;
    return DANN_OK;
} DANN_CATCH_ALL

// shared by dann_query_distance / dann_expand_beam: search-path kernel over a temporary row
// set or stored rows
static int32_t expand_on_device(const dann_index* idx, const IndexView& view, const void* d_query, const uint32_t* ids,
                                uint32_t n, float* out) {
    DevBuf buf;
    DANN_HIP(buf.alloc((size_t)n * 8 + 16));
    uint64_t* d_off = buf.as<uint64_t>();
    uint32_t* d_ids = reinterpret_cast<uint32_t*>(d_off + 2);
    float* d_out = reinterpret_cast<float*>(d_ids + n);
    uint64_t off[2] = {0, n};
    DANN_HIP(hipMemcpyAsync(d_off, off, 16, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(d_ids, ids, (size_t)n * 4, hipMemcpyHostToDevice, idx->main.stream));
    int32_t rc = launch_expand_beam(view, d_query, 1, d_ids, d_off, n, d_out, idx->main.stream);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
}

int32_t dann_query_distance(const dann_query* q, const void* row, uint64_t len, float* out) try {
    if (!q || !row || !out) return DANN_EINVAL;
    const dann_index* idx = q->idx;
    ::dann::ExclusiveGuard lock(idx);
    DeviceGuard guard(idx->device);
    if (len != idx->layer_bytes) {
        set_error("expected slice of length %u - instead got %llu", idx->layer_bytes, (unsigned long long)len);
        return DANN_ELENGTH;
    }
    DevBuf rowbuf;
    DANN_HIP(rowbuf.alloc((len + 15) & ~15ull));
    DANN_HIP(hipMemcpyAsync(rowbuf.p, row, len, hipMemcpyHostToDevice, idx->main.stream));
    IndexView v = idx->view();
    v.rows = rowbuf.as<uint8_t>();
    v.nslots = 1;
    uint32_t zero = 0;
    return expand_on_device(idx, v, q->d_query, &zero, 1, out);
} DANN_CATCH_ALL

int32_t dann_expand_beam(const dann_query* q, const uint32_t* ids, uint32_t n, uint32_t* out_ids, float* out_dists,
                         uint32_t* out_n) try {
    if (!q || !out_n) return DANN_EINVAL;
    const dann_index* idx = q->idx;
    ::dann::ExclusiveGuard lock(idx);
    DeviceGuard guard(idx->device);
    *out_n = 0;
    if (n == 0) return DANN_OK;
    if (!ids || !out_ids || !out_dists) return DANN_EINVAL;
    for (uint32_t i = 0; i < n; ++i)
        if (ids[i] >= idx->nslots) return DANN_EBOUNDS;
    // read_in_bounds(i) -> None for a slot whose tag is not readable: skipped, not counted (provider.rs:681-686)
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i)
        if (!idx->cfg.inline_tags || idx->h_tags[ids[i]] >= 254) out_ids[m++] = ids[i];
    *out_n = m;
    if (m == 0) return DANN_OK;
    return expand_on_device(idx, idx->view(), q->d_query, out_ids, m, out_dists);
} DANN_CATCH_ALL

int32_t dann_expand_beam_batch(const dann_index* cidx, const void* queries, uint32_t nq, const uint32_t* ids,
                               const uint64_t* offsets, float* out_dists) try {
    dann_index* idx = const_cast<dann_index*>(cidx);
    CHECK_IDX(idx);
    if (nq == 0) return DANN_OK;
    if (!queries || !ids || !offsets || !out_dists) return DANN_EINVAL;
    const uint64_t total = offsets[nq];
    uint64_t max_len = 0;
    for (uint32_t i = 0; i < nq; ++i) {
        if (offsets[i + 1] < offsets[i]) return DANN_EINVAL;
        max_len = std::max(max_len, offsets[i + 1] - offsets[i]);
    }
    for (uint64_t i = 0; i < total; ++i)
        if (ids[i] >= idx->nslots) return DANN_EBOUNDS;
    if (total == 0) return DANN_OK;
    DevBuf bq, bo, bi, bd;
    DANN_HIP(bq.alloc((size_t)nq * idx->layer_bytes + 16));
    DANN_HIP(bo.alloc((size_t)(nq + 1) * 8));
    DANN_HIP(bi.alloc(total * 4));
    DANN_HIP(bd.alloc(total * 4));
    DANN_HIP(hipMemcpyAsync(bq.p, queries, (size_t)nq * idx->layer_bytes, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(bo.p, offsets, (size_t)(nq + 1) * 8, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(bi.p, ids, total * 4, hipMemcpyHostToDevice, idx->main.stream));
    int32_t rc = timed(idx, 1, [&] {
        return launch_expand_beam(idx->view(), bq.p, nq, bi.as<uint32_t>(), bo.as<uint64_t>(), max_len, bd.as<float>(),
                                  idx->main.stream);
    });
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(out_dists, bd.p, total * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    if (idx->cfg.inline_tags)  // the positional batch form cannot drop entries: unreadable slots report NaN
        for (uint64_t i = 0; i < total; ++i)
            if (idx->h_tags[ids[i]] < 254) out_dists[i] = __builtin_nanf("");
    return DANN_OK;
} DANN_CATCH_ALL

// ---- search --------------------------------------------------------------------------------
static int32_t pq_ready(const dann_index* idx) {
    if (idx->cfg.dtype == DT_PQ && (!idx->d_pq_pivots || !idx->d_pq_offsets)) {
        set_error("DANN_PQ index has no pivot table: call dann_set_pq_table first");
        return DANN_EINVAL;
    }
    return DANN_OK;
}

static int32_t search_device(dann_index* idx, SearchCtx& ctx, const void* d_queries, const uint32_t* d_qslots, uint32_t nq,
                             uint32_t l_value, uint32_t beam, uint32_t k, uint32_t* d_ids, float* d_dists,
                             dann_search_stats* d_stats, uint32_t* d_rec_ids, float* d_rec_d, uint32_t rec_stride,
                             uint32_t* d_rec_n) {
    if (int32_t prc = pq_ready(idx)) return prc;
    if (idx->cfg.dtype == DT_PQ && d_qslots) {
        set_error("insert-time search (query = stored row) is not defined for DANN_PQ");
        return DANN_EUNSUPPORTED;
    }
    SearchArgs a;
    a.ix = idx->view();
    a.queries = d_queries;
    a.qslots = d_qslots;
    a.nq = nq;
    a.l_value = l_value;
    a.beam_width = beam;
    a.k = k;
    a.ht_entries = auto_visited_entries(idx, l_value, beam);
    a.out_ids = d_ids;
    a.out_dists = d_dists;
    a.stats = d_stats;
    a.rec_ids = d_rec_ids;
    a.rec_dists = d_rec_d;
    a.rec_stride = rec_stride;
    a.rec_n = d_rec_n;
    a.qmap = nullptr;
    a.range_ids = nullptr;
    a.range_d = nullptr;
    a.range_second = nullptr;
    a.range_cap = a.range_max = a.range_thresh = a.has_inner = 0;
    a.radius = a.inner_radius = a.range_slack = 0.0f;
    a.fail_flag = nullptr;
    a.spill = nullptr;
    a.spill_next = nullptr;
    a.spill_slices = a.spill_bits = 0;
    return search_with_retry(idx, ctx, a);
}

// shared access for the Knn search entry points: the index is read-only here, every call runs on its own context
#define CHECK_IDX_SHARED(idx)                               \
    if (!(idx)) {                                           \
        set_error("null index");                            \
        return DANN_EINVAL;                                 \
    }                                                       \
    std::shared_lock<std::shared_mutex> _rd((idx)->rw);     \
    DeviceGuard _guard((idx)->device);                      \
    CtxLease _lease(idx);                                   \
    if (_lease.status != DANN_OK) return _lease.status;     \
    SearchCtx& ctx = *_lease.ctx

static int32_t grow_stage(SearchCtx& ctx, int i, size_t need) {
    if (ctx.stage_bytes[i] >= need) return DANN_OK;
    if (ctx.stage[i]) (void)hipFree(ctx.stage[i]);
    ctx.stage[i] = nullptr;
    ctx.stage_bytes[i] = 0;
    const size_t sz = need + need / 4;
    DANN_HIP(hipMalloc(&ctx.stage[i], sz));
    ctx.stage_bytes[i] = sz;
    return DANN_OK;
}

// The scratch arena of the range / filtered searches (stage[4]) stays resident between calls only up to this size: a
// call with per-query filter bitmaps on a large index can need gigabytes, and an index that nearly fills HBM would
// miss them in its next build or search (the pool keeps up to 16 contexts).  Beyond it the block is released when the
// call returns (a hipFree synchronises the device: the price of a call that large, not of every call).
constexpr size_t kArenaKeepBytes = (size_t)256 << 20;
struct ArenaTrim {
    SearchCtx& ctx;
    ~ArenaTrim() {
        if (ctx.stage_bytes[4] > kArenaKeepBytes) {
            (void)hipStreamSynchronize(ctx.stream);
            (void)hipFree(ctx.stage[4]);
            ctx.stage[4] = nullptr;
            ctx.stage_bytes[4] = 0;
        }
    }
};

int32_t dann_search_batch_device(dann_index* idx, const void* d_queries, uint32_t nq, uint32_t l_value,
                                 uint32_t beam_width, uint32_t k, uint32_t* d_out_ids, float* d_out_dists,
                                 dann_search_stats* d_out_stats) try {
    CHECK_IDX_SHARED(idx);
    if (nq == 0) return DANN_OK;
    if (!d_queries || !d_out_ids || !d_out_dists) return DANN_EINVAL;
    if (!d_out_stats) {  // the overflow retry needs per-query status: context-owned stats when the caller passes none
        if (int32_t rc = grow_stage(ctx, 2, (size_t)nq * sizeof(dann_search_stats))) return rc;
        d_out_stats = reinterpret_cast<dann_search_stats*>(ctx.stage[2]);
    }
    return search_device(idx, ctx, d_queries, nullptr, nq, l_value, beam_width, k, d_out_ids, d_out_dists, d_out_stats,
                         nullptr, nullptr, 0, nullptr);
} DANN_CATCH_ALL

static int32_t first_failed_query(const dann_search_stats* stats, uint32_t nq, uint32_t base) {
    for (uint32_t i = 0; i < nq; ++i) {
        if (stats[i].status == (uint32_t)(-DANN_EINVAL)) {
            set_error("query %u: could not retrieve start point (a start slot is not readable)", base + i);
            return DANN_EINVAL;
        }
        if (stats[i].status) {
            set_error("query %u: per-query scratch exhausted (visited table and spill pool); raise the table "
                      "size with dann_set_visited_bits", base + i);
            return DANN_EOVERFLOW;
        }
    }
    return DANN_OK;
}

// ---- small host-pointer calls: one launch for all callers that are waiting (the queue and the leadership: small_calls.h)
extern "C++" {
namespace {
void grab_error_text(std::string& t) {
    char buf[512];
    buf[0] = 0;
    dann_last_error(buf, sizeof buf);
    t = buf;
}
// one launch for `n` calls (same L, beam, k; `total` queries).  Returns kSmallCallDeclined if the staging cannot be
// mapped (the calls then take the general path one by one).
int32_t small_batch_run(dann_index* idx, SmallCall* const* calls, uint32_t n, uint32_t total, size_t qb) {
    CtxLease lease(idx);
    if (lease.status != DANN_OK) return lease.status;
    SearchCtx& ctx = *lease.ctx;
    if (ctx.h_stage_bytes < kSmallStage) {
        if (ctx.h_stage) (void)hipHostFree(ctx.h_stage);
        ctx.h_stage = nullptr;
        ctx.h_stage_bytes = 0;
        DANN_HIP(hipHostMalloc(&ctx.h_stage, kSmallStage, hipHostMallocMapped));
        ctx.h_stage_bytes = kSmallStage;
    }
    void* dbase = nullptr;
    if (hipHostGetDevicePointer(&dbase, ctx.h_stage, 0) != hipSuccess) {
        // (a block another path of this context allocated without the mapping: once more, mapped)
        (void)hipGetLastError();
        (void)hipHostFree(ctx.h_stage);
        ctx.h_stage = nullptr;
        ctx.h_stage_bytes = 0;
        DANN_HIP(hipHostMalloc(&ctx.h_stage, kSmallStage, hipHostMallocMapped));
        ctx.h_stage_bytes = kSmallStage;
        if (hipHostGetDevicePointer(&dbase, ctx.h_stage, 0) != hipSuccess) {
            (void)hipGetLastError();
            return kSmallCallDeclined;
        }
    }
    const uint32_t k = calls[0]->k;
    const size_t in_b = ((size_t)total * qb + 15) & ~(size_t)15, ids_b = ((size_t)total * k * 4 + 15) & ~(size_t)15;
    uint8_t* const h = reinterpret_cast<uint8_t*>(ctx.h_stage);
    uint8_t* const d = reinterpret_cast<uint8_t*>(dbase);
    size_t off = 0;
    for (uint32_t c = 0; c < n; ++c) {
        memcpy(h + off, calls[c]->queries, (size_t)calls[c]->nq * qb);
        off += (size_t)calls[c]->nq * qb;
    }
    // row types whose kernels read a query more than once (PQ: the table build; SQ-8: the compensation) get the queries
    // in device memory: one copy for the whole group of calls; the results still land in the mapped block
    const int dt = idx->cfg.dtype;
    const void* dq = d;
    if (!(dt == DT_F32 || dt == DT_F16 || dt == DT_U8 || dt == DT_I8)) {
        if (int32_t grc = grow_stage(ctx, 0, in_b + 16)) return grc;
        DANN_HIP(hipMemcpyAsync(ctx.stage[0], h, (size_t)total * qb, hipMemcpyHostToDevice, ctx.stream));
        dq = ctx.stage[0];
    }
    int32_t rc = search_device(idx, ctx, dq, nullptr, total, calls[0]->l_value, calls[0]->beam, k,
                               reinterpret_cast<uint32_t*>(d + in_b), reinterpret_cast<float*>(d + in_b + ids_b),
                               reinterpret_cast<dann_search_stats*>(d + in_b + 2 * ids_b), nullptr, nullptr, 0, nullptr);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipStreamSynchronize(ctx.stream));
    const uint32_t* ri = reinterpret_cast<const uint32_t*>(h + in_b);
    const float* rd = reinterpret_cast<const float*>(h + in_b + ids_b);
    const dann_search_stats* rs = reinterpret_cast<const dann_search_stats*>(h + in_b + 2 * ids_b);
    uint32_t q0 = 0;
    for (uint32_t c = 0; c < n; ++c) {
        SmallCall& r = *calls[c];
        memcpy(r.out_ids, ri + (size_t)q0 * k, (size_t)r.nq * k * 4);
        memcpy(r.out_dists, rd + (size_t)q0 * k, (size_t)r.nq * k * 4);
        if (r.out_stats) memcpy(r.out_stats, rs + q0, (size_t)r.nq * sizeof(dann_search_stats));
        r.rc = first_failed_query(rs + q0, r.nq, 0);
        if (r.rc != DANN_OK) grab_error_text(r.text);
        q0 += r.nq;
    }
    return DANN_OK;
}
}  // namespace
}  // extern "C++"

// queries per chunk of the host-pointer pipeline, and the batch size from which it is used
constexpr uint32_t kHostChunk = 16384;

extern "C++" {
namespace {
// is [p, p + bytes) page-locked host memory the device can reach by DMA (hipHostMalloc / hipHostRegister)?  Then the
// pipeline copies straight from / to it; pageable buffers travel through the context's pinned ring.
bool host_pinned(const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    hipPointerAttribute_t at;
    for (const void* q : {p, (const void*)(reinterpret_cast<const uint8_t*>(p) + bytes - 1)}) {
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();  // (a pageable pointer is an "invalid value" to the runtime: not an error of this call)
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

// Page-locking a caller's pageable buffer for the duration of one call (hipHostRegister, mapped): on this runtime the
// first registration of a range costs ~65 us per MB (3.3 ms for the 51 MB of 100 000 f32 queries), registering the same
// range again ~1 us (scratch/probe_host_register.hip) -- a caller that reuses its buffers from call to call, as a
// serving loop does, can be given the zero-copy launch of page-locked memory at no cost from its second call on.
// Several threads may pass one buffer (a shared query block) at the same time: temporary registrations are counted in a
// process-wide table, the last user unregisters.
struct TempPins {
    std::mutex mu;
    std::map<const void*, std::pair<size_t, uint32_t>> live;  // base -> (bytes, users)
    bool acquire(const void* p, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it != live.end()) {
            if (it->second.first < bytes) return false;  // (registered shorter by another caller: not worth untangling)
            ++it->second.second;
            return true;
        }
        if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterMapped) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        live.emplace(p, std::make_pair(bytes, 1u));
        return true;
    }
    void release(const void* p) {
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) return;
        if (--it->second.second == 0) {
            (void)hipHostUnregister(const_cast<void*>(p));
            live.erase(it);
        }
    }
    // page-locked by the CALLER (stable for the call), not by another thread's temporary registration -- decided under the
    // table's lock: a release on another thread unregisters and erases inside it, so the answer is never "pinned, not ours"
    // for memory that is about to lose its pin
    bool caller_pinned(const void* p, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        return live.count(p) == 0 && host_pinned(p, bytes);
    }
};
TempPins& temp_pins() {
    static TempPins t;
    return t;
}
// the set of temporary registrations of one call; everything acquired is released when it goes out of scope
struct PinScope {
    const void* held[4] = {nullptr, nullptr, nullptr, nullptr};
    int n = 0;
    bool add(const void* p, size_t bytes) {
        if (!temp_pins().acquire(p, bytes)) return false;
        held[n++] = p;
        return true;
    }
    ~PinScope() {
        for (int i = 0; i < n; ++i) temp_pins().release(held[i]);
    }
};
}  // namespace
}  // extern "C++"

int32_t dann_search_batch(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value, uint32_t beam_width,
                          uint32_t k, uint32_t* out_ids, float* out_dists, dann_search_stats* out_stats) try {
    if (!idx) {
        set_error("null index");
        return DANN_EINVAL;
    }
    std::shared_lock<std::shared_mutex> _rd(idx->rw);
    DeviceGuard _guard(idx->device);
    if (nq == 0) return DANN_OK;
    if (!queries || !out_ids || !out_dists) return DANN_EINVAL;
    const size_t qb = idx->cfg.dtype == DT_PQ ? (size_t)idx->cfg.dim * 4 : idx->layer_bytes;  // PQ: f32 queries
    const uint32_t pipeline_dbg = idx->dbg_u32(DANN_DBG_HOST_PIPELINE, 1u);  // 0 off, 1 default, 2 .. 8 lanes
    {   // a small call: one launch reading and writing mapped host memory, shared with the small calls of other threads
        if (pipeline_dbg == 1u && nq <= kSmallCall && small_call_bytes(nq, qb, k) <= kSmallStage / 4) {
            SmallCall me;
            me.queries = queries;
            me.nq = nq;
            me.l_value = l_value;
            me.beam = beam_width;
            me.k = k;
            me.out_ids = out_ids;
            me.out_dists = out_dists;
            me.out_stats = out_stats;
            const int32_t src = small_call(idx->comb, me, qb, [&](SmallCall* const* calls, uint32_t n, uint32_t total, std::string& text) {
                const int32_t rrc = small_batch_run(idx, calls, n, total, qb);
                if (rrc != DANN_OK && rrc != kSmallCallDeclined) grab_error_text(text);
                return rrc;
            });
            if (src != DANN_OK && src != kSmallCallDeclined && !me.text.empty()) set_error("%s", me.text.c_str());
            if (src != kSmallCallDeclined) return src;
        }
    }
    CtxLease _lease(idx);
    if (_lease.status != DANN_OK) return _lease.status;
    SearchCtx& ctx = *_lease.ctx;
    const bool pipeline_off = pipeline_dbg == 0u;
    const uint32_t host_chunk = std::max<uint32_t>(idx->dbg_u32(DANN_DBG_HOST_CHUNK, kHostChunk), 256u);
    const bool chunked = nq >= 2 * host_chunk && !pipeline_off;
    const uint32_t cq = chunked ? host_chunk : nq;  // queries per device pass
    // device staging owned by the context (grow-only): [0] queries, [1] ids | dists | stats in one block (the chunked form:
    // one of each per lane, in the lane's own context)
    const size_t ids_b = ((size_t)cq * k * 4 + 15) & ~(size_t)15, st_b = ((size_t)cq * sizeof(dann_search_stats) + 15) & ~(size_t)15;
    const size_t in_b = (size_t)cq * qb, out_b = 2 * ids_b + st_b;
    if (!chunked) {
        if (int32_t rc = grow_stage(ctx, 0, in_b + 16)) return rc;
        if (int32_t rc = grow_stage(ctx, 1, out_b + 16)) return rc;
    }
    // pinned host staging: copies from / to pageable memory are neither asynchronous nor fast on ROCm 7.2
    const bool pinned = !chunked && in_b + out_b <= (1u << 20);
    const size_t h_need = (size_t)1 << 20;
    if (pinned && ctx.h_stage_bytes < h_need) {
        if (ctx.h_stage) (void)hipHostFree(ctx.h_stage);
        ctx.h_stage = nullptr;
        ctx.h_stage_bytes = 0;
        DANN_HIP(hipHostMalloc(&ctx.h_stage, h_need, hipHostMallocMapped));
        ctx.h_stage_bytes = h_need;
    }
    if (!chunked) {
        void* bq = ctx.stage[0];
        uint8_t* ob = reinterpret_cast<uint8_t*>(ctx.stage[1]);
        uint32_t* bi = reinterpret_cast<uint32_t*>(ob);
        float* bd = reinterpret_cast<float*>(ob + ids_b);
        dann_search_stats* bs = reinterpret_cast<dann_search_stats*>(ob + 2 * ids_b);
        std::vector<dann_search_stats> stats(nq);
        if (pinned) {
            uint8_t* hs = reinterpret_cast<uint8_t*>(ctx.h_stage);
            memcpy(hs, queries, in_b);
            DANN_HIP(hipMemcpyAsync(bq, hs, in_b, hipMemcpyHostToDevice, ctx.stream));
            int32_t rc = search_device(idx, ctx, bq, nullptr, nq, l_value, beam_width, k, bi, bd, bs, nullptr, nullptr, 0, nullptr);
            if (rc != DANN_OK) return rc;
            DANN_HIP(hipMemcpyAsync(hs + in_b, ob, out_b, hipMemcpyDeviceToHost, ctx.stream));
            DANN_HIP(hipStreamSynchronize(ctx.stream));
            memcpy(out_ids, hs + in_b, (size_t)nq * k * 4);
            memcpy(out_dists, hs + in_b + ids_b, (size_t)nq * k * 4);
            memcpy(stats.data(), hs + in_b + 2 * ids_b, (size_t)nq * sizeof(dann_search_stats));
        } else {
            DANN_HIP(hipMemcpyAsync(bq, queries, in_b, hipMemcpyHostToDevice, ctx.stream));
            int32_t rc = search_device(idx, ctx, bq, nullptr, nq, l_value, beam_width, k, bi, bd, bs, nullptr, nullptr, 0, nullptr);
            if (rc != DANN_OK) return rc;
            DANN_HIP(hipMemcpyAsync(out_ids, bi, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx.stream));
            DANN_HIP(hipMemcpyAsync(out_dists, bd, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx.stream));
            DANN_HIP(hipMemcpyAsync(stats.data(), bs, (size_t)nq * sizeof(dann_search_stats), hipMemcpyDeviceToHost, ctx.stream));
            DANN_HIP(hipStreamSynchronize(ctx.stream));
        }
        if (out_stats) memcpy(out_stats, stats.data(), (size_t)nq * sizeof(dann_search_stats));
        return first_failed_query(stats.data(), nq, 0);
    }
    // ---- chunked pipeline: up to three lanes -- the calling thread and two helpers, each with a search context (stream,
    // device staging, pinned ring slot) of its own -- take the chunks round robin; a lane runs copy in, kernel, copy out
    // of its chunk back to back on its stream, and the lanes overlap one another: while one lane's kernel drains (the
    // last queries of a batch leave most of the chip idle) or its host thread copies between the caller's pageable
    // buffers and the ring, another lane's kernel has the chip.  Rounds 3-5 ran the chunks' kernels one after the other
    // with the calling thread's copies between them (11.7 M QPS on 100 000 queries where the device-resident call does
    // 18.6 M).  Buffers the caller page-locked (hipHostMalloc / hipHostRegister) need no ring: the DMA reads and writes
    // them directly.
    // (a buffer page-locked by another thread of this process for the length of ITS call -- temp_pins -- is pageable to us)
    const bool q_direct = temp_pins().caller_pinned(queries, (size_t)nq * qb);
    const bool o_direct = temp_pins().caller_pinned(out_ids, (size_t)nq * k * 4) &&
                          temp_pins().caller_pinned(out_dists, (size_t)nq * k * 4);
    int dev = 0;
    DANN_HIP(hipGetDevice(&dev));
    // ---- no copies at all: page-locked, device-mapped caller buffers are read and written by the search kernel itself
    // (a query is read once, when its wavefront stages it; 51 MB in and 8 MB out over the 5 ms of a 100 000-query launch
    // are a fifth of what the link carries).  One launch for the whole batch -- the chunked lanes below pay for their
    // smaller launches (the last queries of every chunk leave the chip half idle).  Row types whose kernels read the
    // query more than once (PQ: the table build; SQ-8: the compensation) keep the lanes.
    const int dt = idx->cfg.dtype;
    const bool zc_rows = dt == DT_F32 || dt == DT_F16 || dt == DT_U8 || dt == DT_I8;
    // Pageable buffers this index has been handed before (same pointers, same batch) are page-locked for the call and take
    // the same launch -- unless registering them turned out to be expensive on this system (then never again).  A buffer
    // another thread of this process has page-locked the same way counts as pageable here, not as the caller's.
    const size_t q_bytes = (size_t)nq * qb, o_bytes = (size_t)nq * k * 4, s_bytes = (size_t)nq * sizeof(dann_search_stats);
    bool q_zc = q_direct;
    bool i_zc = temp_pins().caller_pinned(out_ids, o_bytes);
    bool d_zc = temp_pins().caller_pinned(out_dists, o_bytes);
    bool s_zc = !out_stats || temp_pins().caller_pinned(out_stats, s_bytes);
    PinScope pins;
    if (zc_rows && !(q_zc && i_zc && d_zc && s_zc) && pipeline_dbg == 1u /* (an explicit lane count: lanes only) */ &&
        idx->host_register_pays.load(std::memory_order_relaxed)) {
        uint32_t registered_before = 0;
        bool seen = false;
        {
            std::lock_guard<std::mutex> lk(idx->stat_mu);
            const dann_index::HostCall key{queries, out_ids, out_dists, nq, 0};
            for (auto& h : idx->host_calls)
                if (h == key) {
                    seen = true;
                    registered_before = h.registered++;
                }
            if (!seen) idx->host_calls[idx->host_calls_next++ % 8u] = key;
        }
        if (seen) {
            const auto t0 = std::chrono::steady_clock::now();
            bool ok = true;
            if (ok && !q_zc) ok = pins.add(queries, q_bytes);
            if (ok && !i_zc) ok = pins.add(out_ids, o_bytes);
            if (ok && !d_zc) ok = pins.add(out_dists, o_bytes);
            if (ok && !s_zc) ok = pins.add(out_stats, s_bytes);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            // the first registration of a range is expensive everywhere; if the second one is too, this runtime does not keep
            // ranges warm and the lanes are the better deal
            if (registered_before >= 1 && ms > 1.0) idx->host_register_pays.store(false, std::memory_order_relaxed);
            if (ok) q_zc = i_zc = d_zc = s_zc = true;
        }
    }
    if (zc_rows && q_zc && i_zc && d_zc && s_zc) {
        void *dq = nullptr, *di = nullptr, *dd = nullptr, *ds = nullptr;
        bool mapped = hipHostGetDevicePointer(&dq, const_cast<void*>(queries), 0) == hipSuccess &&
                      hipHostGetDevicePointer(&di, out_ids, 0) == hipSuccess &&
                      hipHostGetDevicePointer(&dd, out_dists, 0) == hipSuccess &&
                      (!out_stats || hipHostGetDevicePointer(&ds, out_stats, 0) == hipSuccess);
        if (!mapped) (void)hipGetLastError();  // (registered without hipHostRegisterMapped: the lanes copy instead)
        if (mapped) {
            std::vector<dann_search_stats> hstats;
            if (!ds) {  // statistics the caller did not ask for: the call's status is still read from them
                if (int32_t rc = grow_stage(ctx, 1, (size_t)nq * sizeof(dann_search_stats) + 16)) return rc;
                ds = ctx.stage[1];
            }
            int32_t rc = search_device(idx, ctx, dq, nullptr, nq, l_value, beam_width, k, static_cast<uint32_t*>(di),
                                       static_cast<float*>(dd), static_cast<dann_search_stats*>(ds), nullptr, nullptr, 0, nullptr);
            if (rc != DANN_OK) return rc;
            DANN_HIP(hipStreamSynchronize(ctx.stream));
            if (!out_stats) {
                hstats.resize(nq);
                DANN_HIP(hipMemcpy(hstats.data(), ds, (size_t)nq * sizeof(dann_search_stats), hipMemcpyDeviceToHost));
                return first_failed_query(hstats.data(), nq, 0);
            }
            return first_failed_query(out_stats, nq, 0);
        }
    }
    const uint32_t nchunks = (nq + cq - 1) / cq;
    auto chunk_len = [&](uint32_t c) { return std::min(cq, nq - c * cq); };
    struct Lane {
        int32_t rc = DANN_OK;          // a call-level failure (HIP, arguments)
        int32_t failed = DANN_OK;      // the first failed query of this lane's chunks
        uint32_t failed_chunk = ~0u;
        std::string text;              // error text of whichever comes first (set_error is thread-local)
    };
    auto grab_text = [](std::string& t) {
        char buf[512];
        buf[0] = 0;
        dann_last_error(buf, sizeof buf);
        t = buf;
    };
    // everything one lane does, on context `lc`; `first` / `step`: its chunks
    auto run_lane = [&](SearchCtx& lc, uint32_t first, uint32_t step, Lane& ln) -> int32_t {
#define DANN_HIP_RC(call)                                  \
    do {                                                   \
        hipError_t e_ = (call);                            \
        if (e_ != hipSuccess) return hip_fail(e_, #call);  \
    } while (0)
        if (int32_t rc = grow_stage(lc, 0, in_b + 16)) return rc;
        if (int32_t rc = grow_stage(lc, 1, out_b + 16)) return rc;
        if (lc.h_stage_bytes < in_b + out_b) {
            if (lc.h_stage) (void)hipHostFree(lc.h_stage);
            lc.h_stage = nullptr;
            lc.h_stage_bytes = 0;
            DANN_HIP_RC(hipHostMalloc(&lc.h_stage, in_b + out_b, hipHostMallocMapped));
            lc.h_stage_bytes = in_b + out_b;
        }
        uint8_t* const h_in = reinterpret_cast<uint8_t*>(lc.h_stage);
        uint8_t* const h_out = h_in + in_b;
        void* const d_in = lc.stage[0];
        uint8_t* const ob = reinterpret_cast<uint8_t*>(lc.stage[1]);
        for (uint32_t c = first; c < nchunks; c += step) {
            const uint32_t n = chunk_len(c);
            const uint8_t* src = reinterpret_cast<const uint8_t*>(queries) + (size_t)c * cq * qb;
            if (!q_direct) {
                memcpy(h_in, src, (size_t)n * qb);
                src = h_in;
            }
            DANN_HIP_RC(hipMemcpyAsync(d_in, src, (size_t)n * qb, hipMemcpyHostToDevice, lc.stream));
            int32_t rc = search_device(idx, lc, d_in, nullptr, n, l_value, beam_width, k, reinterpret_cast<uint32_t*>(ob),
                                       reinterpret_cast<float*>(ob + ids_b), reinterpret_cast<dann_search_stats*>(ob + 2 * ids_b),
                                       nullptr, nullptr, 0, nullptr);
            if (rc != DANN_OK) return rc;
            if (o_direct) {
                DANN_HIP_RC(hipMemcpyAsync(out_ids + (size_t)c * cq * k, ob, (size_t)n * k * 4, hipMemcpyDeviceToHost, lc.stream));
                DANN_HIP_RC(hipMemcpyAsync(out_dists + (size_t)c * cq * k, ob + ids_b, (size_t)n * k * 4, hipMemcpyDeviceToHost,
                                           lc.stream));
                // the statuses always pass through the ring: the call's return value is read from them
                DANN_HIP_RC(hipMemcpyAsync(h_out + 2 * ids_b, ob + 2 * ids_b, (size_t)n * sizeof(dann_search_stats),
                                           hipMemcpyDeviceToHost, lc.stream));
            } else {
                DANN_HIP_RC(hipMemcpyAsync(h_out, ob, out_b, hipMemcpyDeviceToHost, lc.stream));
            }
            DANN_HIP_RC(hipStreamSynchronize(lc.stream));
            const dann_search_stats* st = reinterpret_cast<const dann_search_stats*>(h_out + 2 * ids_b);
            if (!o_direct) {
                memcpy(out_ids + (size_t)c * cq * k, h_out, (size_t)n * k * 4);
                memcpy(out_dists + (size_t)c * cq * k, h_out + ids_b, (size_t)n * k * 4);
            }
            if (out_stats) memcpy(out_stats + (size_t)c * cq, st, (size_t)n * sizeof(dann_search_stats));
            if (ln.failed == DANN_OK) {
                ln.failed = first_failed_query(st, n, c * cq);
                if (ln.failed != DANN_OK) {
                    ln.failed_chunk = c;
                    grab_text(ln.text);
                }
            }
        }
        return DANN_OK;
#undef DANN_HIP_RC
    };
    constexpr uint32_t kMaxLanes = 8, kDefaultLanes = 3;
    const uint32_t want = std::min<uint32_t>(nchunks, pipeline_dbg >= 2u ? std::min(pipeline_dbg, kMaxLanes) : kDefaultLanes);
    // helper lanes take a context only if one is free or may still be created: sixteen callers all waiting for a second
    // context would wait for one another
    std::unique_ptr<CtxLease> extra[kMaxLanes - 1];
    uint32_t lanes = 1;
    for (uint32_t t = 1; t < want; ++t) {
        extra[lanes - 1].reset(new CtxLease(idx, /*try_only=*/true));
        if (extra[lanes - 1]->status != DANN_OK || !extra[lanes - 1]->ctx) {
            extra[lanes - 1].reset();
            break;
        }
        ++lanes;
    }
    Lane ln[kMaxLanes];
    // (the helpers are joined on every way out of this scope: a joinable std::thread must never be destroyed)
    struct Helpers {
        std::thread th[kMaxLanes - 1];
        ~Helpers() {
            for (auto& t : th)
                if (t.joinable()) t.join();
        }
    } helpers;
    bool started[kMaxLanes] = {true};
    for (uint32_t t = 1; t < lanes; ++t) {
        try {
            helpers.th[t - 1] = std::thread([&, t]() {
                try {
                    (void)hipSetDevice(dev);
                    ln[t].rc = run_lane(*extra[t - 1]->ctx, t, lanes, ln[t]);
                    if (ln[t].rc != DANN_OK) grab_text(ln[t].text);
                } catch (...) {
                    ln[t].rc = DANN_EINTERNAL;
                    ln[t].text = "exception in a lane of the host-pointer pipeline";
                }
            });
            started[t] = true;
        } catch (...) {  // no thread to be had: the calling thread takes that lane's chunks after its own
            started[t] = false;
        }
    }
    ln[0].rc = run_lane(ctx, 0, lanes, ln[0]);
    if (ln[0].rc != DANN_OK) grab_text(ln[0].text);
    for (uint32_t t = 1; t < lanes; ++t)
        if (!started[t] && ln[0].rc == DANN_OK) {
            ln[t].rc = run_lane(ctx, t, lanes, ln[t]);
            if (ln[t].rc != DANN_OK) grab_text(ln[t].text);
        }
    for (uint32_t t = 1; t < lanes; ++t)
        if (helpers.th[t - 1].joinable()) helpers.th[t - 1].join();
    for (uint32_t t = 0; t < lanes; ++t)
        if (ln[t].rc != DANN_OK) {
            set_error("%s", ln[t].text.c_str());
            return ln[t].rc;
        }
    const Lane* worst = nullptr;  // the failed query with the smallest index, whichever lane saw it
    for (uint32_t t = 0; t < lanes; ++t)
        if (ln[t].failed != DANN_OK && (!worst || ln[t].failed_chunk < worst->failed_chunk)) worst = &ln[t];
    if (worst) {
        set_error("%s", worst->text.c_str());
        return worst->failed;
    }
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_range_search_batch(dann_index* idx, const void* queries, uint32_t nq, uint32_t starting_l,
                                uint32_t beam_width, float radius, int32_t has_inner_radius, float inner_radius,
                                float initial_slack, float range_slack, uint32_t max_returned, uint32_t out_cap,
                                uint32_t* out_ids, float* out_dists, dann_search_stats* out_stats,
                                uint32_t* out_second_round) try {
    CHECK_IDX_SHARED(idx);  // read-only: concurrent callers run side by side, each on its own context
    ArenaTrim _trim{ctx};
    // RangeSearchError (range_search.rs:30-45, 93-131)
    if (starting_l == 0 || beam_width == 0) {
        set_error("l_value and beam width cannot be zero");
        return DANN_EINVAL;
    }
    if (max_returned && max_returned < starting_l) {
        set_error("max_returned must be greater than or equal to starting_l");
        return DANN_EINVAL;
    }
    if (!(initial_slack >= 0.0f && initial_slack <= 1.0f)) {
        set_error("initial_search_slack must be between 0 and 1.0");
        return DANN_EINVAL;
    }
    if (!(range_slack >= 1.0f)) {
        set_error("range_search_slack must be greater than or equal to 1.0");
        return DANN_EINVAL;
    }
    if (has_inner_radius && inner_radius > radius) {
        set_error("inner_radius must be less than or equal to radius");
        return DANN_EINVAL;
    }
    if (nq == 0) return DANN_OK;
    if (!queries || !out_ids || !out_dists || out_cap == 0) return DANN_EINVAL;
    uint64_t cap = max_returned ? max_returned : (uint64_t)4 * out_cap + 1024;
    cap = std::min<uint64_t>(cap, idx->nslots);
    cap = std::max<uint64_t>(cap, 1);
    if (int32_t prc = pq_ready(idx)) return prc;
    const size_t qb = idx->cfg.dtype == DT_PQ ? (size_t)idx->cfg.dim * 4 : idx->layer_bytes;
    // scratch of this call: one block of the context's grow-only arena (no hipMalloc / hipFree per call: a hipFree
    // synchronises the whole device and would serialise concurrent callers)
    Carve cv;
    const size_t o_q = cv.take((size_t)nq * qb + 16), o_i = cv.take((size_t)nq * out_cap * 4),
                 o_d = cv.take((size_t)nq * out_cap * 4), o_s = cv.take((size_t)nq * sizeof(dann_search_stats)),
                 o_ri = cv.take((size_t)nq * cap * 4), o_rd = cv.take((size_t)nq * cap * 4), o_sec = cv.take((size_t)nq * 4);
    if (int32_t grc = grow_stage(ctx, 4, cv.off)) return grc;
    const ArenaPtr bq{ctx.stage[4], o_q}, bi{ctx.stage[4], o_i}, bd{ctx.stage[4], o_d}, bs{ctx.stage[4], o_s},
        bri{ctx.stage[4], o_ri}, brd{ctx.stage[4], o_rd}, bsec{ctx.stage[4], o_sec};
    DANN_HIP(hipMemcpyAsync(bq.p, queries, (size_t)nq * qb, hipMemcpyHostToDevice, ctx.stream));
    SearchArgs a;
    a.ix = idx->view();
    a.queries = bq.p;
    a.qslots = nullptr;
    a.nq = nq;
    a.l_value = starting_l;
    a.beam_width = beam_width;
    a.k = out_cap;
    a.ht_entries = auto_visited_entries(idx, std::max<uint32_t>(starting_l, 64), beam_width);
    a.out_ids = bi.as<uint32_t>();
    a.out_dists = bd.as<float>();
    a.stats = bs.as<dann_search_stats>();
    a.rec_ids = nullptr;
    a.rec_dists = nullptr;
    a.rec_stride = 0;
    a.rec_n = nullptr;
    a.qmap = nullptr;
    a.range_ids = bri.as<uint32_t>();
    a.range_d = brd.as<float>();
    a.range_second = bsec.as<uint32_t>();
    a.range_cap = (uint32_t)cap;
    a.range_max = max_returned ? max_returned : 0xFFFFFFFFu;
    a.range_thresh = (uint32_t)((float)starting_l * initial_slack);
    a.has_inner = has_inner_radius ? 1u : 0u;
    a.radius = radius;
    a.inner_radius = inner_radius;
    a.range_slack = range_slack;
    a.fail_flag = nullptr;
    a.spill = nullptr;
    a.spill_next = nullptr;
    a.spill_slices = a.spill_bits = 0;
    int32_t rc = search_with_retry(idx, ctx, a);
    if (rc != DANN_OK) return rc;
    std::vector<dann_search_stats> stats(nq);
    DANN_HIP(hipMemcpyAsync(out_ids, bi.p, (size_t)nq * out_cap * 4, hipMemcpyDeviceToHost, ctx.stream));
    DANN_HIP(hipMemcpyAsync(out_dists, bd.p, (size_t)nq * out_cap * 4, hipMemcpyDeviceToHost, ctx.stream));
    DANN_HIP(hipMemcpyAsync(stats.data(), bs.p, (size_t)nq * sizeof(dann_search_stats), hipMemcpyDeviceToHost,
                            ctx.stream));
    if (out_second_round)
        DANN_HIP(hipMemcpyAsync(out_second_round, bsec.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx.stream));
    DANN_HIP(hipStreamSynchronize(ctx.stream));
    if (out_stats) memcpy(out_stats, stats.data(), (size_t)nq * sizeof(dann_search_stats));
    for (uint32_t i = 0; i < nq; ++i)
        if (stats[i].status) {
            set_error("query %u: range result list or visited scratch exhausted (list capacity %llu)", i,
                      (unsigned long long)cap);
            return DANN_EOVERFLOW;
        }
    return DANN_OK;
} DANN_CATCH_ALL

// ---- filtered searches ---------------------------------------------------------------------------
// compute_adaptive_l (inline_filter_search.rs:283-301): f64, truncating casts, host libm
static uint32_t adaptive_l(uint32_t base_l, uint32_t visited, uint32_t matched, double max_multiplier) {
    if (matched == 0 || visited == 0) return (uint32_t)std::min<double>((double)base_l * max_multiplier, 4.0e9);
    const double specificity = (double)matched / (double)visited;
    double multiplier;
    if (specificity >= 0.5) multiplier = 1.0;
    else if (specificity >= 0.1) multiplier = 2.0;
    else multiplier = std::pow(2.0, -std::log10(specificity));
    multiplier = std::min(std::max(multiplier, 1.0), max_multiplier);
    return (uint32_t)std::min<double>((double)base_l * multiplier, 4.0e9);
}

namespace {
struct FilteredCall {
    const void* queries;
    uint32_t nq, l_value, beam, k;
    const dann_filter* filter;
    uint32_t* out_ids;
    float* out_dists;
    dann_search_stats* out_stats;
    // FilteredRange only
    bool range = false;
    float radius = 0.f, inner_radius = 0.f, initial_slack = 1.f, range_slack = 1.f;
    int32_t has_inner = 0;
    uint32_t max_returned = 0;
    uint32_t* out_second = nullptr;
};
}  // namespace

static int32_t filtered_search(dann_index* idx, SearchCtx& ctx, const FilteredCall& c) {
    const dann_filter* f = c.filter;
    if (!f || !f->bits || (f->mode != DANN_FILTER_INLINE && f->mode != DANN_FILTER_MULTIHOP)) {
        set_error("filter: mode must be DANN_FILTER_INLINE or DANN_FILTER_MULTIHOP and bits non-null");
        return DANN_EINVAL;
    }
    if (c.l_value == 0 || c.beam == 0) {
        set_error("l_value and beam_width must be non-zero");
        return DANN_EINVAL;
    }
    if (f->adaptive_samples && (f->mode != DANN_FILTER_INLINE || c.range)) {
        set_error("AdaptiveL applies to InlineFilterSearch only");
        return DANN_EINVAL;
    }
    if (f->adaptive_samples && !(f->adaptive_scale >= 1.0)) {
        set_error("adaptive L scale factor must be >= 1.0");  // AdaptiveLSearchError::ScaleFactorLessThanOne
        return DANN_EINVAL;
    }
    if (c.range && f->mode != DANN_FILTER_INLINE) {
        set_error("FilteredRange runs on the inline filter search (filtered_range_search.rs:148-156)");
        return DANN_EINVAL;
    }
    if (int32_t prc = pq_ready(idx)) return prc;
    const uint32_t nslots = idx->nslots;
    const uint64_t words = (nslots + 31) / 32;
    if (f->stride_words && f->stride_words < words) {
        set_error("filter stride %llu words is shorter than one bitmap (%llu words)",
                  (unsigned long long)f->stride_words, (unsigned long long)words);
        return DANN_EINVAL;
    }
    hipStream_t st = ctx.stream;
    const size_t qb = idx->cfg.dtype == DT_PQ ? (size_t)idx->cfg.dim * 4 : idx->layer_bytes;
    const bool inl = f->mode == DANN_FILTER_INLINE;
    SearchArgs a;
    a.ix = idx->view();
    a.l_value = c.l_value;
    a.beam_width = c.beam;
    a.k = c.k;
    a.ht_entries = auto_visited_entries(idx, c.l_value, c.beam);
    a.filter_mode = f->mode;
    a.filter_stride = f->stride_words;
    // AdaptiveL: table of new L for every (visited, matched) the decision can see
    std::vector<uint32_t> tab;  // (outlives the asynchronous upload below: the arena is sized after it)
    const uint32_t cmax = std::max<uint32_t>((c.beam * idx->cfg.max_degree + 63u) & ~63u, (idx->cfg.num_start_points + 63u) & ~63u);
    if (f->adaptive_samples) {
        const uint64_t stride = (uint64_t)f->adaptive_samples + cmax;
        if (stride * cmax > (64ull << 20)) {
            set_error("AdaptiveL sample_count %u is too large for the decision table", f->adaptive_samples);
            return DANN_EUNSUPPORTED;
        }
        tab.resize(stride * cmax);
        uint32_t lmax = 0;
        for (uint32_t dv = 0; dv < cmax; ++dv)
            for (uint64_t m = 0; m < stride; ++m) {
                const uint32_t v = f->adaptive_samples + dv;
                const uint32_t nl = m <= v ? adaptive_l(c.l_value, v, (uint32_t)m, f->adaptive_scale) : c.l_value;
                tab[dv * stride + m] = nl;
                lmax = std::max(lmax, nl);
            }
        a.ad_samples = f->adaptive_samples;
        a.ad_stride = (uint32_t)stride;
        a.qcap_max = std::max(lmax, c.l_value + idx->cfg.num_start_points);
    }
    // per-launch scratch: the matched list and its sort keys; queries go through in chunks that keep it small
    uint32_t m_cap = 0, key_cap = 0;
    if (inl) {
        m_cap = f->matched_cap ? f->matched_cap : std::min<uint32_t>(nslots, 8192);
        m_cap = std::min<uint32_t>(std::max<uint32_t>(m_cap, 64), nslots);
        key_cap = 1;
        while (key_cap < m_cap + (c.range ? c.l_value : 0)) key_cap <<= 1;
    }
    uint64_t rcap = 0;
    if (c.range) {
        // matched_within_radius: every matched id of the first round within the radius (<= m_cap) plus the
        // second round's appends, which stop at max_returned
        rcap = c.max_returned ? c.max_returned : (uint64_t)4 * c.k + 1024;
        rcap = std::max<uint64_t>(std::min<uint64_t>(std::max<uint64_t>(rcap, m_cap), nslots), 1);
    }
    const bool tie_rust = idx->prune_tie_order == DANN_TIE_RUST;  // equal distances in the reference's own order
    const uint64_t per_query = (uint64_t)m_cap * 8 + (uint64_t)key_cap * 8 + rcap * 8 + 64 + (tie_rust ? kTieWorkBytes : 0);
    const uint32_t chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(c.nq, (1ull << 30) / per_query));
    // scratch of this call: one block of the context's grow-only arena (see dann_range_search_batch)
    const size_t fwords = f->stride_words ? (size_t)f->stride_words * c.nq : (size_t)words;
    Carve cv;
    const size_t o_q = cv.take((size_t)chunk * qb + 16), o_i = cv.take((size_t)chunk * c.k * 4),
                 o_d = cv.take((size_t)chunk * c.k * 4), o_s = cv.take((size_t)chunk * sizeof(dann_search_stats)),
                 o_f = cv.take(fwords * 4), o_tab = cv.take(tab.size() * 4),
                 o_mi = cv.take(inl ? (size_t)chunk * m_cap * 4 : 0), o_md = cv.take(inl ? (size_t)chunk * m_cap * 4 : 0),
                 o_k = cv.take(inl ? (size_t)chunk * key_cap * 8 : 0), o_tw = cv.take(tie_rust ? (size_t)chunk * kTieWorkBytes : 0),
                 o_ri = cv.take(c.range ? (size_t)chunk * rcap * 4 : 0), o_rd = cv.take(c.range ? (size_t)chunk * rcap * 4 : 0),
                 o_sec = cv.take(c.range ? (size_t)chunk * 4 : 0);
    if (int32_t grc = grow_stage(ctx, 4, cv.off)) return grc;
    void* const ar = ctx.stage[4];
    const ArenaPtr bq{ar, o_q}, bi{ar, o_i}, bd{ar, o_d}, bs{ar, o_s}, bf{ar, o_f}, btab{ar, o_tab}, bmi{ar, o_mi}, bmd{ar, o_md},
        bk{ar, o_k}, btw{ar, o_tw}, bri{ar, o_ri}, brd{ar, o_rd}, bsec{ar, o_sec};
    DANN_HIP(hipMemcpyAsync(bf.p, f->bits, fwords * 4, hipMemcpyHostToDevice, st));
    if (!tab.empty()) {
        DANN_HIP(hipMemcpyAsync(btab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st));
        a.ad_table = btab.as<uint32_t>();
    }
    if (tie_rust) a.tie_work = btw.as<uint8_t>();
    if (inl) {
        a.m_ids = bmi.as<uint32_t>();
        a.m_d = bmd.as<float>();
        a.m_keys = bk.as<unsigned long long>();
        a.m_cap = m_cap;
        a.key_cap = key_cap;
    }
    if (c.range) {
        a.range_ids = bri.as<uint32_t>();
        a.range_d = brd.as<float>();
        a.range_second = bsec.as<uint32_t>();
        a.range_cap = (uint32_t)rcap;
        a.range_max = c.max_returned ? c.max_returned : 0xFFFFFFFFu;
        a.range_thresh = (uint32_t)((float)c.l_value * c.initial_slack);
        a.has_inner = c.has_inner ? 1u : 0u;
        a.radius = c.radius;
        a.inner_radius = c.inner_radius;
        a.range_slack = c.range_slack;
    }
    a.out_ids = bi.as<uint32_t>();
    a.out_dists = bd.as<float>();
    a.stats = bs.as<dann_search_stats>();
    a.queries = bq.p;
    std::vector<dann_search_stats> stats(chunk);
    for (uint32_t off = 0; off < c.nq; off += chunk) {
        const uint32_t n = std::min(chunk, c.nq - off);
        DANN_HIP(hipMemcpyAsync(bq.p, (const uint8_t*)c.queries + (size_t)off * qb, (size_t)n * qb, hipMemcpyHostToDevice, st));
        a.nq = n;
        a.filter = bf.as<uint32_t>() + (size_t)off * f->stride_words;
        int32_t rc = search_with_retry(idx, ctx, a);
        if (rc != DANN_OK) return rc;
        DANN_HIP(hipMemcpyAsync(c.out_ids + (size_t)off * c.k, bi.p, (size_t)n * c.k * 4, hipMemcpyDeviceToHost, st));
        DANN_HIP(hipMemcpyAsync(c.out_dists + (size_t)off * c.k, bd.p, (size_t)n * c.k * 4, hipMemcpyDeviceToHost, st));
        DANN_HIP(hipMemcpyAsync(stats.data(), bs.p, (size_t)n * sizeof(dann_search_stats), hipMemcpyDeviceToHost, st));
        if (c.out_second)
            DANN_HIP(hipMemcpyAsync(c.out_second + off, bsec.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        DANN_HIP(hipStreamSynchronize(st));
        if (c.out_stats) memcpy(c.out_stats + off, stats.data(), (size_t)n * sizeof(dann_search_stats));
        for (uint32_t i = 0; i < n; ++i)
            if (stats[i].status) {
                set_error("query %u: per-query scratch exhausted (matched list %u entries, result list %llu, visited "
                          "table); raise dann_filter.matched_cap / out_cap", off + i, m_cap, (unsigned long long)rcap);
                return DANN_EOVERFLOW;
            }
    }
    return DANN_OK;
}

int32_t dann_filtered_search_batch(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value,
                                   uint32_t beam_width, uint32_t k, const dann_filter* filter, uint32_t* out_ids,
                                   float* out_dists, dann_search_stats* out_stats) try {
    CHECK_IDX_SHARED(idx);
    ArenaTrim _trim{ctx};
    if (nq == 0) return DANN_OK;
    if (!queries || !out_ids || !out_dists || k == 0) return DANN_EINVAL;
    FilteredCall c{queries, nq, l_value, beam_width, k, filter, out_ids, out_dists, out_stats};
    return filtered_search(idx, ctx, c);
} DANN_CATCH_ALL

int32_t dann_filtered_range_search_batch(dann_index* idx, const void* queries, uint32_t nq, uint32_t starting_l,
                                         uint32_t beam_width, float radius, int32_t has_inner_radius,
                                         float inner_radius, float initial_slack, float range_slack,
                                         uint32_t max_returned, uint32_t out_cap, const dann_filter* filter,
                                         uint32_t* out_ids, float* out_dists, dann_search_stats* out_stats,
                                         uint32_t* out_second_round) try {
    CHECK_IDX_SHARED(idx);
    ArenaTrim _trim{ctx};
    // RangeSearchError (range_search.rs:30-45, 93-131)
    if (starting_l == 0 || beam_width == 0) {
        set_error("l_value and beam width cannot be zero");
        return DANN_EINVAL;
    }
    if (max_returned && max_returned < starting_l) {
        set_error("max_returned must be greater than or equal to starting_l");
        return DANN_EINVAL;
    }
    if (!(initial_slack >= 0.0f && initial_slack <= 1.0f)) {
        set_error("initial_search_slack must be between 0 and 1.0");
        return DANN_EINVAL;
    }
    if (!(range_slack >= 1.0f)) {
        set_error("range_search_slack must be greater than or equal to 1.0");
        return DANN_EINVAL;
    }
    if (has_inner_radius && inner_radius > radius) {
        set_error("inner_radius must be less than or equal to radius");
        return DANN_EINVAL;
    }
    if (nq == 0) return DANN_OK;
    if (!queries || !out_ids || !out_dists || out_cap == 0) return DANN_EINVAL;
    FilteredCall c{queries, nq, starting_l, beam_width, out_cap, filter, out_ids, out_dists, out_stats};
    c.range = true;
    c.radius = radius;
    c.inner_radius = inner_radius;
    c.initial_slack = initial_slack;
    c.range_slack = range_slack;
    c.has_inner = has_inner_radius;
    c.max_returned = max_returned;
    c.out_second = out_second_round;
    return filtered_search(idx, ctx, c);
} DANN_CATCH_ALL

int32_t dann_rerank_batch_device(dann_index* idx, const void* d_queries, uint32_t nq, const uint32_t* d_cand_ids,
                                 uint32_t cand_stride, uint32_t k, uint32_t* d_out_ids, float* d_out_dists) try {
    CHECK_IDX(idx);
    if (nq == 0) return DANN_OK;
    if (!d_queries || !d_cand_ids || !d_out_ids || !d_out_dists || k == 0) return DANN_EINVAL;
    int32_t rc = timed(idx, 1, [&] {
        return launch_rerank(idx->view(), d_queries, nq, d_cand_ids, cand_stride, k, d_out_ids, d_out_dists,
                             idx->main.stream);
    });
    return rc;
} DANN_CATCH_ALL

int32_t dann_rerank_batch(dann_index* idx, const void* queries, uint32_t nq, const uint32_t* cand_ids,
                          uint32_t cand_stride, uint32_t k, uint32_t* out_ids, float* out_dists) try {
    CHECK_IDX(idx);
    if (nq == 0) return DANN_OK;
    if (!queries || !cand_ids || !out_ids || !out_dists || k == 0) return DANN_EINVAL;
    DevBuf bq, bc, bi, bd;
    DANN_HIP(bq.alloc((size_t)nq * idx->layer_bytes + 16));
    DANN_HIP(bc.alloc((size_t)nq * cand_stride * 4));
    DANN_HIP(bi.alloc((size_t)nq * k * 4));
    DANN_HIP(bd.alloc((size_t)nq * k * 4));
    DANN_HIP(hipMemcpyAsync(bq.p, queries, (size_t)nq * idx->layer_bytes, hipMemcpyHostToDevice, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(bc.p, cand_ids, (size_t)nq * cand_stride * 4, hipMemcpyHostToDevice, idx->main.stream));
    int32_t rc = launch_rerank(idx->view(), bq.p, nq, bc.as<uint32_t>(), cand_stride, k, bi.as<uint32_t>(),
                               bd.as<float>(), idx->main.stream);
    if (rc != DANN_OK) return rc;
    DANN_HIP(hipMemcpyAsync(out_ids, bi.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(out_dists, bd.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_search_record_batch(dann_index* idx, const uint32_t* slots, uint32_t nq, uint32_t l_value,
                                 uint32_t* rec_ids, float* rec_dists, uint32_t rec_stride, uint32_t* rec_n,
                                 dann_search_stats* out_stats) try {
    CHECK_IDX(idx);
    if (nq == 0) return DANN_OK;
    if (!slots || !rec_ids || !rec_dists || !rec_n || rec_stride == 0) return DANN_EINVAL;
    for (uint32_t i = 0; i < nq; ++i)
        if (slots[i] >= idx->nslots) return DANN_EBOUNDS;
    DevBuf bsl, bri, brd, brn, bs;
    DANN_HIP(bsl.alloc((size_t)nq * 4));
    DANN_HIP(bri.alloc((size_t)nq * rec_stride * 4));
    DANN_HIP(brd.alloc((size_t)nq * rec_stride * 4));
    DANN_HIP(brn.alloc((size_t)nq * 4));
    DANN_HIP(bs.alloc((size_t)nq * sizeof(dann_search_stats)));
    DANN_HIP(hipMemcpyAsync(bsl.p, slots, (size_t)nq * 4, hipMemcpyHostToDevice, idx->main.stream));
    int32_t rc = search_device(idx, idx->main, nullptr, bsl.as<uint32_t>(), nq, l_value, 1, 0, nullptr, nullptr,
                               bs.as<dann_search_stats>(), bri.as<uint32_t>(), brd.as<float>(), rec_stride,
                               brn.as<uint32_t>());
    if (rc != DANN_OK) return rc;
    std::vector<dann_search_stats> stats(nq);
    DANN_HIP(hipMemcpyAsync(rec_ids, bri.p, (size_t)nq * rec_stride * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(rec_dists, brd.p, (size_t)nq * rec_stride * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(rec_n, brn.p, (size_t)nq * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(stats.data(), bs.p, (size_t)nq * sizeof(dann_search_stats), hipMemcpyDeviceToHost,
                            idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    if (out_stats) memcpy(out_stats, stats.data(), (size_t)nq * sizeof(dann_search_stats));
    for (uint32_t i = 0; i < nq; ++i)
        if (stats[i].status) {
            set_error("query %u: visited table or record buffer exhausted", i);
            return DANN_EOVERFLOW;
        }
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_search_record_queries(dann_index* idx, const void* queries, uint32_t nq, uint32_t l_value, uint32_t* rec_ids,
                                   float* rec_dists, uint32_t rec_stride, uint32_t* rec_n, dann_search_stats* out_stats) try {
    CHECK_IDX(idx);
    if (nq == 0) return DANN_OK;
    if (!queries || !rec_ids || !rec_dists || !rec_n || rec_stride == 0) return DANN_EINVAL;
    const size_t qb = idx->cfg.dtype == DT_PQ ? (size_t)idx->cfg.dim * 4 : idx->layer_bytes;
    DevBuf bq, bri, brd, brn, bs;
    DANN_HIP(bq.alloc((size_t)nq * qb + 16));
    DANN_HIP(bri.alloc((size_t)nq * rec_stride * 4));
    DANN_HIP(brd.alloc((size_t)nq * rec_stride * 4));
    DANN_HIP(brn.alloc((size_t)nq * 4));
    DANN_HIP(bs.alloc((size_t)nq * sizeof(dann_search_stats)));
    DANN_HIP(hipMemcpyAsync(bq.p, queries, (size_t)nq * qb, hipMemcpyHostToDevice, idx->main.stream));
    int32_t rc = search_device(idx, idx->main, bq.p, nullptr, nq, l_value, 1, 0, nullptr, nullptr, bs.as<dann_search_stats>(),
                               bri.as<uint32_t>(), brd.as<float>(), rec_stride, brn.as<uint32_t>());
    if (rc != DANN_OK) return rc;
    std::vector<dann_search_stats> stats(nq);
    DANN_HIP(hipMemcpyAsync(rec_ids, bri.p, (size_t)nq * rec_stride * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(rec_dists, brd.p, (size_t)nq * rec_stride * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(rec_n, brn.p, (size_t)nq * 4, hipMemcpyDeviceToHost, idx->main.stream));
    DANN_HIP(hipMemcpyAsync(stats.data(), bs.p, (size_t)nq * sizeof(dann_search_stats), hipMemcpyDeviceToHost,
                            idx->main.stream));
    DANN_HIP(hipStreamSynchronize(idx->main.stream));
    if (out_stats) memcpy(out_stats, stats.data(), (size_t)nq * sizeof(dann_search_stats));
    for (uint32_t i = 0; i < nq; ++i)
        if (stats[i].status) {
            set_error("query %u: visited table or record buffer exhausted", i);
            return DANN_EOVERFLOW;
        }
    return DANN_OK;
} DANN_CATCH_ALL

// ---- on-disk formats ---------------------------------------------------------------------------
struct File {
    FILE* f = nullptr;
    ~File() {
        if (f) fclose(f);
    }
};

int32_t dann_save_graph(const dann_index* idx, const char* path) try {
    CHECK_IDX(idx);
    if (!path) return DANN_EINVAL;
    const uint32_t w = idx->cfg.max_degree + 1;
    std::vector<uint32_t> adj((size_t)idx->nslots * w);
    int32_t rc = dann_download_graph(idx, adj.data(), idx->nslots);
    if (rc != DANN_OK) return rc;
    File out;
    out.f = fopen(path, "wb");
    if (!out.f) {
        set_error("cannot open %s for writing", path);
        return DANN_EINVAL;
    }
    uint64_t file_size = 24;
    for (uint32_t i = 0; i < idx->nslots; ++i) file_size += 4ull * (1 + std::min(adj[(size_t)i * w], idx->cfg.max_degree));
    const uint32_t max_degree = idx->cfg.max_degree, start = idx->cfg.capacity;
    const uint64_t nstart = idx->cfg.num_start_points;
    bool ok = fwrite(&file_size, 8, 1, out.f) == 1 && fwrite(&max_degree, 4, 1, out.f) == 1 &&
              fwrite(&start, 4, 1, out.f) == 1 && fwrite(&nstart, 8, 1, out.f) == 1;
    for (uint32_t i = 0; ok && i < idx->nslots; ++i) {
        const uint32_t len = std::min(adj[(size_t)i * w], idx->cfg.max_degree);
        ok = fwrite(&len, 4, 1, out.f) == 1 && (len == 0 || fwrite(&adj[(size_t)i * w + 1], 4, len, out.f) == len);
    }
    if (!ok) {
        set_error("short write to %s", path);
        return DANN_EINVAL;
    }
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_load_graph(dann_index* idx, const char* path, uint32_t* out_start, uint64_t* out_num_start,
                        uint64_t* out_num_points) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (!path) return DANN_EINVAL;
    File in;
    in.f = fopen(path, "rb");
    if (!in.f) {
        set_error("cannot open %s", path);
        return DANN_EINVAL;
    }
    uint64_t file_size = 0, nstart = 0;
    uint32_t max_degree = 0, start = 0;
    if (fread(&file_size, 8, 1, in.f) != 1 || fread(&max_degree, 4, 1, in.f) != 1 || fread(&start, 4, 1, in.f) != 1 ||
        fread(&nstart, 8, 1, in.f) != 1) {
        set_error("%s: truncated header", path);
        return DANN_ELENGTH;
    }
    const uint32_t w = idx->cfg.max_degree + 1;
    std::vector<uint32_t> adj((size_t)idx->nslots * w, 0u);
    std::vector<uint32_t> buf;
    uint64_t pos = 24, npts = 0;
    while (pos < file_size) {
        uint32_t len = 0;
        if (fread(&len, 4, 1, in.f) != 1) {
            set_error("%s: truncated adjacency list %llu", path, (unsigned long long)npts);
            return DANN_ELENGTH;
        }
        // a list can be neither longer than what the header / this index allow nor than the rest of the file
        if (len > std::max(max_degree, idx->cfg.max_degree) || 4ull * len > file_size - pos) {
            set_error("%s: adjacency list %llu claims %u neighbours (header max degree %u)", path,
                      (unsigned long long)npts, len, max_degree);
            return DANN_ETOOLONG;
        }
        buf.resize(len);
        if (len && fread(buf.data(), 4, len, in.f) != len) {
            set_error("%s: truncated adjacency list %llu", path, (unsigned long long)npts);
            return DANN_ELENGTH;
        }
        if (npts < idx->nslots) {
            if (len > idx->cfg.max_degree) {
                set_error("%s: node %llu has %u neighbours, index max_degree is %u", path, (unsigned long long)npts, len,
                          idx->cfg.max_degree);
                return DANN_ETOOLONG;
            }
            adj[(size_t)npts * w] = len;
            if (len) memcpy(&adj[(size_t)npts * w + 1], buf.data(), (size_t)len * 4);
        }
        pos += 4ull * (1 + len);
        ++npts;
    }
    int32_t rc = dann_upload_graph(idx, adj.data(), idx->nslots);
    if (rc != DANN_OK) return rc;
    if (out_start) *out_start = start;
    if (out_num_start) *out_num_start = nstart;
    if (out_num_points) *out_num_points = npts;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_save_vectors_bin(const dann_index* idx, const char* path, uint32_t first_slot, uint32_t n) try {
    CHECK_IDX(idx);
    if (!path) return DANN_EINVAL;
    if ((uint64_t)first_slot + n > idx->nslots) return DANN_EBOUNDS;
    std::vector<uint8_t> rows((size_t)n * idx->layer_bytes);
    if (n) {
        DANN_HIP(hipMemcpy2DAsync(rows.data(), idx->layer_bytes, idx->d_rows + (size_t)first_slot * idx->cfg.row_stride,
                                  idx->cfg.row_stride, idx->layer_bytes, n, hipMemcpyDeviceToHost, idx->main.stream));
        DANN_HIP(hipStreamSynchronize(idx->main.stream));
    }
    File out;
    out.f = fopen(path, "wb");
    if (!out.f) {
        set_error("cannot open %s for writing", path);
        return DANN_EINVAL;
    }
    // `.bin`: dim counts elements of the stored type (SQ-8 rows are written as dim + 4 bytes)
    const uint32_t dim = idx->cfg.dtype == DT_SQ8 ? idx->layer_bytes : idx->cfg.dim;
    if (fwrite(&n, 4, 1, out.f) != 1 || fwrite(&dim, 4, 1, out.f) != 1 ||
        (rows.size() && fwrite(rows.data(), 1, rows.size(), out.f) != rows.size())) {
        set_error("short write to %s", path);
        return DANN_EINVAL;
    }
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_load_vectors_bin(dann_index* idx, const char* path, uint32_t first_slot, uint32_t* out_n) try {
    CHECK_IDX(idx);
    DANN_MUTATION(idx);
    if (!path) return DANN_EINVAL;
    File in;
    in.f = fopen(path, "rb");
    if (!in.f) {
        set_error("cannot open %s", path);
        return DANN_EINVAL;
    }
    uint32_t n = 0, dim = 0;
    if (fread(&n, 4, 1, in.f) != 1 || fread(&dim, 4, 1, in.f) != 1) return DANN_ELENGTH;
    const uint32_t want_dim = idx->cfg.dtype == DT_SQ8 ? idx->layer_bytes : idx->cfg.dim;
    if (dim != want_dim) {
        set_error("data of dimension %u does not match full precision layer's dimension %u", dim, want_dim);
        return DANN_ELENGTH;
    }
    if ((uint64_t)first_slot + n > idx->cfg.capacity) return DANN_EBOUNDS;
    std::vector<uint8_t> rows((size_t)n * idx->layer_bytes);
    if (rows.size() && fread(rows.data(), 1, rows.size(), in.f) != rows.size()) {
        set_error("%s: truncated payload", path);
        return DANN_ELENGTH;
    }
    int32_t rc = dann_set_elements(idx, first_slot, n, rows.data(), rows.size());
    if (rc != DANN_OK) return rc;
    if (out_n) *out_n = n;
    return DANN_OK;
} DANN_CATCH_ALL

// ---- diagnostics -----------------------------------------------------------------------------
int32_t dann_abi_version(void) { return DANN_ABI_VERSION; }
int32_t dann_kernel_time(const dann_index* idx, int32_t which, double* total_ms, uint64_t* launches) try {
    if (!idx || which < 0 || which > 5) return DANN_EINVAL;
    if (which == 5) {  // the build's MFMA Gram tiles: events on the build stream, resolved here
        ::dann::ExclusiveGuard lock(idx);
        return build_tile_clock(idx, total_ms, launches);
    }
    std::lock_guard<std::mutex> lk(idx->stat_mu);
    if (total_ms) *total_ms = idx->clocks[which].total_ms;
    if (launches) *launches = idx->clocks[which].launches;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_kernel_time_reset(dann_index* idx) try {
    if (!idx) return DANN_EINVAL;
    {
        ::dann::ExclusiveGuard lock(idx);
        build_tile_clock_reset(idx);
    }
    std::lock_guard<std::mutex> lk(idx->stat_mu);
    for (auto& c : idx->clocks) c = KernelClock();
    for (auto& c : idx->families) c = KernelClock();
    return DANN_OK;
} DANN_CATCH_ALL

// ---- development switches and the per-family launch counters (include/dann_debug.h) ----------------------------------
int32_t dann_debug_set(dann_index* idx, int32_t key, double value) try {
    if (!idx || key < 0 || key >= DANN_DBG_COUNT) return DANN_EINVAL;
    idx->dbg[key].store(value, std::memory_order_relaxed);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_debug_get(const dann_index* idx, int32_t key, double* value) try {
    if (!idx || !value || key < 0 || key >= DANN_DBG_COUNT) return DANN_EINVAL;
    *value = idx->dbg[key].load(std::memory_order_relaxed);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_debug_small_call_stats(dann_index* idx, uint64_t* out2) try {
    if (!idx || !out2) return DANN_EINVAL;
    out2[0] = idx->comb.stats[0].load(std::memory_order_relaxed);
    out2[1] = idx->comb.stats[1].load(std::memory_order_relaxed);
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_debug_search_families(const dann_index* idx, uint64_t* out_launches, double* out_ms) try {
    if (!idx) return DANN_EINVAL;
    std::lock_guard<std::mutex> lk(idx->stat_mu);
    for (int f = 0; f < DANN_FAMILY_COUNT; ++f) {
        if (out_launches) out_launches[f] = idx->families[f].launches;
        if (out_ms) out_ms[f] = idx->families[f].total_ms;
    }
    return DANN_OK;
} DANN_CATCH_ALL

const char* dann_debug_family_name(int32_t family) {
    static const char* const names[DANN_FAMILY_COUNT] = {"one_wave", "team", "pair", "persistent", "server", "pq_lut"};
    return family >= 0 && family < DANN_FAMILY_COUNT ? names[family] : nullptr;
}

int32_t dann_set_visited_bits(dann_index* idx, uint32_t bits) try {
    // 0 = automatic; 6..15 = log2(entries); >= 64 = explicit entry count (rounded up to a multiple of 64)
    if (!idx || (bits != 0 && bits < 64 && (bits < 6 || bits > 15)) || bits > 32768) return DANN_EINVAL;
    idx->visited_bits = bits;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_visited_format(dann_index* idx, uint32_t entry_bits) try {
    if (!idx || (entry_bits != 0 && entry_bits != 16 && entry_bits != 32)) return DANN_EINVAL;
    idx->visited_format = entry_bits;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_max_concurrency(dann_index* idx, uint32_t max_queries_in_flight) try {
    if (!idx) return DANN_EINVAL;
    idx->max_concurrency = max_queries_in_flight;
    return DANN_OK;
} DANN_CATCH_ALL

int32_t dann_set_prune_tie_order(dann_index* idx, uint32_t order) try {
    if (!idx || (order != DANN_TIE_POSITION && order != DANN_TIE_RUST)) return DANN_EINVAL;
    dann::ExclusiveGuard g(idx);  // not while a build is running on another thread
    idx->prune_tie_order = order;
    return DANN_OK;
} DANN_CATCH_ALL

}  // extern "C"
