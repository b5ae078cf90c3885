// Beam-search kernel, its device helpers and the launch dispatch.  Included by one translation unit per row type
// (search_f32.hip ... search_pq.hip) so the instantiations compile in parallel, and by search_kernels.hip for the
// host-side sizing helpers.  Everything lives in an anonymous namespace: each includer gets its own copy.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "dann_device.h"
#include "rust_order.h"
#include "dann_internal.h"

namespace dann {
namespace {

constexpr int kWave = 64;
constexpr int kMaxBeam = 16;
constexpr int kGatherRows = 4;
#ifndef DANN_WIDE_ROWS
#define DANN_WIDE_ROWS 2
#endif
constexpr int kWideRows = DANN_WIDE_ROWS;  // rows per lane group and trip in the wide (f16) gather
#ifndef DANN_REG_MERGE
#define DANN_REG_MERGE 16
#endif
constexpr uint32_t kRegMerge = DANN_REG_MERGE;  // survivors handled by the in-register merge
#ifndef DANN_SEQ_INSERT
#define DANN_SEQ_INSERT 2
#endif
constexpr uint32_t kSeqInsert = DANN_SEQ_INSERT;  // survivors inserted one by one, the queue never leaving its registers
constexpr uint32_t kTuneRowPrefetch = 1u;  // SearchArgs::tune bits
constexpr uint32_t kTuneNoSpeculation = 2u;  // teams: no speculative expansion of the predicted next node
constexpr uint32_t kTuneNoSelfStart = 4u;    // teams: the visited wave always waits for the control wave's words
constexpr uint8_t kTagPublished = 254;  // Tag::can_read: tag >= PUBLISHED (diskann-inmem/src/tag.rs:86-133)

// optional per-phase cycle accounting (compile with -DDANN_PHASE_CYCLES; debug only)
#ifdef DANN_PHASE_CYCLES
#define PH_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PH_ADD(idx, t0, t1) ph_acc[idx] += (t1) - (t0)
#else
#define PH_T(var)
#define PH_ADD(idx, t0, t1)
#endif  // rows in flight per lane group in the fixed-length gather

constexpr uint32_t kAdjLandBytes = 256u;
struct SearchLds {
    uint32_t ht_off, cand_id_off, cand_d_off, cand2_id_off, cand2_d_off, adj_off, slots_off, mscr_off, mail_off, stage_off, snew_off, beam_off, q_off, total;
};

__host__ __device__ inline uint32_t round16(uint32_t x) { return (x + 15u) & ~15u; }
// entries of the queue image in LDS: the largest capacity the queue can have during the search
__host__ __device__ inline uint32_t lds_queue_entries(const SearchArgs& a) {
    const uint32_t q = a.l_value + a.ix.nstart;
    return q > a.qcap_max ? q : a.qcap_max;
}

// bytes of the staged query: f32 vector (float rows), raw bytes (integer rows), lookup table (PQ rows)
__host__ __device__ inline uint32_t query_lds_bytes(const IndexView& ix) {
    if (ix.dtype == DT_PQ) return ix.pq_chunks * 1024u;
    if (ix.dtype == DT_U8 || ix.dtype == DT_I8 || ix.dtype == DT_SQ8) return ix.layer_bytes;
    return ix.dim * 4u;
}

// The queue image and the visited table come last, in that order: every other region then sits at an offset that
// depends only on (cmax, query bytes) -- compile-time constants in the plain fixed-length instantiations, where the
// region pointers cost no SGPRs and the LDS instructions carry immediate offsets; the image is sized by the queue's
// real capacity `qcap` (L + start points, or what AdaptiveL may grow it to), not by its register slots, so the table's
// offset is the one run-time offset.  (Every byte counts: at 1 M x 128-byte rows the table caps the queries per CU.)
__host__ __device__ inline SearchLds search_lds_layout(uint32_t ht_entries, uint32_t cmax, uint32_t qcap,
                                                       uint32_t qbytes, bool team = false) {
    SearchLds l;
    uint32_t off = 0;
    l.q_off = off;
    off += round16(qbytes);
    l.cand_id_off = off;
    off += round16(cmax * 4u);
    l.cand_d_off = off;
    off += round16(cmax * 4u);
    // teams: a second candidate buffer -- the visited wave fills it with the next hop's candidates while the gather wave
    // still evaluates the current hop's (the speculative expansion of the predicted next node, §3.5 of DESIGN.md)
    l.cand2_id_off = off;
    if (team) off += round16(cmax * 4u);
    l.cand2_d_off = off;
    if (team) off += round16(cmax * 4u);
    // teams: two landing buffers of 64 dwords for adjacency rows requested ahead of their use (length + at most 63
    // neighbours each; the loads write LDS directly, see adj_fetch_lds)
    l.adj_off = off;
    if (team) off += 2u * kAdjLandBytes;
    // teams: the table slots of the visited wave's speculative inserts (wave 0 takes them back through these), and 64
    // (id, distance) words of scratch for wave 0's merge (the candidate buffer it merges from is already being refilled)
    l.slots_off = off;
    if (team) off += 256u;
    l.mscr_off = off;
    if (team) off += 512u;
    l.mail_off = off;  // teams: the mailbox the four waves of a team talk through (64 words, see kMb*)
    if (team) off += 256u;
    l.beam_off = off;
    off += round16(kMaxBeam * 4u);
    l.stage_off = off;  // the queue image, (id, distance bits) pairs: every merge scatters the register-resident queue
    off += round16(qcap * 8u);  // here and reloads it (one 8-byte LDS access per entry).  One buffer is enough: nothing
                                // is read from it between the first scatter write and the reload (ranks come from
                                // registers or were taken before), and one wave's LDS operations retire in order.
    l.snew_off = l.cand_id_off;  // (unused: the slow merge keeps its sorted survivors in the candidates' own buffer)
    l.ht_off = off;        // 16-byte aligned (wiped with 16-byte stores)
    off += ht_entries * 4u;  // any multiple of 64
    l.total = off;
    return l;
}

// exact visited set: open addressing, linear probing, ds_cmpst.  == hashbrown::HashSet::insert
// (glue.rs:542-549).  Two levels: the LDS table takes ids until it is 75 % full ("open");
// after that it is frozen (lookups only) and new ids go to a spill table in global memory
// claimed from a small pool -- rare, slower, still exact.
enum : int { kPresent = 0, kInserted = 1, kAbsent = 2 };
__device__ __forceinline__ int ht_visit(uint32_t* ht, uint32_t mod, uint32_t id, bool open) {
    // double hashing over a prime number of slots (<= the allocated entries): the wave waits for its slowest
    // lane, and linear probing's cluster tails made that several times the mean probe count
    uint32_t h = __umulhi(id * 2654435761u, mod);
    const uint32_t step = 1u + __umulhi(id * 2246822519u + 0x9E3779B9u, mod - 1u);
    for (;;) {
        uint32_t old = open ? atomicCAS(&ht[h], kEmpty, id) : ht[h];
        if (old == kEmpty) return open ? kInserted : kAbsent;
        if (old == id) return kPresent;
        h += step;
        h = h >= mod ? h - mod : h;
    }
}
// insert into the open table (the hot path of every hop): first probe straight-line for all lanes, the double-hashing
// loop only runs when some lane collided.  Same probe sequence as ht_visit.  Returns true when `id` was not yet present.
__device__ __forceinline__ bool ht_insert_open(uint32_t* ht, uint32_t mod, uint32_t id, bool active) {
    uint32_t h = __umulhi(id * 2654435761u, mod);
    uint32_t old = active ? atomicCAS(&ht[h], kEmpty, id) : id;
    bool isnew = active && old == kEmpty;
    bool pending = active && old != kEmpty && old != id;
    if (ballot64(pending)) {
        const uint32_t step = 1u + __umulhi(id * 2246822519u + 0x9E3779B9u, mod - 1u);
        while (pending) {
            h += step;
            h = h >= mod ? h - mod : h;
            old = atomicCAS(&ht[h], kEmpty, id);
            isnew = old == kEmpty;
            pending = old != kEmpty && old != id;
        }
    }
    return isnew;
}
// the same, also reporting the slot the id went into (for a rollback of speculative inserts: the slot is set back to
// kEmpty, which restores the table exactly -- an insert only ever fills an empty slot)
__device__ __forceinline__ bool ht_insert_open_slot(uint32_t* ht, uint32_t mod, uint32_t id, bool active, uint32_t* slot) {
    uint32_t h = __umulhi(id * 2654435761u, mod);
    uint32_t old = active ? atomicCAS(&ht[h], kEmpty, id) : id;
    bool isnew = active && old == kEmpty;
    bool pending = active && old != kEmpty && old != id;
    if (ballot64(pending)) {
        const uint32_t step = 1u + __umulhi(id * 2246822519u + 0x9E3779B9u, mod - 1u);
        while (pending) {
            h += step;
            h = h >= mod ? h - mod : h;
            old = atomicCAS(&ht[h], kEmpty, id);
            isnew = old == kEmpty;
            pending = old != kEmpty && old != id;
        }
    }
    *slot = h;
    return isnew;
}
// ---- the visited table with 16-bit entries (SearchArgs::ht16) ------------------------------------------------------
// Half the LDS per id, still an exact set.  Ids live below 2^m (m = bits of the slot count of the index); probe k of id
// looks at x_k = id * (A + k * B2) mod 2^m -- A odd, B2 even: every multiplier is odd, so id -> x_k is a bijection of
// [0, 2^m) for every k.  The table is W dwords = W buckets of two 16-bit entries (any W): bucket = floor(x_k * W / 2^m)
// (multiplicative hashing onto [0, W)), so the x_k of one bucket are a contiguous range of at most ceil(2^m / W) <= 2^tb
// values, and an entry stores (k << tb) | (x_k & (2^tb - 1)): within one bucket the low tb bits tell the x_k apart.
// Bucket and entry together give k and x_k, hence the id: two different ids never look alike, whatever the probe they
// were placed by and whichever half of the bucket they sit in.  0xFFFF is "empty" (k stays below 2^(16 - tb) - 1, so no
// entry is all ones).  An insert takes the first empty half (low, then high) of the first probed bucket that has one;
// entries are never removed, so a bucket's high half is never occupied while its low half is empty, and a lookup may
// stop at the first bucket with an empty half.  (Round 5: a probe used to look at ONE 16-bit slot; with two per probe a
// wavefront's inserts of a hop need 1.7 trips through the probe loop instead of 2.9 at the tables' load.)  An id that
// finds kmax full buckets is "exhausted": the caller freezes the table and sends it to the spill table in global
// memory (lookups keep probing kmax buckets first).  LDS has no 16-bit compare-and-swap: an insert swaps the whole
// dword and retries on the same bucket when the other half changed meanwhile.
constexpr uint32_t kHt16A = 0x9E3779B1u, kHt16B2 = 0x3C6EF372u;
enum : int { kHt16Present = 0, kHt16Inserted = 1, kHt16Exhausted = 2 };
struct Ht16 {
    uint32_t shift, buckets, tb, kmax;  // shift = 32 - m: x << shift is x_k as a 32-bit fraction of 2^m
};
// (SearchArgs::ht_prime holds the number of 16-bit entries = 2 * buckets)
__device__ __forceinline__ Ht16 ht16_of(const SearchArgs& a) { return Ht16{a.ht_shift, a.ht_prime >> 1, a.ht_tb, a.ht_kmax}; }
__device__ __forceinline__ uint32_t ht16_bucket(const Ht16& t, uint32_t x) { return __umulhi(x << t.shift, t.buckets); }
__device__ __forceinline__ uint32_t ht16_tag(const Ht16& t, uint32_t x, uint32_t tagmask) {
    return ((x << t.shift) >> t.shift) & tagmask;
}
__device__ __forceinline__ int ht16_insert_open(uint32_t* htw, const Ht16& t, uint32_t id, bool active) {
    const uint32_t tagmask = (1u << t.tb) - 1u;
    uint32_t x = id * kHt16A;
    const uint32_t step = id * kHt16B2;
    uint32_t k = 0;
    int res = kHt16Present;
    bool pending = active;
    uint32_t* wp = htw + ht16_bucket(t, x);
    uint32_t val = ht16_tag(t, x, tagmask);
    uint32_t w = *wp;  // (inactive lanes read some bucket of the table too: no branch around the load)
    while (ballot64(pending)) {
        if (pending) {
            const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
            if (lo == val || hi == val) {
                pending = false;  // present
            } else if (lo == 0xFFFFu || hi == 0xFFFFu) {
                const uint32_t neww = lo == 0xFFFFu ? (w ^ (0xFFFFu ^ val)) : (w ^ ((0xFFFFu ^ val) << 16));
                const uint32_t old = atomicCAS(wp, w, neww);
                if (old == w) {
                    res = kHt16Inserted;
                    pending = false;
                } else {
                    w = old;  // the other half changed meanwhile (another lane of this hop): the same bucket again
                }
            } else {  // both halves hold other ids: next probe
                ++k;
                x += step;
                if (k >= t.kmax) {
                    res = kHt16Exhausted;
                    pending = false;
                } else {
                    wp = htw + ht16_bucket(t, x);
                    val = ht16_tag(t, x, tagmask) | (k << t.tb);
                    w = *wp;
                }
            }
        }
    }
    return res;
}
// lookup in the frozen table: true when `id` is present
__device__ __forceinline__ bool ht16_contains(const uint32_t* htw, const Ht16& t, uint32_t id) {
    const uint32_t tagmask = (1u << t.tb) - 1u;
    uint32_t x = id * kHt16A;
    const uint32_t step = id * kHt16B2;
    for (uint32_t k = 0; k < t.kmax; ++k, x += step) {
        const uint32_t w = htw[ht16_bucket(t, x)], val = ht16_tag(t, x, tagmask) | (k << t.tb);
        const uint32_t lo = w & 0xFFFFu, hi = w >> 16;
        if (lo == val || hi == val) return true;
        if (lo == 0xFFFFu || hi == 0xFFFFu) return false;
    }
    return false;
}

// Spill tables are handed from wave to wave inside a launch, possibly across XCDs (private L2s):
// every probe is an agent-scope atomic, and the table is wiped with write-through (sc1) 16-byte
// stores drained by s_waitcnt before the busy flag is released -- no release/acquire fences, which
// cost microseconds each at this occupancy (MI355X_MICROARCH.md, inter-workgroup visibility).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void spill_wipe(uint32_t* table, uint32_t entries, uint32_t lane) {
    const u32x4 e = {kEmpty, kEmpty, kEmpty, kEmpty};
    for (uint32_t i = lane * 4u; i < entries; i += kWave * 4u) {
        uint32_t* p = table + i;
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(e) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ bool spill_insert(uint32_t* gt, uint32_t mask, uint32_t shift, uint32_t id) {
    uint32_t h = (id * 2246822519u) >> shift;
    for (;;) {
        uint32_t old = atomicCAS(&gt[h], kEmpty, id);
        if (old == kEmpty) return true;
        if (old == id) return false;
        h = (h + 1) & mask;
    }
}

// rust_order::sort_keys_unstable as a real call: inlined into beam_search_kernel it takes hipcc 7.2 down ("SI Fix SGPR
// copies" / instruction selection crash), and the filtered kernels would carry its code in every instantiation
__device__ __attribute__((noinline)) void rust_sort_keys_call(unsigned long long* keys, uint32_t n, void* work) {
    rust_order::sort_keys_unstable(keys, n, work);
}
// total order on non-NaN f32 as unsigned bits (NaN sorts last)
__device__ __forceinline__ uint32_t ordered_bits(float d) {
    const uint32_t u = __builtin_bit_cast(uint32_t, d);
    return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(uint32_t o) {
    return __builtin_bit_cast(float, (o >> 31) ? (o & 0x7FFFFFFFu) : ~o);
}
// ascending bitonic sort of n (power of two) 64-bit keys in global memory by one wave.  Keys written
// by one lane are read by another in the next pass: agent-scope accesses (served by L2) + a drain.
__device__ void wave_sort_keys(unsigned long long* keys, uint32_t n, uint32_t lane) {
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = lane; i < n; i += kWave) {
                const uint32_t p = i ^ j;
                if (p > i) {
                    const unsigned long long x = __hip_atomic_load(keys + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long y = __hip_atomic_load(keys + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) {
                        __hip_atomic_store(keys + i, y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(keys + p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
}
__device__ __forceinline__ unsigned long long key_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void key_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t u32_load(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float f32_load(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT));
}

// MODE selects what is compiled in (every switch below is wave-uniform at run time; as a run-time switch each one
// costs scalar compares, branches and live SGPRs in every hop -- the hop is issue-bound, §3.2 of DESIGN.md):
//   kModePlain    beam_width == 1, no inline tags, max_degree <= 64, no filter: the Knn / Range / record search of
//                 a published store (every benchmark configuration).  Same-box A/B against kModeGeneral on the
//                 same launch: single query L = 64 235 -> 217 us, 1024 concurrent queries +8 %.
//   kModeGeneral  any beam width, inline tags, any degree; no filter
//   kModeFiltered the filtered searches (inline / multihop / AdaptiveL); generic-length instantiations only
// minimum of a non-NaN value over the wave: DPP row shifts inside the rows of 16, then the two row broadcasts
// (lane 63 ends up with the minimum of all lanes)
__device__ __forceinline__ float wave_min_f32(float v) {
    constexpr int kInf = 0x7F800000;
#define DANN_MIN_STEP(CTRL, ROWMASK)                                                                                  \
    {                                                                                                                  \
        const float t = __builtin_bit_cast(                                                                            \
            float, __builtin_amdgcn_update_dpp(kInf, __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));          \
        v = t < v ? t : v;                                                                                             \
    }
    DANN_MIN_STEP(0x111, 0xf)  // row_shr:1
    DANN_MIN_STEP(0x112, 0xf)  // row_shr:2
    DANN_MIN_STEP(0x114, 0xf)  // row_shr:4
    DANN_MIN_STEP(0x118, 0xf)  // row_shr:8   -> lane 15 of every row: the row's minimum
    DANN_MIN_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    DANN_MIN_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef DANN_MIN_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr uint32_t kTeamExit = 0xFFFFFFFFu;  // release word of a team (kMbGo)
// Which physical wavefront of a team's workgroup is the queue wave (the others take the jobs 1, 2, ... in order).  Five
// wavefronts sit on four SIMDs: the first and the last share one.
#ifndef DANN_TEAM_QUEUE_WAVE
#define DANN_TEAM_QUEUE_WAVE 0
#endif
constexpr int kTeamQueueWave = DANN_TEAM_QUEUE_WAVE;
// the team's mailbox (SearchLds mail_off, 64 words).  Hops are numbered from 0 (the start points); pops from 1 (pop n
// returns the node hop n expands).  Words that one wave rewrites while another may still read the previous value are
// double-buffered by the parity of the hop / pop they belong to.
enum : int {
    kMbGo = 0,        // control (hop 0: queue) wave -> gather & visited waves: (hop + 1) | candidates << 20 | buffer << 27
                      // of the hop to work on, or kTeamExit
    kMbVReply = 1,    // visited wave -> control wave: ran | kept << 8 | inserted << 16
    kMbLoaded = 2,    // queue wave -> control wave: hop + 1 of the last hop whose distances it holds in registers
    kMbDStatus = 3,   // control wave -> queue wave, at the end: its status ...
    kMbDCmps = 4,     //   ... and the candidates it had evaluated
    kMbHtCount0 = 5,  // queue wave -> control wave: ids in the visited table after the start points
    kMbSpecSeq = 6,   // control (hop 0: queue) wave -> visited wave: hop + 1 of the hop whose kMbHop words are complete
    kMbHop = 8,       // + 8 * (hop & 1):  +0 candidates | buffer << 16   +1 node for the visited wave (kEmpty: none)
                      //                   +2 ids in the visited table    +3 buffer for the visited wave's candidates
    kMbPop = 24,      // + 8 * (pop & 1):  +0 pop number (written last)  +1 found  +2 node  +3 best unexpanded entry left
                      //                   +4 its distance  +5 the one after it  +6 its distance
};
// (every mailbox word is the same for all lanes of the reader: handing it over as a scalar keeps the branches on it
// scalar branches instead of nests of exec-masked regions)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t mb_load(const uint32_t* p) {
    return uni(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ uint32_t lds_poll_lane(const uint32_t* p) {  // (a per-lane address)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// publishes what this wave wrote to LDS before: one wave's LDS instructions execute in issue order, so all it takes is
// that the compiler keeps them in program order (a release fence would also drain the global-memory counter, and the
// prefetch loads in flight with it)
__device__ __forceinline__ void mb_store(uint32_t* p, uint32_t v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// A mailbox wait that outlasts about a second is a protocol bug or a wave that was held up for that long (preemption
// under oversubscription, a debugger): the team gives the query back -- status DANN_EINTERNAL through the words the
// control wave's regular exit uses, release word set, this wave ends -- and the host re-runs it with one wave per query
// (search_with_retry), instead of a trap that would take the host process down.  The queue wave, parked at the hop's
// barrier, is released when the last helper has ended the same way or has seen the release word.
__device__ __forceinline__ void team_abort(uint32_t* mail) {
    mail[kMbDStatus] = (uint32_t)(-DANN_EINTERNAL);
    mail[kMbDCmps] = 0u;
    asm volatile("" ::: "memory");
    __hip_atomic_store(mail + kMbGo, kTeamExit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_endpgm();
}
// wait until *p != seen (another wave of the team writes it); returns the new value
__device__ __forceinline__ uint32_t mb_wait_change(uint32_t* mail, const uint32_t* p, uint32_t seen) {
    uint32_t v, spins = 0;
    while ((v = mb_load(p)) == seen) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) team_abort(mail);
    }
    asm volatile("" ::: "memory");
    return v;
}
__device__ __forceinline__ void mb_wait_at_least(uint32_t* mail, const uint32_t* p, uint32_t want) {
    uint32_t spins = 0;
    while (mb_load(p) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) team_abort(mail);
    }
    asm volatile("" ::: "memory");
}
constexpr uint32_t kAdjPending = 0xFFFFFFFEu;  // "not landed yet" (never an id: ids stay below 2^31; never a length)
// a gather wave's share of a team gather: NW gather waves split every block of NW * GROUPS * U candidates, U rows per
// lane group in flight (a team has one gather wave, U = 4: the whole 32-neighbour hop in one pass)
template <int DT, int OP, bool NORM, int DIM, int NW, int U>
__device__ __forceinline__ void team_gather_share(const IndexView& ix, uint32_t wi, uint32_t nc, const uint32_t* cand_id,
                                                  float* cand_d, const F4 (&xq)[(DIM > 0 && !Scheme<DT, OP, false>::kInt) ? DIM / (4 * Scheme<DT, OP, false>::G) : 1],
                                                  const uint4& xqi, int xx_pre, const uint8_t* qs, const SqParams& sqp, int g,
                                                  int v) {
    using S = Scheme<DT, OP, false>;
    constexpr int G = S::G, GROUPS = kWave / G;
    using RT = typename RowType<DT>::type;
    for (uint32_t c0 = 0; c0 < nc; c0 += NW * GROUPS * U) {
        uint32_t c[U];
        bool act[U];
        const uint8_t* rowb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            c[u] = c0 + ((uint32_t)u * NW + wi) * GROUPS + (uint32_t)g;
            act[u] = c[u] < nc;
            const uint32_t id = act[u] ? cand_id[c[u]] : 0u;
            rowb[u] = ix.rows + (uint64_t)id * ix.row_stride;
        }
        float out[U];
        if constexpr (!S::kInt) {
            const RT* rows[U];
#pragma unroll
            for (int u = 0; u < U; ++u) rows[u] = reinterpret_cast<const RT*>(rowb[u]);
            group_distance_pre<S::NACC, OP, DIM, U>(xq, rows, act, v, out);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (act[u] && v == 0) cand_d[c[u]] = post_op<OP, NORM>(out[u]);
        } else {
            const uint8_t* rows[U];
#pragma unroll
            for (int u = 0; u < U; ++u) rows[u] = rowb[u];
            group_distance_int_pre<OP, DT == DT_I8, U>(xqi, xx_pre, rows, v, out);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (act[u] && v == 0) cand_d[c[u]] = finish_distance<DT, OP, NORM>(out[u], qs, rows[u], ix.dim, sqp);
        }
    }
}

// synchronisation of one wave with itself (LDS written by some lanes, read by others) where other waves share the
// workgroup: a counter drain instead of the workgroup barrier
__device__ __forceinline__ void team_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// argmin over the lanes of `on` by (distance ascending, lane descending); `on` must not be empty
__device__ __forceinline__ int team_best_lane(bool on, float d) {
    const float m = wave_min_f32(on ? d : __builtin_inff());
    return 63 - __builtin_clzll(ballot64(on && d == m));
}

// ---- a team of five wavefronts per query (latency regime) -----------------------------------------------------------
// The dependent chain of a hop is  pop -> adjacency row -> visited filter -> candidate rows -> merge -> pop.  A team
// gives every link its own wave and overlaps them:
//   wave 0, queue:    merge of hop h's distances, pop, publication of (node, best and second-best unexpanded entry left)
//                     -- the code of the one-wave search (beam_search_one), it also stages the query and writes results;
//   wave 1, control:  when hop h's distances are ready, decides *before the merge* which node hop h + 1 expands (the best
//                     unexpanded entry unless a new candidate is at most as far: lower-bound insert, queue.rs:150-170)
//                     and which node will be best after that; expands where no speculation covers it; starts the hop;
//   wave 2, visited:  while hop h + 1's rows are evaluated, runs the visited filter of the predicted hop h + 2 node into
//                     the spare candidate buffer (taken back by the control wave if the prediction fails);
//   waves 3 and 4, gather: evaluate the candidate rows, half of the hop's candidates each (one wave alone spent half of
//                     its time issuing the vector instructions of ~17 distances from a single SIMD: round 6).
// One workgroup barrier per hop ("distances ready"); everything else goes through the LDS mailbox (kMb*).  The queue, the
// visited set and the order of expansions are exactly those of the one-wave search: the early decisions only predict
// what the pop after the merge returns, and the control wave checks every one of them against the queue wave's
// publication (a mismatch is DANN_EINTERNAL).  The visited table never spills here: a query that would need it reports
// DANN_EOVERFLOW and the host re-runs it with one wave.
template <int DT, int OP, bool NORM, int QS, int DIM>
__device__ __forceinline__ void team_control_wave(const SearchArgs& a, uint8_t* smem, const SearchLds& L, const uint32_t lane) {
    const IndexView& ix = a.ix;
    uint32_t* const mail = reinterpret_cast<uint32_t*>(smem + L.mail_off);
    uint32_t* const ht = reinterpret_cast<uint32_t*>(smem + L.ht_off);
    const uint32_t cstride = L.cand2_id_off - L.cand_id_off;
    const uint32_t ht_mod = a.ht_prime, R = ix.max_degree;
    const uint32_t ht_limit = a.ht_open;
    const bool latency = (a.tune & kTuneRowPrefetch) != 0;
    uint32_t pf_dummy = 0;  // landing register of prefetch loads nothing reads
    auto buf_ids = [&](uint32_t b) { return reinterpret_cast<uint32_t*>(smem + L.cand_id_off + b * cstride); };
    auto buf_d = [&](uint32_t b) { return reinterpret_cast<const float*>(smem + L.cand_d_off + b * cstride); };
    // an adjacency row requested ahead of its use lands in LDS (buffer 0: the node named to the visited wave, buffer 1:
    // a new candidate this wave expands itself) through global_load_lds_dword: lane l's dword goes to M0 + 4 l.  No
    // register is written, so nothing waits for the load before its consumer does (adj_landed here, polling in the
    // visited wave -- hence kAdjPending in buffer 0 first).  One instruction covers the length and max_degree <= 63
    // neighbours (teams require it).
    auto adj_fetch_lds = [&](uint32_t node, uint32_t which) {
        const uint32_t* p = ix.adj + (uint64_t)node * ix.adj_stride + (lane <= R ? lane : R);
        const uint32_t lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(
            __attribute__((address_space(3))) uint8_t*)(smem + L.adj_off + which * kAdjLandBytes));
        // (an earlier request into the same buffer must have landed: its tail would overwrite the marks)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (which == 0u) reinterpret_cast<uint32_t*>(smem + L.adj_off)[lane] = kAdjPending;
        uint32_t m0_saved;
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %2\n\t"
            "global_load_lds_dword %1, off\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(m0_saved)
            : "v"(p), "s"(lds)
            : "memory");
    };
    auto adj_landed = [&](uint32_t which) -> const uint32_t* {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return reinterpret_cast<const uint32_t*>(smem + L.adj_off + which * kAdjLandBytes);
    };
    // argmin over the lanes of `on` by (distance ascending, lane descending); `on` must not be empty
    auto best_lane = [&](bool on, float d) -> int { return team_best_lane(on, d); };

    uint32_t hop = 0;                // the hop in flight / last finished
    uint32_t cur = 0, nc_cur = 0;    // its candidate buffer and count
    uint32_t ht_count = 0, status = 0, cmps = 0;
    uint32_t pf_node = kEmpty;       // node whose adjacency row is in (or on its way to) landing buffer 0
    uint32_t spec_sent = kEmpty, spec_node = kEmpty, spec_nc = 0, spec_new = 0;
    uint32_t early_node = kEmpty;    // the node the hop in flight expands (the queue wave's pop must agree)
    // pf_node's row was requested by the visited wave (self-start): this wave's memory counter does not cover it, so it
    // never reads landing buffer 0 for that node (the row comes from memory in the rare case this wave expands it)
    bool pf_by_visited = false;
    const bool self_start = !(a.tune & kTuneNoSelfStart);

    // start hop `hop + 1` on `nc` candidates in buffer `buf`; the visited wave works on pf_node meanwhile
#ifdef DANN_PHASE_CYCLES
    unsigned long long ph_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // start hop `hop + 1` on `nc` candidates in buffer `buf`: the gather wave first (team_go) -- the hop's critical
    // path --, then the visited wave with the node to work on meanwhile, pf_node (team_spec)
    auto team_go = [&](uint32_t nc, uint32_t buf) {
        ++hop;
        if (lane == 0) mb_store(mail + kMbGo, ((hop + 1u) & 0xFFFFFu) | (nc << 20) | (buf << 27));
        cur = buf;
        nc_cur = nc;
    };
    auto team_spec = [&]() {
        PH_T(pts0);
        // (the visited wave refills the previous hop's buffer once the queue wave holds those distances in registers: it
        // waits for kMbLoaded itself, beside its wait for the adjacency row)
        spec_sent = (pf_node != kEmpty && !(a.tune & kTuneNoSpeculation)) ? pf_node : kEmpty;
        if (lane == 0) {
            uint32_t* w = mail + kMbHop + 8u * (hop & 1u);
            *reinterpret_cast<uint4*>(w) = make_uint4(nc_cur | (cur << 16), spec_sent, ht_count, cur ^ 1u);
            mb_store(mail + kMbSpecSeq, hop + 1u);
        }
        PH_T(pts1);
        PH_ADD(2, pts0, pts1);
    };
    auto spec_rollback = [&]() {
        const uint32_t slot = reinterpret_cast<const uint32_t*>(smem + L.slots_off)[lane];
        if (slot != kEmpty) ht[slot] = kEmpty;
        spec_node = kEmpty;
        team_wave_sync();
    };
    // neighbours of `node` through the visited filter into buffer `buf` (provider.rs:448-454); the row comes from landing
    // buffer `landed` (0 / 1) or, landed < 0, straight from memory
    uint32_t touch_id = kEmpty;  // (per lane) a candidate the last expansion found
    auto expand = [&](uint32_t node, int landed, uint32_t buf) -> uint32_t {
        uint32_t len, id;
        const uint32_t col = lane < R ? lane : R - 1u;
        if (landed >= 0) {
            const uint32_t* row = adj_landed((uint32_t)landed);
            len = uni(row[0]);
            id = row[1u + col];
        } else {
            const uint32_t* row = ix.adj + (uint64_t)node * ix.adj_stride;
            len = uni(row[0]);
            id = row[1u + col];
        }
        len = len < R ? len : R;  // Neighbors::get clamps (neighbors.rs:146-148)
        if (ht_count + len > ht_limit) {
            status = (uint32_t)(-DANN_EOVERFLOW);
            return 0;
        }
        id = lane < len ? id : kEmpty;
        const bool isnew = ht_insert_open(ht, ht_mod, id, id != kEmpty);
        const bool keep = isnew && id < ix.nslots;
        const uint64_t nm = ballot64(isnew), km = ballot64(keep);
        if (keep) buf_ids(buf)[mbcnt(km)] = id;
        touch_id = keep ? id : kEmpty;
        ht_count += (uint32_t)__popcll(nm);
        return (uint32_t)__popcll(km);
    };
    // latency regime: should one of the candidates an expansion has just found be expanded straight away, its adjacency
    // row is in L2 by then.  (Issued after the runner-up's row request, which waits for the loads before it.)
    auto touch_found = [&]() {
        if (latency && touch_id != kEmpty) {
            const uint32_t* a0 = ix.adj + (uint64_t)touch_id * ix.adj_stride;
            const uint32_t* a1 = a0 + R;
            asm volatile(
                "global_load_dword %0, %1, off\n\t"
                "global_load_dword %0, %2, off"
                : "+v"(pf_dummy)
                : "v"(a0), "v"(a1));
        }
        touch_id = kEmpty;
    };
    struct Pub {
        uint32_t found, node, pf_next, pf_next2;
        float pf_next_d, pf_next2_d;
    };
    auto read_pub = [&](uint32_t n) -> Pub {  // publication of pop n (waits for it)
        const uint32_t* w = mail + kMbPop + 8u * (n & 1u);
        mb_wait_at_least(mail, w, n);
        const uint4 lo = *reinterpret_cast<const uint4*>(w), hi = *reinterpret_cast<const uint4*>(w + 4);
        Pub r;
        r.found = uni(lo.y);
        r.node = uni(lo.z);
        r.pf_next = uni(lo.w);
        r.pf_next_d = __builtin_bit_cast(float, uni(hi.x));
        r.pf_next2 = uni(hi.y);
        r.pf_next2_d = __builtin_bit_cast(float, uni(hi.z));
        return r;
    };

    __syncthreads();  // the query is staged, the table wiped, the mailbox cleared (queue wave)
    __syncthreads();  // hop 0 (the start points): distances ready -- the queue wave merges them and pops the first node
    for (;;) {
        bool started = false;
        bool self_started = false;  // this hop's runner-up is requested by the visited wave, not here
        bool overtaken = false;  // (phase statistics only)
        (void)overtaken;
        PH_T(pd0);
        if (hop > 0) {
            // ---- hop `hop`'s distances are ready
            // (everything this decision reads was written before the barrier: one batch of LDS loads, one wait)
            const uint32_t* wp = mail + kMbPop + 8u * (hop & 1u);
            const uint4 plo = *reinterpret_cast<const uint4*>(wp), phi = *reinterpret_cast<const uint4*>(wp + 4);
            const uint32_t r_v = mail[kMbVReply];
            const bool has = lane < nc_cur;
            const float nd = has ? buf_d(cur)[lane] : 0.0f;
            const uint32_t nid = has ? buf_ids(cur)[lane] : kEmpty;
            Pub pub;
            pub.found = uni(plo.y);
            pub.node = uni(plo.z);
            pub.pf_next = uni(plo.w);
            pub.pf_next_d = __builtin_bit_cast(float, uni(phi.x));
            pub.pf_next2 = uni(phi.y);
            pub.pf_next2_d = __builtin_bit_cast(float, uni(phi.z));
            const uint32_t r = uni(r_v);
            spec_node = kEmpty;
            if (spec_sent != kEmpty && (r & 1u)) {
                spec_node = spec_sent;
                spec_nc = (r >> 8) & 0xFFu;
                spec_new = (r >> 16) & 0xFFu;
            }
            if (uni(plo.x) != hop || !pub.found || pub.node != early_node) status = (uint32_t)(-DANN_EINTERNAL);
            if (status) break;
            cmps += nc_cur;
            const uint32_t pf_next = pub.pf_next, pf_next2 = pub.pf_next2;
#ifdef DANN_PHASE_CYCLES
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PH_T(pdl);
            ph_acc[1] += pdl - pd0;
#endif
            if (pf_next != kEmpty) {
                const bool ahead = has && nd <= pub.pf_next_d;  // (a NaN distance never enters the queue)
                uint32_t next, runner, nc;
                if (ballot64(ahead) == 0) {
                    // the next hop expands pf_next; the runner-up after that pop: pf_next2 unless a new candidate is
                    // at most as far
                    next = pf_next;
                    if (spec_node == pf_next) {  // ... and the visited wave has its candidates ready: go
                        ht_count += spec_new;
                        nc = spec_nc;
                        spec_node = kEmpty;
                        team_go(nc, cur ^ 1u);
                        started = true;
                        // the visited wave has taken this very decision (team_visited_wave, self-start): it names the
                        // runner-up computed below and requests its adjacency row itself
                        self_started = self_start && hop >= 2u;  // (hop: the one just started)
                    } else {
                        overtaken = true;  // (statistics: not the short path)
                        if (spec_node != kEmpty) spec_rollback();  // (a runner-up the pop contradicted)
                        nc = expand(next, (pf_node == next && !pf_by_visited) ? 0 : -1, cur ^ 1u);
                        if (status) break;
                    }
                    runner = pf_next2;
                    if (pf_next2 != kEmpty) {
                        const bool ahead2 = has && nd <= pub.pf_next2_d;
                        if (ballot64(ahead2)) runner = (uint32_t)__builtin_amdgcn_readlane((int)nid, best_lane(ahead2, nd));
                    }
                } else {
                    // the next hop expands the closest new candidate (of equal ones the one inserted last): its
                    // adjacency row was touched when the candidate was found and comes from L2; the runner-up is
                    // worked out while it travels
                    overtaken = true;
                    if (spec_node != kEmpty) spec_rollback();
                    const int bj = best_lane(ahead, nd);
                    next = (uint32_t)__builtin_amdgcn_readlane((int)nid, bj);
                    adj_fetch_lds(next, 1);
                    const bool rest = ahead && (int)lane != bj;
                    runner = pf_next;  // stays the runner-up unless a second new candidate is ahead of it too
                    if (ballot64(rest)) runner = (uint32_t)__builtin_amdgcn_readlane((int)nid, best_lane(rest, nd));
                    nc = expand(next, 1, cur ^ 1u);
                    if (status) break;
                }
                early_node = next;
                // the hop's critical path first: the gather waves ("go", as soon as the candidates are there), then what
                // the visited wave needs -- the runner-up's adjacency row on its way, its words --, prefetches last
                if (!started) team_go(nc, cur ^ 1u);
                if (self_started) {
                    pf_node = runner;  // (never the node just expanded: the visited wave requests its row)
                    pf_by_visited = true;
                } else if (pf_node != runner) {  // (else its row is in landing buffer 0 already)
                    pf_node = runner;
                    pf_by_visited = false;
                    if (pf_node != kEmpty) adj_fetch_lds(pf_node, 0);
                }
                team_spec();
                touch_found();
                started = true;
            }
        }
        if (!started) {
            // no unexpanded entry is known to be left (or this is the first hop): the queue wave's merge and pop decide
            if (spec_node != kEmpty) spec_rollback();
            const Pub pub = read_pub(hop + 1u);
            if (hop == 0) ht_count = mb_load(mail + kMbHtCount0);
            if (!pub.found) break;  // the search is over
            const uint32_t nc = expand(pub.node, (pf_node == pub.node && !pf_by_visited) ? 0 : -1, cur ^ 1u);
            if (status) break;
            early_node = pub.node;
            team_go(nc, cur ^ 1u);
            if (pf_node != pub.pf_next) {
                pf_node = pub.pf_next;
                pf_by_visited = false;
                if (pf_node != kEmpty) adj_fetch_lds(pf_node, 0);
            }
            team_spec();
            touch_found();
#ifdef DANN_PHASE_CYCLES
            overtaken = true;
            ph_acc[7] += 1;
#endif
        }
        PH_T(pd1);
#ifdef DANN_PHASE_CYCLES
        ph_acc[overtaken ? 13 : 12] += pd1 - pd0;
        ph_acc[overtaken ? 6 : 5] += 1;
#endif
        __syncthreads();  // "distances ready" of the hop just started
        PH_T(pd2);
        PH_ADD(15, pd1, pd2);
    }
#ifdef DANN_PHASE_CYCLES
    if (lane == 0)
        for (int i = 0; i < 16; ++i)
            if (ph_acc[i]) atomicAdd(&a.phase_cycles[i], ph_acc[i]);
#endif
    if (lane == 0) {
        mail[kMbDStatus] = status;
        mail[kMbDCmps] = cmps;
        mb_store(mail + kMbGo, kTeamExit);
    }
    __syncthreads();  // the release: every wave of the team meets here once more
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(pf_dummy));
}

// the visited wave
__device__ __forceinline__ void team_visited_wave(const SearchArgs& a, uint8_t* smem, const SearchLds& L, const uint32_t lane) {
    const IndexView& ix = a.ix;
    uint32_t* const mail = reinterpret_cast<uint32_t*>(smem + L.mail_off);
    uint32_t* const ht = reinterpret_cast<uint32_t*>(smem + L.ht_off);
    const uint32_t* const landing = reinterpret_cast<const uint32_t*>(smem + L.adj_off);
    uint32_t* const slots = reinterpret_cast<uint32_t*>(smem + L.slots_off);
    const uint32_t cstride = L.cand2_id_off - L.cand_id_off;
    const uint32_t ht_mod = a.ht_prime, R = ix.max_degree;
    const bool touch = (a.tune & kTuneRowPrefetch) && ix.layer_bytes <= 512u;
    uint32_t pf_dummy = 0;
    __syncthreads();  // the query is staged, the table wiped, the mailbox cleared
    // what this wave did in the hop before (the control wave's spec_sent and the reply it reads): node, ran, ids inserted
    uint32_t prev_node = kEmpty, prev_ran = 0, prev_fresh = 0;
    bool released = false;  // the barrier just passed was the release
    const bool self_start = !(a.tune & kTuneNoSelfStart);
    for (uint32_t hop = 0;; ++hop) {
        uint32_t node = kEmpty, table_ids = 0, out_buf = 0;
        // ---- self-start.  On a prepared hop -- the control wave finds that the node to expand next is the one this wave
        // has filtered already -- the longest chain of the hop used to be: control wave (decision, "go", runner-up,
        // adjacency request, words) -> this wave (row lands, filter).  The decision is a function of what lies in LDS when
        // the barrier opens (the pop's publication, the hop's distances) and of this wave's own last reply, so this wave
        // takes the same decision at the same time and, on exactly those hops, names the runner-up and requests its
        // adjacency row itself, without waiting for the words (4 % of a query: DESIGN 3.5).  The control wave, on those hops,
        // records the same runner-up without requesting anything (team_control_wave: `self_started`).
        bool self = false;
        if (self_start && hop >= 2u && prev_ran) {
            const uint32_t h = hop - 1u;  // the hop whose distances are ready
            const uint32_t* wp = mail + kMbPop + 8u * (h & 1u);
            const uint4 plo = *reinterpret_cast<const uint4*>(wp), phi = *reinterpret_cast<const uint4*>(wp + 4);
            const uint4 hw = *reinterpret_cast<const uint4*>(mail + kMbHop + 8u * (h & 1u));
            const float nd0 = reinterpret_cast<const float*>(smem + L.cand_d_off)[lane];
            const float nd1 = reinterpret_cast<const float*>(smem + L.cand_d_off + cstride)[lane];
            const uint32_t ni0 = reinterpret_cast<const uint32_t*>(smem + L.cand_id_off)[lane];
            const uint32_t ni1 = reinterpret_cast<const uint32_t*>(smem + L.cand_id_off + cstride)[lane];
            const uint32_t word = uni(hw.x), nc_h = word & 0xFFFFu, buf_h = (word >> 16) & 1u;
            const uint32_t pf_next = uni(plo.w), pf_next2 = uni(phi.y);
            const float pf_next_d = __builtin_bit_cast(float, uni(phi.x)), pf_next2_d = __builtin_bit_cast(float, uni(phi.z));
            const bool has = lane < nc_h;
            const float nd = has ? (buf_h ? nd1 : nd0) : 0.0f;
            const uint32_t nid = has ? (buf_h ? ni1 : ni0) : kEmpty;
            if (pf_next != kEmpty && prev_node == pf_next && ballot64(has && nd <= pf_next_d) == 0) {
                self = true;
                uint32_t runner = pf_next2;
                if (pf_next2 != kEmpty) {
                    const bool ahead2 = has && nd <= pf_next2_d;
                    if (ballot64(ahead2)) runner = (uint32_t)__builtin_amdgcn_readlane((int)nid, team_best_lane(ahead2, nd));
                }
                node = runner;
                table_ids = uni(hw.z) + prev_fresh;  // the control wave's count after this hop's "go"
                out_buf = buf_h;                     // the buffer of the hop just finished is the spare one now
                if (node != kEmpty) {
                    // (landing buffer 0 is free: the request before this one -- whoever made it -- has landed, this wave
                    // polled for it before it replied `ran`)
                    const uint32_t* p = ix.adj + (uint64_t)node * ix.adj_stride + (lane <= R ? lane : R);
                    const uint32_t lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(
                        __attribute__((address_space(3))) uint8_t*)(smem + L.adj_off));
                    reinterpret_cast<uint32_t*>(smem + L.adj_off)[lane] = kAdjPending;
                    uint32_t m0_saved;
                    asm volatile(
                        "s_waitcnt lgkmcnt(0)\n\t"
                        "s_mov_b32 %0, m0\n\t"
                        "s_mov_b32 m0, %2\n\t"
                        "global_load_lds_dword %1, off\n\t"
                        "s_mov_b32 m0, %0"
                        : "=&s"(m0_saved)
                        : "v"(p), "s"(lds)
                        : "memory");
                }
            }
        }
        // this hop's "go" (or the release).  A self-started hop waits for it too, before its first write to anything the
        // control wave's decision reads (the spare candidate buffer, the slots, the reply): the control wave's LDS loads
        // precede its store of "go" in its own instruction order, which is the order LDS performs them in
        uint32_t g, gspins = 0;
        while ((g = mb_load(mail + kMbGo)) != kTeamExit && (g & 0xFFFFFu) != ((hop + 1u) & 0xFFFFFu)) {
            __builtin_amdgcn_s_sleep(1);
            if (++gspins > (1u << 24)) team_abort(mail);
        }
        asm volatile("" ::: "memory");
        if (g == kTeamExit) {
            if (!self) break;  // (the release is still ahead)
            // self-started, and the control wave ended the search instead of starting the hop (an error status): the
            // barrier ahead is the release; the request, if any, must not stay in flight
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            node = kEmpty;
        }
        if (!self) {
            mb_wait_at_least(mail, mail + kMbSpecSeq, hop + 1u);  // (the gather waves are started first)
            const uint32_t* w = mail + kMbHop + 8u * (hop & 1u);
            node = uni(w[1]);
            table_ids = uni(w[2]);
            out_buf = uni(w[3]);
        }
        uint32_t ran = 0, kept = 0, fresh = 0;
        uint32_t touch_id = kEmpty;  // (per lane) a candidate whose rows are requested once the hop's barrier is behind
        if (node != kEmpty) {
            // this wave refills the previous hop's candidate buffer: the queue wave must hold those distances in registers
            // (it has, long since: one poll, beside the wait for the adjacency row)
            mb_wait_at_least(mail, mail + kMbLoaded, hop);
            // the control wave asked for the node's adjacency row (kAdjPending in every dword first): wait for the
            // length, then for every neighbour slot below it
            // (all 64 dwords of the request, also those beyond the length: a part still in flight would land on top
            // of the *next* request's kAdjPending marks and pass for its data)
            uint32_t len = kAdjPending, val = kAdjPending, spins = 0;
            for (;;) {
                const uint32_t mine = lds_poll_lane(landing + lane);
                if (ballot64(mine == kAdjPending) == 0) {
                    len = mb_load(landing);
                    val = lds_poll_lane(landing + 1u + (lane < R ? lane : R - 1u));
                    break;
                }
                if (++spins > (1u << 16)) {  // (never observed; a speculation skipped costs nothing but time)
                    len = kAdjPending;
                    // (a request of this wave's own must not be left in flight: the next one into the buffer may be the
                    // control wave's, whose counter does not see it)
                    if (self) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (len != kAdjPending) {
                len = len < R ? len : R;
                if (table_ids + len <= a.ht_open) {  // (an expansion would not overflow the table)
                    const uint32_t id = lane < len ? val : kEmpty;
                    uint32_t slot = 0;
                    const bool isnew = ht_insert_open_slot(ht, ht_mod, id, id != kEmpty, &slot);
                    const bool keep = isnew && id < ix.nslots;
                    const uint64_t nm = ballot64(isnew), km = ballot64(keep);
                    uint32_t* const out = reinterpret_cast<uint32_t*>(smem + L.cand_id_off + out_buf * cstride);
                    if (keep) out[mbcnt(km)] = id;
                    slots[lane] = isnew ? slot : kEmpty;
                    ran = 1;
                    kept = (uint32_t)__popcll(km);
                    fresh = (uint32_t)__popcll(nm);
                    if (touch && keep) touch_id = id;
                }
            }
        }
        if (lane == 0) mail[kMbVReply] = ran | (kept << 8) | (fresh << 16);
        prev_node = node;
        prev_ran = ran;
        prev_fresh = fresh;
        __syncthreads();  // distances ready -- or, after a self-started hop the control wave ended instead, the release
        if (mb_load(mail + kMbGo) == kTeamExit) {
            released = true;
            break;
        }
        // latency regime: request the rows of the candidates just prepared (one dword per 128-byte line; nothing ever reads
        // pf_dummy) and their adjacency rows, should one of them be expanded straight away -- after the barrier: this wave's
        // filter is the tail of the hop's longest chain (decision -> adjacency row -> filter), the requests are not, and the
        // gather that wants the rows starts a decision later
        if (touch_id != kEmpty) {
            const uint8_t* prow = ix.rows + (uint64_t)touch_id * ix.row_stride;
            const uint32_t last = ix.layer_bytes >= 4u ? ix.layer_bytes - 4u : 0u;  // (PQ rows of fewer than four chunks)
            const uint8_t* p1 = prow + (128u < last ? 128u : last);
            const uint8_t* p2 = prow + (256u < last ? 256u : last);
            const uint8_t* p3 = prow + (384u < last ? 384u : last);
            const uint8_t* p4 = prow + last;
            const uint32_t* a0 = ix.adj + (uint64_t)touch_id * ix.adj_stride;
            const uint32_t* a1 = a0 + R;
            asm volatile(
                "global_load_dword %0, %1, off\n\t"
                "global_load_dword %0, %2, off\n\t"
                "global_load_dword %0, %3, off\n\t"
                "global_load_dword %0, %4, off\n\t"
                "global_load_dword %0, %5, off\n\t"
                "global_load_dword %0, %6, off\n\t"
                "global_load_dword %0, %7, off"
                : "+v"(pf_dummy)
                : "v"(prow), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(a0), "v"(a1));
        }
    }
    if (!released) __syncthreads();  // the release
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(pf_dummy));
}

// waves 1 .. 3 of a team
template <int DT, int OP, bool NORM, int QS, int DIM, int TEAM>
__device__ __forceinline__ void team_helper(const SearchArgs& a, uint8_t* smem) {
    static_assert(TEAM == 4 || TEAM == 5 || TEAM == 7, "a team: queue, control, visited and 1, 2 or 4 gather wavefronts");
    constexpr int NW = TEAM - 3;  // gather wavefronts: they split the hop's candidates (team_gather_share)
    using S = Scheme<DT, OP, false>;
    constexpr int G = S::G;
    constexpr bool kInt = S::kInt;
    const IndexView& ix = a.ix;
    const uint32_t lane = threadIdx.x & 63u;
    // (the job of a physical wavefront: see kTeamQueueWave)
    const uint32_t pwave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wave = pwave < (uint32_t)kTeamQueueWave ? pwave + 1u : pwave;
    const uint32_t qbytes = kInt ? (uint32_t)DIM + (DT == DT_SQ8 ? 4u : 0u) : (uint32_t)DIM * 4u;
    const SearchLds L = search_lds_layout(a.ht_entries, (uint32_t)kWave, lds_queue_entries(a), qbytes, true);
    if (wave == 1u) {
        team_control_wave<DT, OP, NORM, QS, DIM>(a, smem, L, lane);
        return;
    }
    if (wave == 2u) {
        team_visited_wave(a, smem, L, lane);
        return;
    }
    // ---- the gather wave: four rows per lane group in flight, as the one-wave search
    const uint8_t* qs = smem + L.q_off;
    // the two candidate buffers sit one fixed stride apart; addressing them as base + buffer * stride (never through a
    // table of pointers) keeps every access an LDS instruction -- a selected pointer decays to a flat address, whose
    // loads also wait for the global-memory counter
    const uint32_t cstride = L.cand2_id_off - L.cand_id_off;
    uint32_t* const mail = reinterpret_cast<uint32_t*>(smem + L.mail_off);
    const SqParams sqp{ix.sq_k, ix.sq_shift_norm_sq};
    __syncthreads();  // the query is staged
    const int g = lane / G, v = lane % G;
    constexpr int NTQ = (!kInt) ? DIM / (4 * G) : 1;
    F4 xq[NTQ];
    uint4 xqi = {0u, 0u, 0u, 0u};
    int xx_pre = 0;
    if constexpr (!kInt) {
#pragma unroll
        for (int t = 0; t < NTQ; ++t) xq[t] = load4(reinterpret_cast<const float*>(qs) + t * 4 * G + 4 * v);
    } else {
        xqi = *reinterpret_cast<const uint4*>(qs + 16 * v);
        xx_pre = group_norm_int_pre<DT == DT_I8>(xqi);
    }
    uint32_t seen = 0;
    for (;;) {
        seen = mb_wait_change(mail, mail + kMbGo, seen);
        if (seen == kTeamExit) break;
        const uint32_t nc = (seen >> 20) & 0x7Fu, buf = (seen >> 27) & 1u;
        team_gather_share<DT, OP, NORM, DIM, NW, 4 / NW>(
            ix, wave - 3u, nc, reinterpret_cast<const uint32_t*>(smem + L.cand_id_off + buf * cstride),
            reinterpret_cast<float*>(smem + L.cand_d_off + buf * cstride), xq, xqi, xx_pre, qs, sqp, g, v);
        __syncthreads();  // distances ready
    }
    __syncthreads();  // the release
}

enum : int { kModePlain = 0, kModeGeneral = 1, kModeFiltered = 2 };
// TEAM > 1: this is the queue wave of a team (see team_control_wave)
template <int DT, int OP, bool NORM, int QS, int DIM, int MODE, int TEAM = 1, bool HT16 = false>
__device__ __forceinline__ void beam_search_one(const SearchArgs& a, const uint32_t slot, uint8_t* smem) {
    constexpr bool FILT = MODE == kModeFiltered;
    constexpr bool PLAIN = MODE == kModePlain;
    static_assert(TEAM == 1 || (PLAIN && DIM > 0), "teams serve the plain fixed-length searches");
    static_assert(TEAM == 1 || !HT16, "teams keep the 32-bit visited table (their rollback works on its slots)");
    // synchronisation of this wave with itself (LDS written by some lanes, read by others): the workgroup barrier of a
    // one-wave workgroup, a counter drain when helper waves share the workgroup
    auto WS = [&]() {
        if constexpr (TEAM > 1) team_wave_sync();
        else __syncthreads();
    };
    using S = Scheme<DT, OP, false>;
    constexpr int G = (DIM > 0) ? S::G : S::GS;  // fixed-length path: narrow groups, query slice in registers
    constexpr int GROUPS = kWave / G;
    constexpr bool kInt = S::kInt;
    using QT = typename std::conditional<kInt, uint8_t, float>::type;
    using RT = typename RowType<DT>::type;

    const IndexView& ix = a.ix;
    const uint32_t lane = TEAM > 1 ? (threadIdx.x & 63u) : threadIdx.x;
    const uint32_t qi = a.qmap ? a.qmap[slot] : slot;
    const uint32_t R = ix.max_degree;
    const uint32_t W = PLAIN ? 1u : a.beam_width;
    uint32_t qcap = a.l_value + ix.nstart;  // queue capacity == search_l (scratch.rs:199-207); AdaptiveL may grow it
    // plain mode: W = 1, R <= 64 and at most 64 start points (plain_mode) -> one 64-entry candidate block
    const uint32_t cmax = PLAIN ? (uint32_t)kWave
                                : (((W * R + 63u) & ~63u) > ((ix.nstart + 63u) & ~63u) ? ((W * R + 63u) & ~63u)
                                                                                       : ((ix.nstart + 63u) & ~63u));
    // fixed-length instantiations know the staged query's size (f32 vector, raw bytes, SQ-8: bytes + compensation)
    const uint32_t qbytes = DIM > 0 ? (kInt ? (uint32_t)DIM + (DT == DT_SQ8 ? 4u : 0u) : (uint32_t)DIM * 4u)
                                    : query_lds_bytes(ix);
    const SqParams sqp{ix.sq_k, ix.sq_shift_norm_sq};
    const SearchLds L = search_lds_layout(a.ht_entries, cmax, lds_queue_entries(a), qbytes, TEAM > 1);
    QT* qs = reinterpret_cast<QT*>(smem + L.q_off);
    uint32_t* ht = reinterpret_cast<uint32_t*>(smem + L.ht_off);
    uint32_t* cand_id = reinterpret_cast<uint32_t*>(smem + L.cand_id_off);
    float* cand_d = reinterpret_cast<float*>(smem + L.cand_d_off);
    // teams: the two candidate buffers (base + buffer * stride, never a table of pointers: see team_helper)
    const uint32_t cstride = L.cand2_id_off - L.cand_id_off;
    uint2* stage = reinterpret_cast<uint2*>(smem + L.stage_off);
    auto stage_dist = [&](uint32_t p) -> float { return __builtin_bit_cast(float, stage[p].y); };
    constexpr uint32_t QCAPP = QS * kWave;  // padded queue capacity
    uint32_t* beam = reinterpret_cast<uint32_t*>(smem + L.beam_off);
    uint32_t* mail = reinterpret_cast<uint32_t*>(smem + L.mail_off);  // (teams)
    if constexpr (TEAM > 1) mail[lane] = 0u;

    // ---- stage the query (f16 query widened to f32 once: layers/full.rs:421-423) -------
    {
        if constexpr (DT == DT_PQ) {
            // populate_chunk_distances_impl (fixed_chunk_pq_table.rs:152-192): the lookup table of this
            // query, built straight into LDS: entry (chunk, centroid) = metric(query chunk, pivot chunk)
            const float* q = reinterpret_cast<const float*>(a.queries) + (uint64_t)qi * ix.dim;
            float* lut = reinterpret_cast<float*>(qs);
            const uint32_t total = ix.pq_chunks * 256u;
            for (uint32_t t = lane; t < total; t += kWave) {
                const uint32_t chunk = t >> 8, centroid = t & 255u;
                const uint32_t s0 = ix.pq_offsets[chunk], e0 = ix.pq_offsets[chunk + 1];
                const float raw = simd_op_seq<OP == OP_L2>(q + s0, ix.pq_pivots + (uint64_t)centroid * ix.dim + s0, e0 - s0);
                lut[t] = (OP == OP_L2) ? raw : -raw;
            }
        } else {
        const uint8_t* qsrc = a.qslots ? ix.rows + (uint64_t)a.qslots[qi] * ix.row_stride
                                       : reinterpret_cast<const uint8_t*>(a.queries) + (uint64_t)qi * ix.layer_bytes;
        if constexpr (kInt) {
            for (uint32_t i = lane; i < ix.layer_bytes; i += kWave) reinterpret_cast<uint8_t*>(qs)[i] = qsrc[i];
        } else {
            const RT* src = reinterpret_cast<const RT*>(qsrc);
            for (uint32_t i = lane; i < ix.dim; i += kWave) reinterpret_cast<float*>(qs)[i] = load1(src + i);
        }
        }
    }
    const uint32_t ht_size = a.ht_entries;
    uint32_t status_early = 0;  // (HT16) an insert outside the hop found no slot: the query is re-run with a larger table
    const uint32_t ht_mod = a.ht_prime;  // probing modulus: largest prime <= ht_size (HT16: the number of 16-bit slots)
    const Ht16 h16 = ht16_of(a);
    // insert into the open table outside the hop (start points, the second phase of a range search)
    auto visit_open = [&](uint32_t id) {
        if constexpr (HT16) {
            if (ht16_insert_open(ht, h16, id, id < ix.nslots) == kHt16Exhausted) status_early = (uint32_t)(-DANN_EOVERFLOW);
        } else {
            ht_visit(ht, ht_mod, id, true);
        }
    };
    {   // wipe the visited table with 16-byte stores (ht_size is a multiple of 64, the table 16-byte aligned)
        const u32x4 e4 = {kEmpty, kEmpty, kEmpty, kEmpty};
        for (uint32_t i = lane * 4u; i < ht_size; i += kWave * 4u) *reinterpret_cast<u32x4*>(ht + i) = e4;
    }
    __syncthreads();  // (TEAM: all waves -- the helpers read the staged query after it)

    const int g = lane / G, v = lane % G;
    // query slice of this lane in registers for the fixed-length float path
    constexpr int NTQ = (DIM > 0 && !kInt) ? DIM / (4 * G) : 1;
    F4 xq[NTQ];
    if constexpr (DIM > 0 && !kInt) {
#pragma unroll
        for (int t = 0; t < NTQ; ++t) xq[t] = load4(reinterpret_cast<const float*>(qs) + t * 4 * G + 4 * v);
    }

    // 128-byte integer rows: the lane's 16 query bytes and the query's squared norm in registers
    uint4 xqi = {0u, 0u, 0u, 0u};
    int xx_pre = 0;
    if constexpr (DIM > 0 && kInt) {
        static_assert(!kInt || DIM == 0 || DIM == 128, "integer rows: only the 128-byte length is specialised");
        xqi = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(qs) + 16 * v);
        xx_pre = group_norm_int_pre<DT == DT_I8>(xqi);
    }

    // ---- queue state: entry p at lane p % 64, slot p / 64 ------------------------------
    uint32_t qid[QS];
    float qd[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        qid[s] = kEmpty;
        qd[s] = 0.0f;
    }
    uint32_t size = 0, cmps = 0, hops = 0, ht_count = 0, status = 0, nrec = 0;
    // the adjacency row requested ahead of its use: the node (the best unexpanded queue entry), its row's first dwords
    // per lane (lane 0: the length) and neighbour `lane`.  The length stays in a vector register until it is needed -- a
    // scalar copy would wait for the load on the spot.
    uint32_t pf_node = kEmpty, pf_lenv = 0, pf_val = kEmpty, node0 = kEmpty;
    auto adj_fetch = [&](uint32_t node, uint32_t& lenv, uint32_t& val) {
        const uint32_t* prow = ix.adj + (uint64_t)node * ix.adj_stride;
        lenv = prow[lane <= R ? lane : R];  // (max_degree >= 1; the same cache lines as the neighbours)
        val = prow[1u + (lane < R ? lane : R - 1u)];
    };
    auto adj_len = [&](uint32_t lenv) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)lenv, 0); };
    uint32_t pf_dummy = 0;  // landing register of the row-prefetch loads (latency mode)
    bool lds_open = true;
#ifdef DANN_PHASE_CYCLES
    unsigned long long ph_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    uint32_t* spill = nullptr;
    uint32_t spill_count = 0;
    const uint32_t spill_size = 1u << a.spill_bits, spill_mask = spill_size - 1u, spill_shift = 32u - a.spill_bits;

    // filtered searches: QueryLabelProvider::is_match == one bit per slot (graph/ext/labeled.rs:44-68)
    const uint32_t fmode = FILT ? a.filter_mode : 0u;
    const uint32_t* fbits = a.filter ? a.filter + (uint64_t)qi * a.filter_stride : nullptr;
    auto fmatch = [&](uint32_t id) -> bool { return id < ix.nslots && ((fbits[id >> 5] >> (id & 31u)) & 1u); };
    uint32_t* m_ids = a.m_ids ? a.m_ids + (uint64_t)qi * a.m_cap : nullptr;
    float* m_d = a.m_d ? a.m_d + (uint64_t)qi * a.m_cap : nullptr;
    unsigned long long* m_keys = a.m_keys ? a.m_keys + (uint64_t)qi * a.key_cap : nullptr;
    uint32_t nm = 0, sample_visited = 0, sample_matched = 0;
    bool l_adjusted = false;
    // DANN_TIE_RUST: 64 keys (the rejected candidates of a multihop hop) + the sorter's work area, per query
    unsigned long long* const tie_keys =
        (FILT && a.tie_work) ? reinterpret_cast<unsigned long long*>(a.tie_work + (uint64_t)qi * kTieWorkBytes) : nullptr;
    // `keys[0 .. n).sort_unstable_by(fast_distance)` as the Rust standard library does it (rust_order.h), by one lane; the
    // keys are in global memory (written with agent-scope stores by the whole wave)
    auto rust_sort_keys = [&](unsigned long long* keys, uint32_t n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (lane == 0) rust_sort_keys_call(keys, n, reinterpret_cast<uint8_t*>(tie_keys) + 512);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // inline filter search: accepted candidates of cand[0..nc) go to matched_results in emission order
    auto append_matched = [&](uint32_t nc) -> uint32_t {
        uint32_t added = 0;
        for (uint32_t c0 = 0; c0 < nc; c0 += kWave) {
            const uint32_t c = c0 + lane;
            const bool mt = c < nc && fmatch(cand_id[c]);
            const uint64_t m = ballot64(mt);
            const uint32_t r = nm + added + mbcnt(m);
            if (mt && r < a.m_cap) {
                m_ids[r] = cand_id[c];
                m_d[r] = cand_d[c];
            }
            added += (uint32_t)__popcll(m);
        }
        nm += added;
        if (nm > a.m_cap) status = (uint32_t)(-DANN_EOVERFLOW);
        return added;
    };

    // row prefetch of the latency regime: one dword per 128-byte line of row `id`, all landing in one scratch register
    // nothing ever reads (see the beam loop)
    auto touch_row = [&](uint32_t id, bool on) {
        if (on) {
            const uint8_t* prow = ix.rows + (uint64_t)id * ix.row_stride;
            const uint32_t last = ix.layer_bytes >= 4u ? ix.layer_bytes - 4u : 0u;  // (PQ rows of fewer than four chunks)
            const uint8_t* p1 = prow + (128u < last ? 128u : last);
            const uint8_t* p2 = prow + (256u < last ? 256u : last);
            const uint8_t* p3 = prow + (384u < last ? 384u : last);
            const uint8_t* p4 = prow + last;
            asm volatile(
                "global_load_dword %0, %1, off\n\t"
                "global_load_dword %0, %2, off\n\t"
                "global_load_dword %0, %3, off\n\t"
                "global_load_dword %0, %4, off\n\t"
                "global_load_dword %0, %5, off"
                : "+v"(pf_dummy)  // read-write: the register stays reserved around the whole loop, so a load that
                                  // lands late never finds it holding something else
                : "v"(prow), "v"(p1), "v"(p2), "v"(p3), "v"(p4));
        }
    };

    // distance of every candidate in cand_id[0..nc) -> cand_d.  Stores with inline tags (ix.tag_off, store.rs:133-158):
    // the tag byte of each row is requested together with the row (no extra round trip), an unreadable slot
    // (tag < PUBLISHED) is marked kEmpty and compacted away afterwards -- expand_beam_inner skips it after the
    // visited insert and does not count it (provider.rs:448-473, 681-686).  Returns the number of candidates kept.
    const uint32_t tag_off = PLAIN ? 0u : ix.tag_off;
    auto gather = [&](uint32_t nc) -> uint32_t {
        if constexpr (TEAM > 1) {
            // (only the start points come this way: hop 0 of the team, started by this wave; cand_id is buffer 0)
            if (lane == 0) {
                mail[kMbHop + 0] = nc;
                mail[kMbHop + 1] = kEmpty;
                mail[kMbSpecSeq] = 1u;
                mb_store(mail + kMbGo, 1u | (nc << 20));
            }
            __syncthreads();  // "distances ready"
            return nc;
        } else
        if constexpr (DIM > 0 && !kInt) {
            constexpr int U = kGatherRows;
            for (uint32_t c0 = 0; c0 < nc; c0 += GROUPS * U) {
                const RT* rows[U];
                bool act[U];
                float out[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t c = c0 + u * GROUPS + g;
                    act[u] = c < nc;
                    uint32_t id = act[u] ? cand_id[c] : 0u;
                    rows[u] = reinterpret_cast<const RT*>(ix.rows + (uint64_t)id * ix.row_stride);
                }
                uint8_t tg[U];
                if (tag_off) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        tg[u] = (act[u] && v == 0) ? reinterpret_cast<const uint8_t*>(rows[u])[tag_off] : (uint8_t)255;
                }
                group_distance_pre<S::NACC, OP, DIM, U>(xq, rows, act, v, out);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t c = c0 + u * GROUPS + g;
                    if (act[u] && v == 0) {
                        cand_d[c] = post_op<OP, NORM>(out[u]);  // float rows only
                        if (tag_off && tg[u] < kTagPublished) cand_id[c] = kEmpty;
                    }
                }
            }
        } else if constexpr (DIM > 0 && kInt) {
            constexpr int U = kGatherRows;
            for (uint32_t c0 = 0; c0 < nc; c0 += GROUPS * U) {
                const uint8_t* rows[U];
                bool act[U];
                float out[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t c = c0 + u * GROUPS + g;
                    act[u] = c < nc;
                    uint32_t id = act[u] ? cand_id[c] : 0u;
                    rows[u] = ix.rows + (uint64_t)id * ix.row_stride;
                }
                uint8_t tg[U];
                if (tag_off) {
#pragma unroll
                    for (int u = 0; u < U; ++u) tg[u] = (act[u] && v == 0) ? rows[u][tag_off] : (uint8_t)255;
                }
                group_distance_int_pre<OP, DT == DT_I8, U>(xqi, xx_pre, rows, v, out);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t c = c0 + u * GROUPS + g;
                    if (act[u] && v == 0) {
                        cand_d[c] = finish_distance<DT, OP, NORM>(out[u], reinterpret_cast<const uint8_t*>(qs), rows[u],
                                                                  ix.dim, sqp);
                        if (tag_off && tg[u] < kTagPublished) cand_id[c] = kEmpty;
                    }
                }
            }
        } else if constexpr (DT == DT_PQ) {
            // pq_dist_lookup_single (fixed_chunk_pq_table.rs:82-100): one lane per candidate, table entries
            // added in chunk order in f32; code rows read 16 bytes at a time
            const float* lut = reinterpret_cast<const float*>(qs);
            for (uint32_t c = lane; c < nc; c += kWave) {
                const uint8_t* code = ix.rows + (uint64_t)cand_id[c] * ix.row_stride;
                const uint8_t tg = tag_off ? code[tag_off] : (uint8_t)255;
                float accum = 0.0f;
                for (uint32_t b0 = 0; b0 < ix.pq_chunks; b0 += 16) {
                    const uint4 w = *reinterpret_cast<const uint4*>(code + b0);
                    const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint32_t ch = b0 + i;
                        if (ch < ix.pq_chunks) accum += lut[ch * 256u + ((ws[i >> 2] >> (8 * (i & 3))) & 255u)];
                    }
                }
                cand_d[c] = accum;
                if (tg < kTagPublished) cand_id[c] = kEmpty;
            }
        } else {
            constexpr int U = S::kWide ? kWideRows : kGatherRows;
            for (uint32_t c0 = 0; c0 < nc; c0 += GROUPS * U) {
                const uint8_t* rows[U];
                bool act[U];
                float out[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t c = c0 + u * GROUPS + g;
                    act[u] = c < nc;
                    uint32_t id = act[u] ? cand_id[c] : 0u;
                    rows[u] = ix.rows + (uint64_t)id * ix.row_stride;
                }
                uint8_t tg[U];
                if (tag_off) {
#pragma unroll
                    for (int u = 0; u < U; ++u) tg[u] = (act[u] && v == 0) ? rows[u][tag_off] : (uint8_t)255;
                }
                group_distance_many<DT, OP, false, U>(qs, rows, act, (int)ix.dim, v, out);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t c = c0 + u * GROUPS + g;
                    if (act[u] && v == 0) {
                        cand_d[c] = finish_distance<DT, OP, NORM>(out[u], reinterpret_cast<const uint8_t*>(qs), rows[u],
                                                                  ix.dim, sqp);
                        if (tag_off && tg[u] < kTagPublished) cand_id[c] = kEmpty;
                    }
                }
            }
        }
        if (!tag_off) return nc;
        // drop the unreadable slots, emission order kept (forward compaction: writes never pass the reads)
        uint32_t w = 0;
        WS();
        for (uint32_t c0 = 0; c0 < nc; c0 += kWave) {
            const uint32_t c = c0 + lane;
            const uint32_t id = c < nc ? cand_id[c] : kEmpty;
            const float d = c < nc ? cand_d[c] : 0.0f;
            const bool ok = id != kEmpty;
            const uint64_t m = ballot64(ok);
            WS();
            if (ok) {
                const uint32_t r = w + mbcnt(m);
                cand_id[r] = id;
                cand_d[r] = d;
            }
            w += (uint32_t)__popcll(m);
            WS();
        }
        return w;
    };

    // merge cand[m0 .. m0+n) (n <= 64) into the queue.
    // Exactness: the sequential inserts keep the best `qcap` elements under the total order
    // (distance asc, insertion time desc), so (1) when the queue is full a candidate worse
    // than its last element can be dropped up front (queue.rs:142-146), (2) a surviving
    // candidate j lands at  #{old e: d_e < d_j} + #{surviving i: d_i < d_j or (d_i == d_j, i > j)},
    // (3) an old element e moves up by #{surviving j: d_j <= d_e}.
    // (the candidates in registers: lane j holds candidate j of n; cbi / cbd: n words each of LDS the slow path may use
    // to compact the survivors -- the candidates' own buffer)
    // distance of queue entry p (wave-uniform), from the registers
    auto queue_dist = [&](uint32_t p) -> float {
        float r = 0.0f;
#pragma unroll
        for (int s = 0; s < QS; ++s)
            if ((p >> 6) == (uint32_t)s)
                r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd[s]), (int)(p & 63u)));
        return r;
    };
    bool stage_stale = false;  // the LDS image of the queue is behind the registers (only the slow path reads it)
    auto merge_regs = [&](bool has, float nd, uint32_t nid, const uint32_t n, uint32_t* cbi, float* cbd) {
        bool nvalid = has && !(nd != nd);  // NaN distances are ignored (queue.rs:131-134)
        if (size == qcap && qcap > 0) nvalid = nvalid && !(queue_dist(size - 1) < nd);
        const uint64_t km = ballot64(nvalid);
        const uint32_t nv = (uint32_t)__popcll(km);
#ifdef DANN_PHASE_CYCLES
        ph_acc[8] += nv;
        ph_acc[9] += nv > kRegMerge ? 1 : 0;
        ph_acc[10] += 1;
#endif
        if (nv == 0) return;
        PH_T(phm0);
        if (QS <= 4 && nv <= kSeqInsert) {
            // a few survivors (the steady state of a full queue): the sequential inserts themselves, in emission order,
            // on the register-resident queue.  The lower bound is a ballot count; "move the tail up by one" is a DPP
            // wave shift per slot, lane 63 of one slot carried into lane 0 of the next.  No LDS, no cross-lane
            // round trip.  (Entries at or beyond `size` hold don't-care values, as everywhere.)
            for (uint64_t mm = km; mm; mm &= mm - 1) {
                const int j = __builtin_ctzll(mm);
                const int dj_bits = __builtin_amdgcn_readlane(__builtin_bit_cast(int, nd), j);
                const float dj = __builtin_bit_cast(float, dj_bits);
                const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)nid, j);
                uint32_t pos = 0;
#pragma unroll
                for (int s = 0; s < QS; ++s)
                    pos += (uint32_t)__popcll(ballot64(((uint32_t)(s * kWave) + lane < size) && qd[s] < dj));
                if (pos >= qcap) continue;  // behind a full queue's last entry (it was equal to it when tested, not any more)
#pragma unroll
                for (int s = QS - 1; s >= 0; --s) {
                    if (pos >= (uint32_t)((s + 1) * kWave) || size < (uint32_t)(s * kWave)) continue;  // slot untouched
                    const uint32_t p = (uint32_t)(s * kWave) + lane;
                    // lane l <- lane l - 1; lane 0 <- lane 63 of the slot below (the `old` operand of the DPP move)
                    const int cd = s > 0 ? __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd[s > 0 ? s - 1 : 0]), 63) : 0;
                    const int ci = s > 0 ? __builtin_amdgcn_readlane((int)qid[s > 0 ? s - 1 : 0], 63) : 0;
                    const int sd = __builtin_amdgcn_update_dpp(cd, __builtin_bit_cast(int, qd[s]), 0x138, 0xf, 0xf, false);
                    const int si = __builtin_amdgcn_update_dpp(ci, (int)qid[s], 0x138, 0xf, 0xf, false);
                    if (p > pos) {
                        qd[s] = __builtin_bit_cast(float, sd);
                        qid[s] = (uint32_t)si;
                    } else if (p == pos) {
                        qd[s] = dj;
                        qid[s] = idj;
                    }
                }
                size = size < qcap ? size + 1u : qcap;
            }
            stage_stale = true;
            PH_T(phm1s);
            PH_ADD(11, phm0, phm1s);
            return;
        }
        uint32_t shift[QS];
        uint32_t pos_new = 0;
        if (QS <= 4 && nv <= kRegMerge) {
            // few survivors (queues beyond 256 entries always take the LDS path below, whose cost does not grow with
            // the number of register slots).  One pass over them gives every survivor its rank among the survivors
            // (`before`) and its lower bound in the queue (a ballot count per slot, wave-uniform, kept in the
            // survivor's own lane), and every queue entry e the number of survivors that go in front of it
            // (shift_e = #{j: d_j <= d_e}).  No LDS before the scatter.
            uint32_t before = 0, lbound = 0;
#pragma unroll
            for (int s = 0; s < QS; ++s) shift[s] = 0;
            for (uint64_t mm = km; mm; mm &= mm - 1) {
                const int j = __builtin_ctzll(mm);
                const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nd), j));
                before += (dj < nd) ? 1u : 0u;
                before += ((dj == nd) & ((uint32_t)j > lane)) ? 1u : 0u;
                uint32_t lb = 0;
#pragma unroll
                for (int s = 0; s < QS; ++s) {
                    shift[s] += (dj <= qd[s]) ? 1u : 0u;  // entries >= size: never scattered
                    lb += (uint32_t)__popcll(ballot64(((uint32_t)(s * kWave) + lane < size) && qd[s] < dj));
                }
                lbound = (int)lane == j ? lb : lbound;
            }
            has = nvalid;
            pos_new = lbound + before;
        } else {
        if (stage_stale) {  // the lower-bound search below reads the queue's LDS image
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                const uint32_t p = (uint32_t)(s * kWave) + lane;
                if (p < size) stage[p] = make_uint2(qid[s], __builtin_bit_cast(uint32_t, qd[s]));
            }
            WS();
        }
        if (nv != n) {  // compact the survivors, emission order preserved
            const uint32_t cj = mbcnt(km);
            WS();
            if (nvalid) {
                cbd[cj] = nd;
                cbi[cj] = nid;
            }
            WS();
            has = lane < nv;
            nd = has ? cbd[lane] : 0.0f;
            nid = has ? cbi[lane] : kEmpty;
        }
        // rank among the survivors
        uint32_t before = 0;
        for (uint32_t jj = 0; jj < nv; ++jj) {
            const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nd), jj));
            before += ((dj < nd) | ((dj == nd) & (jj > lane))) ? 1u : 0u;
        }
        // (the sorted survivors go where the candidates came from: cbi's content is in registers by now)
        float* const snew = reinterpret_cast<float*>(cbi);
        if (has) snew[before] = nd;
        WS();
        // old elements: shift = #{new <= d_e}  (upper bound in snew[0..nv))
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            uint32_t lo = 0;
#pragma unroll
            for (uint32_t step = 64; step > 0; step >>= 1) {
                const uint32_t t = lo + step;
                if (t <= nv && snew[t - 1] <= qd[s]) lo = t;
            }
            shift[s] = lo;
        }
        // new elements: #{old < d_j}  (lower bound in the queue image)
        uint32_t lb = 0;
#pragma unroll
        for (uint32_t step = QCAPP; step > 0; step >>= 1) {
            const uint32_t t = lb + step;
            if (t <= size && stage_dist(t - 1) < nd) lb = t;
        }
        pos_new = before + lb;
        }
        PH_T(phm1);
        PH_ADD(11, phm0, phm1);
        // scatter into the queue image, then reload
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const uint32_t p = (uint32_t)(s * kWave) + lane;
            if (p < size) {
                const uint32_t np = p + shift[s];
                if (np < qcap) stage[np] = make_uint2(qid[s], __builtin_bit_cast(uint32_t, qd[s]));
            }
        }
        if (has && pos_new < qcap) stage[pos_new] = make_uint2(nid, __builtin_bit_cast(uint32_t, nd));
        const uint32_t total = size + nv;
        size = total < qcap ? total : qcap;
        stage_stale = false;
        WS();
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const uint32_t p = (uint32_t)(s * kWave) + lane;
            if (p < size) {
                const uint2 e = stage[p];
                qid[s] = e.x;
                qd[s] = __builtin_bit_cast(float, e.y);
            }
        }
    };
    auto merge = [&](uint32_t m0, uint32_t n) {
        const bool has = lane < n;
        merge_regs(has, has ? cand_d[m0 + lane] : 0.0f, has ? cand_id[m0 + lane] : kEmpty, n, cand_id + m0, cand_d + m0);
    };

    // expand `nb` nodes of beam[]: adjacency rows in pop order, ids in stored order, visited filter
    // (provider.rs:448-454); survivors go to cand_id[0..nc)
    // accept_only (expand_beam_accept_only, labeled.rs:196-214,284-291): ids that do not match the filter
    // are skipped *before* the visited set sees them
    auto claim_spill = [&]() {
        if (spill) return;
        uint32_t slice = kEmpty;
        if (a.spill) {
            // slices are recycled inside a launch: busy flag per slice, rotating start
            if (lane == 0) {
                uint32_t* busy = a.spill_next + 16;
                uint32_t s = atomicAdd(a.spill_next, 1u) % a.spill_slices;
                for (uint32_t t = 0; t < 2u * a.spill_slices; ++t) {
                    if (atomicCAS(&busy[s], 0u, 1u) == 0u) {
                        slice = s;
                        break;
                    }
                    s = (s + 1 == a.spill_slices) ? 0u : s + 1;
                }
            }
            slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)slice);
        }
        if (slice < a.spill_slices) spill = a.spill + ((uint64_t)slice << a.spill_bits);
    };
    auto expand = [&](uint32_t nb, bool accept_only = false, bool node_in_reg = false) -> uint32_t {
        uint32_t nc = 0;
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t node = node_in_reg ? node0 : beam[b];  // the main loop's single pop stays in a register
            const uint32_t* arow = ix.adj + (uint64_t)node * ix.adj_stride;
            const bool hit = (node == pf_node);
#ifdef DANN_PHASE_CYCLES
            ph_acc[hit ? 5 : 6] += 1;
#endif
            uint32_t len = hit ? adj_len(pf_lenv) : arow[0];
            len = len < R ? len : R;  // Neighbors::get clamps (neighbors.rs:146-148)
            if (lds_open && ht_count + len > a.ht_open) {
                // freeze the LDS table, claim a spill table (kept once claimed)
                lds_open = false;
                claim_spill();
            }
            if (!lds_open && (!spill || spill_count + len > spill_size - (spill_size >> 2))) {
                status = (uint32_t)(-DANN_EOVERFLOW);
                break;
            }
            PH_T(phv0);
            for (uint32_t j0 = 0; j0 < len; j0 += kWave) {
                const uint32_t j = j0 + lane;
                const bool inb = j < len;
                const uint32_t id = hit ? (inb ? pf_val : kEmpty) : (inb ? arow[1 + j] : kEmpty);
                bool isnew = false;
                // (HT16: the table holds ids below 2^m only; an id beyond the index is never a candidate anyway)
                const bool act = inb && id != kEmpty && (!accept_only || fmatch(id)) && (!HT16 || id < ix.nslots);
                if (lds_open) {
                    if constexpr (HT16) {
                        const int r = ht16_insert_open(ht, h16, id, act);
                        isnew = act && r == kHt16Inserted;
                        const bool exh = act && r == kHt16Exhausted;
                        if (ballot64(exh)) {  // (rare) no slot among this id's probes: freeze the table, the id goes to the spill table
                            lds_open = false;
                            claim_spill();
                            if (!spill) {
                                status = (uint32_t)(-DANN_EOVERFLOW);
                                break;
                            }
                            if (exh) isnew = spill_insert(spill, spill_mask, spill_shift, id);
                        }
                    } else {
                        isnew = ht_insert_open(ht, ht_mod, id, act);
                    }
                } else if (act) {  // frozen LDS table: lookup, then the spill table in global memory
                    if constexpr (HT16) isnew = !ht16_contains(ht, h16, id) && spill_insert(spill, spill_mask, spill_shift, id);
                    else isnew = ht_visit(ht, ht_mod, id, false) == kAbsent && spill_insert(spill, spill_mask, spill_shift, id);
                }
                const bool keep = isnew && id < ix.nslots;
                const uint64_t nm = ballot64(isnew), km = ballot64(keep);
                if (keep) cand_id[nc + mbcnt(km)] = id;
                nc += (uint32_t)__popcll(km);
                if (lds_open) ht_count += (uint32_t)__popcll(nm);
                else spill_count += (uint32_t)__popcll(nm);
            }
            PH_T(phv1);
            PH_ADD(7, phv0, phv1);
            if (HT16 && status) break;
        }
        return nc;
    };

    // ---- start points: frozen slots [capacity, capacity + nstart) (index.rs:1950-1958) ---
    {
        const uint32_t ns = ix.nstart;
        for (uint32_t i = lane; i < ns; i += kWave) {
            cand_id[i] = ix.capacity + i;
            visit_open(ix.capacity + i);
        }
        if (HT16 && ballot64(status_early != 0)) status = (uint32_t)(-DANN_EOVERFLOW);
        ht_count = ns;
        WS();
        // start points are created FROZEN (store.rs:766-772) and the host refuses to unpublish them; should one be
        // unreadable all the same, the search fails as the reference's does
        const uint32_t nsk = gather(ns);
        if (nsk != ns) status = (uint32_t)(-DANN_EINVAL);  // "could not retrieve start point" (provider.rs:408-431)
        WS();
        // the filtered searches do not count the start points as comparisons (inline_filter_search.rs:186-197)
        cmps = fmode ? 0u : nsk;
        if (fmode == DANN_FILTER_INLINE) append_matched(nsk);
        for (uint32_t m0 = 0; m0 < nsk; m0 += kWave) merge(m0, (nsk - m0) < (uint32_t)kWave ? (nsk - m0) : (uint32_t)kWave);
    }

    uint32_t* const rec_i = a.rec_ids ? a.rec_ids + (uint64_t)qi * a.rec_stride : nullptr;  // wave-uniform
    float* const rec_d = a.rec_ids ? a.rec_dists + (uint64_t)qi * a.rec_stride : nullptr;
    // W == 1 pop (queue.rs:297-313), one scan: the first unexpanded entry is popped (node0: kept in a register, no LDS
    // round trip); the second one (pf_next, distance pf_next_d) is the node the next hop will expand unless a new
    // candidate gets in front of it, the third one (pf_next2) the node after that
    auto pop_one = [&](uint32_t& pf_next, float& pf_next_d, uint32_t& pf_next2, float& pf_next2_d) -> uint32_t {
        uint32_t got = 0;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            if (got >= 3u) continue;
            const bool cand = ((uint32_t)(s * kWave) + lane < size) && !(qid[s] & kVisitedBit);
            uint64_t m = ballot64(cand);
            if (got == 0u && m) {
                const int l = __builtin_ctzll(m);
                const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)qid[s], l);
                if ((int)lane == l) qid[s] |= kVisitedBit;
                node0 = id;
                if (rec_i) {  // VisitedSearchRecord::record (search/record.rs:86-93)
                    if (nrec < a.rec_stride) {
                        const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd[s]), l));
                        if (lane == 0) {
                            rec_i[nrec] = id;
                            rec_d[nrec] = d;
                        }
                    } else {
                        status = (uint32_t)(-DANN_EOVERFLOW);
                    }
                    ++nrec;
                }
                got = 1;
                m &= m - 1;
            }
            if (got == 1u && m) {
                const int l2 = __builtin_ctzll(m);
                pf_next = (uint32_t)__builtin_amdgcn_readlane((int)qid[s], l2);
                pf_next_d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd[s]), l2));
                got = 2;
                m &= m - 1;
            }
            if (got == 2u && m) {
                const int l3 = __builtin_ctzll(m);
                pf_next2 = (uint32_t)__builtin_amdgcn_readlane((int)qid[s], l3);
                pf_next2_d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd[s]), l3));
                got = 3;
            }
        }
        return got ? 1u : 0u;
    };
    // ---- beam loop of a team's queue wave (see team_control_wave): pop, publish, wait for the hop's distances, merge -----
    if constexpr (TEAM > 1) {
        mb_store(mail + kMbHtCount0, ht_count);
        if (lane == 0) mail[kMbLoaded] = 1u;  // (the start points' distances went into the queue above)
        for (uint32_t npop = 1;; ++npop) {
            PH_T(ph0);
            uint32_t pf_next = kEmpty, pf_next2 = kEmpty;
            float pf_next_d = 0.0f, pf_next2_d = 0.0f;
            const uint32_t nb = pop_one(pf_next, pf_next_d, pf_next2, pf_next2_d);
            hops += nb;
            if (lane == 0) {
                uint32_t* w = mail + kMbPop + 8u * (npop & 1u);
                w[1] = nb;
                w[2] = node0;
                w[3] = pf_next;
                w[4] = __builtin_bit_cast(uint32_t, pf_next_d);
                w[5] = pf_next2;
                w[6] = __builtin_bit_cast(uint32_t, pf_next2_d);
                mb_store(w, npop);
            }
            PH_T(ph1);
            PH_ADD(0, ph0, ph1);
            __syncthreads();  // "distances ready" of hop npop -- or the release
            PH_T(ph2);
            PH_ADD(14, ph1, ph2);
            // (one batch of LDS loads, one wait: the release word, the hop's word and BOTH candidate buffers -- which of
            // them the hop used is in the word; a buffer holds 64 entries whatever the hop's count)
            const uint32_t go_v = __hip_atomic_load(mail + kMbGo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t word_v = mail[kMbHop + 8u * (npop & 1u)];
            const float nd0 = reinterpret_cast<const float*>(smem + L.cand_d_off)[lane];
            const float nd1 = reinterpret_cast<const float*>(smem + L.cand_d_off + cstride)[lane];
            const uint32_t ni0 = reinterpret_cast<const uint32_t*>(smem + L.cand_id_off)[lane];
            const uint32_t ni1 = reinterpret_cast<const uint32_t*>(smem + L.cand_id_off + cstride)[lane];
            if (uni(go_v) == kTeamExit) break;
            const uint32_t word = uni(word_v);
            const uint32_t nc = word & 0xFFFFu, buf = (word >> 16) & 1u;
            const bool has = lane < nc;
            const float nd = has ? (buf ? nd1 : nd0) : 0.0f;
            const uint32_t nid = has ? (buf ? ni1 : ni0) : kEmpty;
            WS();
            if (lane == 0) mb_store(mail + kMbLoaded, npop + 1u);  // the buffer may be refilled now
            merge_regs(has, nd, nid, nc, reinterpret_cast<uint32_t*>(smem + L.mscr_off),
                       reinterpret_cast<float*>(smem + L.mscr_off + 256u));
            PH_T(ph4);
            PH_ADD(3, ph2, ph4);
            PH_ADD(4, ph0, ph4);
        }
        const uint32_t dstatus = mail[kMbDStatus];
        if (!status) status = dstatus;
        cmps += mail[kMbDCmps];
    } else
    // ---- beam loop ----------------------------------------------------------------------
    for (;;) {
        PH_T(ph0);
        // pop up to W closest unexpanded entries (queue.rs:297-313)
        uint32_t nb = 0;
        uint32_t pf_next = kEmpty;  // W == 1: the best entry still unexpanded after this pop (the prefetch target)
        float pf_next_d = 0.0f;
        if (W == 1) {
            uint32_t pf_next2 = kEmpty;
            float pf_next2_d = 0.0f;
            nb = pop_one(pf_next, pf_next_d, pf_next2, pf_next2_d);
        } else
        for (uint32_t w = 0; w < W; ++w) {
            bool found = false;
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                if (found) continue;
                const bool cand = ((uint32_t)(s * kWave) + lane < size) && !(qid[s] & kVisitedBit);
                const uint64_t m = ballot64(cand);
                if (m) {
                    const int l = __builtin_ctzll(m);
                    const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)qid[s], l);
                    const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd[s]), l));
                    if ((int)lane == l) qid[s] |= kVisitedBit;
                    if (lane == 0) {
                        beam[nb] = id;
                        if (rec_i && nrec < a.rec_stride) {  // VisitedSearchRecord::record (search/record.rs:86-93)
                            rec_i[nrec] = id;
                            rec_d[nrec] = d;
                        }
                    }
                    if (rec_i) {
                        if (nrec >= a.rec_stride) status = (uint32_t)(-DANN_EOVERFLOW);
                        ++nrec;
                    }
                    ++nb;
                    found = true;
                }
            }
            if (!found) break;
        }
        if (nb == 0 || status) break;
        hops += nb;
        if (W > 1) WS();
        PH_T(ph1);
        PH_ADD(0, ph0, ph1);

        const uint32_t nc_seen = expand(nb, false, W == 1);
        if (status) break;
        WS();
        PH_T(ph2);
        PH_ADD(1, ph1, ph2);
        // speculative adjacency prefetch: while the candidate rows are in flight, fetch the
        // adjacency row of the best unexpanded entry of the *current* queue; if no new
        // candidate beats it, the next hop starts without a dependent HBM round trip.
        pf_node = kEmpty;
        if (PLAIN || (W == 1 && R <= (uint32_t)kWave)) {
            pf_node = pf_next;
            if (pf_node != kEmpty) adj_fetch(pf_node, pf_lenv, pf_val);
        }
        const uint32_t nc = gather(nc_seen);
        // latency mode (few queries in flight, bandwidth to spare): touch the rows of the predicted next node's
        // neighbours, one dword per 128-byte line, so that the next hop's gather is served by L2 instead of HBM when
        // the prediction holds.  Pure prefetch: no visited-set side effect, nothing ever waits for these loads.  They
        // all land in one scratch register the compiler keeps reserved (the empty asm "uses" it one hop later, after
        // that hop's gather has drained the in-order load queue; the loop exit drains it explicitly).
        asm volatile("" ::"v"(pf_dummy));
        if ((a.tune & kTuneRowPrefetch) && pf_node != kEmpty && ix.layer_bytes <= 512u) {
            const uint32_t plen = adj_len(pf_lenv);
            touch_row(pf_val, lane < (plen < R ? plen : R) && pf_val < ix.nslots);
        }
        WS();
        PH_T(ph3);
        PH_ADD(2, ph2, ph3);
        cmps += nc;
        if (fmode == DANN_FILTER_MULTIHOP) {
            // multihop_search_internal (multihop_filter_search.rs:172-236); nc <= 64 (checked by the host)
            const bool hasc = lane < nc;
            const uint32_t cid = hasc ? cand_id[lane] : kEmpty;
            const float cd = hasc ? cand_d[lane] : 0.0f;
            const bool acc = hasc && fmatch(cid);
            const bool rej = hasc && !acc;
            const uint64_t am = ballot64(acc), rm = ballot64(rej);
            const uint32_t na = (uint32_t)__popcll(am);
            // rejected nodes closest first, at most max_degree / 2 of them expand a second hop.  Equal distances: by
            // emission order (a stable sort), or -- DANN_TIE_RUST, when the hop has any -- where the reference's
            // `sort_unstable_by(fast_distance)` leaves them (multihop_filter_search.rs:207; tie_sorted below)
            uint32_t rank = 0;
            bool tied = rej && cd != cd;
            for (uint64_t mm = rm; mm; mm &= mm - 1) {
                const int j = __builtin_ctzll(mm);
                const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cd), j));
                rank += ((dj < cd) | ((dj == cd) & ((uint32_t)j < lane))) ? 1u : 0u;
                tied |= rej & (dj == cd) & ((uint32_t)j != lane);
            }
            const uint32_t nrej = (uint32_t)__popcll(rm);
            const uint32_t nsel = nrej < R / 2 ? nrej : R / 2;
            const bool tie_sorted = tie_keys && ballot64(tied);
            uint32_t tie_src = lane;  // position `lane` of the sorted rejected list holds the candidate of lane tie_src
            if (tie_sorted) {
                const uint32_t e = mbcnt(rm);  // emission order of the rejected candidates
                if (rej) key_store(tie_keys + e, ((unsigned long long)ordered_bits(cd) << 32) | lane);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                rust_sort_keys(tie_keys, nrej);
                if (lane < nrej) tie_src = (uint32_t)key_load(tie_keys + lane);
            }
            WS();
            if (acc) {  // accepted one-hop neighbours, emission order kept
                const uint32_t r = mbcnt(am);
                cand_id[r] = cid;
                cand_d[r] = cd;
            }
            WS();
            if (na) merge(0, na);
            WS();
            if (tie_sorted) {
                const uint32_t scid = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(tie_src << 2), (int)cid);
                if (lane < nsel) cand_id[lane] = scid;
            } else if (rej && rank < nsel) {
                cand_id[rank] = cid;
            }
            WS();
            const uint32_t sel = lane < nsel ? cand_id[lane] : kEmpty;
            WS();
            const uint32_t gsz = (cmax / R) < (uint32_t)kMaxBeam ? (cmax / R) : (uint32_t)kMaxBeam;
            for (uint32_t b0 = 0; b0 < nsel && !status; b0 += gsz) {
                const uint32_t gb = nsel - b0 < gsz ? nsel - b0 : gsz;
                for (uint32_t t = 0; t < gb; ++t) {
                    const uint32_t node = (uint32_t)__builtin_amdgcn_readlane((int)sel, (int)(b0 + t));
                    if (lane == 0) beam[t] = node;
                }
                WS();
                const uint32_t nc2_seen = expand(gb, true);
                if (status) break;
                WS();
                const uint32_t nc2 = gather(nc2_seen);
                WS();
                cmps += nc2;
                for (uint32_t m0 = 0; m0 < nc2; m0 += kWave)
                    merge(m0, (nc2 - m0) < (uint32_t)kWave ? (nc2 - m0) : (uint32_t)kWave);
                WS();
            }
            if (status) break;
            hops += nsel;
        } else {
            if (fmode == DANN_FILTER_INLINE) {
                sample_matched += append_matched(nc);
                sample_visited += nc;
                if (status) break;
            }
            for (uint32_t m0 = 0; m0 < nc; m0 += kWave) merge(m0, (nc - m0) < (uint32_t)kWave ? (nc - m0) : (uint32_t)kWave);
            // AdaptiveL (inline_filter_search.rs:262-277): one resize, decided from the hit rate so far; the new
            // L comes from a table the host filled with compute_adaptive_l (f64 log10 / powf of the host libm)
            if (FILT && a.ad_samples && !l_adjusted && sample_visited >= a.ad_samples) {
                l_adjusted = true;
                const uint32_t new_l = a.ad_table[(uint64_t)(sample_visited - a.ad_samples) * a.ad_stride + sample_matched];
                if (new_l > a.l_value) {  // NeighborPriorityQueue::reconfigure (queue.rs:339-353)
                    qcap = new_l;
                    if (size > qcap) size = qcap;
                }
            }
        }
        PH_T(ph4);
        PH_ADD(3, ph3, ph4);
        PH_ADD(4, ph0, ph4);
    }

    if (a.tune & kTuneRowPrefetch) {  // no prefetch load may still be in flight when its landing register is released
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" ::"v"(pf_dummy));
    }
    // ---- graph::search::Range second phase (range_search.rs:283-316, 424-470) -----------------------
    uint32_t range_written = 0, range_second = 0;
    // ---- inline filter search: matched_results sorted by distance (inline_filter_search.rs:279) ------------
    // sort_unstable_by(fast_distance): key = ordered distance bits << 32 | push index, sorted by a network -- the
    // order of a stable sort.  A list without equal distances has one sorted order; one with ties is, under
    // DANN_TIE_RUST, sorted again from its push order the way the reference's unstable sort does it (rust_sort_keys).
    uint32_t nkeys = 0;
    if (fmode == DANN_FILTER_INLINE && !status) {
        nkeys = 1;
        while (nkeys < nm) nkeys <<= 1;
        __threadfence_block();
        auto fill = [&]() {
            for (uint32_t i = lane; i < nkeys; i += kWave)
                key_store(m_keys + i, i < nm ? ((unsigned long long)ordered_bits(f32_load(m_d + i)) << 32) | i : ~0ull);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        fill();
        if (nm > 1) wave_sort_keys(m_keys, nkeys, lane);
        if (tie_keys && nm > 1) {
            bool tied = false;
            for (uint32_t i = lane; i + 1 < nm; i += kWave) {
                const uint32_t o0 = (uint32_t)(key_load(m_keys + i) >> 32), o1 = (uint32_t)(key_load(m_keys + i + 1) >> 32);
                const float f0 = from_ordered_bits(o0), f1 = from_ordered_bits(o1);
                tied |= (f0 == f1) | (f1 != f1);  // (-0.0 == +0.0; a NaN is Equal to everything)
            }
            if (ballot64(tied)) {
                fill();
                rust_sort_keys(m_keys, nm);
            }
        }
    }
    if (a.range_ids && fmode == DANN_FILTER_INLINE && !status) {
        // ---- FilteredRange (filtered_range_search.rs:140-330) -------------------------------------------
        uint32_t* wids = a.range_ids + (uint64_t)qi * a.range_cap;  // matched_within_radius
        float* wds = a.range_d + (uint64_t)qi * a.range_cap;
        // the matched entries within the radius are a prefix of the sorted list
        uint32_t nw = 0;
        for (uint32_t i0 = 0; i0 < nm; i0 += kWave) {
            const uint32_t i = i0 + lane;
            uint32_t id = kEmpty;
            float d = 0.0f;
            bool in = false;
            if (i < nm) {
                const unsigned long long key = key_load(m_keys + i);
                d = from_ordered_bits((uint32_t)(key >> 32));
                id = u32_load(m_ids + (uint32_t)key);
                in = d <= a.radius;
            }
            const uint64_t m = ballot64(in);
            const uint32_t r = nw + mbcnt(m);
            if (in && r < a.range_cap) {
                wids[r] = id;
                wds[r] = d;
            }
            nw += (uint32_t)__popcll(m);
        }
        if (nw > a.range_cap) status = (uint32_t)(-DANN_EOVERFLOW);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // in_range = (first starting_l queue entries + matched) within the radius, sorted by (distance, id),
        // duplicates removed (:170-181)
        uint32_t n2 = 0;
        const uint32_t take = size < a.l_value ? size : a.l_value;
        if (!status) {
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                const uint32_t p = (uint32_t)(s * kWave) + lane;
                const bool in = p < take && qd[s] <= a.radius;
                const uint64_t m = ballot64(in);
                const uint32_t r = n2 + mbcnt(m);
                if (in && r < a.key_cap)
                    key_store(m_keys + r, ((unsigned long long)ordered_bits(qd[s]) << 32) | (qid[s] & ~kVisitedBit));
                n2 += (uint32_t)__popcll(m);
            }
            for (uint32_t i0 = 0; i0 < nw; i0 += kWave) {
                const uint32_t i = i0 + lane;
                if (i < nw && n2 + i < a.key_cap)
                    key_store(m_keys + n2 + i, ((unsigned long long)ordered_bits(f32_load(wds + i)) << 32) | u32_load(wids + i));
            }
            n2 += nw;
            if (n2 > a.key_cap) status = (uint32_t)(-DANN_EOVERFLOW);
        }
        uint32_t nf = 0;  // frontier length; frontier ids live in m_ids[] (the push-order list is dead now)
        if (!status) {
            uint32_t np2 = 1;
            while (np2 < n2) np2 <<= 1;
            for (uint32_t i = n2 + lane; i < np2; i += kWave) key_store(m_keys + i, ~0ull);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (n2 > 1) wave_sort_keys(m_keys, np2, lane);
            for (uint32_t i0 = 0; i0 < n2; i0 += kWave) {
                const uint32_t i = i0 + lane;
                bool keep = false;
                uint32_t id = kEmpty;
                if (i < n2) {
                    const unsigned long long key = key_load(m_keys + i);
                    id = (uint32_t)key;
                    keep = i == 0 || (uint32_t)key_load(m_keys + i - 1) != id;
                }
                const uint64_t m = ballot64(keep);
                const uint32_t r = nf + mbcnt(m);
                if (keep && r < a.m_cap) m_ids[r] = id;
                nf += (uint32_t)__popcll(m);
            }
            if (nf > a.m_cap) status = (uint32_t)(-DANN_EOVERFLOW);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!status && nf >= a.range_thresh && nw < a.range_max) {
            range_second = 1;
            // visited := ids of in_range; range_frontier := in_range (:190-199)
            WS();
            for (uint32_t i = lane; i < ht_size; i += kWave) ht[i] = kEmpty;
            if (spill) spill_wipe(spill, spill_size, lane);
            WS();
            lds_open = true;
            ht_count = 0;
            spill_count = 0;
            pf_node = kEmpty;
            for (uint32_t i0 = 0; i0 < nf && !status; i0 += kWave) {
                const uint32_t i = i0 + lane;
                const uint32_t cnt = (nf - i0) < (uint32_t)kWave ? (nf - i0) : (uint32_t)kWave;
                if (lds_open && ht_count + cnt > a.ht_open) status = (uint32_t)(-DANN_EOVERFLOW);
                else if (i < nf) visit_open(u32_load(m_ids + i));
                if (HT16 && ballot64(status_early != 0)) status = (uint32_t)(-DANN_EOVERFLOW);
                ht_count += cnt;
            }
            WS();
            const float nav = a.radius * a.range_slack;
            uint32_t front = 0;
            // filtered_range_search_internal (:263-330): cmps and hops keep accumulating
            while (!status && front < nf && nw < a.range_max) {
                const uint32_t nb = nf - front < W ? nf - front : W;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < nb) beam[lane] = u32_load(m_ids + front + lane);
                front += nb;
                WS();
                const uint32_t nc_seen = expand(nb);
                if (status) break;
                WS();
                const uint32_t nc = gather(nc_seen);
                WS();
                cmps += nc;
                hops += nb;
                for (uint32_t c0 = 0; c0 < nc; c0 += kWave) {
                    const uint32_t c = c0 + lane;
                    const bool has = c < nc;
                    const uint32_t id = has ? cand_id[c] : kEmpty;
                    const float d = has ? cand_d[c] : 0.0f;
                    const bool fr = has && d <= nav;
                    const uint64_t fm = ballot64(fr);
                    const uint32_t rf = nf + mbcnt(fm);
                    if (fr && rf < a.m_cap) m_ids[rf] = id;
                    nf += (uint32_t)__popcll(fm);
                    const bool mt = fr && d <= a.radius && fmatch(id);
                    const uint64_t mm = ballot64(mt);
                    const uint32_t rw = nw + mbcnt(mm);
                    if (mt && rw < a.range_max && rw < a.range_cap) {
                        wids[rw] = id;
                        wds[rw] = d;
                    }
                    uint32_t add = (uint32_t)__popcll(mm);
                    if (nw + add > a.range_max) add = a.range_max - nw;
                    nw += add;
                }
                if (nf > a.m_cap || nw > a.range_cap) status = (uint32_t)(-DANN_EOVERFLOW);
            }
        }
        // matched_within_radius.take(max_returned) -> start points dropped -> inner radius -> output
        if (!status && a.out_ids) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint32_t* oi = a.out_ids + (uint64_t)qi * a.k;
            float* od = a.out_dists + (uint64_t)qi * a.k;
            const uint32_t lim = nw < a.range_max ? nw : a.range_max;
            for (uint32_t i0 = 0; i0 < lim; i0 += kWave) {
                const uint32_t i = i0 + lane;
                uint32_t id = kEmpty;
                float d = 0.0f;
                if (i < lim) {
                    id = u32_load(wids + i);
                    d = f32_load(wds + i);
                }
                const bool ok = i < lim && id < ix.capacity && !(a.has_inner && d <= a.inner_radius);
                const uint64_t m = ballot64(ok);
                const uint32_t r = range_written + mbcnt(m);
                if (ok && r < a.k) {
                    oi[r] = id;
                    od[r] = d;
                }
                range_written += (uint32_t)__popcll(m);
            }
            if (range_written > a.k) status = (uint32_t)(-DANN_EOVERFLOW);  // the reference's output Vec is unbounded
            range_written = range_written < a.k ? range_written : a.k;
            for (uint32_t r = range_written + lane; r < a.k; r += kWave) {
                oi[r] = kEmpty;
                od[r] = __builtin_inff();
            }
        }
    }
    if (a.range_ids && fmode != DANN_FILTER_INLINE && !status) {
        uint32_t* rids = a.range_ids + (uint64_t)qi * a.range_cap;
        float* rds = a.range_d + (uint64_t)qi * a.range_cap;
        const uint32_t max_ret = a.range_max < a.range_cap ? a.range_max : a.range_cap;  // list capacity
        // in_range = the first starting_l queue entries within the radius (start points included)
        uint32_t nr = 0;
        const uint32_t take = size < a.l_value ? size : a.l_value;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const uint32_t p = (uint32_t)(s * kWave) + lane;
            const bool in = p < take && qd[s] <= a.radius;
            const uint64_t m = ballot64(in);
            const uint32_t r = nr + mbcnt(m);
            if (in && r < a.range_cap) {
                rids[r] = qid[s] & ~kVisitedBit;
                rds[r] = qd[s];
            }
            nr += (uint32_t)__popcll(m);
        }
        if (nr > a.range_cap) status = (uint32_t)(-DANN_EOVERFLOW);
        const uint32_t init_hops = hops;
        if (!status && nr >= a.range_thresh && nr < a.range_max) {
            range_second = 1;
            // visited := ids of in_range only (range_search.rs:297-301)
            WS();
            for (uint32_t i = lane; i < ht_size; i += kWave) ht[i] = kEmpty;
            if (spill) spill_wipe(spill, spill_size, lane);
            WS();
            lds_open = true;
            ht_count = 0;
            spill_count = 0;
            pf_node = kEmpty;
            for (uint32_t i0 = 0; i0 < nr && !status; i0 += kWave) {
                const uint32_t i = i0 + lane;
                const uint32_t cnt = (nr - i0) < (uint32_t)kWave ? (nr - i0) : (uint32_t)kWave;
                if (lds_open && ht_count + cnt > a.ht_open) status = (uint32_t)(-DANN_EOVERFLOW);
                else if (i < nr) visit_open(rids[i]);
                if (HT16 && ballot64(status_early != 0)) status = (uint32_t)(-DANN_EOVERFLOW);
                ht_count += cnt;
            }
            WS();
            const float rlimit = a.radius * a.range_slack;
            uint32_t front = 0;
            while (!status && front < nr && nr < a.range_max) {
                // next beam: up to W ids from the front of the frontier (== in_range in arrival order)
                uint32_t nb = nr - front < W ? nr - front : W;
                __threadfence_block();  // rids[] written and re-read by this wave only
                if (lane < nb) beam[lane] = rids[front + lane];
                front += nb;
                WS();
                const uint32_t nc_seen = expand(nb);
                if (status) break;
                WS();
                const uint32_t nc = gather(nc_seen);
                WS();
                hops += nb;
                // append survivors in emission order while the list has room
                for (uint32_t c0 = 0; c0 < nc; c0 += kWave) {
                    const uint32_t c = c0 + lane;
                    const bool in = c < nc && cand_d[c] <= rlimit;
                    const uint64_t m = ballot64(in);
                    const uint32_t r = nr + mbcnt(m);
                    if (in && r < a.range_max) {
                        if (r < a.range_cap) {
                            rids[r] = cand_id[c];
                            rds[r] = cand_d[c];
                        }
                    }
                    uint32_t add = (uint32_t)__popcll(m);
                    if (nr + add > a.range_max) add = a.range_max - nr;
                    nr += add;
                    if (nr > a.range_cap) status = (uint32_t)(-DANN_EOVERFLOW);
                }
            }
            hops = init_hops + hops;  // the reference adds the cumulative counter to the initial one (:308-314)
        }
        (void)max_ret;
        // post-process: start points dropped, inner/outer radius filter, output buffer capacity k
        if (!status && a.out_ids) {
            __threadfence_block();
            uint32_t* oi = a.out_ids + (uint64_t)qi * a.k;
            float* od = a.out_dists + (uint64_t)qi * a.k;
            for (uint32_t i0 = 0; i0 < nr; i0 += kWave) {
                const uint32_t i = i0 + lane;
                uint32_t id = kEmpty;
                float d = 0.0f;
                if (i < nr) {
                    id = rids[i];
                    d = rds[i];
                }
                const bool ok = i < nr && id < ix.capacity && !(a.has_inner && d <= a.inner_radius) && d <= a.radius;
                const uint64_t m = ballot64(ok);
                const uint32_t r = range_written + mbcnt(m);
                if (ok && r < a.k) {
                    oi[r] = id;
                    od[r] = d;
                }
                range_written += (uint32_t)__popcll(m);
            }
            range_written = range_written < a.k ? range_written : a.k;
            for (uint32_t r = range_written + lane; r < a.k; r += kWave) {
                oi[r] = kEmpty;
                od[r] = __builtin_inff();
            }
        }
    }

#ifdef DANN_PHASE_CYCLES
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&a.phase_cycles[i], ph_acc[i]);
#endif
    if (spill) {  // hand the spill table back clean
        WS();
        spill_wipe(spill, spill_size, lane);
        WS();
        if (lane == 0) atomicExch(a.spill_next + 16 + (uint32_t)((spill - a.spill) >> a.spill_bits), 0u);
    }
    // ---- results: best entries in order, start points dropped (provider.rs:933-944) -----
    uint32_t written = range_written;
    if (a.out_ids && !a.range_ids && fmode == DANN_FILTER_INLINE) {
        // matched_results.take(l_value) -> Translate drops start points -> first k (inline_filter_search.rs:131-139)
        uint32_t* oi = a.out_ids + (uint64_t)qi * a.k;
        float* od = a.out_dists + (uint64_t)qi * a.k;
        const uint32_t lim = status ? 0u : (nm < a.l_value ? nm : a.l_value);
        for (uint32_t i0 = 0; i0 < lim; i0 += kWave) {
            const uint32_t i = i0 + lane;
            uint32_t id = kEmpty;
            float d = 0.0f;
            if (i < lim) {
                const unsigned long long key = key_load(m_keys + i);
                d = from_ordered_bits((uint32_t)(key >> 32));
                id = u32_load(m_ids + (uint32_t)key);
            }
            const bool res = i < lim && id < ix.capacity;
            const uint64_t m = ballot64(res);
            const uint32_t r = written + mbcnt(m);
            if (res && r < a.k) {
                oi[r] = id;
                od[r] = d;
            }
            written += (uint32_t)__popcll(m);
        }
        written = written < a.k ? written : a.k;
        for (uint32_t r = written + lane; r < a.k; r += kWave) {
            oi[r] = kEmpty;
            od[r] = __builtin_inff();
        }
    } else if (a.out_ids && !a.range_ids) {
        uint32_t* oi = a.out_ids + (uint64_t)qi * a.k;
        float* od = a.out_dists + (uint64_t)qi * a.k;
        uint32_t taken = 0;  // multihop: entries that are not rejected start points, first l_value of them
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const uint32_t p = (uint32_t)(s * kWave) + lane;
            const uint32_t id = qid[s] & ~kVisitedBit;
            bool res = p < size && id < ix.capacity;
            if (fmode == DANN_FILTER_MULTIHOP) {
                // best.iter().filter(not a rejected start point).take(l_value) (multihop_filter_search.rs:88-95)
                const bool cnt = p < size && !(id >= ix.capacity && !fmatch(id));
                const uint64_t cm = ballot64(cnt);
                res = res && (taken + mbcnt(cm)) < a.l_value;
                taken += (uint32_t)__popcll(cm);
            }
            const uint64_t m = ballot64(res);
            const uint32_t r = written + mbcnt(m);
            if (res && r < a.k) {
                oi[r] = id;
                od[r] = qd[s];
            }
            written += (uint32_t)__popcll(m);
        }
        written = written < a.k ? written : a.k;
        for (uint32_t r = written + lane; r < a.k; r += kWave) {
            oi[r] = kEmpty;
            od[r] = __builtin_inff();
        }
    }
    if (lane == 0) {
        if (a.stats) {
            dann_search_stats st;
            st.cmps = cmps;
            st.hops = hops;
            // Translate::post_process counts a push only while the buffer still has room afterwards
            // (provider.rs:933-944, search_output_buffer.rs:107-124): k - 1 when the buffer of length k fills
            st.result_count = (!a.range_ids && a.k && written == a.k) ? a.k - 1u : written;
            st.written = written;
            st.status = status;
            a.stats[qi] = st;
        }
        if (status && a.fail_flag) __hip_atomic_store(a.fail_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.rec_n) a.rec_n[qi] = nrec;
        // (the maximum only grows: a plain read that already shows a value >= nrec makes the atomic unnecessary, and a
        // stale smaller one only costs the atomic it would have cost anyway -- atomics on one address are served one
        // after another, and every insert search of a batch ends here: profiles/r05_scan_atomics.txt)
        if (a.rec_max && nrec > __hip_atomic_load(a.rec_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.rec_max, nrec);
        if (a.range_second) a.range_second[qi] = range_second;
    }
}

// ---- persistent server (dann_server_start / dann_search_submit / dann_search_wait) ----------------------------------
__device__ __forceinline__ uint32_t sys_load_u32(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store_u32(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// wave 0: turns the host's publication words into the device-side `avail` counter; leaves on a stop request or
// after idle_timeout_us without a new submission
__device__ void server_dispatch(const ServerView& sv) {
    const uint32_t lane = threadIdx.x;
    unsigned long long avail = __hip_atomic_load(sv.d_avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long last = wall_clock64();
    const long long born = last;
    const long long idle_ticks = (long long)sv.idle_timeout_us * (long long)sv.ticks_per_us;
    const long long resident_ticks = (long long)sv.max_resident_us * (long long)sv.ticks_per_us;
    uint32_t backoff = 1;
    // second bound, should the wall clock not advance: an idle iteration costs at least a PCIe round trip (~1 us)
    uint32_t idle_iters = 0;
    const uint32_t max_idle_iters = sv.idle_timeout_us * 4u + 1024u;
    for (;;) {
        const unsigned long long t = avail + lane;
        const uint32_t expect = server_lap_tag(t, sv.ring_shift);
        const bool ready = (sys_load_u32(sv.h_pub + (uint32_t)(t & (sv.ring - 1u))) >> 20) == expect;
        const uint64_t m = ballot64(ready);
        const uint32_t n = m == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~m);  // published tickets form a prefix
        if (n) {
            avail += n;
            if (lane == 0) __hip_atomic_store(sv.d_avail, avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = wall_clock64();
            backoff = 1;
            idle_iters = 0;
            if (last - born <= resident_ticks) continue;
            // resident for max_resident_us under a steady stream of tickets: leave as on the idle timeout -- the workers
            // serve what has been counted into `avail`, later tickets wait in the ring for the relaunch
        }
        const bool stop = sys_load_u32(sv.h_ctl) != 0u;
        const bool idle = wall_clock64() - last > idle_ticks || ++idle_iters > max_idle_iters;
        const bool old = wall_clock64() - born > resident_ticks;
        if (stop || idle || old) {
            if (lane == 0) {
                sys_store_u32(sv.h_ctl + 1, 1u);  // tells the host to relaunch on the next submission
                __hip_atomic_store(sv.d_stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        __builtin_amdgcn_s_sleep(16);
        if (backoff < 8u) ++backoff;
        for (uint32_t b = 1; b < backoff; ++b) __builtin_amdgcn_s_sleep(64);
    }
}

// One wave per query (grid = nq); LOOP = 1 (PERSIST, dann_set_max_concurrency): `grid` persistent waves take the
// launch's queries one after the other from a shared counter -- a server that keeps N queries in flight is not held
// up by the slowest query of each batch of N (the tail is 2.3x the mean search at N = 1024); LOOP = 2: the waves
// serve the submission ring of dann_server_start until told to stop.  Results do not depend on it.  Separate
// instantiations (plain mode only): the loop around the body costs the one-wave-per-query launch 3-5 % when it is
// compiled into the same kernel.
template <int DT, int OP, bool NORM, int QS, int DIM, int MODE, int LOOP, int TEAM = 1, bool HT16 = false>
#ifndef DANN_SEARCH_KERNEL_ATTR
#define DANN_SEARCH_KERNEL_ATTR
#endif
__global__ __launch_bounds__(kWave * TEAM) DANN_SEARCH_KERNEL_ATTR void beam_search_kernel(SearchArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if constexpr (TEAM > 1) {
        static_assert(LOOP == 0, "teams serve one-shot launches");
        if ((threadIdx.x >> 6) != (uint32_t)kTeamQueueWave) {
            team_helper<DT, OP, NORM, QS, DIM, TEAM>(a, smem);
            return;
        }
        beam_search_one<DT, OP, NORM, QS, DIM, MODE, TEAM>(a, blockIdx.x, smem);
    } else if constexpr (LOOP == 0) {
        beam_search_one<DT, OP, NORM, QS, DIM, MODE, 1, HT16>(a, blockIdx.x, smem);
    } else if constexpr (LOOP == 1) {
        uint32_t slot = blockIdx.x;
        for (;;) {
            uint32_t nxt = 0;  // the ticket for the search after this one is drawn now: its round trip hides behind the search
            if (threadIdx.x == 0) nxt = atomicAdd(a.work_next, 1u);
            beam_search_one<DT, OP, NORM, QS, DIM, MODE, 1, HT16>(a, slot, smem);
            __syncthreads();  // the next query reuses this wave's LDS
            slot = gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
            if (slot >= a.nq) break;
        }
    } else {
        const ServerView& sv = a.srv;
        if (blockIdx.x == 0) {
            server_dispatch(sv);
            return;
        }
        const uint32_t lane = threadIdx.x, w = blockIdx.x - 1u;
        uint8_t* myq = sv.d_q + (size_t)w * sv.qstride;
        for (;;) {
            // ---- draw a ticket, wait until the host has published it
            uint32_t tlo = 0, thi = 0;
            if (lane == 0) {
                const unsigned long long t0 = atomicAdd(sv.d_head, 1ull);
                tlo = (uint32_t)t0;
                thi = (uint32_t)(t0 >> 32);
            }
            tlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)tlo);
            thi = (uint32_t)__builtin_amdgcn_readfirstlane((int)thi);
            const unsigned long long t = ((unsigned long long)thi << 32) | tlo;
            bool leave = false;
            uint32_t naps = 0;
            for (;;) {
                if (__hip_atomic_load(sv.d_avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > t) break;
                if (__hip_atomic_load(sv.d_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    leave = true;  // the ticket stays unserved: the host resets head to avail before it relaunches
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if (naps < 16u) ++naps;
                for (uint32_t b = 0; b < naps; ++b) __builtin_amdgcn_s_sleep(32);
            }
            if (leave) break;
            // the ring entry names the result slot (slots are handed out by the host independently of ticket order, so
            // that a caller who collects its tickets late, or out of order, holds up nobody's submission); taking the
            // entry is acknowledged at once -- the position is free for the next lap long before the search ends
            const uint32_t pos = (uint32_t)(t & (sv.ring - 1u));
            uint32_t entry = 0;
            if (lane == 0) {
                entry = sys_load_u32(sv.h_pub + pos);
                sys_store_u32(sv.h_ack + pos, entry >> 20);  // (the stored value depends on the load: it cannot pass it)
            }
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)entry) & 0xFFFFFu;
            // ---- stage the query: host ring -> this worker's device buffer (system-scope 16-byte loads)
            {
                const uint8_t* src = sv.h_queries + (size_t)slot * sv.qstride;
                for (uint32_t o = lane * 16u; o < sv.qbytes; o += kWave * 16u) {
                    u32x4 v;
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(src + o) : "memory");
                    *reinterpret_cast<u32x4*>(myq + o) = v;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // the search reads the staged query back with plain loads: drop this CU's (stale) L1 copy of the buffer
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            beam_search_one<DT, OP, NORM, QS, DIM, MODE, 1, HT16>(a, w, smem);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- result: this worker's device rows -> the host ring, then the completion word
            {
                const uint32_t* oi = a.out_ids + (size_t)w * a.k;
                const float* od = a.out_dists + (size_t)w * a.k;
                for (uint32_t r = lane; r < a.k; r += kWave) {
                    sys_store_u32(sv.h_res_ids + (size_t)slot * a.k + r, u32_load(oi + r));
                    sys_store_u32(reinterpret_cast<uint32_t*>(sv.h_res_d) + (size_t)slot * a.k + r,
                                  __builtin_bit_cast(uint32_t, f32_load(od + r)));
                }
                if (lane < 5u) {
                    const uint32_t* st = reinterpret_cast<const uint32_t*>(a.stats + w);
                    sys_store_u32(reinterpret_cast<uint32_t*>(sv.h_res_stats + slot) + lane, u32_load(st + lane));
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the result words before the completion word
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) sys_store_u32(sv.h_done + slot, (uint32_t)t + 1u);
            }
            __syncthreads();  // the next query reuses this wave's LDS
        }
    }
}

// what kModePlain assumes (checked by the host for every launch)
inline bool plain_mode(const SearchArgs& a) {
    return !a.filter_mode && a.beam_width == 1 && a.ix.tag_off == 0 && a.ix.max_degree <= (uint32_t)kWave &&
           a.ix.nstart <= (uint32_t)kWave;
}

#ifndef DANN_TEAM_WAVES
#define DANN_TEAM_WAVES 5
#endif
constexpr int kTeam = DANN_TEAM_WAVES;  // wavefronts per query in the latency regime (SearchArgs::team)

template <int DT, int OP, bool NORM, int QS, int DIM, int MODE, int LOOP = 0, int TEAM = 1, bool HT16 = false>
int32_t launch_one(const SearchArgs& a, size_t lds, hipStream_t stream, int* regs_out) {
    if constexpr (MODE == kModePlain && LOOP == 0 && TEAM == 1 && DIM > 0 && QS <= 4 && DT != DT_PQ) {
        if (a.team && !a.srv.ring && !a.grid && !regs_out)
            return launch_one<DT, OP, NORM, QS, DIM, MODE, 0, kTeam>(a, lds, stream, regs_out);
    }
    if constexpr (MODE == kModePlain && LOOP == 0 && TEAM == 1) {
        if (a.srv.ring && !regs_out) {
            if constexpr (QS <= 4) return launch_one<DT, OP, NORM, QS, DIM, MODE, 2>(a, lds, stream, regs_out);
            set_error("the search server supports L + start points <= 256");
            return DANN_EUNSUPPORTED;
        }
        if (a.grid && !regs_out) return launch_one<DT, OP, NORM, QS, DIM, MODE, 1>(a, lds, stream, regs_out);
    }
    if constexpr (MODE == kModePlain && TEAM == 1 && !HT16) {  // 16-bit visited-table entries (SearchArgs::ht16)
        if (a.ht16 && !regs_out) return launch_one<DT, OP, NORM, QS, DIM, MODE, LOOP, TEAM, true>(a, lds, stream, regs_out);
    }
    if (a.ht16 && !HT16 && !regs_out) {
        set_error("internal: 16-bit visited table requested for a kernel that has none");
        return DANN_EINTERNAL;
    }
    auto kern = beam_search_kernel<DT, OP, NORM, QS, DIM, MODE, LOOP, TEAM, HT16>;
    if (regs_out) {  // query only: VGPRs of the instantiation this launch would use
        hipFuncAttributes attr;
        hipError_t e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kern));
        if (e != hipSuccess) return hip_fail(e, "hipFuncGetAttributes");
        *regs_out = attr.numRegs;
        return DANN_OK;
    }
    if (lds > 64 * 1024) {  // raise the dynamic LDS limit of this instantiation once per device (160 KiB per CU)
        static bool raised[64] = {};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !raised[dev]) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
            if (dev >= 0 && dev < 64) raised[dev] = true;
        }
    }
    const uint32_t grid = LOOP == 2 ? a.srv.workers + 1u : LOOP == 1 ? a.grid : a.nq;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWave * TEAM), lds, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "beam_search_kernel launch");
    return DANN_OK;
}

template <int DT, int OP, bool NORM, int DIM, int MODE>
int32_t launch_qs2(const SearchArgs& a, uint32_t qcap, size_t lds, hipStream_t stream, int* regs_out) {
    if (qcap <= 64) return launch_one<DT, OP, NORM, 1, DIM, MODE>(a, lds, stream, regs_out);
    if (qcap <= 128) return launch_one<DT, OP, NORM, 2, DIM, MODE>(a, lds, stream, regs_out);
    if (qcap <= 256) return launch_one<DT, OP, NORM, 4, DIM, MODE>(a, lds, stream, regs_out);
    if (qcap <= 512) return launch_one<DT, OP, NORM, 8, DIM, MODE>(a, lds, stream, regs_out);
    if (qcap <= 1024) return launch_one<DT, OP, NORM, 16, DIM, MODE>(a, lds, stream, regs_out);
    set_error("search list size L + start points = %u exceeds the supported maximum of 1024", qcap);
    return DANN_EUNSUPPORTED;
}

template <int DT, int OP, bool NORM, int DIM>
int32_t launch_qs(const SearchArgs& a, uint32_t qcap, size_t lds, hipStream_t stream, int* regs_out) {
    if (a.filter_mode) return launch_qs2<DT, OP, NORM, 0, kModeFiltered>(a, qcap, lds, stream, regs_out);
    if (plain_mode(a))
        return launch_qs2<DT, OP, NORM, DIM, kModePlain>(a, qcap, lds, stream, regs_out);
    return launch_qs2<DT, OP, NORM, DIM, kModeGeneral>(a, qcap, lds, stream, regs_out);
}

template <int DT>
int32_t launch_dt(const SearchArgs& a, uint32_t qcap, size_t lds, hipStream_t stream, int* regs_out) {
    int op;
    bool norm;
    if (!resolve_metric(a.ix.dtype, a.ix.metric, &op, &norm)) {
        set_error("metric %d is not defined for dtype %d", a.ix.metric, a.ix.dtype);
        return DANN_EUNSUPPORTED;
    }
    if (op == OP_L2) {
        if constexpr (DT == DT_F32 || DT == DT_F16) {
            if (a.ix.dim == 128) return launch_qs<DT, OP_L2, false, 128>(a, qcap, lds, stream, regs_out);
        }
        if constexpr (DT == DT_SQ8) {
            if (norm) {
                if (a.ix.dim == 128) return launch_qs<DT, OP_L2, true, 128>(a, qcap, lds, stream, regs_out);
                return launch_qs<DT, OP_L2, true, 0>(a, qcap, lds, stream, regs_out);
            }
        }
        if constexpr (DT == DT_U8 || DT == DT_I8 || DT == DT_SQ8) {  // 128-byte integer rows (C-int8)
            if (a.ix.dim == 128) return launch_qs<DT, OP_L2, false, 128>(a, qcap, lds, stream, regs_out);
        }
        return launch_qs<DT, OP_L2, false, 0>(a, qcap, lds, stream, regs_out);
    }
    if (op == OP_IP) {
        if constexpr (DT == DT_F32 || DT == DT_F16) {
            if (norm) return launch_qs<DT, OP_IP, true, 0>(a, qcap, lds, stream, regs_out);
        }
        if constexpr (DT == DT_U8 || DT == DT_I8 || DT == DT_SQ8) {
            if (a.ix.dim == 128) return launch_qs<DT, OP_IP, false, 128>(a, qcap, lds, stream, regs_out);
        }
        return launch_qs<DT, OP_IP, false, 0>(a, qcap, lds, stream, regs_out);
    }
    if constexpr (DT == DT_U8 || DT == DT_I8) {
        if (a.ix.dim == 128) return launch_qs<DT, OP_COS, false, 128>(a, qcap, lds, stream, regs_out);
    }
    if constexpr (DT != DT_SQ8 && DT != DT_PQ) return launch_qs<DT, OP_COS, false, 0>(a, qcap, lds, stream, regs_out);
    return DANN_EUNSUPPORTED;
}

// does a team instantiation exist for this launch?  (launch_one: plain mode, a fixed-length kernel -- 128-element rows
// of the metric's specialised form --, at most 256 queue entries, not PQ rows; launch_dt's case analysis)
inline bool team_shape(const SearchArgs& a) {
    int op;
    bool norm;
    const int dt = a.ix.dtype;
    if (!plain_mode(a) || dt == DT_PQ || a.ix.dim != 128u || !resolve_metric(dt, a.ix.metric, &op, &norm)) return false;
    if (std::max(a.l_value + a.ix.nstart, a.qcap_max) > 256u) return false;
    const bool ints = dt == DT_U8 || dt == DT_I8 || dt == DT_SQ8;
    if (op == OP_L2) return true;
    if (op == OP_IP) return ints;  // (float rows: inner product and CosineNormalized run the generic-length kernel)
    return dt == DT_U8 || dt == DT_I8;
}

uint32_t cmax_of(const SearchArgs& a) {
    uint32_t c1 = (a.beam_width * a.ix.max_degree + 63u) & ~63u, c2 = (a.ix.nstart + 63u) & ~63u;
    return c1 > c2 ? c1 : c2;
}
uint32_t qs_of(uint32_t qcap) { return qcap <= 64 ? 1 : qcap <= 128 ? 2 : qcap <= 256 ? 4 : qcap <= 512 ? 8 : 16; }

}  // namespace
}  // namespace dann
