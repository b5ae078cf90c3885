// The beam search over PQ code rows with the query's lookup table in registers (pq_search_kernel).
//
// Why.  beam_search_kernel<DT_PQ> keeps the lookup table of a query -- chunks x 256 f32, 16 KiB at 16 chunks -- in LDS:
// six queries per CU, one dependent LDS lookup per chunk and candidate, and two dependent memory round trips per hop
// (adjacency row, then 16-byte code rows of which whole 64-byte sectors are fetched).  Measured at 1 M x 16 chunks,
// L = 96 (profiles/r04_final_pq_summary.json): 0.02 of the HBM peak, 4.7 x its algorithmic bytes.
//
// What this kernel does instead (one wavefront per query, plain Knn search, degree <= 64; at most 16 chunks in round 5,
// at most 64 since round 6 -- NG = 1 .. 4 groups of 16 chunks, 64 table registers each: 4 / 2 / 2 / 1 wavefronts per SIMD):
//   * the table lives in 64 vector registers per lane: entry (chunk c, centroid b) in register 4 c + (b >> 6) of lane
//     b & 63.  A lookup is four ds_bpermute_b32 (every lane pulls from lane b & 63 of the chunk's four registers) and a
//     select on b >> 6 -- the LDS crossbar, but no LDS footprint: ~10 KB of LDS per query, and the register budget of
//     four wavefronts per SIMD = 16 queries per CU.  Table entries are added in chunk order in f32, starting from 0.0
//     (pq_dist_lookup_single, fixed_chunk_pq_table.rs:82-100); chunks the index does not have read a zero table
//     (x + 0.0 == x bit for bit: the running sum is never -0.0);
//   * lane i serves neighbour i of the expanded node from the adjacency row to the merge: no compaction, no candidate
//     buffer in LDS;
//   * optional packed layout (dann_pq_pack_neighbors; IndexView::pq_pack): a node's row holds its adjacency list AND its
//     neighbours' code rows, 64-byte aligned -- one contiguous read per hop, requested a hop ahead for the predicted
//     next node, instead of 1 + degree dependent gathers of 64-byte sectors.
//
// What stays exactly as in beam_search_one (search_kernel_impl.h), the statement of the algorithm: queue rule and merge
// (queue.rs:130-171; the three merge paths are the ones of beam_search_one, candidates in lane = adjacency order), pop
// order (queue.rs:297-313), adjacency clamp (neighbors.rs:146-148), exact visited set (16-bit table in LDS, frozen at
// 75 % load, continued in a spill table), counters and the result rule (provider.rs:933-944).  A query that exhausts
// table and spill pool reports DANN_EOVERFLOW and is re-run through beam_search_kernel by search_with_retry.
#pragma once
#include "search_pair_impl.h"

namespace dann {
namespace {

constexpr uint32_t kPqLutChunks = 64;  // chunks the register-resident table covers at most (4 registers each)
// groups of 16 chunks = code-row dwordx4 loads = 64 table registers: the instantiation a chunk count takes
__host__ __device__ inline uint32_t pq_lut_groups(uint32_t chunks) { return (chunks + 15u) / 16u; }

// LDS of one query: the queue image ((id, distance) pairs: the merge scatters the register-resident queue here and
// reloads it), two 64-word buffers of the merge's slow path, the visited table
struct PqLds {
    uint32_t stage_off, cbi_off, cbd_off, ht_off, total;
};
// (ov_words: the overflow table of the 16-bit table directly behind it -- ov_insert, search_pair_impl.h)
__host__ __device__ inline PqLds pq_lds_layout(uint32_t qs, uint32_t ht_words, uint32_t ov_words) {
    PqLds l;
    l.stage_off = 0;
    l.cbi_off = qs * 64u * 8u;
    l.cbd_off = l.cbi_off + 256u;
    l.ht_off = l.cbd_off + 256u;
    l.total = l.ht_off + (ht_words + ov_words) * 4u;
    return l;
}

// one table entry: populate_chunk_distances_impl (fixed_chunk_pq_table.rs:152-192) -- the arithmetic of the LDS form
// (simd_op_seq: the reference's f32 kernel, four accumulators of eight lanes, then (a0 + a1) + (a2 + a3) and the sum
// tree).  Chunks of exactly eight elements -- 128 dimensions in 16 chunks -- are one block of accumulator 0: the other
// three accumulators stay +0.0, so (a0 + a1) + (a2 + a3) = (a0 + 0.0) + 0.0.  For L2, a0 = fma(c, c, 0.0) is never -0.0
// and both additions are identities; for the inner product a0 + 0.0 turns a -0.0 product into +0.0 and the second
// addition is an identity.  Same bits as simd_op_seq, a quarter of its instructions.
typedef float pq_f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int OP>
__device__ __forceinline__ float pq_lut_entry(const float* q, const float* pivot, uint32_t len) {
    if (len == 8u) {
        const pq_f4u qa = *reinterpret_cast<const pq_f4u*>(q), qb = *reinterpret_cast<const pq_f4u*>(q + 4);
        const pq_f4u pa = *reinterpret_cast<const pq_f4u*>(pivot), pb = *reinterpret_cast<const pq_f4u*>(pivot + 4);
        const float x[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        const float y[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
        float s[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            if (OP == OP_L2) {
                const float c = x[l] - y[l];
                s[l] = __builtin_fmaf(c, c, 0.0f);
            } else {
                s[l] = __builtin_fmaf(x[l], y[l], 0.0f) + 0.0f;
            }
        }
        const float raw = ((s[0] + s[4]) + (s[2] + s[6])) + ((s[1] + s[5]) + (s[3] + s[7]));
        return (OP == OP_L2) ? raw : -raw;
    }
    const float raw = simd_op_seq<OP == OP_L2>(q, pivot, len);
    return (OP == OP_L2) ? raw : -raw;
}

// Four chunks = one dword of a code row: accum += table[c][byte c of w], c = 0 .. 3, in this order.  Written as one
// block of instructions: the compiler's own schedule either keeps all 64 permute results of a row alive (64 registers
// the table needs) or waits for every chunk's four permutes before it issues the next four.  Here the sixteen permutes of
// the dword go out back to back and are consumed as they return (LDS instructions of one wave return in order: the
// counter tells how many are still out); the select on centroid >> 6 is three bit-field inserts under all-ones /
// all-zeros masks (v_bfe_i32 of bit 6 and bit 7) -- 7 vector instructions per chunk, no compare, no condition-code hazard.
__device__ __forceinline__ void pq_lut_word(float& accum, uint32_t w, float t0, float t1, float t2, float t3, float t4,
                                            float t5, float t6, float t7, float t8, float t9, float t10, float t11,
                                            float t12, float t13, float t14, float t15) {
    uint32_t a0, a1, a2, a3, m6, m7, r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
    const uint32_t two = 2u;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshlrev_b32_sdwa %[a0], %[two], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %[a1], %[two], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %[a2], %[two], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %[a3], %[two], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "ds_bpermute_b32 %[r0], %[a0], %[t0]\n\t"
        "ds_bpermute_b32 %[r1], %[a0], %[t1]\n\t"
        "ds_bpermute_b32 %[r2], %[a0], %[t2]\n\t"
        "ds_bpermute_b32 %[r3], %[a0], %[t3]\n\t"
        "ds_bpermute_b32 %[r4], %[a1], %[t4]\n\t"
        "ds_bpermute_b32 %[r5], %[a1], %[t5]\n\t"
        "ds_bpermute_b32 %[r6], %[a1], %[t6]\n\t"
        "ds_bpermute_b32 %[r7], %[a1], %[t7]\n\t"
        "ds_bpermute_b32 %[r8], %[a2], %[t8]\n\t"
        "ds_bpermute_b32 %[r9], %[a2], %[t9]\n\t"
        "ds_bpermute_b32 %[r10], %[a2], %[t10]\n\t"
        "ds_bpermute_b32 %[r11], %[a2], %[t11]\n\t"
        "ds_bpermute_b32 %[r12], %[a3], %[t12]\n\t"
        "ds_bpermute_b32 %[r13], %[a3], %[t13]\n\t"
        "ds_bpermute_b32 %[r14], %[a3], %[t14]\n\t"
        "ds_bpermute_b32 %[r15], %[a3], %[t15]\n\t"
        "v_bfe_i32 %[m6], %[w], 6, 1\n\t"
        "v_bfe_i32 %[m7], %[w], 7, 1\n\t"
        "s_waitcnt lgkmcnt(12)\n\t"
        "v_bfi_b32 %[r0], %[m6], %[r1], %[r0]\n\t"
        "v_bfi_b32 %[r2], %[m6], %[r3], %[r2]\n\t"
        "v_bfi_b32 %[r0], %[m7], %[r2], %[r0]\n\t"
        "v_add_f32 %[acc], %[acc], %[r0]\n\t"
        "v_bfe_i32 %[m6], %[w], 14, 1\n\t"
        "v_bfe_i32 %[m7], %[w], 15, 1\n\t"
        "s_waitcnt lgkmcnt(8)\n\t"
        "v_bfi_b32 %[r4], %[m6], %[r5], %[r4]\n\t"
        "v_bfi_b32 %[r6], %[m6], %[r7], %[r6]\n\t"
        "v_bfi_b32 %[r4], %[m7], %[r6], %[r4]\n\t"
        "v_add_f32 %[acc], %[acc], %[r4]\n\t"
        "v_bfe_i32 %[m6], %[w], 22, 1\n\t"
        "v_bfe_i32 %[m7], %[w], 23, 1\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_bfi_b32 %[r8], %[m6], %[r9], %[r8]\n\t"
        "v_bfi_b32 %[r10], %[m6], %[r11], %[r10]\n\t"
        "v_bfi_b32 %[r8], %[m7], %[r10], %[r8]\n\t"
        "v_add_f32 %[acc], %[acc], %[r8]\n\t"
        "v_bfe_i32 %[m6], %[w], 30, 1\n\t"
        "v_bfe_i32 %[m7], %[w], 31, 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_bfi_b32 %[r12], %[m6], %[r13], %[r12]\n\t"
        "v_bfi_b32 %[r14], %[m6], %[r15], %[r14]\n\t"
        "v_bfi_b32 %[r12], %[m7], %[r14], %[r12]\n\t"
        "v_add_f32 %[acc], %[acc], %[r12]"
        : [acc] "+v"(accum), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [m6] "=&v"(m6), [m7] "=&v"(m7),
          [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), [r6] "=&v"(r6),
          [r7] "=&v"(r7), [r8] "=&v"(r8), [r9] "=&v"(r9), [r10] "=&v"(r10), [r11] "=&v"(r11), [r12] "=&v"(r12),
          [r13] "=&v"(r13), [r14] "=&v"(r14), [r15] "=&v"(r15)
        : [w] "v"(w), [two] "v"(two), [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3), [t4] "v"(t4), [t5] "v"(t5),
          [t6] "v"(t6), [t7] "v"(t7), [t8] "v"(t8), [t9] "v"(t9), [t10] "v"(t10), [t11] "v"(t11), [t12] "v"(t12),
          [t13] "v"(t13), [t14] "v"(t14), [t15] "v"(t15));
}

// sum over the 16 NG chunk tables of one code row of 16 NG bytes (chunk order, f32, from 0.0).  Every lane must be active:
// a permute delivers zero from a source lane that is switched off.
template <int NG>
__device__ __forceinline__ float pq_lut_sum(const float (&lut)[64 * NG], const uint4 (&code)[NG]) {
    float accum = 0.0f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float* t = lut + 64 * g;
        const uint4 w = code[g];
        pq_lut_word(accum, w.x, t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11], t[12], t[13], t[14],
                    t[15]);
        pq_lut_word(accum, w.y, t[16], t[17], t[18], t[19], t[20], t[21], t[22], t[23], t[24], t[25], t[26], t[27], t[28],
                    t[29], t[30], t[31]);
        pq_lut_word(accum, w.z, t[32], t[33], t[34], t[35], t[36], t[37], t[38], t[39], t[40], t[41], t[42], t[43], t[44],
                    t[45], t[46], t[47]);
        pq_lut_word(accum, w.w, t[48], t[49], t[50], t[51], t[52], t[53], t[54], t[55], t[56], t[57], t[58], t[59], t[60],
                    t[61], t[62], t[63]);
    }
    return accum;
}

template <int OP, int QS, bool PACK, int NG>
__device__ __forceinline__ void pq_search_body(const SearchArgs& a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const IndexView& ix = a.ix;
    const uint32_t lane = threadIdx.x;
    const uint32_t qi = a.qmap ? a.qmap[blockIdx.x] : blockIdx.x;
    const uint32_t R = ix.max_degree, ns = ix.nstart, qcap = a.l_value + ns;  // qcap <= 64 QS
    constexpr uint32_t kOverflow = (uint32_t)(-DANN_EOVERFLOW);
    constexpr uint32_t QCAPP = QS * kWave;

    const PqLds L = pq_lds_layout(QS, a.ht_entries, a.ht_ov);
    uint2* const stage = reinterpret_cast<uint2*>(smem + L.stage_off);
    uint32_t* const cbi = reinterpret_cast<uint32_t*>(smem + L.cbi_off);
    float* const cbd = reinterpret_cast<float*>(smem + L.cbd_off);
    uint32_t* const ht = reinterpret_cast<uint32_t*>(smem + L.ht_off);
    auto stage_dist = [&](uint32_t p) -> float { return __builtin_bit_cast(float, stage[p].y); };
    const Ht16 h16 = ht16_of(a);
    const uint32_t ht_limit = a.ht_open;  // ids the open table takes (75 % of its entries by default)
    uint32_t* const ov = ht + a.ht_entries;  // overflow table (ids whose probes are all taken), wiped with the table
    const uint32_t ov_mask = a.ht_ov - 1u, ov_limit = a.ht_ov ? a.ht_ov - 1u : 0u;
    uint32_t ovc = 0;
    {
        const u32x4 e4 = {kEmpty, kEmpty, kEmpty, kEmpty};
        for (uint32_t i = lane * 4u; i < a.ht_entries + a.ht_ov; i += kWave * 4u) *reinterpret_cast<u32x4*>(ht + i) = e4;
    }

    // ---- the query's lookup table, into registers: lut[4 c + j] of lane l = entry (chunk c, centroid 64 j + l).  One
    // chunk per trip of a rolled loop (four entries per lane), the array rotated by four registers per trip: register
    // indices stay compile-time constants, the entry arithmetic is instantiated four times, not 64.
    constexpr int NLUT = 64 * NG;
    float lut[NLUT];
#pragma unroll
    for (int k = 0; k < NLUT; ++k) lut[k] = 0.0f;
    {
        const float* q = reinterpret_cast<const float*>(a.queries) + (uint64_t)qi * ix.dim;
#pragma nounroll
        for (uint32_t c = 0; c < 16u * NG; ++c) {
            float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (c < ix.pq_chunks) {
                const uint32_t s0 = ix.pq_offsets[c], e0 = ix.pq_offsets[c + 1];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[j] = pq_lut_entry<OP>(q + s0, ix.pq_pivots + (uint64_t)(64u * j + lane) * ix.dim + s0, e0 - s0);
            }
#pragma unroll
            for (int k = 0; k < NLUT - 4; ++k) lut[k] = lut[k + 4];
#pragma unroll
            for (int j = 0; j < 4; ++j) lut[NLUT - 4 + j] = v[j];
        }
    }
    __syncthreads();  // the table is wiped

    // ---- queue state: entry p at lane p % 64, slot p / 64 (registers); everything else is wave-uniform ----------------
    uint32_t qid[QS];
    float qd[QS];
#pragma unroll
    for (int s = 0; s < QS; ++s) {
        qid[s] = kEmpty;
        qd[s] = 0.0f;
    }
    uint32_t size = 0, cmps = 0, hops = 0, htc = 0, spc = 0, status = 0;
    bool open = true;
    uint32_t* spill = nullptr;
    const uint32_t spill_size = 1u << a.spill_bits, spill_mask = spill_size - 1u, spill_shift = 32u - a.spill_bits;
    const uint32_t spill_limit = spill_size - (spill_size >> 2);
    auto claim_spill = [&]() {
        if (spill) return;
        uint32_t slice = kEmpty;
        if (a.spill) {
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

    // distance of queue entry p (wave-uniform), from the registers
    auto queue_dist = [&](uint32_t p) -> float {
        float r = 0.0f;
#pragma unroll
        for (int s = 0; s < QS; ++s)
            if ((p >> 6) == (uint32_t)s)
                r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qd[s]), (int)(p & 63u)));
        return r;
    };
    // ---- merge of one hop's candidates (lane j holds candidate j, `has` = it exists) into the queue: beam_search_one's
    // merge_regs (see there for the rule and its proof), three paths by the number of survivors
    bool stage_stale = false;  // the LDS image of the queue is behind the registers (only the slow path reads it)
    auto merge = [&](bool has, float nd, uint32_t nid) {
        bool nvalid = has && !(nd != nd);  // NaN distances are ignored (queue.rs:131-134)
        if (size == qcap && qcap > 0) nvalid = nvalid && !(queue_dist(size - 1) < nd);
        const uint64_t km = ballot64(nvalid);
        const uint32_t nv = (uint32_t)__popcll(km);
        if (nv == 0) return;
        if (nv <= kSeqInsert) {
            // a few survivors (the steady state of a full queue): the sequential inserts themselves, in emission order,
            // on the register-resident queue -- lower bound by ballot count, "move the tail up by one" by DPP wave shift
            for (uint64_t mm = km; mm; mm &= mm - 1) {
                const int j = __builtin_ctzll(mm);
                const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nd), j));
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
            return;
        }
        uint32_t shift[QS];
        uint32_t pos_new = 0;
        if (nv <= kRegMerge) {
            // one pass over the survivors: rank among the survivors, lower bound in the queue (ballot count per slot),
            // and for every queue entry the number of survivors that go in front of it
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
                __syncthreads();
            }
            // compact the survivors, emission order preserved
            const uint32_t cj = mbcnt(km);
            if (nvalid) {
                cbd[cj] = nd;
                cbi[cj] = nid;
            }
            __syncthreads();
            has = lane < nv;
            nd = has ? cbd[lane] : 0.0f;
            nid = has ? cbi[lane] : kEmpty;
            // rank among the survivors
            uint32_t before = 0;
            for (uint32_t jj = 0; jj < nv; ++jj) {
                const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nd), jj));
                before += ((dj < nd) | ((dj == nd) & (jj > lane))) ? 1u : 0u;
            }
            __syncthreads();
            float* const snew = reinterpret_cast<float*>(cbi);  // (cbi's content is in registers by now)
            if (has) snew[before] = nd;
            __syncthreads();
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
            __syncthreads();  // (the image is read before the scatter below rewrites it)
        }
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
        __syncthreads();
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

    // the row of `node`: length (every lane), neighbour `lane`, and -- packed layout -- that neighbour's code row
    const uint32_t jn = lane < R ? lane : R - 1u;
    auto fetch = [&](uint32_t node, uint32_t& lenv, uint32_t& idv, uint4 (&codev)[NG]) {
        if constexpr (PACK) {
            const uint8_t* prow = ix.pq_pack + (uint64_t)node * ix.pq_pack_stride;
            lenv = *reinterpret_cast<const uint32_t*>(prow);
            idv = reinterpret_cast<const uint32_t*>(prow)[1u + jn];
#pragma unroll
            for (int g = 0; g < NG; ++g)
                codev[g] = *reinterpret_cast<const uint4*>(prow + ix.pq_pack_codes + 16u * NG * jn + 16u * g);
        } else {
            const uint32_t* arow = ix.adj + (uint64_t)node * ix.adj_stride;
            lenv = arow[0];
            idv = arow[1u + jn];
        }
    };
    auto code_row = [&](uint32_t id, uint4 (&codev)[NG]) {  // (PQ rows: 16 NG bytes at a stride of at least that)
#pragma unroll
        for (int g = 0; g < NG; ++g) codev[g] = *reinterpret_cast<const uint4*>(ix.rows + (uint64_t)id * ix.row_stride + 16u * g);
    };

    // ---- start points: frozen slots [capacity, capacity + nstart) (index.rs:1950-1958), the candidates of "hop 0" ----
    {
        const bool on = lane < ns;
        const uint32_t id = ix.capacity + (on ? lane : 0u);
        if (ballot64(ht16_insert_flat(ht, h16, id, on, cbi + lane) == 2u)) status = kOverflow;  // (a table of >= 128 slots: never)
        htc = ns;
        uint4 w[NG];
        code_row(id, w);
        const float d = pq_lut_sum<NG>(lut, w);
        cmps = ns;
        merge(on, d, id);
    }

    uint32_t pfn = kEmpty, pf_len = 0, pf_id = kEmpty;  // the row requested ahead: its node (kEmpty: none)
    uint4 pf_code[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) pf_code[g] = make_uint4(0u, 0u, 0u, 0u);
    while (!status) {
        // ---- pop: the closest unexpanded entry (queue.rs:297-313) and the one after it, the node the next hop expands
        // unless a new candidate gets in front of it
        uint32_t node = kEmpty, next = kEmpty;
        {
            uint32_t got = 0;
#pragma unroll
            for (int s = 0; s < QS; ++s) {
                if (got >= 2u) continue;
                uint64_t m = ballot64(((uint32_t)(s * kWave) + lane < size) && !(qid[s] & kVisitedBit));
                if (got == 0u && m) {
                    const int l = __builtin_ctzll(m);
                    node = (uint32_t)__builtin_amdgcn_readlane((int)qid[s], l);
                    if ((int)lane == l) qid[s] |= kVisitedBit;
                    got = 1;
                    m &= m - 1;
                }
                if (got == 1u && m) {
                    next = (uint32_t)__builtin_amdgcn_readlane((int)qid[s], __builtin_ctzll(m));
                    got = 2;
                }
            }
            if (!got) break;
        }
        ++hops;

        // ---- the node's row: requested a hop ahead when the prediction held; then the request for the predicted next one
        uint32_t lenv, idv;
        uint4 codev[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) codev[g] = make_uint4(0u, 0u, 0u, 0u);
        if (node != pfn) {
            fetch(node, lenv, idv, codev);
        } else {
            lenv = pf_len;
            idv = pf_id;
#pragma unroll
            for (int g = 0; g < NG; ++g) codev[g] = pf_code[g];
        }
        pfn = next;
        fetch(next != kEmpty ? next : 0u, pf_len, pf_id, pf_code);
        uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)lenv);
        len = len < R ? len : R;  // Neighbors::get clamps (neighbors.rs:146-148)

        // ---- visited filter.  The open table takes ids up to 75 % of its slots; then it is frozen and new ids go to a
        // spill table in global memory
        if (open && (htc + len > ht_limit || (a.ht_ov && ovc + len > ov_limit))) {
            open = false;
            claim_spill();
        }
        if (!open && (!spill || spc + len > spill_limit)) {
            status = kOverflow;
            break;
        }
        const bool inb = lane < len;
        const uint32_t id = inb ? idv : kEmpty;
        const bool act = inb && id < ix.nslots;  // (the 16-bit table holds ids below 2^m; an id beyond the index is never a candidate)
        bool isnew = false;
        if (open) {
            const uint32_t r = ht16_insert_flat(ht, h16, id, act, cbi + lane);  // (cbi: idle outside the merge)
            isnew = r == 1u;
            if (ballot64(r == 2u)) {  // (rare) no slot among an id's probes
                if (a.ht_ov) {  // the overflow table takes it (room for this hop's ids: checked above)
                    if (r == 2u) isnew = ov_insert(ov, ov_mask, id);
                    ovc += (uint32_t)__popcll(ballot64(r == 2u && isnew));
                } else {  // the table is frozen, the id goes to the spill table
                    open = false;
                    claim_spill();
                    if (!spill) {
                        status = kOverflow;
                        break;
                    }
                    if (r == 2u) isnew = spill_insert(spill, spill_mask, spill_shift, id);
                }
            }
        } else if (act) {
            isnew = !ht16_contains(ht, h16, id) && !(a.ht_ov && ov_contains(ov, ov_mask, id)) &&
                    spill_insert(spill, spill_mask, spill_shift, id);
        }
        const uint32_t nnew = (uint32_t)__popcll(ballot64(isnew));
        if (open) htc += nnew;
        else spc += nnew;
        cmps += nnew;

        // ---- distances: the new neighbours' code rows (already here in the packed layout), sixteen table lookups each
        if constexpr (!PACK) code_row(isnew ? id : 0u, codev);
        const float d = pq_lut_sum<NG>(lut, codev);
        merge(isnew, d, id);
    }

    // ---- the spill table goes back clean -------------------------------------------------------------------------------
    if (spill) {
        __syncthreads();
        spill_wipe(spill, spill_size, lane);
        __syncthreads();
        if (lane == 0) atomicExch(a.spill_next + 16 + (uint32_t)((spill - a.spill) >> a.spill_bits), 0u);
    }
    // ---- results: best entries in order, start points dropped (provider.rs:933-944) -----------------------------------
    uint32_t written = 0;
    if (a.out_ids) {
        uint32_t* oi = a.out_ids + (uint64_t)qi * a.k;
        float* od = a.out_dists + (uint64_t)qi * a.k;
#pragma unroll
        for (int s = 0; s < QS; ++s) {
            const uint32_t p = (uint32_t)(s * kWave) + lane;
            const uint32_t id = qid[s] & ~kVisitedBit;
            const bool res = p < size && id < ix.capacity;
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
            st.result_count = (a.k && written == a.k) ? a.k - 1u : written;
            st.written = written;
            st.status = status;
            a.stats[qi] = st;
        }
        if (status && a.fail_flag) __hip_atomic_store(a.fail_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The kernels: one per table size, compiled for the wavefronts per SIMD its registers allow (512 per SIMD lane): 64
// table registers + ~64 of state at four wavefronts, 128 / 192 + state at two, 256 + state at one (the table's upper half
// in accumulation registers, moved through v_accvgpr_read in front of its permutes).
#ifndef DANN_PQ_KERNEL_ATTR
#define DANN_PQ_KERNEL_ATTR(W) __attribute__((amdgpu_waves_per_eu(W, W)))
#endif
template <int OP, int QS, bool PACK>
__global__ __launch_bounds__(kWave) DANN_PQ_KERNEL_ATTR(4) void pq_search_kernel(SearchArgs a) {
    pq_search_body<OP, QS, PACK, 1>(a);
}
template <int OP, int QS, bool PACK>
__global__ __launch_bounds__(kWave) DANN_PQ_KERNEL_ATTR(2) void pq_search_kernel_g2(SearchArgs a) {
    pq_search_body<OP, QS, PACK, 2>(a);
}
template <int OP, int QS, bool PACK>
__global__ __launch_bounds__(kWave) DANN_PQ_KERNEL_ATTR(2) void pq_search_kernel_g3(SearchArgs a) {
    pq_search_body<OP, QS, PACK, 3>(a);
}
template <int OP, int QS, bool PACK>
__global__ __launch_bounds__(kWave) DANN_PQ_KERNEL_ATTR(1) void pq_search_kernel_g4(SearchArgs a) {
    pq_search_body<OP, QS, PACK, 4>(a);
}

// what the kernel serves (host side; the table geometry is checked by the caller)
inline bool pq_lut_shape(const SearchArgs& a) {
    return a.ix.dtype == DT_PQ && plain_mode(a) && !a.team && !a.grid && !a.srv.ring && !a.rec_ids && !a.range_ids &&
           !a.qslots && a.out_ids && a.ix.pq_chunks <= kPqLutChunks && a.ix.row_stride % 16u == 0u &&
           a.ix.row_stride >= 16u * pq_lut_groups(a.ix.pq_chunks) &&
           std::max(a.l_value + a.ix.nstart, a.qcap_max) <= 4u * (uint32_t)kWave && a.ix.nstart >= 1u &&
           (a.ix.metric == M_L2 || a.ix.metric == M_IP);
}
inline uint32_t pq_lut_qs(const SearchArgs& a) {
    const uint32_t q = a.l_value + a.ix.nstart;
    return q <= 64u ? 1u : q <= 128u ? 2u : 4u;
}
// queries per CU the registers of a table size allow
inline uint32_t pq_lut_waves_per_cu(uint32_t chunks) {
    const uint32_t g = pq_lut_groups(chunks);
    return g <= 1u ? 16u : g <= 3u ? 8u : 4u;
}

template <int NG>
inline int32_t launch_pq_lut_g(const SearchArgs& a, size_t lds, hipStream_t stream) {
    auto go = [&](auto kern) -> int32_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               160 * 1024);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        }
        hipLaunchKernelGGL(kern, dim3(a.nq), dim3(kWave), lds, stream, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "pq_search_kernel launch");
        return DANN_OK;
    };
    const bool l2 = a.ix.metric == M_L2, pack = a.ix.pq_pack != nullptr;
    const uint32_t qs = pq_lut_qs(a);
#define DANN_PQ_K(OP, QS, PK)                                          \
    [&]() -> int32_t {                                                 \
        if constexpr (NG == 1) return go(pq_search_kernel<OP, QS, PK>); \
        else if constexpr (NG == 2) return go(pq_search_kernel_g2<OP, QS, PK>); \
        else if constexpr (NG == 3) return go(pq_search_kernel_g3<OP, QS, PK>); \
        else return go(pq_search_kernel_g4<OP, QS, PK>);               \
    }()
#define DANN_PQ_GO(OP, QS) return pack ? DANN_PQ_K(OP, QS, true) : DANN_PQ_K(OP, QS, false)
    if (l2) {
        if (qs == 1) DANN_PQ_GO(OP_L2, 1);
        if (qs == 2) DANN_PQ_GO(OP_L2, 2);
        DANN_PQ_GO(OP_L2, 4);
    }
    if (qs == 1) DANN_PQ_GO(OP_IP, 1);
    if (qs == 2) DANN_PQ_GO(OP_IP, 2);
    DANN_PQ_GO(OP_IP, 4);
#undef DANN_PQ_GO
#undef DANN_PQ_K
}

}  // namespace
}  // namespace dann
