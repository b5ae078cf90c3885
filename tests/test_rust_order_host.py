"""diskann_amd/csrc/rust_order.h (the product's sequential walk of Rust's select_nth_unstable_by + sort_unstable_by, used
by the prune kernels under dann_set_prune_tie_order(idx, DANN_TIE_RUST)) is plain C++: compiled here for the host and
compared, position by position, with the checker's independent restatement (oracle/rust_unstable_sort.h -- the one the
reference's grid_insert goldens pin) on random pools with few distinct distances.  The GPU side of the same statement is
tests/test_gpu_tie_order.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include "rust_order.h"
#include <cstring>
// the kernels hand the walk the LDS region of the sort keys: 8 bytes per pool slot, pool slots = the next power of two
// at or above the pool's length -- less than kWorkBytes for pools of fewer than 64 slots
extern "C" int ro_sorted_neighbors_lds(const float* d, unsigned P, unsigned max, unsigned short* out) {
    unsigned pcap = 1;
    while (pcap < P) pcap <<= 1;
    const unsigned region = pcap * 8u;
    static unsigned char work[4096 * 8 + 64];
    std::memset(work, 0xA5, sizeof work);
    for (unsigned i = 0; i < P; ++i) out[i] = (unsigned short)i;
    const bool fb = dann::rust_order::sorted_neighbors(out, d, P, max, work);
    for (unsigned i = region; i < sizeof work; ++i)
        if (work[i] != 0xA5) return -1;  // wrote past the region of the sort keys
    return fb ? 1 : 0;
}
extern "C" int ro_sorted_neighbors(const float* d, unsigned P, unsigned max, unsigned short* out) {
    alignas(8) unsigned char work[dann::rust_order::kWorkBytes + 16];
    std::memset(work, 0xA5, sizeof work);
    for (unsigned i = 0; i < P; ++i) out[i] = (unsigned short)i;
    const bool fb = dann::rust_order::sorted_neighbors(out, d, P, max, work);
    for (unsigned i = dann::rust_order::kWorkBytes; i < sizeof work; ++i)
        if (work[i] != 0xA5) return -1;  // wrote past its work area
    return fb ? 1 : 0;
}
// the filtered searches' form: whole 8-byte keys (ordered distance bits << 32 | push index) sorted in place
extern "C" int ro_sort_keys(const float* d, unsigned n, unsigned* out_index) {
    static unsigned long long keys[1 << 16];
    alignas(8) unsigned char work[dann::rust_order::kKeyWorkBytes + 16];
    std::memset(work, 0xA5, sizeof work);
    if (n > (1u << 16)) return -2;
    for (unsigned i = 0; i < n; ++i) {
        unsigned u;
        std::memcpy(&u, d + i, 4);
        const unsigned o = (u >> 31) ? ~u : (u | 0x80000000u);  // ordered_bits (search_kernel_impl.h)
        keys[i] = ((unsigned long long)o << 32) | i;
    }
    dann::rust_order::sort_keys_unstable(keys, n, work);
    for (unsigned i = 0; i < n; ++i) out_index[i] = (unsigned)keys[i];
    for (unsigned i = dann::rust_order::kKeyWorkBytes; i < sizeof work; ++i)
        if (work[i] != 0xA5) return -1;
    return 0;
}
// M. D. McIlroy's adversary ("A killer adversary for quicksort", 1999) against THIS sort: elements are indices, their
// values are decided lazily ("gas" until a comparison forces one to freeze), so that every pivot the algorithm picks
// turns out to be among the smallest values left -- the input that results drives the quicksort through its 2 log2(n)
// bad partitions into heapsort, and the selection through its sixteen rounds into the fallback.  what = 0: sort_unstable,
// 1: select_nth(n, index).  Returns the killer values (a permutation of 0 .. n-1) in out_val.
namespace {
struct Adversary {
    unsigned* val;
    unsigned gas, nsolid = 0, candidate = 0;
    bool less(unsigned x, unsigned y) {
        if (val[x] == gas && val[y] == gas) {
            if (x == candidate) val[x] = nsolid++;
            else val[y] = nsolid++;
        }
        if (val[x] == gas) candidate = x;
        else if (val[y] == gas) candidate = y;
        return val[x] < val[y];
    }
};
struct AdvLess {
    Adversary* a;
    bool operator()(unsigned x, unsigned y) const { return a->less(x, y); }
};
}  // namespace
extern "C" int ro_adversary(unsigned n, int what, unsigned index, unsigned* out_val) {
    static unsigned elems[1 << 16];
    alignas(8) unsigned char work[4096];
    if (n > (1u << 16)) return -2;
    for (unsigned i = 0; i < n; ++i) {
        elems[i] = i;
        out_val[i] = n;  // gas
    }
    Adversary adv{out_val, n};
    if (what == 0 && n > 2) out_val[1] = adv.nsolid++;  // (element 1 is the smallest: the run detection stops after two elements)
    dann::rust_order::SorterT<unsigned, unsigned, AdvLess, 66, 12> s(elems, AdvLess{&adv}, work);
    if (what == 0) s.sort_unstable(0, n);
    else s.select_nth(n, index);
    for (unsigned i = 0; i < n; ++i)
        if (out_val[i] == n) out_val[i] = adv.nsolid++;
    return s.fallback ? 1 : 0;
}
// KeyLess against the float comparison it stands for
extern "C" int ro_key_less(float a, float b) {
    unsigned ua, ub;
    std::memcpy(&ua, &a, 4);
    std::memcpy(&ub, &b, 4);
    const unsigned long long ka = (unsigned long long)((ua >> 31) ? ~ua : (ua | 0x80000000u)) << 32;
    const unsigned long long kb = (unsigned long long)((ub >> 31) ? ~ub : (ub | 0x80000000u)) << 32;
    return dann::rust_order::KeyLess{}(ka, kb) ? 1 : 0;
}
"""


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("rust_order")
    src = d / "harness.cpp"
    src.write_text(HARNESS)
    so = d / "libro.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror",
                           "-I", os.path.join(ROOT, "diskann_amd", "csrc"), str(src), "-o", str(so)])
    lib = C.CDLL(str(so))
    lib.ro_sorted_neighbors.restype = C.c_int
    lib.ro_sorted_neighbors.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    lib.ro_sorted_neighbors_lds.restype = C.c_int
    lib.ro_sorted_neighbors_lds.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    lib.ro_sort_keys.restype = C.c_int
    lib.ro_sort_keys.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
    lib.ro_key_less.restype = C.c_int
    lib.ro_key_less.argtypes = [C.c_float, C.c_float]
    lib.ro_adversary.restype = C.c_int
    lib.ro_adversary.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_void_p]
    return lib


def _product(lib, d, mx):
    d = np.ascontiguousarray(d, np.float32)
    out = np.zeros(max(d.size, 1), np.uint16)
    rc = lib.ro_sorted_neighbors(d.ctypes.data, d.size, mx, out.ctypes.data)
    assert rc >= 0, "rust_order wrote past its work area"
    return out[:min(mx, d.size)].astype(np.uint32)


def test_product_walk_equals_the_checkers_restatement(host_lib):
    rng = np.random.default_rng(17)
    sizes = [0, 1, 2, 3, 8, 9, 13, 16, 17, 18, 20, 21, 22, 31, 32, 33, 34, 63, 64, 65, 71, 72, 127, 128, 129, 255, 300,
             511, 512, 513, 750, 1000, 2047, 2048, 4095, 4096]
    cases = 0
    for n in sizes:
        for levels in (1, 2, 3, 5, 17, 10 ** 6):
            for shape in range(4):
                d = rng.integers(0, levels, n).astype(np.float32)
                if shape == 1:
                    d = np.sort(d)
                elif shape == 2:
                    d = np.sort(d)[::-1].copy()
                elif shape == 3 and n > 4:
                    d[: n // 2] = np.sort(d[: n // 2])  # a long run, then noise
                ids = np.arange(n, dtype=np.uint32)
                for mx in sorted({0, 1, 2, n // 3 + 1, max(n - 1, 0), n, n + 7, 750}):
                    want, _ = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, ids, d, mx)
                    got = _product(host_lib, d, mx)
                    assert np.array_equal(got, want), (n, levels, shape, mx)
                    cases += 1
    assert cases > 5000


def test_specials_compare_like_partial_cmp(host_lib):
    """NaN compares Equal to everything (fast_distance), -0.0 == +0.0: both restatements must walk the same way"""
    rng = np.random.default_rng(23)
    for n in (5, 40, 200):
        for _ in range(20):
            d = rng.integers(-2, 3, n).astype(np.float32)
            d[rng.integers(0, n, max(n // 10, 1))] = np.nan
            d[rng.integers(0, n, max(n // 10, 1))] = -0.0
            d[rng.integers(0, n, max(n // 20, 1))] = np.inf
            ids = np.arange(n, dtype=np.uint32)
            want, _ = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, ids, d, n)
            assert np.array_equal(_product(host_lib, d, n), want)


def test_key_comparison_is_the_float_comparison(host_lib):
    """KeyLess works on the order-preserving bits in integer arithmetic (the float form crashes hipcc 7.2 inside the
    quicksort): it must be `a < b` on f32 -- -0.0 == +0.0, a NaN is less than nothing and nothing is less than it"""
    vals = [0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, -np.nan, 1e-45, -1e-45, 3.4e38, -3.4e38, 1.5, 1.5000001]
    rng = np.random.default_rng(3)
    vals += list(rng.standard_normal(40).astype(np.float32)) + list(rng.integers(0, 2 ** 32, 60, dtype=np.uint64).astype(np.uint32).view(np.float32))
    for a in vals:
        for b in vals:
            want = bool(np.float32(a) < np.float32(b))
            assert bool(host_lib.ro_key_less(np.float32(a), np.float32(b))) == want, (a, b)


def test_key_sort_equals_the_checkers_sort_unstable(host_lib):
    """`sort_unstable_by(fast_distance)` over whole (distance, payload) elements -- the filtered searches' matched list and
    the multihop search's rejected candidates -- : the product's in-place key form against the checker's restatement on
    lists with few distinct distances, runs, NaNs and signed zeros, lengths up to 20 000 (beyond the 16-bit form)"""
    rng = np.random.default_rng(29)
    cases = 0
    for n in [0, 1, 2, 5, 19, 20, 21, 22, 32, 33, 34, 63, 64, 65, 100, 129, 257, 700, 1000, 4097, 20000]:
        for levels in (1, 2, 3, 7, 10 ** 6):
            for shape in range(5):
                d = rng.integers(0, levels, n).astype(np.float32)
                if shape == 1:
                    d = np.sort(d)
                elif shape == 2:
                    d = np.sort(d)[::-1].copy()
                elif shape == 3 and n > 4:
                    d[: n // 2] = np.sort(d[: n // 2])
                elif shape == 4 and n > 4:
                    d[rng.integers(0, n, n // 8 + 1)] = np.nan
                    d[rng.integers(0, n, n // 8 + 1)] = -0.0
                ids = np.arange(n, dtype=np.uint32)
                want, _ = oracle.rust_sort(oracle.RUST_SORT_UNSTABLE, ids, d)
                got = np.zeros(max(n, 1), np.uint32)
                dd = np.ascontiguousarray(d, np.float32)
                assert host_lib.ro_sort_keys(dd.ctypes.data, n, got.ctypes.data) == 0
                assert np.array_equal(got[:n], want), (n, levels, shape)
                cases += 1
    assert cases > 400


def _post_conditions(order, d, keep):
    """what SortedNeighbors::new promises whatever the order of equal keys: the first `keep` entries are distinct pool
    positions, sorted by distance, and no entry left out is closer than the last one kept"""
    d = np.asarray(d, np.float32)
    order = np.asarray(order, np.int64)
    assert len(set(order.tolist())) == len(order)
    dd = d[order]
    assert np.all(dd[:-1] <= dd[1:]) if len(dd) > 1 else True
    if 0 < len(order) < d.size:
        rest = np.setdiff1d(np.arange(d.size), order)
        assert d[rest].min() >= dd[-1]


def test_paths_the_goldens_do_not_reach_selection_loop_descending_runs(host_lib):
    """The reference's grid_insert goldens never run a pool through the selection's partition loop (no pool exceeds
    max_occlusion_size there), strictly descending runs, or the first-minimum case.  Pools of 751 .. 4096 candidates with
    heavy ties, selected down to 750 / 100 / 1 and sorted: both restatements keep the promise of SortedNeighbors::new
    (sorted prefix, nothing closer left out) and agree position by position; the checker's path counters show that the
    loop, both of its partitions, its 16-entry insertion sort, reversed runs and the first-minimum swap all ran."""
    rng = np.random.default_rng(61)
    before = oracle.rust_sort_paths()
    for n in (751, 752, 800, 1023, 1500, 2048, 3000, 4096):
        for levels in (2, 5, 40, 10 ** 6):
            for shape in range(3):
                d = rng.integers(0, levels, n).astype(np.float32)
                if shape == 1:
                    d = np.sort(d)[::-1].copy() + np.arange(n, dtype=np.float32)[::-1] * (levels > 1000)  # strictly descending when tie-free
                elif shape == 2:
                    d[: n // 3] = np.sort(d[: n // 3])[::-1]
                ids = np.arange(n, dtype=np.uint32)
                for mx in (750, 100, 17, 1):
                    want, _ = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, ids, d, mx)
                    got = _product(host_lib, d, mx)
                    _post_conditions(want, d, mx)
                    _post_conditions(got, d, mx)
                    assert np.array_equal(got, want), (n, levels, shape, mx)
    # strictly descending whole slices (sort_unstable reverses them) beyond the insertion-sort length
    for n in (21, 33, 200, 750, 5000):
        d = np.arange(n, 0, -1).astype(np.float32)
        want, _ = oracle.rust_sort(oracle.RUST_SORT_UNSTABLE, np.arange(n, dtype=np.uint32), d)
        got = np.zeros(n, np.uint32)
        assert host_lib.ro_sort_keys(d.ctypes.data, n, got.ctypes.data) == 0
        assert np.array_equal(got, want) and np.array_equal(got, np.arange(n - 1, -1, -1))
    after = oracle.rust_sort_paths()
    for path in ("select_loop", "select_partition_lt", "select_partition_le", "select_insertion_16", "select_min",
                 "run_reversed"):
        assert after[path] > before[path], path


def test_adversarial_inputs_reach_heapsort_and_the_selections_fallback(host_lib):
    """McIlroy's adversary played against the product's own walk yields the inputs on which every pivot is bad: the sort
    falls through its 2 log2(n) levels into heapsort, the selection through its sixteen rounds into the fallback (both
    restatements: a sort of the range instead of core's median_of_medians -- the documented deviation).  On those inputs,
    and on tied versions of them, both restatements return sorted, complete results and agree."""
    before = oracle.rust_sort_paths()
    fb_before = oracle.rust_sort_fallbacks()
    for n in (2000, 4096, 20000):
        val = np.zeros(n, np.uint32)
        assert host_lib.ro_adversary(n, 0, 0, val.ctypes.data) == 0
        assert sorted(val.tolist()) == list(range(n))
        for div in (1, 3, 50):
            d = (val // div).astype(np.float32)
            ids = np.arange(n, dtype=np.uint32)
            want, wd = oracle.rust_sort(oracle.RUST_SORT_UNSTABLE, ids, d)
            assert np.all(wd[:-1] <= wd[1:]) and sorted(want.tolist()) == list(range(n))
            got = np.zeros(n, np.uint32)
            assert host_lib.ro_sort_keys(d.ctypes.data, n, got.ctypes.data) == 0
            assert np.array_equal(got, want), (n, div)
    mid = oracle.rust_sort_paths()
    assert mid["heapsort"] > before["heapsort"]
    # the selection: index in the middle of a 4 000-entry pool (the 16-bit form the prune kernels use)
    n, index = 4000, 2000
    val = np.zeros(n, np.uint32)
    assert host_lib.ro_adversary(n, 1, index, val.ctypes.data) == 1  # the walk itself reached its fallback on the way
    for div in (1, 4):
        d = (val // div).astype(np.float32)
        want, _ = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, np.arange(n, dtype=np.uint32), d, index + 1)
        got = _product(host_lib, d, index + 1)
        _post_conditions(want, d, index + 1)
        _post_conditions(got, d, index + 1)
        assert np.array_equal(got, want), div
    after = oracle.rust_sort_paths()
    assert after["select_fallback"] > mid["select_fallback"] and oracle.rust_sort_fallbacks() > fb_before


def test_the_walk_stays_inside_the_lds_region_the_kernels_give_it(host_lib):
    """prune_sorted_pool / sort_pool_wave pass the region of the sort keys as the work area: 8 bytes per pool slot.  For
    every pool length the walk's merge buffer and stacks stay inside it (the arrays behind it in LDS are the pool's ids
    and distances)"""
    rng = np.random.default_rng(41)
    for n in list(range(1, 140)) + [255, 256, 257, 511, 512, 1000, 2048, 4096]:
        for levels in (1, 2, 4, 10 ** 6):
            for rep in range(3):
                d = np.ascontiguousarray(rng.integers(0, levels, n), np.float32)
                if rep == 1:
                    d = np.sort(d)[::-1].copy()
                out = np.zeros(n, np.uint16)
                for mx in (n, max(n // 2, 1), 750):
                    rc = host_lib.ro_sorted_neighbors_lds(d.ctypes.data, n, mx, out.ctypes.data)
                    assert rc >= 0, (n, levels, rep, mx)
                    want, _ = oracle.rust_sort(oracle.RUST_SORTED_NEIGHBORS, np.arange(n, dtype=np.uint32), d, mx)
                    assert np.array_equal(out[:min(mx, n)].astype(np.uint32), want)


def test_both_restatements_under_address_and_ub_sanitizers(tmp_path):
    """tests/src/rust_order_sanitize.cpp: 24 000 tied pools (lengths 0 .. 4096) through the product's walk on exact-size
    heap blocks and through the checker's restatement, compiled with -fsanitize=address,undefined: clean, and equal"""
    exe = tmp_path / "sanitize"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "diskann_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
           os.path.join(ROOT, "tests", "src", "rust_order_sanitize.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr.lower():
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.startswith("ok 24000 cases"), run.stdout + run.stderr
