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
