// Sanitizer run of the two restatements of Rust's select_nth_unstable_by + sort_unstable_by (tests/test_rust_order_host.py):
// the product's walk (diskann_amd/csrc/rust_order.h) over exact-size heap blocks -- pool positions, and a work area of
// 8 bytes per pool slot as the kernels give it -- against the checker's (oracle/rust_unstable_sort.h), 24 000 tied pools.
// Built with -fsanitize=address,undefined: any access outside the blocks, any overflow or misaligned access aborts.
#include "rust_order.h"
#include "rust_unstable_sort.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
struct E { uint32_t id; float d; };
int main() {
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    long cases = 0;
    for (int rep = 0; rep < 6000; ++rep) {
        unsigned n = rep < 200 ? rep : (unsigned)(rnd() % 4097);
        unsigned levels = (unsigned[]){1, 2, 3, 7, 50, 1000000}[rnd() % 6];
        std::vector<float> d(n);
        for (auto& x : d) x = (float)(rnd() % levels);
        if (rep % 7 == 0) for (unsigned i = 0; i < n; ++i) d[i] = (float)(n - i);
        if (rep % 11 == 0) for (unsigned i = 0; i < n; ++i) d[i] = (float)(i / 3);
        unsigned pcap = 1; while (pcap < n) pcap <<= 1;
        std::vector<uint16_t> ord(n ? n : 1);              // exact-size heap blocks: ASan sees any overrun
        std::vector<unsigned char> work(pcap * 8u);
        for (unsigned mx : {n, n / 2 + 1, 750u, 1u}) {
            for (unsigned i = 0; i < n; ++i) ord[i] = (uint16_t)i;
            dann::rust_order::sorted_neighbors(ord.data(), d.data(), n, mx, work.data());
            std::vector<E> v(n);
            for (unsigned i = 0; i < n; ++i) v[i] = {i, d[i]};
            rust_sort::sorted_neighbors(v, mx, [](const E& a, const E& b) { return a.d < b.d; });
            for (size_t i = 0; i < v.size(); ++i) if (v[i].id != ord[i]) { printf("MISMATCH n=%u mx=%u i=%zu\n", n, mx, i); return 1; }
            ++cases;
        }
    }
    printf("ok %ld cases\n", cases);
    return 0;
}
