// cost of page-locking a caller's pageable buffer for the duration of one call: hipHostRegister / hipHostUnregister
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main() {
    for (size_t mb : {1, 8, 51}) {
        size_t bytes = mb << 20;
        void* p = aligned_alloc(4096, bytes);
        memset(p, 1, bytes);
        hipFree(0);
        for (int rep = 0; rep < 4; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped);
            auto t1 = std::chrono::steady_clock::now();
            void* d = nullptr;
            hipHostGetDevicePointer(&d, p, 0);
            hipHostUnregister(p);
            auto t2 = std::chrono::steady_clock::now();
            printf("%zu MB rep %d: register %.3f ms (rc %d), unregister %.3f ms\n", mb, rep,
                   std::chrono::duration<double, std::milli>(t1 - t0).count(), (int)e,
                   std::chrono::duration<double, std::milli>(t2 - t1).count());
        }
        free(p);
    }
    return 0;
}
