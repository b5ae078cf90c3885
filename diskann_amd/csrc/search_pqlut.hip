// pq_search_kernel instantiations (search_pq_impl.h: PQ rows, the query's lookup table in registers) for codes of
// 1 .. 16 chunks (64 table registers); one translation unit per table size so that they compile side by side
#include "search_pq_impl.h"

namespace dann {
int32_t launch_search_pqlut_g1(const SearchArgs& a, size_t lds, hipStream_t stream) { return launch_pq_lut_g<1>(a, lds, stream); }
int32_t launch_search_pqlut(const SearchArgs& a, size_t lds, hipStream_t stream) {
    switch (pq_lut_groups(a.ix.pq_chunks)) {
        case 0:
        case 1: return launch_search_pqlut_g1(a, lds, stream);
        case 2: return launch_search_pqlut_g2(a, lds, stream);
        case 3: return launch_search_pqlut_g3(a, lds, stream);
        case 4: return launch_search_pqlut_g4(a, lds, stream);
    }
    set_error("internal: the register-resident PQ table covers at most %u chunks", kPqLutChunks);
    return DANN_EINTERNAL;
}
}  // namespace dann
