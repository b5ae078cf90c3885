// pq_search_kernel instantiations (search_pq_impl.h: PQ rows, the query's lookup table in registers) for codes of
// 49 .. 64 chunks (256 table registers); one translation unit per table size so that they compile side by side
#include "search_pq_impl.h"

namespace dann {
int32_t launch_search_pqlut_g4(const SearchArgs& a, size_t lds, hipStream_t stream) { return launch_pq_lut_g<4>(a, lds, stream); }
}  // namespace dann
