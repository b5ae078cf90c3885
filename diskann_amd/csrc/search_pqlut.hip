// pq_search_kernel instantiations (search_pq_impl.h: PQ rows, the query's lookup table in registers); its own
// translation unit so that it compiles beside the beam_search_kernel instantiations of search_pq.hip
#include "search_pq_impl.h"

namespace dann {
int32_t launch_search_pqlut(const SearchArgs& a, size_t lds, hipStream_t stream) { return launch_pq_lut(a, lds, stream); }
}  // namespace dann
