// pair_search_kernel instantiations (search_pair_impl.h: two queries per wavefront, 128-byte integer rows); its own
// translation unit so that it compiles beside the beam_search_kernel instantiations of search_{u8,i8,sq8}.hip
#include "search_pair_impl.h"

namespace dann {
int32_t launch_search_pair(const SearchArgs& a, size_t lds, hipStream_t stream) {
    switch (a.ix.dtype) {
        case DT_U8: return launch_pair_dt<DT_U8>(a, lds, stream);
        case DT_I8: return launch_pair_dt<DT_I8>(a, lds, stream);
        case DT_SQ8: return launch_pair_dt<DT_SQ8>(a, lds, stream);
    }
    set_error("internal: two queries per wavefront serve 128-byte integer rows");
    return DANN_EINTERNAL;
}
}  // namespace dann
