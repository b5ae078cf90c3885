// beam_search_kernel instantiations for DT_SQ8 rows (see search_kernel_impl.h; split per row type so the
// translation units compile in parallel)
#include "search_kernel_impl.h"

namespace dann {
int32_t launch_search_sq8(const SearchArgs& a, uint32_t qcap, size_t lds, hipStream_t stream, int* regs_out) {
    return launch_dt<DT_SQ8>(a, qcap, lds, stream, regs_out);
}
}  // namespace dann
