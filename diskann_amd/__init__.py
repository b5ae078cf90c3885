"""diskann_amd: MI355X-native batched distance / beam-search / RobustPrune path behind the
surface of the reference's diskann-inmem provider (include/dann.h)."""
from ._ffi import (F32, F16, U8, I8, SQ8, PQ, COSINE, INNER_PRODUCT, L2, COSINE_NORMALIZED, IBC_ALL, IBC_NONE, TIE_POSITION, TIE_RUST, BUILD_MFMA_BACKEDGE, BUILD_MFMA_POOL, BUILD_ROW_KERNEL_ONLY, BuildConfig,
                   Config, DannError, SearchStats, lib)
from .provider import FILTER_INLINE, FILTER_MULTIHOP, Knn, Provider, build_config, NP_DTYPE, STATS_DTYPE, sq8_compress, sq8_train, pq_build_lut, pq_scan, pq_compress, pq_lloyds, pq_kmeanspp, pq_train, pq_rolling_sum_stats

from .sharding import Comm, MultiProvider

__all__ = ["Comm", "MultiProvider", "F32", "F16", "U8", "I8", "SQ8", "PQ", "sq8_compress", "sq8_train", "pq_build_lut", "pq_scan", "pq_compress", "pq_lloyds", "pq_kmeanspp", "pq_train", "pq_rolling_sum_stats", "COSINE", "INNER_PRODUCT", "L2", "COSINE_NORMALIZED", "IBC_ALL", "IBC_NONE", "TIE_POSITION", "TIE_RUST", "BUILD_MFMA_BACKEDGE", "BUILD_MFMA_POOL", "BUILD_ROW_KERNEL_ONLY",
           "BuildConfig", "Config", "DannError", "SearchStats", "lib", "Knn", "Provider", "build_config", "NP_DTYPE",
           "STATS_DTYPE", "FILTER_INLINE", "FILTER_MULTIHOP"]
