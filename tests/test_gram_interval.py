"""The MFMA prunes decide `d_ik / d_jk > alpha` from Gram-matrix distances d' whenever the error interval
[d' - E, d' + E], E = c1 (|x|^2 + |y|^2) + c2 |d'| (csrc/build_kernels.hip: gram_c1_chained, gram_c2_for_dim,
`first_exceed`), decides it, and re-evaluate the pair with the bit-exact row kernel otherwise.  The graph is only
identical to the reference's if E really bounds |d' - d_ref|.  This test evaluates d' with the oracle's restatement of
the kernel's Gram arithmetic (orc_gram_chain: one f32 FMA chain per entry -- checked bit for bit against the MFMA kernel
in tests/test_gpu_build.py) and d_ref with the reference's pair kernel, on pairs built to cancel (near-duplicates far
from the origin, mixed scales), and checks the bound.  CPU only."""
import numpy as np
import pytest

import oracle


def _sets(rng, dim):
    base = rng.normal(0, 1, (1, dim)).astype(np.float32)
    yield "gaussian", rng.normal(0, 1, (48, dim)).astype(np.float32)
    yield "offset cluster", (base * 30 + rng.normal(0, 0.05, (48, dim))).astype(np.float32)   # |x|^2 >> d
    yield "near duplicates", (base * 5 + rng.normal(0, 1e-3, (48, dim))).astype(np.float32)
    yield "mixed scales", (rng.normal(0, 1, (48, dim)) * 10.0 ** rng.integers(-3, 3, (48, 1))).astype(np.float32)
    yield "sift-like", np.floor(np.abs(rng.normal(0, 40, (48, dim)))).astype(np.float32)       # integer-valued rows
    yield "sparse", (rng.normal(0, 1, (48, dim)) * (rng.random((48, dim)) < 0.1)).astype(np.float32)


def _c1_chained(dim):
    """csrc/build_kernels.hip gram_c1_chained: 1.05 (K + 4) 2^-24 with K = dim rounded up to 32"""
    return np.float32(1.05) * np.float32(((dim + 31) // 32 * 32) + 4) * np.float32(2.0 ** -24)


def _c2_for_dim(dim):
    """csrc/build_kernels.hip gram_c2_for_dim: max(3e-6, (dim / 8 + 16) 2^-24)"""
    return max(np.float32(3.0e-6), np.float32(dim // 8 + 16) * np.float32(2.0 ** -24))


@pytest.mark.parametrize("dtype", [oracle.F32, oracle.F16])
@pytest.mark.parametrize("dim", [32, 100, 128, 260, 768, 1536])
def test_chained_gram_distance_error_is_inside_its_interval(dim, dtype):
    """The three-kernel pool prune (gram_tiles_kernel): one f32 FMA chain over the whole row per Gram entry (restated by
    orc_gram_chain, checked bit for bit against the kernel in tests/test_gpu_build.py), norms in f64, interval constants
    that grow with the row length.  d_ref is the reference's own pair kernel for the row type (f16 rows: widened to f32,
    Strategy2x4)."""
    rng = np.random.default_rng(7100 + dim)
    c1, c2 = _c1_chained(dim), _c2_for_dim(dim)
    worst = 0.0
    for name, rows in _sets(rng, dim):
        if dtype == oracle.F16:
            rows = rows.astype(np.float16)
            if not np.isfinite(rows).all():
                continue
        wide = rows.astype(np.float32)
        g = oracle.gram_chain(wide)
        nrm = (wide.astype(np.float64) ** 2).sum(1).astype(np.float32)  # the kernel accumulates the norms in f64
        n = rows.shape[0]
        for i in range(n):
            for j in range(i):  # the sweep only asks for j < i
                nsum = np.float32(nrm[i] + nrm[j])
                dp = np.float32(nsum - np.float32(np.float32(2.0) * g[i, j]))
                e = np.float32(c1 * nsum + c2 * np.abs(dp))
                d_ref = oracle.distance(dtype, oracle.L2, rows[i], rows[j])
                err = abs(float(dp) - d_ref)
                assert err <= float(e), (name, dim, i, j, float(dp), d_ref, float(e))
                if float(e) > 0:
                    worst = max(worst, err / float(e))
                dpi = np.float32(-g[i, j])
                ei = np.float32(c1 * nsum + c2 * np.abs(dpi))
                di_ref = oracle.distance(dtype, oracle.INNER_PRODUCT, rows[i], rows[j])
                assert abs(float(dpi) - di_ref) <= float(ei), (name, dim, i, j, "ip")
    assert worst < 1.0
