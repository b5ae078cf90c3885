"""CPU model of how gram_tiles_kernel (diskann_amd/csrc/build_kernels.hip) deals the tiles of one candidate list to the
eight waves of its workgroup: tile q of the row-major enumeration goes to wave q % 8 (odd workgroups count their waves in
reverse).  Checked for every shape the launches can have: each tile of the lower-triangular tiling is computed exactly once,
no wave owns more than three (the kernel's accumulator budget), the narrow instantiation is chosen exactly when no wave
can own more than one, and the deal is never worse -- and for most shapes better -- balanced over the four SIMDs than the
wave-per-row-block assignment it replaced."""
import itertools

WAVES = 8


def tiles_of(tr, tc):
    """row-major enumeration: row block b holds column blocks 0 .. min(b, tc - 1)"""
    return [(b, c) for b in range(tr) for c in range(min(b, tc - 1) + 1)]


def decode(q, tr, tc):
    """the kernel's loop: subtract row-block counts until q falls inside one"""
    b = 0
    while True:
        cnt = min(b, tc - 1) + 1
        if q < cnt:
            return b, q
        q -= cnt
        b += 1
        assert b < tr


def deal(tr, tc, odd):
    T = len(tiles_of(tr, tc))
    owned = {}
    for wave in range(WAVES):
        wv = WAVES - 1 - wave if odd else wave
        nt = (T - wv + 7) // 8 if T > wv else 0
        owned[wave] = [decode(wv + 8 * t, tr, tc) for t in range(nt)]
    return owned


def simd_load(owned):
    load = [0, 0, 0, 0]
    for wave, ts in owned.items():
        load[wave % 4] += len(ts)
    return load


def test_every_tile_once_and_at_most_three_per_wave():
    for tr, tc in itertools.product(range(1, 9), range(1, 4)):
        if tc > tr:
            continue
        want = tiles_of(tr, tc)
        for odd in (False, True):
            owned = deal(tr, tc, odd)
            got = [t for ts in owned.values() for t in ts]
            assert sorted(got) == sorted(want), (tr, tc, odd)
            assert len(got) == len(set(got))
            assert max(len(ts) for ts in owned.values()) <= 3
            assert all(c <= b for b, c in got)  # lower triangle (and the diagonal blocks) only
            # nobody owns two tiles more than anybody else
            assert max(len(ts) for ts in owned.values()) - min(len(ts) for ts in owned.values()) <= 1


def test_narrow_instantiation_iff_one_tile_per_wave():
    """launch_gram_tiles picks gram_tiles_kernel<RT, 1> when tmax <= 8 (tmax = tiles of a full ng x mg item)"""
    for ng, mg in ((32, 32), (64, 64), (96, 96), (128, 96), (160, 96), (256, 96), (256, 64), (256, 32), (96, 32)):
        tr, tc = ng // 32, min(mg, ng) // 32
        tmax = len(tiles_of(tr, tc))
        narrow = tmax <= 8
        for tr_item in range(1, tr + 1):  # items of the launch are no longer than ng rows
            for odd in (False, True):
                owned = deal(tr_item, min(tc, tr_item), odd)
                if narrow:
                    assert max(len(ts) for ts in owned.values()) <= 1, (ng, mg, tr_item)


def test_balance_over_simds_against_one_wave_per_row_block():
    better = 0
    for tr in range(1, 9):
        tc = min(tr, 3)
        old = [0, 0, 0, 0]
        for wave in range(tr):  # until round 4: wave w owned row block w
            old[wave % 4] += min(wave, tc - 1) + 1
        new = simd_load(deal(tr, tc, False))
        assert sum(new) == sum(old)
        assert max(new) <= max(old), (tr, old, new)
        better += max(new) < max(old)
        # two workgroups share a CU: the reversed deal of the odd one puts its extra tiles on other SIMDs
        both = [a + b for a, b in zip(new, simd_load(deal(tr, tc, True)))]
        assert max(both) - min(both) <= 1, (tr, both)
    assert better >= 4
