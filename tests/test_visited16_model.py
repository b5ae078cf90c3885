"""CPU model of the 16-bit visited table (search_kernel_impl.h: ht16_insert_open / ht16_contains, search_kernels.hip:
ht16_geometry).  The device code stores 16 bits per id and still has to be an exact set -- NotInMut is a
hashbrown::HashSet (diskann/src/graph/glue.rs:524-561).  The argument: probe k of id looks at x_k = id * (A + k * B2)
mod 2^m with an odd multiplier, a bijection of [0, 2^m); bucket = floor(x_k * W / 2^m) for a table of W dwords = W
buckets of two 16-bit entries (any count), so the x_k of one bucket are at most ceil(2^m / W) consecutive values and
their low tb bits tell them apart; entry = (k, low tb bits of x_k): bucket and entry give back x_k and k, hence the id,
whichever half of the bucket the entry sits in.  An insert takes the first empty half (low, then high) of the first
probed bucket that has one; a lookup stops at the first bucket with an empty half.  This file checks that argument on
the same constants and the same geometry rule, and replays inserts against a Python set."""
import numpy as np

A, B2 = 0x9E3779B1, 0x3C6EF372


def geometry(words, nslots):
    if words < 32 or words > 65536:
        return None
    m = 1
    while m < 32 and (1 << m) < nslots:
        m += 1
    if m >= 32:
        return None
    per_bucket = ((1 << m) + words - 1) // words
    tb = 0
    while (1 << tb) < per_bucket:
        tb += 1
    if tb > 14:
        return None
    return dict(idmask=(1 << m) - 1, tb=tb, kmax=min((1 << (16 - tb)) - 1, 64), buckets=words, m=m)


def probe(g, ident, k):
    x = (ident * (A + k * B2)) & 0xFFFFFFFF & g["idmask"]
    return (x * g["buckets"]) >> g["m"], (x & ((1 << g["tb"]) - 1)) | (k << g["tb"])


def test_bucket_and_entry_determine_the_id():
    for words, nslots in ((32, 100), (64, 4001), (256, 70000), (1024, 1 << 20), (2048, 1_000_001), (4096, 10_000_001),
                          (928, 1_000_001), (768, 1_000_001), (1504, 10_000_001), (36, 4001)):
        g = geometry(words, nslots)
        assert g is not None
        ids = np.arange(min(nslots, 1 << 18), dtype=np.uint64)
        if nslots > ids.size:  # a sample that reaches the top of the id range
            ids = np.unique(np.concatenate([ids, np.random.default_rng(1).integers(0, nslots, 1 << 18).astype(np.uint64),
                                            np.arange(nslots - 1000, nslots, dtype=np.uint64)]))
        seen = {}
        for k in range(min(g["kmax"], 6)):
            x = (ids * np.uint64((A + k * B2) & 0xFFFFFFFF)) & np.uint64(g["idmask"])
            slot = (x * np.uint64(g["buckets"])) >> np.uint64(g["m"])
            entry = (x & np.uint64((1 << g["tb"]) - 1)) | np.uint64(k << g["tb"])
            assert entry.max() < 0xFFFF, "0xFFFF is the empty mark"
            assert slot.max() < g["buckets"]
            key = slot * np.uint64(65536) + entry
            assert np.unique(key).size == ids.size, (words, nslots, k)  # injective for this k
            for kk, prev in seen.items():  # and no (slot, entry) of probe k equals one of another probe number
                assert not np.intersect1d(prev, key).size, (k, kk)
            seen[k] = key


def test_geometry_limits():
    assert geometry(48, 100) is not None        # any bucket count
    assert geometry(16, 100) is None
    assert geometry(1024, 100_000_001) is None  # 27 id bits over 2^10 buckets: no room for a probe number
    g = geometry(8192, 100_000_001)             # 2^13 buckets: 14 tag bits, 3 probes of two places each
    assert g and g["tb"] == 14 and g["kmax"] == 3
    g = geometry(2048, 1_000_001)
    assert g and g["tb"] == 9 and g["kmax"] == 64
    g = geometry(1024, 1000)                    # more buckets than ids: a bucket per id
    assert g and g["tb"] == 0


def test_replay_against_a_set():
    """insert / lookup exactly as the device does (first empty half of the first probed bucket that has one, exhausted
    after kmax full buckets)"""
    rng = np.random.default_rng(7)
    for words, nslots, fill in ((64, 5001, 96), (256, 1 << 20, 380), (256, 3_000_000, 384), (928, 1_000_001, 1390)):
        g = geometry(words, nslots)
        table = {}  # bucket -> [low, high]
        truth, exhausted = set(), set()
        stream = rng.integers(0, nslots, fill * 3)
        for ident in map(int, stream):
            if len(truth) >= fill:
                break
            res = None
            for k in range(g["kmax"]):
                b, entry = probe(g, ident, k)
                cur = table.setdefault(b, [])
                if entry in cur:
                    res = "present"
                    break
                if len(cur) < 2:
                    cur.append(entry)
                    res = "inserted"
                    break
            if res is None:
                exhausted.add(ident)  # the device sends these to the spill table
                continue
            assert (res == "present") == (ident in truth), ident
            truth.add(ident)
        # lookups: everything inserted is found, nothing else is
        for ident in list(truth) + [int(i) for i in rng.integers(0, nslots, 2000)]:
            found = False
            for k in range(g["kmax"]):
                b, entry = probe(g, ident, k)
                cur = table.get(b, [])
                if entry in cur:
                    found = True
                    break
                if len(cur) < 2:
                    break
            assert found == (ident in truth), ident
