"""The control wave of a team (csrc/search_kernel_impl.h, team_control_wave) decides which node the next hop expands --
and which node will be the best unexpanded entry after that -- from the queue wave's last publication and the hop's new
distances, *before* the merge.  This file restates the rule and the reference's queue (queue.rs:130-171, 297-313) in
plain Python and checks on tie-heavy random states that every decision the rule commits to equals what the pops after
the sequential inserts return (host logic, no GPU).  Also the sequential-insert form of the merge on the
register-resident queue (`kSeqInsert` path)."""
import math

import numpy as np

from test_merge_rule import sequential_insert


def pops(queue, expanded, n):
    """the first n unexpanded entries in queue order (queue.rs:297-313 pops them one by one)"""
    return [q for q in queue if q[1] not in expanded][:n]


def control_rule(pub, cands):
    """pub = (pf_next, pf_next2): the best and second-best unexpanded entries left after the last pop, as (d, id) or None.
    Returns (next, runner): each an id, or None where the rule does not commit (the control wave then waits for the
    queue wave's pop / leaves the visited wave without work)."""
    pf_next, pf_next2 = pub
    if pf_next is None:
        return None, None
    live = [(d, j, i) for j, (d, i) in enumerate(cands) if not math.isnan(d)]

    def best(sel):  # distance ascending, of equal ones the one emitted last
        return min(sel, key=lambda t: (t[0], -t[1]))

    ahead = [t for t in live if t[0] <= pf_next[0]]
    if not ahead:
        nxt = pf_next[1]
        runner = None
        if pf_next2 is not None:
            ahead2 = [t for t in live if t[0] <= pf_next2[0]]
            runner = best(ahead2)[2] if ahead2 else pf_next2[1]
        return nxt, runner
    b = best(ahead)
    rest = [t for t in ahead if t[1] != b[1]]
    return b[2], (best(rest)[2] if rest else pf_next[1])


def test_early_decisions_equal_the_pops_after_the_merge():
    rng = np.random.default_rng(424242)
    decided = runner_checked = 0
    for trial in range(6000):
        cap = int(rng.integers(1, 24))
        nq = int(rng.integers(0, cap + 1))
        levels = int(rng.integers(1, 10))  # few distinct distances: ties everywhere
        qd = np.sort(rng.integers(0, levels, nq)).astype(np.float32)
        queue = [(float(qd[e]), 1000 + e) for e in range(nq)]
        expanded = {i for _, i in queue if rng.random() < 0.5}
        left = pops(queue, expanded, 2)
        pub = (left[0] if len(left) > 0 else None, left[1] if len(left) > 1 else None)
        nc = int(rng.integers(0, 40))
        cd = rng.integers(0, levels + 2, nc).astype(np.float32)
        cd[rng.random(nc) < 0.05] = np.nan
        cands = [(float(cd[j]), j) for j in range(nc)]
        nxt, runner = control_rule(pub, cands)
        merged = sequential_insert(queue, cap, cands)
        after = pops(merged, expanded, 2)
        if nxt is not None:
            decided += 1
            assert after and after[0][1] == nxt, (trial, queue, expanded, cands, nxt, after)
            # the runner-up is a prediction the kernel verifies (a miss costs a speculation, never a result); the rule
            # is exact whenever the predicted entry is still inside the queue after the merge
            if runner is not None and any(i == runner for _, i in merged):
                runner_checked += 1
                assert len(after) > 1 and after[1][1] == runner, (trial, queue, expanded, cands, runner, after)
        else:
            assert pub[0] is None
    assert decided > 3000 and runner_checked > 2000


def seq_insert_registers(qd, qid, size, cap, cands):
    """the kSeqInsert path: per survivor a lower bound by counting, then `shift the tail up by one` on fixed-size arrays"""
    qd, qid = list(qd), list(qid)
    full = size == cap and cap > 0
    worst = qd[size - 1] if size else None
    for d, i in cands:
        if math.isnan(d) or (full and worst < d):
            continue
        pos = sum(1 for e in range(size) if qd[e] < d)
        if pos >= cap:
            continue
        for p in range(len(qd) - 1, pos, -1):
            qd[p], qid[p] = qd[p - 1], qid[p - 1]
        qd[pos], qid[pos] = d, i
        size = min(size + 1, cap)
    return [(qd[p], qid[p]) for p in range(size)]


def test_register_resident_inserts_equal_the_reference_queue():
    rng = np.random.default_rng(99)
    for trial in range(3000):
        cap = int(rng.integers(1, 30))
        nq = int(rng.integers(0, cap + 1))
        levels = int(rng.integers(1, 8))
        qd = np.sort(rng.integers(0, levels, nq)).astype(np.float32)
        queue = [(float(qd[e]), 1000 + e) for e in range(nq)]
        nc = int(rng.integers(0, 6))
        cd = rng.integers(0, levels + 2, nc).astype(np.float32)
        cd[rng.random(nc) < 0.1] = np.nan
        cands = [(float(cd[j]), j) for j in range(nc)]
        slots = 64
        regs_d = [q[0] for q in queue] + [0.0] * (slots - nq)
        regs_i = [q[1] for q in queue] + [-1] * (slots - nq)
        assert seq_insert_registers(regs_d, regs_i, nq, cap, cands) == sequential_insert(queue, cap, cands), (trial, queue, cands)
