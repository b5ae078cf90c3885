"""A model of the search server's submission protocol (csrc/server.hip), host logic only: callers submit queries and
collect their tickets in any order; resident waves take ring entries in sequence order.

The first form of the server tied the result slot to the ticket (slot = ticket mod ring; a submit waited until the slot's
previous ticket had been collected).  A caller that collected out of order -- or was descheduled with tickets outstanding
-- could then hold the slot another caller's ticket needed, while its own newer ticket waited *behind* that one in the
in-order ring: a deadlock (seen as a 30-second timeout of the out-of-order test about once in seven runs).  The present
form hands out result slots from a free list and frees a ring position as soon as a wave has taken its entry.  The model
runs both under the same adversarial schedules and checks that the present form always drains."""
import random


def run(form, ring, callers, depth, total, seed, max_steps=200000):
    """Returns True if every ticket was served and collected.  One step = one randomly chosen actor makes one move if it
    can.  `form`: "tied" (slot = seq mod ring) or "free" (free slots, positions acknowledged when taken)."""
    rng = random.Random(seed)
    published = {}          # seq -> slot (entries visible to the device)
    next_seq = 0
    taken = 0               # the device takes entries strictly in sequence order
    served = set()          # seqs whose result is in their slot
    collected = set()
    free_slots = list(range(ring))
    slot_busy_until_collected = {}   # tied form: slot -> seq occupying it
    class Caller:
        def __init__(self):
            self.out = []            # outstanding (seq) of this caller
            self.pending = None      # a ticket number drawn but not yet published (tied form: waiting for its slot)
            self.submitted = 0
    cs = [Caller() for _ in range(callers)]
    per_caller = total // callers
    for step in range(max_steps):
        if len(collected) == per_caller * callers:
            return True
        actor = rng.randrange(callers + 1)
        if actor == callers:         # the device: take the next entry if it is published, serve it at once
            if taken in published:
                served.add(taken)
                taken += 1
            continue
        c = cs[actor]
        want_submit = c.submitted < per_caller and len(c.out) < depth
        if c.pending is not None:    # tied form: blocked in submit until the slot's previous occupant was collected
            s = c.pending
            slot = s % ring
            if slot_busy_until_collected.get(slot) is None:
                slot_busy_until_collected[slot] = s
                published[s] = slot
                c.out.append(s)
                c.pending = None
                c.submitted += 1
            continue                 # (a caller blocked in submit does nothing else: that is the point)
        if want_submit and (not c.out or rng.random() < 0.7):
            if form == "tied":
                c.pending = next_seq
                next_seq += 1
            else:
                if not free_slots:
                    continue         # back-pressure: `ring` tickets uncollected
                slot = free_slots.pop()
                s = next_seq
                next_seq += 1
                # position s mod ring: free once the entry of s - ring was *taken* (s - ring < taken)
                if s - ring >= taken:
                    free_slots.append(slot)   # (the real submit spins here holding its number; the model retries)
                    next_seq -= 1
                    continue
                published[s] = slot
                c.out.append(s)
                c.submitted += 1
            continue
        if c.out:                    # collect: the newest first (the adversarial order), only if it has been served
            s = c.out[-1]
            if s in served:
                c.out.pop()
                collected.add(s)
                if form == "tied":
                    slot_busy_until_collected[s % ring] = None
                else:
                    free_slots.append(published[s])
    return False


def test_free_slots_always_drain_where_tied_slots_can_deadlock():
    tied_stuck = 0
    for seed in range(60):
        assert run("free", ring=8, callers=4, depth=3, total=400, seed=seed), seed
        tied_stuck += not run("tied", ring=8, callers=4, depth=3, total=400, seed=seed)
    assert tied_stuck > 0   # the schedule family does reach the old form's deadlock
