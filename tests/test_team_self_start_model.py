"""A model of the handshake between a team's control, visited and queue waves around the visited wave's SELF-START
(diskann_amd/csrc/search_kernel_impl.h: team_control_wave, team_visited_wave), explored over every
interleaving of the waves' LDS accesses between two barriers.

What the kernel relies on, and what is checked here for every schedule:
  1. both waves take the same decision ("prepared hop": the node to expand next is the one the visited wave filtered) --
     they read the same words, none of which is rewritten before both have read it;
  2. on a self-started hop the visited wave writes nothing the control wave's decision reads (the spare candidate buffer,
     the slots, its reply) before the control wave has read it -- the guard is the hop's "go", which the control wave
     stores after its loads (one wave's LDS accesses are performed in issue order);
  3. the spare candidate buffer is not refilled before the queue wave holds its distances in registers (kMbLoaded);
  4. landing buffer 0 never has two adjacency requests in flight, and nobody reads it while a request of another wave's is;
  5. when the control wave ends the search instead of starting the hop (error status), every wave meets exactly one more
     barrier (the release) -- no wave is left waiting at a barrier nobody else comes to.

The model is a restatement, not the kernel: it is there to make the argument of DESIGN 3.5 checkable.  The kernel itself
is compared with the oracle by tests/test_gpu_parity.py (every team case, with the self-start on and off)."""
import itertools

import pytest


class Lds:
    """the words the three waves share, with who-wrote-what bookkeeping for the checks"""

    def __init__(self, hop, prepared, exits):
        self.hop = hop                      # the hop whose distances are ready when the barrier opens
        self.prepared = prepared            # the control wave's decision (a function of the words below)
        self.exits = exits                  # the control wave ends the search at this decision (error status)
        self.go = hop                       # kMbGo's hop number (hop + 1 after this hop's "go"); "exit" on release
        self.spec_seq = hop
        self.loaded = hop - 1               # kMbLoaded: last hop whose buffer the queue wave holds in registers
        self.reply_gen = hop                # generation of kMbVReply / slots / spare buffer contents
        self.control_read_inputs = False
        self.landing_in_flight = None       # which wave's request is in flight into landing buffer 0
        self.landing_node = "spec(h)"       # what the buffer holds
        self.errors = []


def control_steps(l, self_start_enabled):
    """the control wave between barrier h and barrier h + 1, as a list of atomic LDS steps"""
    def read_inputs():
        # one batch of loads: pop publication, reply, the hop's distances -- all of generation `hop`
        if l.reply_gen != l.hop:
            l.errors.append("control read a reply / slots / buffer the visited wave had already rewritten")
        l.control_read_inputs = True
    yield read_inputs
    if l.exits:
        def release():
            l.go = "exit"
        yield release
        return

    def go():
        l.go = l.hop + 1
    def request():
        if l.landing_in_flight is not None:
            l.errors.append("two adjacency requests in flight into landing buffer 0")
        l.landing_in_flight = "control"
        l.landing_node = "runner(h+1)"
    if l.prepared:
        yield go                                              # fast path: "go" first
        # self-started: records the runner-up, requests nothing (pf_by_visited); with the self-start switched off
        # (DANN_DBG_TUNE_OFF bit 64) it requests the runner-up's row itself, as through round 5
        if not self_start_enabled:
            yield request
    else:
        def expand():                                         # reads landing buffer 0 only for its own requests
            if l.landing_in_flight == "visited":
                l.errors.append("control read landing buffer 0 while the visited wave's request was in flight")
        yield expand
        yield go
        yield request

    def words():
        l.spec_seq = l.hop + 1
    yield words


def queue_steps(l):
    def load_buffer():
        l.loaded = l.hop                                      # holds hop's distances in registers
    yield load_buffer

    def merge_pop_publish():
        pass
    yield merge_pop_publish


def visited_steps(l, self_start_enabled):
    """generator of steps; a step returns False when it has to be retried (a poll that did not succeed)"""
    state = {"self": False, "left": False}

    def decide():
        # reads the same words as the control wave's decision; they are of generation `hop` until this wave rewrites them
        state["self"] = self_start_enabled and l.prepared
        if state["self"]:
            if l.landing_in_flight is not None:
                l.errors.append("two adjacency requests in flight into landing buffer 0")
            l.landing_in_flight = "visited"
            l.landing_node = "runner(h+1)"
        return True
    yield decide

    def wait_go():
        if l.go == "exit":
            state["left"] = True
            return True
        return l.go == l.hop + 1
    yield wait_go

    def wait_words():
        if state["left"] or state["self"]:
            return True
        return l.spec_seq == l.hop + 1
    yield wait_words

    def wait_loaded():
        if state["left"]:
            return True
        return l.loaded >= l.hop
    yield wait_loaded

    def landed():
        if state["left"]:
            return True
        # the row arrives some time after it was requested; the poll sees it (the model lets it land here)
        if l.landing_in_flight is None and l.landing_node != "runner(h+1)":
            return False                                      # (not requested yet: waited mode before the control wave's request)
        l.landing_in_flight = None
        return True
    yield landed

    def write_outputs():
        if state["left"]:
            return True
        if not l.control_read_inputs:
            l.errors.append("the visited wave rewrote reply / slots / spare buffer before the control wave had read them")
        if l.loaded < l.hop:
            l.errors.append("the spare candidate buffer was refilled before the queue wave had loaded it")
        l.reply_gen = l.hop + 1
        return True
    yield write_outputs
    state["done"] = True


def explore(prepared, exits, self_start_enabled):
    """all interleavings of the three waves' steps between two barriers; returns (schedules, errors)"""
    errors, schedules = set(), 0

    def run(order):
        l = Lds(hop=5, prepared=prepared, exits=exits)
        gens = {"c": iter(list(control_steps(l, self_start_enabled))), "q": iter(list(queue_steps(l))), "v": visited_steps(l, self_start_enabled)}
        pending = {"c": None, "q": None, "v": None}
        done = set()
        steps = 0
        it = itertools.cycle(order)
        stalled = 0
        while len(done) < 3:
            w = next(it)
            if w in done:
                continue
            if pending[w] is None:
                try:
                    pending[w] = next(gens[w])
                except StopIteration:
                    done.add(w)
                    continue
            r = pending[w]()
            if r is False:
                stalled += 1
                if stalled > 200:
                    l.errors.append(f"deadlock: wave {w} waits for something nobody provides")
                    break
                continue
            stalled = 0
            pending[w] = None
            steps += 1
        # every wave arrives at exactly one barrier after its steps (the per-hop one, or the release): checked by
        # construction of the step lists -- what could go wrong is a wave that never gets there (deadlock above)
        return l.errors

    # schedules: every sequence over {c, q, v} of length 8 used as a round-robin priority pattern covers all relative
    # orders of the <= 5 + 2 + 6 steps that matter (who reaches which step first)
    for order in itertools.product("cqv", repeat=8):
        if len(set(order)) < 3:
            continue
        schedules += 1
        for e in run(order):
            errors.add(e)
    return schedules, errors


@pytest.mark.parametrize("prepared", [True, False])
@pytest.mark.parametrize("exits", [False, True])
@pytest.mark.parametrize("self_start", [True, False])
def test_self_start_handshake_has_no_bad_interleaving(prepared, exits, self_start):
    schedules, errors = explore(prepared, exits, self_start)
    assert schedules > 1000
    assert not errors, sorted(errors)


def test_the_model_sees_what_the_go_guard_prevents():
    """the same exploration with the visited wave's wait for "go" removed: on a prepared, self-started hop there are
    schedules in which it rewrites its reply before the control wave has read it -- the model is not vacuous"""
    global visited_steps
    real = visited_steps

    def without_guard(l, self_start_enabled):
        for step in real(l, self_start_enabled):
            if step.__name__ == "wait_go":
                yield lambda: True
            else:
                yield step
    visited_steps = without_guard
    try:
        _, errors = explore(prepared=True, exits=False, self_start_enabled=True)
    finally:
        visited_steps = real
    assert any("before the control wave had read them" in e for e in errors), errors
