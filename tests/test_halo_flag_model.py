"""The halo-flag form of the single-frame sharded step (csrc/comm.hip, gh_common.h) as a discrete-event model, checked under
random timing: N ranks, each with its caller's stream and its side stream, the two polled dwords (`started`, `flag`), RCCL's
send / recv rendezvous between neighbours, hosts that run ahead of or lag behind their GPUs, ranks that are late by whole steps.

What real multi-GPU hardware would have to show and a one-GPU box cannot: (1) no deadlock -- every step of every rank completes
whatever the timing; (2) the exchange of step e sends the rows the producer of step e wrote (read-after-write across the two
streams without an event); (3) it never overwrites halo rows a kernel may still read (write-after-read); (4) the shard kernel of
step e reads halo rows that hold its neighbours' step-e rows; (5) the caller may overwrite the shard right after the step: the
producer of step e+1 never runs before exchange e has read its rows.  The model executes exactly the ordering rules the code
relies on: in-order streams, `started` stored when the shard kernel BEGINS, the side stream's wait for it, the exchange completing
only when both neighbours have posted theirs, `flag` set behind the exchange, edge bands that let the kernel finish only behind
`flag`.  No GPU, no timing assumptions: durations and host delays are random."""
import random

import pytest


class Rank:
    def __init__(self, r, n, steps, rng):
        self.r, self.n, self.steps = r, n, steps
        self.up, self.down = (r - 1 if r > 0 else None), (r + 1 if r < n - 1 else None)
        # host: when each step's operations are enqueued (monotone; may be far ahead of or behind the GPU)
        t, self.enq = 0.0, []
        for _ in range(steps):
            t += rng.choice([0.0, 0.1, 1.0, 5.0, 40.0]) * rng.random()
            self.enq.append(t)
        self.started = 0          # dword stored by the shard kernel's first workgroup
        self.flag = 0             # dword set behind the exchange
        self.rows_version = 0     # what the shard's edge rows hold (written by the producer of step e)
        self.halo_version = {"up": 0, "down": 0}
        # main stream: P(e) producer of the frame rows, K(e) shard kernel; side stream: W(e), X(e), F(e)
        self.main = [(op, e) for e in range(1, steps + 1) for op in ("P", "K")]
        # (a rank without neighbours -- world of one -- launches its kernel and nothing else: sharded_step's first return)
        self.side = [(op, e) for e in range(1, steps + 1) for op in ("W", "X", "F")] if n > 1 else []
        self.mi = self.si = 0
        self.k_running = None     # (e, earliest finish time) while K(e) runs
        self.x_posted = 0         # highest e whose exchange this rank has posted (its sends are readable, its recvs armed)
        self.x_done = 0
        self.k_done = 0
        self.dur = lambda: rng.choice([0.5, 1.0, 3.0, 10.0]) * (0.2 + rng.random())
        self.busy_until_main = self.busy_until_side = 0.0
        self.sent = {}            # e -> version of the rows this rank's exchange e read when it was posted


def _run(ranks):
    n, steps = ranks[0].n, ranks[0].steps
    now, guard = 0.0, 0
    while any(k.mi < len(k.main) or k.si < len(k.side) or k.k_running for k in ranks):
        guard += 1
        assert guard < 200000, "no progress: deadlock"
        progressed = False
        for k in ranks:
            # ---- the caller's stream (in order) ----
            if k.k_running:
                e, t_min = k.k_running
                has_nb = k.up is not None or k.down is not None
                if now >= t_min and (not has_nb or k.flag >= e):      # the edge bands leave their wait only behind `flag`
                    # (4) the halo rows the kernel read hold the neighbours' rows of THIS step
                    if k.up is not None:
                        assert k.halo_version["up"] == e, (k.r, e, "stale upper halo")
                    if k.down is not None:
                        assert k.halo_version["down"] == e, (k.r, e, "stale lower halo")
                    k.k_running, k.k_done = None, e
                    progressed = True
            elif k.mi < len(k.main) and now >= k.busy_until_main:
                op, e = k.main[k.mi]
                if now >= k.enq[e - 1]:
                    if op == "P":
                        # (5) the producer of step e overwrites the rows exchange e-1 sent: that exchange must have read them
                        assert k.x_done >= e - 1 or (k.up is None and k.down is None), (k.r, e, "producer overtook the exchange")
                        k.rows_version = e
                        k.busy_until_main = now + k.dur()
                    else:
                        k.started = e                                  # stored when the kernel BEGINS: everything before it on s is done
                        k.k_running = (e, now + k.dur())
                    k.mi += 1
                    progressed = True
            # ---- the side stream (in order) ----
            if k.si < len(k.side) and now >= k.busy_until_side:
                op, e = k.side[k.si]
                if now >= k.enq[e - 1]:
                    if op == "W":
                        if k.started >= e:
                            k.si += 1
                            progressed = True
                    elif op == "X":
                        if k.x_posted < e:
                            # posting the exchange: (2) the rows it sends are the producer's rows of step e ...
                            assert k.rows_version == e, (k.r, e, "exchange read rows of step %d" % k.rows_version)
                            # ... (3) and the halo rows it will overwrite are no longer read: the previous kernel has finished
                            assert k.k_done >= e - 1, (k.r, e, "exchange overwrites halo rows a kernel still reads")
                            k.sent[e] = k.rows_version
                            k.x_posted = e
                            progressed = True
                        nbs = [ranks[i] for i in (k.up, k.down) if i is not None]
                        if all(nb.x_posted >= e for nb in nbs):       # rendezvous: both neighbours have posted their step-e exchange
                            if k.up is not None:
                                k.halo_version["up"] = ranks[k.up].sent[e]
                            if k.down is not None:
                                k.halo_version["down"] = ranks[k.down].sent[e]
                            k.x_done = e
                            k.busy_until_side = now + k.dur()
                            k.si += 1
                            progressed = True
                    else:
                        k.flag = e
                        k.si += 1
                        progressed = True
        if not progressed:
            # advance time to the next moment something can change
            cands = [t for k in ranks for t in (k.busy_until_main, k.busy_until_side) if t > now]
            cands += [k.k_running[1] for k in ranks if k.k_running and k.k_running[1] > now]
            cands += [k.enq[e - 1] for k in ranks for e in range(1, steps + 1) if k.enq[e - 1] > now]
            assert cands, "no progress and nothing pending: deadlock at t=%.2f %s" % (
                now, [(k.r, k.main[k.mi] if k.mi < len(k.main) else None, k.side[k.si] if k.si < len(k.side) else None, k.started, k.flag) for k in ranks])
            now = min(cands)
    for k in ranks:
        assert k.k_done == steps and (k.x_done == steps or n == 1)
    return now


def simulate(n, steps, seed):
    rng = random.Random(seed)
    return _run([Rank(r, n, steps, rng) for r in range(n)])


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_halo_flag_handshake_is_deadlock_free_and_ordered(n):
    for seed in range(150):
        simulate(n, steps=6, seed=seed * 17 + n)


def test_the_model_catches_a_broken_protocol():
    """Negative control: without the side stream's wait for `started` (an exchange enqueued on a second stream with nothing that orders
    it behind the caller's stream) the exchange reads rows the producer has not written yet, or overwrites halo rows a kernel still
    reads -- the model must say so for some timings."""
    broken = 0
    for seed in range(60):
        rng = random.Random(seed)
        ranks = [Rank(r, 3, 4, rng) for r in range(3)]
        for k in ranks:
            k.side = [(op, e) for (op, e) in k.side if op != "W"]
        try:
            _run(ranks)
        except AssertionError:
            broken += 1
    assert broken > 30, "a side stream that does not wait for `started` went unnoticed"


# ---- the stream step (gh_sharded_*_stream_step_dev): the halo rows of frame k + 1 travel beside the launch on frame k ---------------

def _run_stream(n, frames, ring, seed):
    """Frames 0 .. frames-1 cycle through `ring` buffers.  Caller's stream of a rank: P(0), then per step k: P(k + 1) (the next frame's
    rows, into buffer (k + 1) % ring), K(k).  Side stream: per step k behind `started` >= k + 1: [the cold-start exchange of frame 0 at
    k = 0,] the exchange of frame k + 1, each followed by the flag number of its frame.  K(k) may finish only behind the flag of frame k."""
    rng = random.Random(seed)
    R = [Rank(r, n, frames, rng) for r in range(n)]
    for k in R:
        k.main = [("P", 0)] + [x for j in range(frames) for x in ((("P", j + 1),) if j + 1 < frames else ()) + (("K", j),)]
        k.side = []
        for j in range(frames):
            if j == 0:
                k.side += [("W", 0), ("X", 0), ("F", 0)]
            if j + 1 < frames:
                k.side += [("W", j), ("X", j + 1), ("F", j + 1)]
        k.buf_version = {}           # ring slot -> frame whose rows it holds
        k.halo = {}                  # ring slot -> {"up": frame, "down": frame}
        k.flag = -1
        k.started = -1
        k.x_posted, k.x_done, k.k_done = -1, -1, -1
        k.reading = None             # ring slot the running kernel reads
    now, guard = 0.0, 0
    enq = lambda k, j: k.enq[min(max(j, 0), frames - 1)]
    while any(k.mi < len(k.main) or k.si < len(k.side) or k.k_running for k in R):
        guard += 1
        assert guard < 400000, "no progress: deadlock"
        progressed = False
        for k in R:
            has_nb = k.up is not None or k.down is not None
            if k.k_running:
                j, t_min = k.k_running
                if now >= t_min and (not has_nb or k.flag >= j):
                    slot = j % ring
                    if k.up is not None:
                        assert k.halo[slot]["up"] == j, (k.r, j, "stale upper halo", k.halo[slot])
                    if k.down is not None:
                        assert k.halo[slot]["down"] == j, (k.r, j, "stale lower halo", k.halo[slot])
                    assert k.buf_version[slot] == j, (k.r, j, "the frame was overwritten under the kernel")
                    k.k_running, k.k_done, k.reading = None, j, None
                    progressed = True
            elif k.mi < len(k.main) and now >= k.busy_until_main:
                op, j = k.main[k.mi]
                if now >= enq(k, j - 1 if op == "P" else j):
                    if op == "P":
                        slot = j % ring
                        # the rows this overwrites were sent by the exchange of frame j - ring: it must have read them
                        assert j - ring < 0 or not has_nb or k.x_done >= j - ring, (k.r, j, "producer overtook the exchange of the frame it replaces")
                        k.buf_version[slot] = j
                        k.busy_until_main = now + k.dur()
                    else:
                        k.started = j
                        k.reading = j % ring
                        k.k_running = (j, now + k.dur())
                    k.mi += 1
                    progressed = True
            if k.si < len(k.side) and now >= k.busy_until_side:
                op, j = k.side[k.si]
                step_of = j if op == "W" else max(j - 1, 0)
                if now >= enq(k, step_of):
                    if op == "W":
                        if k.started >= j:
                            k.si += 1
                            progressed = True
                    elif op == "X":
                        slot = j % ring
                        if k.x_posted < j:
                            assert k.buf_version.get(slot) == j, (k.r, j, "exchange read rows of frame %s" % k.buf_version.get(slot))
                            assert k.reading != slot or k.k_running is None or k.k_running[0] == j, (k.r, j, "exchange overwrites halo rows a kernel still reads")
                            k.sent[j] = j
                            k.x_posted = j
                            progressed = True
                        nbs = [R[i] for i in (k.up, k.down) if i is not None]
                        if all(nb.x_posted >= j for nb in nbs):
                            k.halo[slot] = {"up": R[k.up].sent[j] if k.up is not None else None, "down": R[k.down].sent[j] if k.down is not None else None}
                            k.x_done = j
                            k.busy_until_side = now + k.dur()
                            k.si += 1
                            progressed = True
                    else:
                        k.flag = j
                        k.si += 1
                        progressed = True
        if not progressed:
            cands = [t for k in R for t in (k.busy_until_main, k.busy_until_side) if t > now]
            cands += [k.k_running[1] for k in R if k.k_running and k.k_running[1] > now]
            cands += [t for k in R for t in k.enq if t > now]
            assert cands, "no progress and nothing pending: deadlock"
            now = min(cands)
    for k in R:
        assert k.k_done == frames - 1


@pytest.mark.parametrize("n,ring", [(2, 2), (3, 3), (8, 2), (8, 8), (4, 5)])
def test_stream_step_handshake_is_deadlock_free_and_ordered(n, ring):
    for seed in range(80):
        _run_stream(n, frames=7, ring=ring, seed=seed * 13 + n + ring)


def test_stream_step_needs_two_buffers_per_rank():
    """With ONE buffer the next frame's rows would be written while the launch on the current frame still reads them: the real code
    passes next = NULL for a ring of one and exchanges beside the launch instead; the model, forced to prefetch into the buffer in
    use, must object."""
    bad = 0
    for seed in range(40):
        try:
            _run_stream(3, frames=5, ring=1, seed=seed)
        except AssertionError:
            bad += 1
    assert bad > 30
