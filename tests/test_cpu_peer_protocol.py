"""The peer-flag transport's reuse argument, checked mechanically (no GPU): a small explicit-state model of what
compose_amd/csrc/cedr_mpi.cpp's PeerFlagComm does to the mailboxes -- every store, counter add, wait and read of every
rank as its own atomic step -- explored over ALL interleavings of the ranks. The model restates the protocol (it does not
run the kernels): what it proves is the design argument written next to PeerFlagComm, the part of the transport a one-GPU
box cannot stress (ranks on different GPUs drift apart far more than processes sharing one).

  collectives: slots [parity][source] and one arrival counter per parity in every rank's mailbox; collective e uses
    parity e & 1 and waits for (e/2 + 1) * arrivals-per-collective; both forms (with / without the rank's own block).
  point-to-point: per ordered pair and parity a slot, a flag (message number + 1) and an ack (consumed number + 1);
    a sender reuses a slot after the ack of the message two before.

Each model has a negative control (one parity only / no ack wait) in which the search must FIND a stale read -- the
checker can see the failure it is there to exclude."""
import pytest


def explore(init, actions):
    """Depth-first search over all interleavings. actions(state) -> list of (next_state | 'VIOLATION: ...')."""
    seen, stack = {init}, [init]
    finals = 0
    while stack:
        st = stack.pop()
        nxt = actions(st)
        if not nxt:
            finals += 1
        for n in nxt:
            if isinstance(n, str):
                return n, len(seen)
            if n not in seen:
                seen.add(n)
                stack.append(n)
    return None, len(seen)


def collective_model(P, E, with_self, parities):
    """A rank's program for collective e: for every destination d (all ranks, or all but itself): STORE its block into
    d's slot [par][me], then ADD 1 to d's counter [par]; WAIT until its own counter [par] >= (e // parities + 1) * narr;
    READ every source's slot (asserting it holds collective e's block). par = e % parities."""
    narr = P if with_self else P - 1
    dests = [[d for d in range(P) if with_self or d != r] for r in range(P)]
    # a rank's micro-steps per collective: 2 per destination, 1 wait, 1 read per source
    per = [2 * len(dests[r]) + 1 + len(dests[r]) for r in range(P)]

    def actions(st):
        pcs, slots, ctrs = st
        out = []
        for r in range(P):
            pc = pcs[r]
            e, k = divmod(pc, per[r])
            if e >= E:
                continue
            par = e % parities
            nd = len(dests[r])
            if k < 2 * nd:
                d = dests[r][k // 2]
                if k % 2 == 0:  # store
                    s2 = list(slots)
                    s2[(d * parities + par) * P + r] = e
                    out.append((pcs[:r] + (pc + 1,) + pcs[r + 1:], tuple(s2), ctrs))
                else:  # add
                    c2 = list(ctrs)
                    c2[d * parities + par] += 1
                    out.append((pcs[:r] + (pc + 1,) + pcs[r + 1:], slots, tuple(c2)))
            elif k == 2 * nd:  # wait
                if ctrs[r * parities + par] >= (e // parities + 1) * narr:
                    out.append((pcs[:r] + (pc + 1,) + pcs[r + 1:], slots, ctrs))
            else:  # read source
                src = dests[r][k - 2 * nd - 1]
                got = slots[(r * parities + par) * P + src]
                if got != e:
                    return ["VIOLATION: rank %d, collective %d, read source %d's slot and found collective %d's block"
                            % (r, e, src, got)]
                out.append((pcs[:r] + (pc + 1,) + pcs[r + 1:], slots, ctrs))
        return out
    init = (tuple([0] * P), tuple([-1] * (P * parities * P)), tuple([0] * (P * parities)))
    return init, actions


@pytest.mark.parametrize("P,E", [(2, 6), (3, 6), (4, 4)])  # (4 ranks x 4 collectives: ~180 000 states)
@pytest.mark.parametrize("with_self", [True, False])
def test_two_parities_are_enough_for_the_collectives(P, E, with_self):
    bad, nstates = explore(*collective_model(P, E, with_self, 2))
    assert bad is None, bad
    assert nstates > 100


@pytest.mark.parametrize("with_self", [True, False])
def test_one_parity_is_not(with_self):
    """Negative control: without the alternation a fast rank's next block lands in a slot a slow rank has not read yet
    (or its arrival is counted for the wrong collective)."""
    bad, _ = explore(*collective_model(3, 4, with_self, 1))
    assert bad is not None and bad.startswith("VIOLATION")


def p2p_model(M, wait_for_ack):
    """One ordered pair. Sender, message m (parity m & 1): [wait ack[par] >= m - 1 if m >= 2]; store slot[par] = m;
    flag[par] = m + 1. Receiver: wait flag[par] >= m + 1; read slot[par] (must be m); ack[par] = m + 1."""
    def actions(st):
        ps, pr, slot, flag, ack = st
        out = []
        m, k = divmod(ps, 3)
        if m < M:
            par = m & 1
            if k == 0:
                if not wait_for_ack or m < 2 or ack[par] >= m - 1:
                    out.append((ps + 1, pr, slot, flag, ack))
            elif k == 1:
                s2 = list(slot); s2[par] = m
                out.append((ps + 1, pr, tuple(s2), flag, ack))
            else:
                f2 = list(flag); f2[par] = m + 1
                out.append((ps + 1, pr, slot, tuple(f2), ack))
        m, k = divmod(pr, 3)
        if m < M:
            par = m & 1
            if k == 0:
                if flag[par] >= m + 1:
                    out.append((ps, pr + 1, slot, flag, ack))
            elif k == 1:
                if slot[par] != m:
                    return ["VIOLATION: message %d expected, the slot holds %d" % (m, slot[par])]
                out.append((ps, pr + 1, slot, flag, ack))
            else:
                a2 = list(ack); a2[par] = m + 1
                out.append((ps, pr + 1, slot, flag, tuple(a2)))
        return out
    return (0, 0, (-1, -1), (0, 0), (0, 0)), actions


def test_point_to_point_slots_are_reused_only_after_the_ack():
    bad, nstates = explore(*p2p_model(8, True))
    assert bad is None, bad
    assert nstates > 50
    bad, _ = explore(*p2p_model(8, False))  # negative control: no ack wait -> a slot is overwritten before it is read
    assert bad is not None and bad.startswith("VIOLATION")
