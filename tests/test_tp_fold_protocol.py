"""The folded tensor-parallel decode's exchange protocol (fastllama_amd/csrc/tp_tail.h) as a small model under a random
scheduler: G ranks run the decode launches of a few tokens and layers; every launch is a bag of micro-events -- reads of the slices of its
input vector, stores of its own slice into every rank's region -- in ANY order (its workgroups run concurrently: one may already send its rows
while another still reads), followed by the tail: publish this rank's epoch of the exchange kind in every peer's flag word, wait until every
peer's epoch has arrived, end of launch.  Memory is modelled as version tags; a read must find exactly the version the launch is entitled to
(not the previous exchange's, not the next one's).  The model is the argument of tp_tail.h written down so that it can be run: what keeps a
slot from being overwritten while it is still read is that its only full-width reader is the producer of the NEXT exchange.

The same scheduler finds the violation when the argument is broken (no wait in the tail; one flag word for all four kinds), which is what
makes the green runs mean something.  CPU only: this checks the protocol, not the kernels (tests/test_model_gpu.py does that, bit for bit)."""
import random

import pytest

KINDS = ("q", "x2", "h", "x")          # the four exchanges of a layer: attention planes, wo rows, silu features, w2 rows


class Violation(Exception):
    pass


def rank_program(r, G, tokens, layers, rng, wait=True, one_flag=False):
    """Micro-events of rank r, in program order between launches and in random order inside a launch.
    ('read', slot, src, version) | ('put', slot, dst, version) | ('flag', kind, dst, epoch) | ('wait', kind, epoch) | ('local', slot, version)"""
    epoch = {k: 0 for k in KINDS}

    def launch(reads, puts, kind):
        ev = [("read", s, src, v) for (s, v) in reads for src in range(G) for _ in range(2)]      # every slice is read by more than one workgroup
        ev += [("put", s, dst, v) for (s, v) in puts for dst in range(G)]
        rng.shuffle(ev)
        yield from ev
        if kind is None:
            return
        epoch[kind] += 1
        fk = "all" if one_flag else kind
        e = sum(epoch.values()) if one_flag else epoch[kind]
        for dst in rng.sample(range(G), G):
            if dst != r:
                yield ("flag", fk, dst, e)
        if wait:
            yield ("wait", fk, e)

    for t in range(tokens):
        yield ("local", "x", (t, 0))                                          # the token's embedding row: every rank writes all of x itself
        for l in range(layers):
            yield from launch([("x", (t, l))], [], None)                      # wq|wk|wv: reads x, output stays local
            yield from launch([], [("q", (t, l))], "q")                       # attention: its heads' Q8_0 blocks
            yield from launch([("q", (t, l))], [("x2", (t, l))], "x2")        # wo (+ its rows of x as residual: own slice)
            yield from launch([("x2", (t, l))], [("h", (t, l))], "h")         # w1|w3
            yield from launch([("h", (t, l))], [("x", (t, l + 1))], "x")      # w2 (+ its rows of x2 as residual: own slice)
        yield from launch([("x", (t, layers))], [], None)                     # lm-head reads the last layer's x


def simulate(G, tokens, layers, seed, **variant):
    rng = random.Random(seed)
    mem = [{k: [None] * G for k in KINDS} for _ in range(G)]                  # mem[rank][slot][source slice] = version
    flags = [dict() for _ in range(G)]                                        # flags[rank][(kind, src)] = epoch
    progs = [rank_program(r, G, tokens, layers, random.Random(seed * 1000 + r), **variant) for r in range(G)]
    nxt = [next(p, None) for p in progs]
    steps = 0
    while any(e is not None for e in nxt):
        ready = []
        for r, e in enumerate(nxt):
            if e is None:
                continue
            if e[0] == "wait":
                if all(flags[r].get((e[1], s), 0) >= e[2] for s in range(G) if s != r):
                    ready.append(r)
            else:
                ready.append(r)
        if not ready:
            raise Violation("deadlock: " + repr(nxt))
        r = rng.choice(ready)
        e = nxt[r]
        if e[0] == "read":
            _, slot, src, want = e
            got = mem[r][slot][src]
            if got != want:
                raise Violation(f"rank {r} read {slot}[{src}] = {got}, entitled to {want} (step {steps})")
        elif e[0] == "put":
            _, slot, dst, v = e
            mem[dst][slot][r] = v                                             # this rank's slice of the slot, in rank dst's region
        elif e[0] == "local":
            _, slot, v = e
            mem[r][slot] = [v] * G
        elif e[0] == "flag":
            _, kind, dst, ep = e
            flags[dst][(kind, r)] = max(flags[dst].get((kind, r), 0), ep)
        nxt[r] = next(progs[r], None)
        steps += 1
    return steps


@pytest.mark.parametrize("G", [2, 3, 8])
def test_no_slice_is_read_before_it_arrived_or_after_it_was_overwritten(G):
    for seed in range(120 if G < 8 else 25):
        simulate(G, tokens=3, layers=3, seed=seed)


def test_the_scheduler_finds_the_race_when_the_tail_does_not_wait():
    with pytest.raises(Violation):
        for seed in range(200):
            simulate(2, tokens=2, layers=2, seed=seed, wait=False)


def test_one_flag_word_for_all_kinds_still_orders_the_exchanges():
    """(a variant, not the shipped layout: with a single running count per source rank the waits are at least as strict -- the shipped
    per-kind words exist so that a launch's record is self-contained, not for safety)"""
    for seed in range(60):
        simulate(3, tokens=2, layers=3, seed=seed, one_flag=True)


def test_a_rank_that_stops_publishing_blocks_its_peers_instead_of_letting_them_read_stale_slices():
    """the bounded spin of the kernels turns this into an error after FL_P2P_TIMEOUT_MS (tests/test_model_gpu.py); in the model it is a deadlock"""
    G = 2

    def crippled(seed):
        rng = random.Random(seed)
        mem = [{k: [None] * G for k in KINDS} for _ in range(G)]
        flags = [dict() for _ in range(G)]
        progs = [rank_program(0, G, 2, 2, random.Random(seed)), iter(())]      # rank 1 never runs
        nxt = [next(p, None) for p in progs]
        while nxt[0] is not None:
            e = nxt[0]
            if e[0] == "wait" and not all(flags[0].get((e[1], s), 0) >= e[2] for s in range(G) if s != 0):
                return "blocked"
            if e[0] == "read" and mem[0][e[1]][e[2]] != e[3]:
                return "stale read"
            if e[0] == "put":
                mem[e[2]][e[1]][0] = e[3]
            elif e[0] == "local":
                mem[0][e[1]] = [e[2]] * G
            nxt[0] = next(progs[0], None)
        return "finished"

    assert all(crippled(s) == "blocked" for s in range(20))


def test_multi_pass_row_partition_covers_every_quad_once_in_order():
    """The decode GEMV's multi-pass rows (gemv1_q4_exact_llc.hip, slice_of): a row of NQ quads is taken in npass = ceil(NQ / (NK QPW)) passes, pass p
    owning quads [p NQ / npass, (p + 1) NQ / npass), wave k of the pass its k-th NK-th of those.  The chain order is the row's block order only if these
    slices tile [0, NQ) in ascending (pass, wave) order, and a wave's slice must fit its QPW register quads -- for EVERY row length, not only the LLaMA
    ones the GPU tests run (the formula mirrored here; (NK, QPW) = the shipped forms)."""
    for NK, QPW in ((4, 6), (4, 7), (4, 8), (8, 4)):
        PQ = NK * QPW
        for NQ in range(1, 700):
            npass = (NQ + PQ - 1) // PQ
            nxt = 0
            for p in range(npass):
                lo = (p * NQ) // npass
                nqp = ((p + 1) * NQ) // npass - lo
                assert 0 < nqp <= PQ
                for k in range(NK):
                    qlo = lo + (k * nqp) // NK
                    nq = lo + ((k + 1) * nqp) // NK - qlo
                    assert 0 <= nq <= QPW, (NK, QPW, NQ, p, k, nq)
                    assert qlo == nxt
                    nxt += nq
            assert nxt == NQ
