"""Replays tests/golden/kb_shrink_transcript.npz (the reference's real shard-proof transcript, see
tests/golden/make_transcript.py) through any challenger with the observe / sample / sample_bits /
check_witness interface on Montgomery words."""
import os

import numpy as np

import pyoracle as orc

TAPE = np.load(os.path.join(os.path.dirname(__file__), "golden", "kb_shrink_transcript.npz"))


def replay(ch, stop_before_op=None):
    """Returns (#ops replayed, #pinned sample/witness checks). Raises AssertionError on the first mismatch."""
    ops, data = TAPE["ops"], TAPE["data"]
    data_m = orc.to_monty(data)
    pinned = 0
    for k, (op, arg, off, pin) in enumerate(ops):
        if stop_before_op is not None and k == stop_before_op:
            return k, pinned
        if op == 0:
            if arg:                                          # arg == 0: a barrier between two absorbs
                ch.observe(data_m[off:off + arg])
        elif op == 1:
            got = orc.from_monty(np.array([ch.sample() for _ in range(arg)], dtype=np.uint32))
            assert np.array_equal(got, data[off:off + arg]), ("sample", k)
            pinned += int(pin)
        elif op == 2:
            assert ch.sample_bits(int(arg)) == int(data[off]), ("sample_bits", k)
            pinned += int(pin)
        else:
            assert ch.check_witness(int(arg), int(data_m[off])), ("check_witness", k)
            pinned += 1
    return len(ops), pinned


def pow_op_index(bits=16):
    """Index of the CHECK_WITNESS op with the given bit count (the BaseFold query-phase grind)."""
    ops = TAPE["ops"]
    idx = [k for k, (op, arg, _, _) in enumerate(ops) if op == 3 and arg == bits]
    assert len(idx) == 1
    return idx[0]


def gkr_proof_bytes():
    a, b = (int(x) for x in TAPE["gkr_range"])
    return TAPE["shard_head"].tobytes()[a:b]


def shard_proof_bytes():
    """bincode(ShardProof) of the reference's real proof with the BaseFold openings restricted to 12 queries."""
    return TAPE["shard_head"].tobytes() + TAPE["basefold_proof_q12"].tobytes() + TAPE["jagged_tail"].tobytes()
