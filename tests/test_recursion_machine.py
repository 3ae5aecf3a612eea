"""The hand-transcribed recursion (compress / shrink) machine, pinned on the reference's own proof.

`sp1_amd/machines/recursion.py` restates the eight `Air::eval` bodies of `RecursionAir::compress_machine()`
(/root/reference/crates/recursion/machine/src/machine.rs:L89-L105) as constraint / interaction programs. The proof in
/root/reference/sp1-gpu/crates/perf/recursion_records/shrink_input.bin was made by the Rust prover over that machine,
so the oracle's restated `ShardVerifier::verify_shard` run in FULL mode on its bytes checks exactly the two
chip-dependent equations the transcript-only tests could not:
  * the zerocheck closing equation  (/root/reference/crates/hypercube/src/verifier/shard.rs:L343-L396):
    Σ_chips λ-RLC of eq(ζ, r)·(C_chip(openings; α) − padded_row_adjustment·geq + Σ_j γ^(j+1)·opening_j) == final eval;
  * the LogUp-GKR interaction check (/root/reference/crates/hypercube/src/logup_gkr/verifier.rs:L268-L352):
    numerator / denominator of the circuit's input layer recomputed from every interaction of every chip at the
    opened columns.
Acceptance therefore pins constraint order and signs, column layouts, interaction order / kinds / multiplicities of
the transcription — and the oracle's chip-dependent verifier code — on reference-produced data."""
import copy
import json

import numpy as np
import pytest

import machine_check as MC
import pyoracle as orc
import transcript_tape as tt
from sp1_amd import machine as machine_fmt
from sp1_amd.air import ASSERT_ZERO, VCol
from sp1_amd.machines import load_recursion_compress, recursion as R, recursion_trace as RT
from test_oracle_golden import GOLD, M

T = tt.TAPE
SHAPES = [("BaseAlu", 8, 3), ("ExtAlu", 8, 12), ("MemoryConst", 6, 1), ("MemoryVar", 4, 8), ("Poseidon2WideDeg3", 49, 179),
          ("PrefixSumChecks", 9, 15), ("PublicValues", 10, 1), ("Select", 8, 5)]     # read off the real proof's openings


def _shapes_only(machine):
    return [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None)
            for a, i in machine]


def _verify_real_proof(machine):
    k = int(T["shard_start_op"])
    ch = orc.Challenger()
    assert tt.replay(ch, stop_before_op=k)[0] == k                     # vk.observe_into
    L, lsh = len(T["jagged_z_row"]), len(T["stack_point"])
    rc = orc.shard_verify(_shapes_only(machine), M(GOLD["vk_preprocessed_commit"]), tt.shard_proof_bytes(), L, lsh, ch, 2, 12, 16)
    return rc, ch


def test_machine_shape_matches_the_real_proof():
    m = R.compress_machine()
    assert [(a.name, a.prep_width, a.main_width) for a, _ in m] == SHAPES
    assert [(a.name, i.name) for a, i in m] == [(n, n) for n, _, _ in SHAPES]
    p2 = dict((a.name, (a, i)) for a, i in m)["Poseidon2WideDeg3"]
    assert p2[0].num_constraints == 1 + 8 * 16 + 19 + 16 and (len(p2[1].sends), len(p2[1].receives)) == (16, 16)


def test_full_verifier_accepts_the_reference_shard_proof_with_the_transcribed_machine():
    rc, ch = _verify_real_proof(R.compress_machine())
    assert rc == 0
    # ... and ends in the tape's transcript state (12 of the 124 query indices sampled)
    ref = orc.Challenger()
    stop = len(T["ops"]) - 124 + 12
    assert tt.replay(ref, stop_before_op=stop)[0] == stop
    assert np.array_equal(ref.state(), ch.state())


def test_json_dump_is_current_and_verifies():
    m = load_recursion_compress()
    fresh = machine_fmt.dump_machine(R.compress_machine())
    with open(R.__file__.replace("recursion.py", "recursion_compress.json")) as f:
        on_disk = json.load(f)
    on_disk.pop("source")
    assert on_disk == json.loads(json.dumps(fresh)), "run python -m sp1_amd.machines.dump"
    assert _verify_real_proof(m)[0] == 0


def _flip_constraint_sign(machine, chip, which):
    air = dict((a.name, a) for a, _ in machine)[chip]
    _reorder_asserts(air, list(range(air.num_constraints)), negate=which)


@pytest.mark.parametrize("chip,which", [("Select", 1), ("BaseAlu", 2), ("ExtAlu", 9), ("PrefixSumChecks", 3), ("PublicValues", 0),
                                        ("Poseidon2WideDeg3", 77), ("Poseidon2WideDeg3", 140)])
def test_a_flipped_constraint_sign_is_rejected_by_the_real_proof(chip, which):
    """The real proof distinguishes x - y from y - x in every chip: the zerocheck closing equation fails (code 2xx)."""
    m = copy.deepcopy(R.compress_machine())
    _flip_constraint_sign(m, chip, which)
    rc, _ = _verify_real_proof(m)
    assert 200 <= rc < 300


def _reorder_asserts(air, perm, negate=None):
    """All value instructions first (SSA order kept), then the asserts in the order perm (one optionally negated)."""
    remap, out, asserts = {}, [], []
    for j, (op, x, y) in enumerate(air.instrs):
        if op == ASSERT_ZERO:
            asserts.append(x)
            continue
        remap[j] = len(out)
        if op >= 4:
            x = remap[x]
            if op in (4, 5, 6):
                y = remap[y]
        out.append((op, x, y))
    ops = [remap[asserts[k]] for k in perm]
    if negate is not None:
        out.append((7, ops[negate], 0))                               # NEG
        ops[negate] = len(out) - 1
    air.instrs = out + [(ASSERT_ZERO, x, 0) for x in ops]
    air.__dict__.pop("_array_cache", None)                            # to_array() memoises per instruction count


def test_swapped_constraint_order_is_rejected_by_the_real_proof():
    m = copy.deepcopy(R.compress_machine())
    air = dict((a.name, a) for a, _ in m)["Select"]
    _reorder_asserts(air, [0, 1, 2])
    assert _verify_real_proof(m)[0] == 0                              # hoisting the values changes nothing
    _reorder_asserts(air, [1, 0, 2])
    assert 200 <= _verify_real_proof(m)[0] < 300


@pytest.mark.parametrize("what", ["memconst_layout", "sends_order", "send_receive_role", "single_padding"])
def test_a_wrong_interaction_is_rejected_by_the_real_proof(what):
    m = copy.deepcopy(R.compress_machine())
    its = dict((i.name, i) for _, i in m)
    if what == "memconst_layout":                                     # (addr, value) instead of (value, addr)
        kind, vals, mult = its["MemoryConst"].sends[0]
        its["MemoryConst"].sends[0] = (kind, [VCol.prep(0)] + [VCol.prep(1 + i) for i in range(4)], mult)
    elif what == "sends_order":
        s = its["Poseidon2WideDeg3"].sends
        s[0], s[1] = s[1], s[0]
    elif what == "send_receive_role":
        i = its["Select"]
        i.sends, i.receives = i.receives, i.sends
    else:                                                             # receive_single pads with zeros, not with the value
        kind, vals, mult = its["BaseAlu"].receives[0]
        its["BaseAlu"].receives[0] = (kind, [vals[0], vals[1], vals[1], vals[1], vals[1]], mult)
    for _, i in m:
        i.__dict__.pop("_array_cache", None)
    rc, _ = _verify_real_proof(m)
    assert 100 <= rc < 200


COUNTS = {"BaseAlu": 70, "ExtAlu": 90, "MemoryConst": 50, "MemoryVar": 40, "Poseidon2WideDeg3": 20, "PrefixSumChecks": 33,
          "Select": 100}


def test_generated_traces_satisfy_the_machine_row_by_row():
    tabs, pv = RT.generate(COUNTS, seed=3)
    pvc = MC.from_monty(pv)
    bus = []
    for air, it in R.compress_machine():
        prep, main = (MC.from_monty(x) for x in tabs[air.name])
        assert prep.shape[0] == main.shape[0] and prep.shape[0] % 16 == 0
        assert not MC.constraint_values(air, prep, main, pvc).any(), air.name
        bus.append((it, prep, main))
    assert MC.bus_imbalance(bus) == {}
    # and the checker itself notices a broken cell / a broken multiplicity
    prep, main = (MC.from_monty(x) for x in tabs["ExtAlu"])
    main[5, 4] = (main[5, 4] + 1) % MC.PP                             # in1[0]: read with multiplicity 1
    ext = dict((a.name, a) for a, _ in R.compress_machine())["ExtAlu"]
    assert MC.constraint_values(ext, prep, main, pvc).any()
    assert MC.bus_imbalance([(i, MC.from_monty(tabs[i.name][0]), main if i.name == "ExtAlu" else MC.from_monty(tabs[i.name][1]))
                             for _, i in R.compress_machine()]) != {}


def test_vectorised_poseidon2_rows_match_the_oracle_permutation():
    rng = np.random.default_rng(5)
    x = rng.integers(0, MC.P, size=(7, 16)).astype(np.uint64)
    rows = RT.poseidon2_rows(x)
    for r in range(7):
        want = orc.from_monty(orc.permute(orc.to_monty(x[r].astype(np.uint32))))
        assert np.array_equal(rows[r, R.P2_OUT(0):R.P2_OUT(0) + 16], want.astype(np.uint64))


def test_ext_inverse():
    rng = np.random.default_rng(6)
    a = rng.integers(0, MC.P, size=(50, 4)).astype(np.uint64)
    one = RT.ext_mul(a, RT.ext_inv(a))
    assert np.array_equal(one, np.tile(np.array([1, 0, 0, 0], np.uint64), (50, 1)))


@pytest.mark.parametrize("seed,L,lsh,batch", [(1, 8, 7, 4), (2, 9, 5, 3)])
def test_oracle_proves_and_fully_verifies_a_recursion_shard(seed, L, lsh, batch):
    tabs, pv = RT.generate(COUNTS, seed=seed)
    m = R.compress_machine()
    chips = [(a, i, tabs[a.name][1], tabs[a.name][0]) for a, i in m]
    prep = orc.JaggedRound([c[3] for c in chips], L, lsh, batch, 1)
    ch = orc.Challenger()
    ch.observe(prep.commit)
    v = ch.clone()
    blob = orc.shard_prove(chips, pv, prep, L, lsh, batch, ch, 1, 5, 4)
    assert orc.shard_verify(_shapes_only(m), prep.commit, blob, L, lsh, v, 1, 5, 4) == 0
    assert np.array_equal(v.state(), ch.state())
    # one wrong cell in a Poseidon2 row: the proof of the broken trace is rejected
    bad = {k: (p.copy(), mm.copy()) for k, (p, mm) in tabs.items()}
    bad["Poseidon2WideDeg3"][1][3, 150] ^= 1
    chips = [(a, i, bad[a.name][1], bad[a.name][0]) for a, i in m]
    ch = orc.Challenger()
    ch.observe(prep.commit)
    v = ch.clone()
    blob = orc.shard_prove(chips, pv, prep, L, lsh, batch, ch, 1, 5, 4)
    assert orc.shard_verify(_shapes_only(m), prep.commit, blob, L, lsh, v, 1, 5, 4) != 0


def test_the_wrap_machine_chips_compose_the_permutation_and_balance_their_memory():
    """The wrap machine replaces the wide Poseidon2 chip by Poseidon2LinearLayer / Poseidon2SBox / ExtFeltConvert rows
    (recursion/machine/src/chips/poseidon2_helper/). Here one permutation is laid out as such rows — every value that crosses
    between rows goes through the Memory bus at a fresh address —: the linear layers and cubes, taken from the chips' own
    interaction expressions, compose to the oracle's permutation; every constraint vanishes; with the round-constant additions
    (BaseAlu's job in a real program) stood in for by MemoryVar rows, the bus balances."""
    rng = np.random.default_rng(11)
    chips = dict((a.name, (a, i)) for a, i in R.wrap_machine())
    assert sorted(chips) == ["BaseAlu", "ExtAlu", "ExtFeltConvert", "MemoryConst", "MemoryVar", "Poseidon2LinearLayer", "Poseidon2SBox", "PublicValues", "Select"]
    rc = R._round_constants()
    lin_it, sbox_it, conv_it = chips["Poseidon2LinearLayer"][1], chips["Poseidon2SBox"][1], chips["ExtFeltConvert"][1]
    lin_rows, sbox_rows, var_events, next_addr = [], [], [], [100]

    def fresh(n):
        next_addr[0] += n
        return list(range(next_addr[0] - n, next_addr[0]))

    def linear(state, addrs, external):
        """One Poseidon2LinearLayer row reading `state` (16 values at 4 block addresses); returns the values its send carries."""
        out_addrs = fresh(4)
        prep = np.array(addrs + out_addrs + [int(external), int(not external)], dtype=np.uint64)
        main = np.array(state, dtype=np.uint64)
        lin_rows.append((prep, main))
        sends = [s for s in lin_it.sends if s[2].apply(prep, main) == 1]
        assert len(sends) == 4
        vals = [v.apply(prep, main) for s in sends for v in s[1][1:]]
        assert [s[1][0].apply(prep, main) for s in sends] == out_addrs
        return vals, out_addrs

    def sbox(block, addr, external):
        out_addr = fresh(1)[0]
        cubes = [pow(int(x), 3, MC.P) for x in block]
        prep = np.array([addr, out_addr, int(external), int(not external)], dtype=np.uint64)
        main = np.array(list(block) + cubes, dtype=np.uint64)
        sbox_rows.append((prep, main))
        (send,) = [s for s in sbox_it.sends if s[2].apply(prep, main) == 1]
        return [v.apply(prep, main) for v in send[1][1:]], out_addr

    def add_constants(state, addrs, consts):
        """Stand-in for the BaseAlu rows that add round constants: the old blocks are consumed, new ones supplied (MemoryVar
        multiplicities -1 / +1 keep the bus honest about what was read and written)."""
        new = [(int(x) + int(c)) % MC.P for x, c in zip(state, consts)]
        out = fresh(4)
        for k in range(4):
            var_events.append((addrs[k], state[4 * k:4 * k + 4], MC.P - 1))
            var_events.append((out[k], new[4 * k:4 * k + 4], 1))
        return new, out

    x = [int(v) for v in rng.integers(0, MC.P, size=16)]
    addrs = fresh(4)
    for k in range(4):
        var_events.append((addrs[k], x[4 * k:4 * k + 4], 1))             # the input appears in memory
    state, addrs = linear(x, addrs, True)
    for r in list(range(4)) + ["internal"] + list(range(4, 8)):
        if r == "internal":
            for j in range(20):
                state, addrs = add_constants(state, addrs, [rc[4 + j][0]] + [0] * 15)
                blk, a0 = sbox(state[:4], addrs[0], False)
                state, addrs = linear(blk + state[4:], [a0] + addrs[1:], False)
            continue
        state, addrs = add_constants(state, addrs, rc[r] if r < 4 else rc[24 + (r - 4)])
        blocks = [sbox(state[4 * k:4 * k + 4], addrs[k], True) for k in range(4)]
        state, addrs = linear([v for b, _ in blocks for v in b], [a for _, a in blocks], True)
    want = orc.from_monty(orc.permute(orc.to_monty(np.array(x, dtype=np.uint32))))
    assert state == [int(v) for v in want]
    # ExtFeltConvert: the first output block read as an extension element, its coordinates written as four base elements
    felt_addrs = fresh(4)
    conv_prep = np.array([addrs[0]] + felt_addrs + [1, 1, 1, 1, 1], dtype=np.uint64)
    conv_main = np.array(state[:4], dtype=np.uint64)
    for k in range(1, 4):
        var_events.append((addrs[k], state[4 * k:4 * k + 4], MC.P - 1))   # the rest of the output is consumed
    for k in range(4):
        var_events.append((felt_addrs[k], [state[k], 0, 0, 0], MC.P - 1))
    # tables: constraints vanish, the Memory bus balances
    tables = []
    for name, rows in (("Poseidon2LinearLayer", lin_rows), ("Poseidon2SBox", sbox_rows), ("ExtFeltConvert", [(conv_prep, conv_main)])):
        air, it = chips[name]
        prep, main = np.stack([p for p, _ in rows]), np.stack([m for _, m in rows])
        assert not MC.constraint_values(air, prep, main, np.zeros(R.NUM_PUBLIC_VALUES, dtype=np.uint64)).any(), name
        tables.append((it, prep, main))
    if len(var_events) % 2:
        var_events.append((0, [0, 0, 0, 0], 0))
    ev = np.array([[a, m] + list(v) for a, v, m in var_events], dtype=np.uint64).reshape(-1, 2, 6)
    var_prep = ev[:, :, :2].reshape(-1, 4)
    var_main = ev[:, :, 2:].reshape(-1, 8)
    tables.append((chips["MemoryVar"][1], var_prep, var_main))
    assert MC.bus_imbalance(tables) == {}
    sbox_main = tables[1][2].copy()
    sbox_main[3, 5] = (sbox_main[3, 5] + 1) % MC.P                            # one cube off by one: its constraint and the bus both see it
    assert MC.constraint_values(chips["Poseidon2SBox"][0], tables[1][1], sbox_main, np.zeros(R.NUM_PUBLIC_VALUES, dtype=np.uint64)).any()
    assert MC.bus_imbalance([tables[0], (tables[1][0], tables[1][1], sbox_main)] + tables[2:]) != {}
