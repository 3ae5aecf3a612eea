"""`ExecutionRecord::eval_public_values` as data (sp1_amd/machines/public_values.py) and the public-values leg of the oracle's
LogUp-GKR / shard verifiers (oracle/kb_gkr.hpp verify_public_values, kb_shard.hpp) — VERDICT r5 "missing #1".

What pins the transcription: the word layout equals `PublicValues<[T;4],[T;3],[T;4],T>` field by field (offsets recomputed here
from the struct's declaration order and sizes), the assert / send / receive counts equal a hand count of record.rs:L879-L1531,
and semantics: the public values of executed shards (every shard kind of real guests: tests/test_riscv_exec.py) satisfy every
constraint and close every bus, each single wrong word is caught by the constraint or the bus the reference checks it with, and
the cross-shard chain of `SP1Prover::verify` accepts a real run and names the broken link of a tampered one."""
import os
import struct
import sys

import numpy as np
import pytest

import machine_check as MC
import pyoracle as orc
from sp1_amd.machines import public_values as PVM
from sp1_amd.machines import riscv as R
from sp1_amd.machines import riscv_exec as X
from sp1_amd.machines import riscv_trace as RT

torch = pytest.importorskip("torch")


def test_layout_is_the_reference_struct():
    """hypercube/src/air/public_values.rs:L33-L168 without `mprotect`: W1 = [T; 4], W2 = [T; 3], W3 = [T; 4]."""
    W1, W2, W3, T = 4, 3, 4, 1
    decl = [("prev_committed_value_digest", 8 * W1), ("committed_value_digest", 8 * W1), ("prev_deferred_proofs_digest", 8 * T),
            ("deferred_proofs_digest", 8 * T), ("pc_start", W2), ("next_pc", W2), ("prev_exit_code", T), ("exit_code", T),
            ("is_execution_shard", T), ("previous_init_addr", W2), ("last_init_addr", W2), ("previous_finalize_addr", W2),
            ("last_finalize_addr", W2), ("previous_init_page_idx", W2), ("last_init_page_idx", W2), ("previous_finalize_page_idx", W2),
            ("last_finalize_page_idx", W2), ("initial_timestamp", W3), ("last_timestamp", W3), ("is_timestamp_high_eq", T),
            ("inv_timestamp_high", T), ("is_timestamp_low_eq", T), ("inv_timestamp_low", T), ("global_init_count", T),
            ("global_finalize_count", T), ("global_page_prot_init_count", T), ("global_page_prot_finalize_count", T), ("global_count", T),
            ("global_cumulative_sum", 14 * T), ("prev_commit_syscall", T), ("commit_syscall", T), ("prev_commit_deferred_syscall", T),
            ("commit_deferred_syscall", T), ("initial_timestamp_inv", T), ("last_timestamp_inv", T), ("is_first_execution_shard", T),
            ("is_untrusted_programs_enabled", T), ("proof_nonce", 4 * T), ("empty", 4 * T)]
    off = 0
    for name, size in decl:
        assert (PVM.PV[name], PVM.PV_LEN[name]) == (off, size), name
        off += size
    assert off == PVM.NUM_PV_ELTS == 160 and off % 8 == 0 and PVM.PROOF_MAX_NUM_PVS == 187
    # the words the chips read (riscv_more.py: SyscallInstrs) are the same offsets
    from sp1_amd.machines import riscv_more as M
    assert (M.PV_COMMITTED_VALUE_DIGEST, M.PV_DEFERRED_PROOFS_DIGEST, M.PV_EXIT_CODE, M.PV_COMMIT_SYSCALL, M.PV_COMMIT_DEFERRED_SYSCALL) == \
        tuple(PVM.PV[n] for n in ("committed_value_digest", "deferred_proofs_digest", "exit_code", "commit_syscall", "commit_deferred_syscall"))
    # timestamps: [bits 32..48, 24..32, 16..24, 0..16] (public_values.rs:L641-L652); pcs / addresses: three 16-bit limbs
    assert PVM.timestamp_limbs((0xABCD << 32) | (0x12 << 24) | (0x34 << 16) | 0x5679) == [0xABCD, 0x12, 0x34, 0x5679]
    assert PVM.timestamp_of(PVM.timestamp_limbs(123456789012)) == 123456789012
    assert PVM.addr_limbs(0x0001_2345_6789) == [0x6789, 0x2345, 0x0001]


def test_program_counts_follow_the_reference_body():
    """Asserts, by eval_* function (record.rs): empty 4 | state 14 | first execution shard 61 | exit code 2 | committed value digest
    2 + 1 + 1 + 32 + 32 x 32 + 32 | deferred proofs digest 2 + 1 + 1 + 8 + 8 x 8 + 8 | page protection 2. Sends: 6 + 6 byte checks of the
    state, 32 of the digests, 6 + 6 of the address chains, the State send, the three chain heads and the two page-protection heads;
    receives: the State receive and the five chain ends."""
    air, it = PVM.program()
    assert air.main_width == air.prep_width == 0 and it.main_width == PVM.NUM_PV_ELTS
    assert air.num_constraints == 4 + 14 + 61 + 2 + (4 + 32 + 1024 + 32) + (4 + 8 + 64 + 8) + 2
    kinds = lambda lst: sorted(k for k, _, _ in lst)
    chains = [R.GLOBAL_ACC, PVM.MEMORY_GLOBAL_INIT_CONTROL, PVM.MEMORY_GLOBAL_FINALIZE_CONTROL, PVM.PAGE_PROT_GLOBAL_INIT_CONTROL,
              PVM.PAGE_PROT_GLOBAL_FINALIZE_CONTROL]
    assert kinds(it.sends) == sorted([R.BYTE] * 56 + [R.STATE] + chains) and kinds(it.receives) == sorted([R.STATE] + chains)
    # `max_interaction_kinds_values` (verifier.rs:L120-L124): GlobalAccumulation carries 15 values; nothing wider is sent here
    assert PVM.max_interaction_arity() == 16 == max(len(v) + 1 for _, v, _ in it.sends + it.receives)
    # only PUBLIC / CONST loads: the program is a statement about the public words alone
    from sp1_amd.air import LOAD_MAIN, LOAD_PREP
    assert not any(op in (LOAD_MAIN, LOAD_PREP) for op, _, _ in air.instrs)


def _fibonacci_shards(n=300, cycles=3000, with_key=False):
    ex = X.Executor(X.guest_file("fibonacci.elf"), stdin=[struct.pack("<Q", n)])
    out = list(X.program_shards(ex, cycles))
    if with_key:                                               # + the verifying key's digest: the memory image's initialisation
        return out, out[0][5].pc_start, X.verifying_key_words(ex, out[0][5].pc_start)[3:]
    return out, out[0][5].pc_start


def _violations(pv):
    air, _ = PVM.program()
    row = np.asarray(pv, dtype=np.uint64)
    return int(np.count_nonzero(MC.constraint_values(air, None, row[None, :PVM.NUM_PV_ELTS], row)))


def test_every_word_of_an_executed_shard_is_bound():
    """Shard 1 of a fibonacci run (an execution shard in the middle of the program) and the run's memory shard: the public values
    satisfy eval_public_values and close the buses; changing any one of the words the reference constrains breaks a constraint
    or unbalances the bus its message travels on."""
    shards, _ = _fibonacci_shards()
    for kind, machine, tabs, publics, _, _ in (shards[1], shards[-1]):
        assert not _violations(publics) and MC.check_shard(machine, tabs, publics, PVM.program()) == ([], 0)
        bound = ["pc_start", "next_pc", "initial_timestamp", "last_timestamp", "is_execution_shard", "global_count", "global_cumulative_sum",
                 "previous_init_addr", "last_init_addr", "previous_finalize_addr", "last_finalize_addr", "global_init_count",
                 "global_finalize_count", "is_timestamp_high_eq", "is_timestamp_low_eq", "empty"]
        for name in bound:
            for j in range(PVM.PV_LEN[name]):
                pv = publics.clone()
                w = PVM.PV[name] + j
                pv[w] = (pv[w] + 1) % MC.P
                bad, imbalance = MC.check_shard(machine, tabs, pv, PVM.program())
                assert bad or imbalance, (kind, name, j)
    # the words an execution shard may set freely are exactly the ones the NEXT shard's prev_* words must repeat, or the verifying
    # key fixes (is_untrusted_programs_enabled: the two page-protection chain ends cancel at either multiplicity) — verify.rs
    kind, machine, tabs, publics, _, _ = shards[1]
    for name, val in (("exit_code", 7), ("is_untrusted_programs_enabled", 1)):
        pv = publics.clone()
        pv[PVM.PV[name]] = val
        assert MC.check_shard(machine, tabs, pv, PVM.program()) == ([], 0)
    pv[PVM.PV["is_untrusted_programs_enabled"]] = 2
    assert _violations(pv)
    # ... but not a non-execution shard: there prev_exit_code == exit_code, the digests and commit flags stand still
    kind, machine, tabs, publics, _, _ = shards[-1]
    for name in ("exit_code", "commit_syscall", "commit_deferred_syscall", "committed_value_digest", "deferred_proofs_digest"):
        pv = publics.clone()
        pv[PVM.PV[name]] = (pv[PVM.PV[name]] + 1) % 2
        assert _violations(pv), name


def test_state_constraints_case_by_case():
    base = PVM.no_memory_events(PVM.set_global(PVM.set_state(PVM.blank(), 0x200000, 0x200400, 1 + 8 * 5, 1 + 8 * 9000000, 0, True), 0, None))
    assert not _violations(base) and PVM.get(base, "is_first_execution_shard") == 0
    first = PVM.no_memory_events(PVM.set_global(PVM.set_state(PVM.blank(), 0x200000, 0x200400, 1, 1 + 8 * 70, 0, True), 0, None))
    assert not _violations(first) and PVM.get(first, "is_first_execution_shard") == 1
    for name, val in (("prev_exit_code", 1), ("prev_commit_syscall", 1), ("previous_init_addr", [5, 0, 0]), ("prev_deferred_proofs_digest", [0] * 7 + [9])):
        pv = list(first)                                      # the first execution shard starts from nothing
        PVM.put(pv, name, val)
        assert _violations(pv), name
    pv = list(base)                                           # an execution shard's clock moves ...
    PVM.put(pv, "last_timestamp", PVM.get(pv, "initial_timestamp"))
    assert _violations(pv)
    still = PVM.no_memory_events(PVM.set_global(PVM.initialized_state(0x200000), 0, None))
    assert not _violations(still)                             # ... a precompile shard's does not, nor does its pc
    pv = list(still)
    PVM.put(pv, "next_pc", [4, 0x20, 0])
    assert _violations(pv)
    pv = list(base)                                           # a committed digest stays once COMMIT was called or it is non-zero
    PVM.put(pv, "prev_commit_syscall", 1)
    PVM.put(pv, "commit_syscall", 1)
    PVM.put(pv, "committed_value_digest", [3] + [0] * 31)
    assert _violations(pv)
    pv = list(base)
    PVM.put(pv, "prev_committed_value_digest", [0] * 31 + [200])
    assert _violations(pv)
    PVM.put(pv, "committed_value_digest", [0] * 31 + [200])
    assert not _violations(pv)
    pv = list(base)                                           # an exit code that is set stays
    PVM.put(pv, "prev_exit_code", 2)
    PVM.put(pv, "exit_code", 3)
    assert _violations(pv)
    done = PVM.no_memory_events(PVM.set_global(PVM.finalized_state(1 + 8 * 77, PVM.HALT_PC, 0, list(range(1, 9)), [0] * 8), 0, None))
    assert not _violations(done) and PVM.get(done, "commit_syscall") == PVM.get(done, "prev_commit_deferred_syscall") == 1


def test_the_chain_across_shards_is_the_reference_verify():
    shards, entry, vk_digest = _fibonacci_shards(with_key=True)
    kinds = [s[0] for s in shards]
    pvs = [[int(v) for v in s[3]] for s in shards]
    order = X.proof_order(kinds)
    assert PVM.verify_proof_public_values([pvs[i] for i in order], entry, vk_digest) is None
    assert PVM.verify_proof_public_values([pvs[i] for i in order], entry) == "global cumulative sum is not zero"     # without the key's digest
    assert "vk.pc_start" in PVM.verify_proof_public_values([pvs[i] for i in order], entry + 4, vk_digest)
    assert "first execution shard is not set" == PVM.verify_proof_public_values([pvs[i] for i in order[1:]], entry, vk_digest)       # first shard missing
    assert "invalid initial timestamp" == PVM.verify_proof_public_values([pvs[i] for i in [order[0]] + order[2:]], entry, vk_digest) # a middle shard missing
    assert "execution should have halted" in PVM.verify_proof_public_values([pvs[i] for i in order[:-2]], entry, vk_digest)           # stops before HALT
    assert "never initialized" in PVM.verify_proof_public_values([pvs[i] for i in order[:-1]], entry, vk_digest)                      # no memory shard
    swapped = [pvs[i] for i in order]
    swapped[0], swapped[1] = swapped[1], swapped[0]
    assert PVM.verify_proof_public_values(swapped, entry, vk_digest) is not None
    for name, msg in (("global_cumulative_sum", "global cumulative sum is not zero"), ("prev_exit_code", "prev_exit_code"),
                      ("previous_finalize_addr", "previous_finalize_addr"), ("proof_nonce", "proof_nonce")):
        bad = [list(pvs[i]) for i in order]
        bad[1][PVM.PV[name]] = (bad[1][PVM.PV[name]] + 1) % MC.P
        assert msg in PVM.verify_proof_public_values(bad, entry, vk_digest), name
    assert "length" in PVM.verify_proof_public_values([pvs[i][:160] for i in order], entry, vk_digest)


def test_oracle_logup_gkr_verifier_public_values_leg():
    """gkr_verify with the machine's eval_public_values: accepts the shard's own values (cumulative sum = minus what they send),
    code 4 for values that send something else, code 9 for values that break their own constraints; `beta_seed_dim` takes the
    wider of the chips' messages and the widest kind eval_public_values may send."""
    shards, _ = _fibonacci_shards(20, 1 << 20)
    kind, machine, tabs, publics, _, _ = shards[-1]          # the memory shard: 6 chips, the widest chip message has 12 words
    chips = [(i, RT.to_monty_np(tabs[a.name][1]), RT.to_monty_np(tabs[a.name][0]) if tabs[a.name][0] is not None else None) for a, i in machine]
    heights = [c[1].shape[0] for c in chips]
    L = 17
    ch = orc.Challenger()
    v = ch.clone()
    orc.set_gkr_sparse(True)
    try:
        blob = orc.gkr_prove(chips, L, ch)
    finally:
        orc.set_gkr_sparse(False)
    shapes = [(i, np.zeros((0, i.main_width), np.uint32), None) for i, _, _ in chips]
    pvp = PVM.verifier_program()
    assert orc.gkr_verify(shapes, heights, L, blob, v.clone(), pvp, RT.to_monty_np(publics)) == 0
    assert orc.gkr_verify(shapes, heights, L, blob, v.clone()) == 4                    # a machine without public-value interactions: sum != 0
    other = publics.clone()
    other[PVM.PV["last_finalize_addr"]] += 8
    assert orc.gkr_verify(shapes, heights, L, blob, v.clone(), pvp, RT.to_monty_np(other)) == 4
    other = publics.clone()
    other[PVM.PV["commit_syscall"]] = 0
    assert orc.gkr_verify(shapes, heights, L, blob, v.clone(), pvp, RT.to_monty_np(other)) == 9
    assert orc.gkr_verify(shapes, heights, L, blob, v.clone(), pvp, RT.to_monty_np(publics[:100])) == 9   # fewer words than the machine has
    # `beta_seed_dim` = max(chips' widest message, the widest kind eval_public_values may send) (verifier.rs:L112-L126): every
    # cluster of the RISC-V machine holds the Program chip, whose 17-word message is wider than GlobalAccumulation's 16, so the
    # second term never binds here — which is also why the reference's PROVER can ignore it (prover.rs:L84)
    assert max(len(vs) + 1 for _, i in machine for _, vs, _ in i.sends + i.receives) == 17 > PVM.max_interaction_arity()
