"""GPU parity (-m gpu) on the field / curve precompile chips added at the end of round 5 (sp1hip_rv64_precompile_events): proof
bytes == the oracle prover's and the oracle's verify_shard accepts, shard kind by shard kind — the four kinds whose constraint
programs differ most from the secp256k1 / UINT256_MUL chips that tests/test_gpu_riscv_exec.py already proves on the GPU:

    bls12381_fp     48-limb operands, witness offset 2^15, the operation-selected polynomial of FieldOpCols::eval_variable
    bn254_fp        the same chip at 32 limbs (103 registers in the planner's kept-column orders: it takes the rematerialising
                    schedule since the threshold moved to 64)
    ed_decompress   FieldSqrtCols (a FieldOpCols checked against another operation's result), one-coefficient operands
    uint256_ops     a + b + c / a * b + c with carry: five memory slices, three register reads, modulus 2^256

The file sorts last on purpose: it was written when the round's GPU minutes were spent (the CPU side — every constraint on every
row, every bus, the zerocheck planner's compiled program against the SSA — is tests/test_riscv_precompiles.py and
tests/test_zc_compiler.py), so its first run is the driver's; a failure here does not hide the rest of the suite behind `-x`."""
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

import rv_asm as A  # noqa: E402
from sp1_amd.machines import riscv_exec as X  # noqa: E402
from sp1_amd.machines import riscv_more as M  # noqa: E402
from test_gpu_riscv_exec import prove_both  # noqa: E402
from test_riscv_precompiles import DATA, ED_B, call, ed_add, words  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from sp1_amd import api as a
    torch.cuda.set_device(0)
    return a


def test_the_tower_edwards_and_carry_precompile_shards_match_the_oracle(api):
    bls, bn = M.BLS12381_P, M.BN254_P
    b2 = ed_add(ED_B, ED_B)
    top = (1 << 256) - 1
    data = words(3 ** 230 % bls, 6) + words(bls - 5, 6)                       # 0: bls12-381 Fp operands (96 bytes)
    data += words(7 ** 90 % bn, 4) + words(bn - 11, 4)                       # 96: bn254 Fp operands (64 bytes)
    data += bytes(32) + words(b2[1], 4)                                      # 160: ed25519 x slot, y(2B)
    data += words(top, 4) + words(top - 5, 4) + words(top, 4) + bytes(64)    # 224: a, b, c, d, e
    prog = A.li(28, DATA)
    prog += call(0x00010122, 0, 48) + call(0x00010121, 0, 48)                # bls12-381: x <- x * y, then x <- x - y
    prog += call(0x00010126, 96, 128) + call(0x00010128, 96, 96)             # bn254: x <- x + y, then x <- x * x
    prog += [A.enc("addi", 10, 28, 160), A.enc("addi", 11, 0, 1)] + A.li(5, 0x00000108) + [A.enc("ecall")]
    prog += [A.enc("addi", 12, 28, 288), A.enc("addi", 13, 28, 320), A.enc("addi", 14, 28, 352)] + call(0x00010131, 224, 256)
    prog += [A.enc("addi", 12, 28, 288), A.enc("addi", 13, 28, 224), A.enc("addi", 14, 28, 352)] + call(0x00010130, 224, 256)   # d over a
    ex = X.Executor(A.elf(prog + A.halt(0), data=data + bytes(32)), stdin=[])
    seen, gevs = [], []
    for kind, machine, tabs, publics, gev, sh in X.program_shards(ex, 1 << 20, device="cuda"):
        seen.append(kind)
        gevs.append(gev)
        if kind not in ("core", "memory"):                                   # those two kinds are test_gpu_riscv_exec.py's
            prove_both(api, machine, tabs, publics, 17, 12, 8, 1, 5, 4)
    assert seen == ["core", "bn254_fp", "bls12381_fp", "ed_decompress", "uint256_ops", "memory"]
    assert not X.global_events_balance(gevs + [X.image_events(ex)])


def test_field_operation_pieces_over_several_workgroups_match_the_oracle(api):
    """The polynomial-identity pieces of FieldOpCols (zc_poly.hpp) at a height where a round has several workgroups per identity
    and the last one is partial: 700 SECP256K1_DOUBLE calls and 700 bn254 Fp multiplications in a loop (the bivariate rounds see
    175 row quads, round 2 88 row pairs, ... down to one; three-factor terms with selectors in the Fp chip). Proof bytes == the
    oracle's, the verifier accepts with the shards' public values."""
    import struct
    G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798, 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
    M64 = (1 << 64) - 1
    w4 = lambda v: b"".join(struct.pack("<Q", (v >> (64 * i)) & M64) for i in range(4))
    bn = M.BN254_P
    data = w4(G[0]) + w4(G[1]) + w4(7 ** 90 % bn) + w4(bn - 11)              # 0: the point, 64: the Fp operands
    body = [A.enc("addi", 10, 28, 0), A.enc("addi", 11, 0, 0)] + A.li(5, 0x0000010B) + [A.enc("ecall")]        # p <- 2 p
    body += call(0x00010128, 64, 96)                                                                           # x <- x * y over bn254's Fp
    n = 700
    prog = A.li(28, DATA) + A.li(29, n) + body + [A.enc("addi", 29, 29, -1), A.enc("bne", 29, 0, -4 * (len(body) + 1))]
    ex = X.Executor(A.elf(prog + A.halt(0), data=data + bytes(32)), stdin=[])
    seen, gevs = [], []
    for kind, machine, tabs, publics, gev, sh in X.program_shards(ex, 1 << 22, device="cuda"):
        seen.append(kind)
        gevs.append(gev)
        if kind in ("secp256k1_double", "bn254_fp"):
            assert int(tabs[{"secp256k1_double": "Secp256k1DoubleAssign", "bn254_fp": "Bn254FpOpAssign"}[kind]][1].shape[0]) == -(-n // 32) * 32      # (heights are multiples of 32)
            prove_both(api, machine, tabs, publics, 17, 13, 8, 1, 5, 4)
    assert seen == ["core", "secp256k1_double", "bn254_fp", "memory"]
    assert not X.global_events_balance(gevs + [X.image_events(ex)])
