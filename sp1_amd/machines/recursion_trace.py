"""Satisfying traces for the recursion (compress / shrink) machine of `recursion.py`, generated on the host.

The reference fills these tables from the `ExecutionRecord` of its recursion VM
(/root/reference/crates/recursion/machine/src/chips/*: `generate_preprocessed_trace_into` from the program's
instructions, `generate_trace_into` from the events; Poseidon2 rows by
/root/reference/crates/hypercube/src/operations/poseidon2/trace.rs:L29-L152). That VM (and the recursion programs it
runs) is Rust and out of scope; this module produces what it would hand to the prover for a *random straight-line
recursion program* of a requested shape: every row satisfies its chip's constraints and the Memory bus balances
(each address is written once with multiplicity = the number of later reads, reads carry multiplicity 1).
It is vectorised (numpy, canonical integers -> Montgomery words at the end) so that the reference's real compress
shape (`REFERENCE_COMPRESS_HEIGHTS`, 8.9e7 cells, read off the proof in shrink_input.bin) is generated in seconds.

Layout conventions are those of `recursion.py` (column maps cited there). Row padding follows the reference:
all-zero rows, except Poseidon2 padding rows, which hold the permutation trace of the zero state
(poseidon2_wide/trace.rs:L76-L83) under an all-zero preprocessed row.
"""
import numpy as np

from .recursion import INTERNAL_DIAG, NUM_PUBLIC_VALUES, P, P2_EXT, P2_INT, P2_OUT, P2_S0, P2_WIDTH, PV_DIGEST_OFFSET, R_INV, \
    _round_constants

U = np.uint64
PP = U(P)

# heights of the eight tables in the reference's real compress proof (tests/golden: shrink_input.bin)
REFERENCE_COMPRESS_HEIGHTS = {"BaseAlu": 592384, "ExtAlu": 795264, "MemoryConst": 492320, "MemoryVar": 661984,
                              "Poseidon2WideDeg3": 158368, "PrefixSumChecks": 224608, "PublicValues": 16, "Select": 1087808}


def to_monty(x):
    return ((np.asarray(x, dtype=U) << U(32)) % PP).astype(np.uint32)


def _mul(a, b):
    return (a * b) % PP


def _pow(a, e):
    r = np.ones_like(a)
    while e:
        if e & 1:
            r = _mul(r, a)
        a = _mul(a, a)
        e >>= 1
    return r


def _inv(a):
    return _pow(a, P - 2)


def ext_mul(a, b):
    """[n, 4] x [n, 4] in F[x]/(x^4 - 3), canonical uint64."""
    out = np.zeros_like(a)
    for i in range(4):
        for j in range(4):
            t = _mul(a[:, i], b[:, j])
            if i + j >= 4:
                t = (t * U(3)) % PP
            out[:, (i + j) % 4] = (out[:, (i + j) % 4] + t) % PP
    return out


def ext_inv(a):
    """a = A + B x with A = a0 + a2 y, B = a1 + a3 y, y = x^2, y^2 = 3: a (A - B x) = A^2 - y B^2 in F[y]."""
    def qmul(u, v):                                    # (u0 + u1 y)(v0 + v1 y)
        return ((_mul(u[0], v[0]) + U(3) * _mul(u[1], v[1])) % PP, (_mul(u[0], v[1]) + _mul(u[1], v[0])) % PP)
    A, B = (a[:, 0], a[:, 2]), (a[:, 1], a[:, 3])
    A2, B2 = qmul(A, A), qmul(B, B)
    n0 = (A2[0] + PP * U(3) - U(3) * B2[1] % PP) % PP       # A^2 - y B^2, y (b0 + b1 y) = 3 b1 + b0 y
    n1 = (A2[1] + PP - B2[0]) % PP
    d = _inv((_mul(n0, n0) + PP * U(3) - U(3) * _mul(n1, n1) % PP) % PP)
    i0, i1 = _mul(n0, d), _mul((PP - n1) % PP, d)           # 1 / (n0 + n1 y)
    c = np.stack([A[0], (PP - B[0]) % PP, A[1], (PP - B[1]) % PP], axis=1)   # A - B x
    q = np.stack([i0, np.zeros_like(i0), i1, np.zeros_like(i0)], axis=1)
    return ext_mul(c, q)


def _ext_linear(s):
    """[n, 16] external linear layer (circulant of M4 + column sums), canonical."""
    t = np.empty_like(s)
    for j in range(0, 16, 4):
        x0, x1, x2, x3 = (s[:, j + k] for k in range(4))
        t[:, j] = (2 * x0 + 3 * x1 + x2 + x3) % PP
        t[:, j + 1] = (x0 + 2 * x1 + 3 * x2 + x3) % PP
        t[:, j + 2] = (x0 + x1 + 2 * x2 + 3 * x3) % PP
        t[:, j + 3] = (3 * x0 + x1 + x2 + 2 * x3) % PP
    sums = (t[:, 0:4] + t[:, 4:8] + t[:, 8:12] + t[:, 12:16]) % PP
    return (t + np.tile(sums, 4)) % PP


def poseidon2_rows(inputs):
    """populate_perm (operations/poseidon2/trace.rs:L29-L152): [n, 16] canonical inputs -> [n, 179] canonical rows."""
    rc = np.array(_round_constants(), dtype=U)
    n = inputs.shape[0]
    row = np.zeros((n, P2_WIDTH), dtype=U)
    cube = lambda x: _mul(_mul(x, x), x)
    s = inputs.astype(U)
    for r in range(8):
        row[:, P2_EXT(r, 0):P2_EXT(r, 0) + 16] = s
        if r == 0:
            s = _ext_linear(s)
        s = _ext_linear(cube((s + rc[r if r < 4 else 24 + r - 4]) % PP))
        if r == 3:
            row[:, P2_INT(0):P2_INT(0) + 16] = s
            diag = np.array([(d * R_INV) % P for d in INTERNAL_DIAG], dtype=U)
            for k in range(20):
                s[:, 0] = cube((s[:, 0] + rc[4 + k][0]) % PP)
                tot = _mul(s.sum(axis=1) % PP, U(R_INV))
                s = (tot[:, None] + _mul(s, diag[None, :])) % PP
                if k < 19:
                    row[:, P2_S0(k)] = s[:, 0]
    row[:, P2_OUT(0):P2_OUT(0) + 16] = s
    return row


def _pad32(n):
    return max(-(-n // 32) * 32, 16)


class _Memory:
    """Write-once addresses with read counting. Values are Blocks (4 felts); a 'felt' address holds [v, 0, 0, 0]."""

    def __init__(self, rng):
        self.rng = rng
        self.vals = np.zeros((0, 4), dtype=U)
        self.reads = np.zeros(0, dtype=np.int64)
        self.felt, self.bits = np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)

    def write(self, blocks, felt=False, bit=False):
        a0 = self.vals.shape[0]
        self.vals = np.concatenate([self.vals, blocks.astype(U)])
        self.reads = np.concatenate([self.reads, np.zeros(blocks.shape[0], np.int64)])
        addrs = np.arange(a0, a0 + blocks.shape[0], dtype=np.int64)
        if felt:
            self.felt = np.concatenate([self.felt, addrs])
        if bit:
            self.bits = np.concatenate([self.bits, addrs])
        return addrs

    def pick(self, n, pool=None):
        src = np.arange(self.vals.shape[0]) if pool is None else pool
        addrs = src[self.rng.integers(0, len(src), size=n)]
        np.add.at(self.reads, addrs, 1)
        return addrs


def generate(counts, seed=0, digest=None):
    """counts: {chip name: number of real rows} (MemoryVar rows hold 2 events each; PublicValues is always 16 rows
    with 8 real ones). digest: 8 canonical field elements the shard commits as its public-values digest (written by 8
    extra MemoryConst rows and read by the PublicValues chip); random memory cells by default. Returns
    ({name: (prep, main)} row-major Montgomery uint32 tables padded like the reference, public values [187] Montgomery)."""
    rng = np.random.default_rng(seed)
    mem = _Memory(rng)
    rnd = lambda *shape: rng.integers(0, P, size=shape, dtype=np.int64).astype(U)
    n = {k: int(counts.get(k, 0)) for k in REFERENCE_COMPRESS_HEIGHTS}
    # addresses are offset so that address 0 (what padding rows "point at" with multiplicity 0) is never written
    A0 = 1

    def blocks_of(k, kind):            # kind 0: ext, 1: felt, 2: bit
        b = rnd(k, 4)
        b[kind >= 1, 1:] = 0
        b[kind == 2, 0] = rng.integers(0, 2, size=int((kind == 2).sum())).astype(U)
        return b

    def write_mixed(k):
        kind = rng.integers(0, 3, size=k)
        if k >= 3:
            kind[:3] = (0, 1, 2)       # every pool non-empty
        b = blocks_of(k, kind)
        addrs = mem.write(b)
        mem.felt = np.concatenate([mem.felt, addrs[kind >= 1]])
        mem.bits = np.concatenate([mem.bits, addrs[kind == 2]])
        return addrs, b

    writers = []                       # (table name, prep column of the multiplicity, row indices, addresses)
    T = {}
    # MemoryConst: prep = value[4], addr, mult
    k = n["MemoryConst"]
    addrs, b = write_mixed(k)
    digest_addrs = None
    if digest is not None:
        db = np.zeros((8, 4), dtype=U)
        db[:, 0] = np.asarray(digest, dtype=U) % PP
        digest_addrs = mem.write(db, felt=True)
        addrs, b, k = np.concatenate([addrs, digest_addrs]), np.concatenate([b, db]), k + 8
    prep = np.zeros((_pad32(k), 6), dtype=U)
    prep[:k, 0:4], prep[:k, 4] = b, addrs + A0
    T["MemoryConst"] = [prep, np.zeros((_pad32(k), 1), dtype=U)]
    writers.append(("MemoryConst", 5, np.arange(k), addrs))
    # MemoryVar: prep = (addr, mult) x 2, main = value x 2
    k = n["MemoryVar"]
    prep, main = np.zeros((_pad32(k), 4), dtype=U), np.zeros((_pad32(k), 8), dtype=U)
    for e in range(2):
        addrs, b = write_mixed(k)
        prep[:k, 2 * e], main[:k, 4 * e:4 * e + 4] = addrs + A0, b
        writers.append(("MemoryVar", 2 * e + 1, np.arange(k), addrs))
    T["MemoryVar"] = [prep, main]
    assert len(mem.felt) and len(mem.bits), "need MemoryConst / MemoryVar rows to seed the memory"
    felt_of = lambda addrs: mem.vals[addrs, 0]
    # BaseAlu: prep = addrs {out, in1, in2}, is_add, is_sub, is_mul, is_div, mult; main = out, in1, in2
    k = n["BaseAlu"]
    a1, a2 = mem.pick(k, mem.felt), mem.pick(k, mem.felt)
    v1, v2 = felt_of(a1), felt_of(a2)
    op = rng.integers(0, 4, size=k)
    op[(op == 3) & (v2 == 0)] = 0                                    # no division by zero
    out = np.select([op == 0, op == 1, op == 2], [(v1 + v2) % PP, (v1 + PP - v2) % PP, _mul(v1, v2)], _mul(v1, _inv(v2)))
    ob = np.zeros((k, 4), dtype=U)
    ob[:, 0] = out
    base_out = (k, ob)
    prep, main = np.zeros((_pad32(k), 8), dtype=U), np.zeros((_pad32(k), 3), dtype=U)
    prep[:k, 1], prep[:k, 2] = a1 + A0, a2 + A0
    prep[np.arange(k), 3 + op] = 1
    main[:k, 0], main[:k, 1], main[:k, 2] = out, v1, v2
    T["BaseAlu"] = [prep, main]
    # ExtAlu: same prep; main = out[4], in1[4], in2[4]
    k = n["ExtAlu"]
    a1, a2 = mem.pick(k), mem.pick(k)
    v1, v2 = mem.vals[a1], mem.vals[a2]
    op = rng.integers(0, 4, size=k)
    op[(op == 3) & (v2.sum(axis=1) == 0)] = 0
    out = np.where((op == 0)[:, None], (v1 + v2) % PP, (v1 + PP - v2) % PP)
    is_mul, is_div = op == 2, op == 3
    out[is_mul] = ext_mul(v1[is_mul], v2[is_mul])
    out[is_div] = ext_mul(v1[is_div], ext_inv(v2[is_div]))
    prep_e, main_e = np.zeros((_pad32(k), 8), dtype=U), np.zeros((_pad32(k), 12), dtype=U)
    prep_e[:k, 1], prep_e[:k, 2] = a1 + A0, a2 + A0
    prep_e[np.arange(k), 3 + op] = 1
    main_e[:k, 0:4], main_e[:k, 4:8], main_e[:k, 8:12] = out, v1, v2
    T["ExtAlu"] = [prep_e, main_e]
    ext_out = (k, out)
    # Select: prep = is_real, addrs {bit, out1, out2, in1, in2}, mult1, mult2; main = bit, out1, out2, in1, in2
    k = n["Select"]
    ab, a1, a2 = mem.pick(k, mem.bits), mem.pick(k, mem.felt), mem.pick(k, mem.felt)
    bit, v1, v2 = felt_of(ab), felt_of(a1), felt_of(a2)
    o1, o2 = np.where(bit == 1, v2, v1), np.where(bit == 1, v1, v2)
    prep_s, main_s = np.zeros((_pad32(k), 8), dtype=U), np.zeros((_pad32(k), 5), dtype=U)
    prep_s[:k, 0], prep_s[:k, 1], prep_s[:k, 4], prep_s[:k, 5] = 1, ab + A0, a1 + A0, a2 + A0
    main_s[:k] = np.stack([bit, o1, o2, v1, v2], axis=1)
    T["Select"] = [prep_s, main_s]
    # Poseidon2Wide: prep = input addr[16], output (addr, mult)[16], is_real
    k = n["Poseidon2WideDeg3"]
    ain = np.stack([mem.pick(k, mem.felt) for _ in range(16)], axis=1) if k else np.zeros((0, 16), np.int64)
    h = _pad32(k)
    rows = poseidon2_rows(np.concatenate([mem.vals[ain, 0].reshape(k, 16), np.zeros((h - k, 16), dtype=U)]))
    prep_p = np.zeros((h, 49), dtype=U)
    prep_p[:k, 0:16], prep_p[:k, 48] = ain + A0, 1
    T["Poseidon2WideDeg3"] = [prep_p, rows]
    p2_out = rows[:k, P2_OUT(0):P2_OUT(0) + 16]
    # PrefixSumChecks: prep = x1_mem, x2_mem, acc_addr, next_acc_addr, next_acc_mult, felt_acc_addr,
    #                  felt_next_acc_addr, felt_next_acc_mult, is_real; main = x1, x2[4], acc[4], new_acc[4], felt_acc, felt_new_acc
    k = n["PrefixSumChecks"]
    ax1, ax2, aacc, afa = mem.pick(k, mem.bits), mem.pick(k), mem.pick(k), mem.pick(k, mem.felt)
    x1, x2, acc, fa = felt_of(ax1), mem.vals[ax2], mem.vals[aacc], felt_of(afa)
    fac = (2 * _mul(x1[:, None], x2) + PP - x2) % PP                 # 1 - (x1 + x2) + 2 x1 x2, coefficient-wise
    fac[:, 0] = (fac[:, 0] + 1 + PP - x1) % PP
    new_acc, fna = ext_mul(acc, fac), (x1 + 2 * fa) % PP
    prep_x, main_x = np.zeros((_pad32(k), 9), dtype=U), np.zeros((_pad32(k), 15), dtype=U)
    prep_x[:k, 0], prep_x[:k, 1], prep_x[:k, 2], prep_x[:k, 5], prep_x[:k, 8] = ax1 + A0, ax2 + A0, aacc + A0, afa + A0, 1
    main_x[:k, 0], main_x[:k, 1:5], main_x[:k, 5:9], main_x[:k, 9:13], main_x[:k, 13], main_x[:k, 14] = x1, x2, acc, new_acc, fa, fna
    T["PrefixSumChecks"] = [prep_x, main_x]
    # the outputs of this wave become memory now (so nothing above could read them: one straight-line layer)
    k, ob = base_out
    addrs = mem.write(ob, felt=True)
    T["BaseAlu"][0][:k, 0] = addrs + A0
    writers.append(("BaseAlu", 7, np.arange(k), addrs))
    k, ob = ext_out
    addrs = mem.write(ob)
    T["ExtAlu"][0][:k, 0] = addrs + A0
    writers.append(("ExtAlu", 7, np.arange(k), addrs))
    k = n["Select"]
    for col, mcol, o in ((2, 6, o1), (3, 7, o2)):
        ob = np.zeros((k, 4), dtype=U)
        ob[:, 0] = o
        addrs = mem.write(ob, felt=True)
        prep_s[:k, col] = addrs + A0
        writers.append(("Select", mcol, np.arange(k), addrs))
    k = n["Poseidon2WideDeg3"]
    for i in range(16):
        ob = np.zeros((k, 4), dtype=U)
        ob[:, 0] = p2_out[:, i]
        addrs = mem.write(ob, felt=True)
        prep_p[:k, 16 + 2 * i] = addrs + A0
        writers.append(("Poseidon2WideDeg3", 16 + 2 * i + 1, np.arange(k), addrs))
    k = n["PrefixSumChecks"]
    addrs = mem.write(new_acc)
    prep_x[:k, 3] = addrs + A0
    writers.append(("PrefixSumChecks", 4, np.arange(k), addrs))
    ob = np.zeros((k, 4), dtype=U)
    ob[:, 0] = fna
    addrs = mem.write(ob, felt=True)
    prep_x[:k, 6] = addrs + A0
    writers.append(("PrefixSumChecks", 7, np.arange(k), addrs))
    # PublicValues: 16 rows, the first 8 commit digest word i: prep = pv_idx[8], addr, mult; main = the element
    if digest_addrs is None:
        apv = mem.pick(8, mem.felt)
    else:
        apv = digest_addrs
        np.add.at(mem.reads, apv, 1)
    prep_v, main_v = np.zeros((16, 10), dtype=U), np.zeros((16, 1), dtype=U)
    prep_v[np.arange(8), np.arange(8)] = 1
    prep_v[:8, 8], prep_v[:8, 9], main_v[:8, 0] = apv + A0, 1, felt_of(apv)
    T["PublicValues"] = [prep_v, main_v]
    publics = rnd(NUM_PUBLIC_VALUES)
    publics[PV_DIGEST_OFFSET:PV_DIGEST_OFFSET + 8] = felt_of(apv)
    # every reader is known now: write multiplicities = read counts
    for name, col, rows_, addrs in writers:
        T[name][0][rows_, col] = mem.reads[addrs].astype(U)
    return {name: (to_monty(p), to_monty(m)) for name, (p, m) in T.items()}, to_monty(publics)
