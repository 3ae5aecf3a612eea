"""Pure-Python KoalaBear mini-oracle (TEST INFRASTRUCTURE ONLY — never imported by the product).

Small, slow, obviously-correct restatement in *canonical* integers (no Montgomery words) of the
pieces of the reference that its own golden artifact can pin:

* field / extension: `p = 2^31 - 2^24 + 1`, `EF = F[x]/(x^4 - 3)`
  (/root/reference/crates/primitives/src/lib.rs:L28-L38)
* Poseidon2 width 16, 8 full + 20 partial rounds, x^3
  (/root/reference/slop/crates/koala-bear/src/koala_bear_poseidon2.rs:L20-L63, round structure as
  restated in /root/reference/crates/hypercube/src/operations/poseidon2/{trace.rs:L29-L152,
  air.rs:L17-L66}; internal diagonal [-2,1,2,4,..,2^13,2^15] and the 2^-32 factor as in
  /root/reference/sp1-gpu/crates/sys/include/poseidon2/poseidon2_kb31_16.cuh:L118-L140)
* PaddingFreeSponge<16,8,8> leaf hash and TruncatedPermutation 2-to-1 compression
  (koala_bear_poseidon2.rs:L33-L41)
* Merkle path recomputation (/root/reference/slop/crates/merkle-tree/src/tcs.rs:L102-L188)
* the BaseFold query fold (/root/reference/slop/crates/basefold/src/verifier.rs:L323-L388)

Used by tests/golden/make_golden.py (fixture extraction) and by the CPU tests as a third,
independent implementation against which the C++ oracle is cross-checked.
"""
import os
import re

P = 0x7F000001
R_INV = pow(1 << 32, -1, P)           # the Montgomery 2^-32 factor of the internal layer
TWO_ADICITY = 24
GEN_2_24 = 0x6AC49F88                 # two_adic_generator(24), canonical
EXT_W = 3                              # x^4 = 3

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load_rc():
    txt = open(os.path.join(_HERE, "kb_poseidon2_rc.inc")).read()
    vals = [int(h, 16) for h in re.findall(r"0x([0-9a-f]{8})u", txt)]
    assert len(vals) == 28 * 16
    return [vals[r * 16:(r + 1) * 16] for r in range(28)]


RC = _load_rc()
INTERNAL_DIAG = [P - 2] + [1 << k for k in range(14)] + [1 << 15]


def two_adic_generator(bits):
    assert bits <= TWO_ADICITY
    return pow(GEN_2_24, 1 << (TWO_ADICITY - bits), P)


def _m4(x):
    x0, x1, x2, x3 = x
    return [(2 * x0 + 3 * x1 + x2 + x3) % P, (x0 + 2 * x1 + 3 * x2 + x3) % P,
            (x0 + x1 + 2 * x2 + 3 * x3) % P, (3 * x0 + x1 + x2 + 2 * x3) % P]


def external_linear(s):
    t = []
    for j in range(0, 16, 4):
        t += _m4(s[j:j + 4])
    sums = [(t[k] + t[k + 4] + t[k + 8] + t[k + 12]) % P for k in range(4)]
    return [(t[j] + sums[j % 4]) % P for j in range(16)]


def internal_linear(s):
    tot = sum(s) % P
    return [((tot + INTERNAL_DIAG[i] * s[i]) * R_INV) % P for i in range(16)]


def permute(state):
    s = external_linear(list(state))
    for r in range(4):
        s = [pow((s[i] + RC[r][i]) % P, 3, P) for i in range(16)]
        s = external_linear(s)
    for r in range(20):
        s[0] = pow((s[0] + RC[4 + r][0]) % P, 3, P)
        s = internal_linear(s)
    for r in range(24, 28):
        s = [pow((s[i] + RC[r][i]) % P, 3, P) for i in range(16)]
        s = external_linear(s)
    return s


def hash_felts(xs):
    s = [0] * 16
    for i in range(0, len(xs), 8):
        chunk = xs[i:i + 8]
        s[:len(chunk)] = chunk
        s = permute(s)
    return s[:8]


def compress(left, right):
    return permute(list(left) + list(right))[:8]


def merkle_root_from_path(index, leaf_values, path):
    node = hash_felts(list(leaf_values))
    for sib in path:
        node = compress(node, sib) if index & 1 == 0 else compress(sib, node)
        index >>= 1
    return node, index


# ---- extension field F[x]/(x^4 - 3), elements are 4-lists of canonical ints -------------------

def ext_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def ext_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def ext_mul(a, b):
    out = [0] * 7
    for i in range(4):
        for j in range(4):
            out[i + j] += a[i] * b[j]
    return [(out[k] + EXT_W * (out[k + 4] if k < 3 else 0)) % P for k in range(4)]


def ext_scale(a, c):
    return [(x * c) % P for x in a]


def ext_from_base(c):
    return [c % P, 0, 0, 0]


def ext_inv(a):
    # solve a*b = 1 via the 4x4 multiplication matrix (tiny Gaussian elimination mod P)
    m = [[0] * 4 for _ in range(4)]
    for j in range(4):
        e = [0] * 4
        e[j] = 1
        col = ext_mul(a, e)
        for i in range(4):
            m[i][j] = col[i]
    rhs = [1, 0, 0, 0]
    n = 4
    for c in range(n):
        piv = next(r for r in range(c, n) if m[r][c] % P)
        m[c], m[piv] = m[piv], m[c]
        rhs[c], rhs[piv] = rhs[piv], rhs[c]
        inv = pow(m[c][c], -1, P)
        m[c] = [(v * inv) % P for v in m[c]]
        rhs[c] = (rhs[c] * inv) % P
        for r in range(n):
            if r != c and m[r][c]:
                f = m[r][c]
                m[r] = [(v - f * w) % P for v, w in zip(m[r], m[c])]
                rhs[r] = (rhs[r] - f * rhs[c]) % P
    return rhs


def reverse_bits_len(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def fold_query(e0, e1, beta, x0):
    """verifier.rs:L364-L374 — interpolate through (x0,e0),(-x0,e1), evaluate at beta."""
    x1 = (P - x0) % P
    inv = pow((x1 - x0) % P, -1, P)
    t = ext_mul(ext_sub(beta, ext_from_base(x0)), ext_scale(ext_sub(e1, e0), inv))
    return ext_add(e0, t)


KAT_PERM_ZERO = [145589356, 1876041682, 1734203622, 499355069, 673349476, 595701365, 270340205,
                 131707822, 1236787881, 1085405948, 2065733208, 1999012278, 2062318124, 1616707536,
                 324813015, 749520722]


class Challenger:
    """DuplexChallenger<KoalaBear, Poseidon2, 16, 8> on canonical ints (pure Python, small cases only).
    Type: /root/reference/slop/crates/challenger/src/lib.rs:L25-L87; semantics as restated in
    /root/reference/sp1-gpu/crates/sys/include/challenger/challenger.cuh:L13-L118 (the p3-challenger
    source is an un-vendored dependency). Pinned by tests/golden/make_transcript.py: replaying the real
    shard proof's Fiat-Shamir transcript reproduces its sumcheck points, fold betas, grinding witnesses
    and query indices."""

    def __init__(self):
        self.state = [0] * 16
        self.inp = []
        self.out = []

    def _duplex(self):
        for i, v in enumerate(self.inp):
            self.state[i] = v
        self.inp = []
        self.state = permute(self.state)
        self.out = list(self.state[:8])

    def observe(self, x):
        self.out = []
        self.inp.append(int(x) % P)
        if len(self.inp) == 8:
            self._duplex()

    def observe_many(self, xs):
        for x in xs:
            self.observe(x)

    def sample(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def sample_ext(self):
        return [self.sample() for _ in range(4)]

    def sample_bits(self, bits):
        return self.sample() & ((1 << bits) - 1)

    def check_witness(self, bits, w):
        self.observe(w)
        return self.sample_bits(bits) == 0
