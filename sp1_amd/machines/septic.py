"""Vectorised (torch.int64, canonical residues) arithmetic for the RISC-V machine's Global chip traces: the septic extension
F_p^7 = F_p[z] / (z^7 - 3 z - 5), the curve y^2 = x^3 + 45 x + 41 z^3 over it, `lift_x` and the running digest sum —
restating /root/reference/crates/hypercube/src/{septic_extension.rs, septic_curve.rs, septic_digest.rs} and the Poseidon2
row population of operations/poseidon2/trace.rs (same algorithm as recursion_trace.poseidon2_rows, in torch).

The Frobenius matrices are COMPUTED here (z^(p i) by square-and-multiply in Python integers) rather than copied from the
reference's tables; square roots take the norm route: for n in F_q, q = p^7, r = (q - 1) / (p - 1) is odd, so
n^((r + 1) / 2) squared is N(n) n with N(n) = n^r in F_p, and sqrt(n) = n^((r + 1) / 2) / sqrt_p(N(n));
(r + 1) / 2 = 1 + p ((p + 1) / 2) (1 + p^2 + p^4) turns the big exponent into 30 squarings and three Frobenius maps.
"""
import torch

from ..air import P
from .recursion import INTERNAL_DIAG, P2_EXT, P2_INT, P2_OUT, P2_S0, P2_WIDTH, R_INV, _round_constants

I64 = torch.int64


# ---- Python-integer septic arithmetic (set-up only)
def _smul(a, b):
    res = [0] * 13
    for i in range(7):
        for j in range(7):
            res[i + j] += a[i] * b[j]
    ret = res[:7]
    for i in range(7, 13):
        ret[i - 7] += res[i] * 5
        ret[i - 6] += res[i] * 3
    return [x % P for x in ret]


def _spow(a, e):
    r = [1, 0, 0, 0, 0, 0, 0]
    while e:
        if e & 1:
            r = _smul(r, a)
        a = _smul(a, a)
        e >>= 1
    return r


_ZP = _spow([0, 1, 0, 0, 0, 0, 0], P)                         # z^p
_FROB1 = [[1, 0, 0, 0, 0, 0, 0]]
for _ in range(6):
    _FROB1.append(_smul(_FROB1[-1], _ZP))                      # row i = z^(p i)


def _mat_pow(m, k):
    """rows of the k-th Frobenius power: (z^i) -> frob^k(z^i), composing the linear map k times."""
    out = [[int(i == j) for j in range(7)] for i in range(7)]
    for _ in range(k):
        out = [[sum(row[t] * m[t][j] for t in range(7)) % P for j in range(7)] for row in out]
    return out


_FROB = {k: _mat_pow(_FROB1, k) for k in range(1, 7)}


# ---- torch
def fpow(x, e):
    r = torch.ones_like(x)
    while e:
        if e & 1:
            r = r * x % P
        x = x * x % P
        e >>= 1
    return r


def finv(x):
    return fpow(x % P, P - 2)


def smul(a, b):
    """[n, 7] x [n, 7]"""
    prod = a[:, :, None] * b[:, None, :] % P                   # [n, 7, 7]
    res = torch.zeros((a.shape[0], 13), dtype=I64, device=a.device)
    for i in range(7):
        res[:, i:i + 7] += prod[:, i, :]
    res %= P
    ret = res[:, :7].clone()
    ret[:, 0:6] += res[:, 7:13] * 5
    ret[:, 1:7] += res[:, 7:13] * 3
    return ret % P


def frob(a, k=1):
    m = torch.tensor(_FROB[k], dtype=I64, device=a.device)     # [7, 7]: row i = frob^k(z^i)
    return (a[:, :, None] * m[None, :, :] % P).sum(dim=1) % P


def norm_parts(a):
    """(a^(r - 1), N(a)) with r = 1 + p + ... + p^6: a^(r - 1) = prod_{i=1..6} frob^i(a)."""
    t = smul(frob(a, 1), frob(a, 2))                           # a^(p + p^2)
    t = smul(smul(t, frob(t, 2)), frob(t, 4))                  # ^(1 + p^2 + p^4)
    return t, smul(t, a)[:, 0]


def sinv(a):
    t, n = norm_parts(a)
    return t * finv(n)[:, None] % P


def fsqrt(a):
    """Tonelli-Shanks in F_p (p - 1 = 2^24 127) for quadratic residues a (vectorised; garbage for non-residues)."""
    S, Q = 24, 127
    c = torch.full_like(a, pow(3, Q, P))
    t = fpow(a, Q)
    r = fpow(a, (Q + 1) // 2)
    for i in range(1, S):
        b = t
        for _ in range(S - 1 - i):
            b = b * b % P
        m = b != 1
        r = torch.where(m, r * c % P, r)
        c = c * c % P
        t = torch.where(m, t * c % P, t)
    return r


def is_square_p(a):
    return fpow(a, (P - 1) // 2) == 1


def ssqrt(n):
    """(root [n, 7], ok [n]) — a square root of every n that has one."""
    _, nn = norm_parts(n)
    ok = is_square_p(nn)
    w, sq = n.clone(), n.clone()
    for i in range(1, 30):                                     # w = n^((p + 1) / 2) = n^(1 + 2^23 + ... + 2^29)
        sq = smul(sq, sq)
        if i >= 23:
            w = smul(w, sq)
    f1 = frob(w, 1)
    cand = smul(smul(smul(f1, frob(w, 3)), frob(w, 5)), n)     # n^((r + 1) / 2)
    root = cand * finv(fsqrt(torch.where(ok, nn, torch.ones_like(nn))))[:, None] % P
    return root, ok


def curve_rhs(x):
    out = (smul(smul(x, x), x) + 45 * x) % P
    out[:, 3] = (out[:, 3] + 41) % P
    return out


def ec_add(p1, p2):
    """add_incomplete (septic_curve.rs:L58-L63); points as ([n, 7], [n, 7])."""
    slope = smul((p2[1] - p1[1]) % P, sinv((p2[0] - p1[0]) % P))
    x3 = (smul(slope, slope) - p1[0] - p2[0]) % P
    y3 = (smul(slope, (p1[0] - x3) % P) - p1[1]) % P
    return x3, y3


def _ext_linear(s):
    t = torch.empty_like(s)
    for j in range(0, 16, 4):
        x0, x1, x2, x3 = (s[:, j + k] for k in range(4))
        t[:, j] = 2 * x0 + 3 * x1 + x2 + x3
        t[:, j + 1] = x0 + 2 * x1 + 3 * x2 + x3
        t[:, j + 2] = x0 + x1 + 2 * x2 + 3 * x3
        t[:, j + 3] = 3 * x0 + x1 + x2 + 2 * x3
    t %= P
    sums = (t[:, 0:4] + t[:, 4:8] + t[:, 8:12] + t[:, 12:16]) % P
    return (t + sums.repeat(1, 4)) % P


def poseidon2_rows(inputs):
    """populate_perm (operations/poseidon2/trace.rs:L29-L152): [n, 16] canonical inputs -> [n, 179] rows."""
    dev = inputs.device
    rc = torch.tensor(_round_constants(), dtype=I64, device=dev)
    row = torch.zeros((inputs.shape[0], P2_WIDTH), dtype=I64, device=dev)
    cube = lambda x: x * x % P * x % P
    s = inputs % P
    diag = torch.tensor([(d * R_INV) % P for d in INTERNAL_DIAG], dtype=I64, device=dev)
    for r in range(8):
        row[:, P2_EXT(r, 0):P2_EXT(r, 0) + 16] = s
        if r == 0:
            s = _ext_linear(s)
        s = _ext_linear(cube((s + rc[r if r < 4 else 24 + r - 4]) % P))
        if r == 3:
            row[:, P2_INT(0):P2_INT(0) + 16] = s
            for k in range(20):
                s = s.clone()
                s[:, 0] = cube((s[:, 0] + rc[4 + k][0]) % P)
                tot = s.sum(dim=1) % P * R_INV % P
                s = (tot[:, None] + s * diag[None, :]) % P
                if k < 19:
                    row[:, P2_S0(k)] = s[:, 0]
    row[:, P2_OUT(0):P2_OUT(0) + 16] = s
    return row


def lift_x(message, kind, is_receive):
    """SepticCurve::lift_x + GlobalInteractionOperation::get_digest (operations/global_interaction.rs:L33-L46,
    septic_curve.rs:L124-L163): message [n, 8], kind [n], is_receive [n] bool -> (x, y, offset, permutation rows [n, 179])."""
    n, dev = message.shape[0], message.device
    m = torch.zeros((n, 16), dtype=I64, device=dev)
    m[:, :8] = message
    m[:, 0] += kind << 24
    x = torch.zeros((n, 7), dtype=I64, device=dev)
    y = torch.zeros((n, 7), dtype=I64, device=dev)
    off = torch.zeros(n, dtype=I64, device=dev)
    rows = torch.zeros((n, P2_WIDTH), dtype=I64, device=dev)
    todo = torch.arange(n, device=dev)
    lim = 63 << 24
    for offset in range(256):
        if todo.numel() == 0:
            break
        trial = m[todo].clone()
        trial[:, 7] += offset << 16
        pr = poseidon2_rows(trial)
        xt = pr[:, P2_OUT(0):P2_OUT(0) + 7]
        root, ok = ssqrt(curve_rhs(xt))
        y6 = root[:, 6]
        neg6 = (P - y6) % P
        # the root whose last coordinate lies in [1, 63 2^24] is the "receive" point; exception when neither root's does
        recv_root = torch.where(((y6 >= 1) & (y6 <= lim))[:, None], root, (P - root) % P)
        valid = ok & (((y6 >= 1) & (y6 <= lim)) | ((neg6 >= 1) & (neg6 <= lim)))
        sel = todo[valid]
        yy = recv_root[valid]
        yy = torch.where(is_receive[sel][:, None], yy, (P - yy) % P)       # a send carries the negated point
        x[sel], y[sel], off[sel], rows[sel] = xt[valid], yy, offset, pr[valid]
        todo = todo[~valid]
    assert todo.numel() == 0, "lift_x: no curve point within 256 offsets"
    return x, y, off, rows


def prefix_sums(start, px, py):
    """cumulative[i] = start + P_0 + ... + P_i (a Hillis-Steele scan over the group law; the incomplete formulas are safe with
    overwhelming probability, as in the reference's parallel scan)."""
    n = px.shape[0]
    sx, sy = px.clone(), py.clone()
    # fold `start` into the first point
    fx, fy = ec_add((start[0][None, :], start[1][None, :]), (sx[:1], sy[:1]))
    sx[0], sy[0] = fx[0], fy[0]
    d = 1
    while d < n:
        ax, ay = ec_add((sx[:-d], sy[:-d]), (sx[d:], sy[d:]))
        sx = torch.cat([sx[:d], ax])
        sy = torch.cat([sy[:d], ay])
        d *= 2
    return sx, sy
