"""Satisfying traces for the shards that are NOT made of CPU instructions (riscv_more.py's chips):

  * `precompile_shard` — a KECCAK_PERMUTE precompile shard as the reference builds it (crates/core/executor/src/record.rs splits
    precompile events into their own shards): KeccakPermuteControl (one row per syscall: syscall receive, 25 reads + 25 writes of
    the state words), KeccakPermute (24 rows per syscall: one Keccak-f round each), SyscallPrecompile (the Global receive of the
    syscall), MemoryLocal (one row per touched word), Global (every global interaction, septic-curve digest), Program, Byte,
    Range: the reference's Keccak shape cluster, closed by the shard's public values (`eval_public_values`, public_values.py).
    The permutation is COMPUTED here (keccak_f_rows: theta / rho / pi / chi / iota on 64-bit lanes, every intermediate the AIR
    names) and checked against hashlib's SHA3-256 in the tests.
  * `memory_shard` — global memory initialisation / finalisation (MemoryGlobalInit, MemoryGlobalFinalize, Global, Program, Byte,
    Range; the ends of the two control chains and of the accumulation chain are public values).

As in riscv_trace.py nothing is fitted: tests/machine_check.py requires every constraint of every chip to vanish on every row and
every bus to balance. References per function.
"""
import numpy as np
import torch

from ..air import AirProgram, InteractionProgram, P, VCol
from . import public_values as PVM
from . import riscv as R
from . import riscv_more as M
from . import riscv_trace as RT

I64 = torch.int64
MASK16 = 0xFFFF
U64 = (1 << 64) - 1
ROTL = lambda v, n: ((v << n) | (v >> (64 - n))) & U64 if n else v


def keccak_f_rows(state):
    """One Keccak-f[1600] permutation of `state` (25 u64, index x + 5 y) as the 24 rows of `KeccakCols` values the AIR names
    (keccak256/air.rs; p3-keccak-air's generate_trace_rows): per round a (input lanes), c, c_prime, a_prime, a_prime_prime,
    a_prime_prime_0_0_bits, a_prime_prime_prime_0_0. Returns (rows, final state); lanes are python ints."""
    a = list(state)
    rows = []
    for rnd in range(24):
        A = [[a[x + 5 * y] for x in range(5)] for y in range(5)]                     # A[y][x]
        C = [A[0][x] ^ A[1][x] ^ A[2][x] ^ A[3][x] ^ A[4][x] for x in range(5)]
        Cp = [C[x] ^ C[(x + 4) % 5] ^ ROTL(C[(x + 1) % 5], 1) for x in range(5)]   # C'[x, z] = C[x, z] ^ C[x-1, z] ^ C[x+1, z-1]
        Ap = [[A[y][x] ^ C[x] ^ Cp[x] for x in range(5)] for y in range(5)]          # A' = A ^ D, D[x] = C[x-1] ^ ROT(C[x+1], 1) = C ^ C'
        Bm = [[0] * 5 for _ in range(5)]                                             # B[y][x]; B[y', 2x+3y] = ROT(A'[x, y], r[x][y]) with x' = y
        for x in range(5):
            for y in range(5):
                Bm[(2 * x + 3 * y) % 5][y] = ROTL(Ap[y][x], M.KECCAK_R[x][y])
        App = [[Bm[y][x] ^ (~Bm[y][(x + 1) % 5] & U64 & Bm[y][(x + 2) % 5]) for x in range(5)] for y in range(5)]
        appp00 = App[0][0] ^ M.KECCAK_RC[rnd]
        rows.append({"a": A, "c": C, "c_prime": Cp, "a_prime": Ap, "a_prime_prime": App, "appp00": appp00})
        nxt = [App[y][x] for y in range(5) for x in range(5)]
        nxt[0] = appp00
        a = [nxt[x + 5 * y] for y in range(5) for x in range(5)]
    return rows, a


def _bits(v):
    return [(v >> z) & 1 for z in range(64)]


def _limbs(v):
    return [(v >> (16 * i)) & MASK16 for i in range(4)]


def _limbs_np(v):
    return np.stack([(v >> np.uint64(16 * i)) & np.uint64(MASK16) for i in range(4)], axis=1).astype(np.int64)


def _inv_np(v):
    """Inverses mod P of an int64 array (0 -> 0): v^(P - 2) by square and multiply, products below 2^62."""
    base, out, e = np.asarray(v, dtype=np.int64) % P, np.ones(np.shape(v), dtype=np.int64), P - 2
    while e:
        if e & 1:
            out = out * base % P
        base = base * base % P
        e >>= 1
    return np.where(np.asarray(v) % P == 0, 0, out)


def _rotl(v, n):
    """64-bit rotate left of int64 tensors (two's complement bit patterns)."""
    if n == 0:
        return v
    return (v << n) | RT.srl(v, torch.full_like(v, 64 - n))


def keccak_round_tensors(state):
    """The same 24 rounds as keccak_f_rows, vectorised over permutations: state [n, 25] int64 (lane x + 5 y as a bit pattern).
    Returns a list of 24 dicts of tensors (a [n,5,5] as [y][x], c [n,5], c_prime [n,5], a_prime [n,5,5], a_prime_prime [n,5,5],
    appp00 [n]) and the final state [n, 25]."""
    a = state.clone()
    rows = []
    for rnd in range(24):
        A = a.reshape(-1, 5, 5)                                                       # [n, y, x]
        C = A[:, 0] ^ A[:, 1] ^ A[:, 2] ^ A[:, 3] ^ A[:, 4]                          # [n, x]
        Cp = torch.stack([C[:, x] ^ C[:, (x + 4) % 5] ^ _rotl(C[:, (x + 1) % 5], 1) for x in range(5)], dim=1)
        Ap = A ^ C[:, None, :] ^ Cp[:, None, :]
        Bm = torch.zeros_like(Ap)
        for x in range(5):
            for y in range(5):
                Bm[:, (2 * x + 3 * y) % 5, y] = _rotl(Ap[:, y, x], M.KECCAK_R[x][y])
        App = torch.stack([Bm[:, :, x] ^ (~Bm[:, :, (x + 1) % 5] & Bm[:, :, (x + 2) % 5]) for x in range(5)], dim=2)
        appp00 = App[:, 0, 0] ^ RT._S64(M.KECCAK_RC[rnd])
        rows.append({"a": A, "c": C, "c_prime": Cp, "a_prime": Ap, "a_prime_prime": App, "appp00": appp00})
        nxt = App.clone()
        nxt[:, 0, 0] = appp00
        a = nxt.reshape(-1, 25)
    return rows, a


def _bits_t(v):
    """[...] int64 -> [..., 64] bits, little-endian."""
    z = torch.arange(64, device=v.device)
    return (v[..., None] >> z) & 1


def _limbs_t(v):
    return torch.stack([(v >> (16 * i)) & MASK16 for i in range(4)], dim=-1)


def keccak_permute_table(clk, addr, pre, dev):
    """KeccakPermuteChip::generate_trace_into (keccak256/trace.rs:L63-L162): 24 rows per event, padding rows = the rows of a
    permutation of the zero state with is_real = 0. clk, addr: [n] int64; pre: [n, 25] int64 lane bit patterns."""
    air, _ = R.chip("KeccakPermute")
    L = air.layout
    n_ev = pre.shape[0]
    n = 24 * n_ev
    tb = RT.Table(air, n, dev)
    n_pad = tb.main.shape[0] - n
    # one extra "event" (the zero state) supplies the padding rows
    allpre = torch.cat([pre, torch.zeros((1, 25), dtype=I64, device=dev)]) if n_pad else pre
    rounds, post = keccak_round_tensors(allpre)
    E = allpre.shape[0]
    view = torch.zeros((E, 24, air.main_width), dtype=I64, device=dev)
    pre_l = _limbs_t(allpre.reshape(E, 5, 5))                                            # [E, y, x, 4]
    for rnd, kv in enumerate(rounds):
        row = view[:, rnd]
        row[:, L["keccak.step_flags"] + rnd] = 1
        row[:, L["keccak.export"]] = int(rnd == 23)
        c0 = L["keccak.preimage.0.0"]
        row[:, c0:c0 + 100] = pre_l.reshape(E, 100)
        c0 = L["keccak.a.0.0"]
        row[:, c0:c0 + 100] = _limbs_t(kv["a"]).reshape(E, 100)
        c0 = L["keccak.c.0"]
        row[:, c0:c0 + 320] = _bits_t(kv["c"]).reshape(E, 320)
        c0 = L["keccak.c_prime.0"]
        row[:, c0:c0 + 320] = _bits_t(kv["c_prime"]).reshape(E, 320)
        c0 = L["keccak.a_prime.0.0"]
        row[:, c0:c0 + 1600] = _bits_t(kv["a_prime"]).reshape(E, 1600)
        c0 = L["keccak.a_prime_prime.0.0"]
        row[:, c0:c0 + 100] = _limbs_t(kv["a_prime_prime"]).reshape(E, 100)
        c0 = L["keccak.a_prime_prime_0_0_bits"]
        row[:, c0:c0 + 64] = _bits_t(kv["a_prime_prime"][:, 0, 0])
        c0 = L["keccak.a_prime_prime_prime_0_0_limbs"]
        row[:, c0:c0 + 4] = _limbs_t(kv["appp00"])
        row[:n_ev, L["clk_high"]], row[:n_ev, L["clk_low"]] = clk >> 24, clk & 0xFFFFFF
        row[:n_ev, L["state_addr"]:L["state_addr"] + 3] = _limbs_t(addr)[:, :3]
        row[:n_ev, L["index"]], row[:n_ev, L["is_real"]] = rnd, 1
    flat = view.reshape(E * 24, air.main_width)
    tb.main[:] = flat[:tb.main.shape[0]]
    assert n_pad <= 24
    return tb, post[:n_ev]


def _mem_access_t(tb, prefix, prev_val, t_prev, t_cur):
    """MemoryAccessCols::populate (memory/consistency/trace.rs:L36-L101), one column group, all rows."""
    ph, pl, ch, cl = t_prev >> 24, t_prev & 0xFFFFFF, t_cur >> 24, t_cur & 0xFFFFFF
    same = ph == ch
    d = torch.where(same, cl - pl, ch - ph) - 1
    assert bool((d >= 0).all())
    tb.set(prefix + ".prev_value", _limbs_t(prev_val))
    tb.set(prefix + ".prev_high", ph)
    tb.set(prefix + ".prev_low", pl)
    tb.set(prefix + ".compare_low", same.to(I64))
    tb.set(prefix + ".diff_low_limb", d & MASK16)
    tb.set(prefix + ".diff_high_limb", d >> 16)


def precompile_shard(n_events, seed=0, device="cpu", clk0=(5 << 24) + 1001):
    """A KECCAK_PERMUTE precompile shard of `n_events` random syscalls (see precompile_shard_from)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    # events: distinct, 8-aligned, non-overlapping state addresses >= 2^16 (200-byte states in 256-byte slots); increasing clocks
    # (= 1 mod 8 like every instruction's)
    slots = torch.randperm(4 * n_events + 4, generator=gen, device=dev)[:n_events]
    addr = 0x20_0000 + 256 * slots
    clk = clk0 + 8 * 40 * torch.arange(n_events, device=dev)
    pre = torch.randint(RT.MIN64, (1 << 63) - 1, (n_events, 25), generator=gen, device=dev, dtype=I64)
    t_prev = torch.randint(1, clk0 - 8, (n_events, 25), generator=gen, device=dev, dtype=I64)       # last accesses, in earlier shards
    return precompile_shard_from(clk, addr, pre, t_prev, dev)[:3]


def precompile_shard_from(clk, addr, pre, t_prev, device="cpu", ctx=None):
    """The KECCAK_PERMUTE precompile shard of the syscalls (clk [n], state pointer addr [n], state read pre [n, 25], previous
    timestamps of its words t_prev [n, 25]): (machine, tables, publics, global events) — the first three like riscv_trace.generate
    (vectorised torch.int64: CPU in the tests, the GPU in the bench). Reads happen at clk, writes at clk + 1 (keccak256/permute.rs)."""
    dev = torch.device(device)
    n_events = int(clk.shape[0])
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    kp, post = keccak_permute_table(clk, addr, pre, dev)
    tr.tables["KeccakPermute"] = kp
    # KeccakPermuteControl (controller.rs:L155-L237)
    ct = RT.Table(R.chip("KeccakPermuteControl")[0], n_events, dev)
    ct.set("clk_high", clk >> 24); ct.set("clk_low", clk & 0xFFFFFF); ct.set("is_real", 1)
    al = _syscall_addr_t(ct, "state_addr", addr)
    for i in range(25):
        ct.set("addrs.%d.value" % i, _limbs_t(addr + 8 * i)[:, :3])
        _mem_access_t(ct, "initial_memory_access.%d" % i, pre[:, i], t_prev[:, i], clk)
        _mem_access_t(ct, "final_memory_access.%d" % i, pre[:, i], clk, clk + 1)
        ct.set("final_value.%d" % i, _limbs_t(post[:, i]))
    tr.tables["KeccakPermuteControl"] = ct
    wa = (addr[:, None] + 8 * torch.arange(25, device=dev)[None, :]).reshape(-1)
    return _close_precompile_shard(tr, M.SYS_KECCAK_PERMUTE, clk, al, wa, t_prev.reshape(-1), (clk[:, None] + 1).expand(-1, 25).reshape(-1),
                                   pre.reshape(-1), post.reshape(-1), ctx=ctx)


def _close_precompile_shard(tr, syscall_id, clk, ptr_limbs, word_addr, t_initial, t_final, v_initial, v_final, arg2_limbs=None, ctx=None):
    """What every precompile shard has around its own chips: SyscallPrecompile (one row per call), MemoryLocal (one row per touched
    word: state before the call's first and after its last access), the Global chip over their events, the Program table of the
    run (`ctx`: riscv_trace.RunContext; nothing executes here, its multiplicities are zero), the byte tables — the shard's shape
    cluster (riscv/mod.rs:L560-L597) — and its public values: the program's initial state (`update_initialized_state`)."""
    dev = tr.dev
    ctx = ctx if ctx is not None else RT.RunContext()
    # SyscallPrecompile (syscall/chip.rs:L196-L254)
    st = RT.Table(R.chip("SyscallPrecompile")[0], int(clk.shape[0]), dev)
    st.set("clk_high", clk >> 24); st.set("clk_low", clk & 0xFFFFFF); st.set("syscall_id", syscall_id)
    st.set("arg1", ptr_limbs[:, :3]); st.set("is_real", 1)
    if arg2_limbs is not None:
        st.set("arg2", arg2_limbs[:, :3])
    tr.tables["SyscallPrecompile"] = st
    # MemoryLocal (memory/local.rs:L98-L250)
    ml = RT.Table(R.chip("MemoryLocal")[0], int(word_addr.shape[0]), dev)
    ml.set("addr", _limbs_t(word_addr)[:, :3])
    ml.set("initial_clk_high", t_initial >> 24); ml.set("initial_clk_low", t_initial & 0xFFFFFF)
    ml.set("final_clk_high", t_final >> 24); ml.set("final_clk_low", t_final & 0xFFFFFF)
    for tag, v in (("initial", v_initial), ("final", v_final)):
        l = _limbs_t(v)
        ml.set(tag + "_value", l)
        ml.set(tag + "_value_lower", l[:, 2] & 0xFF)
        ml.set(tag + "_value_upper", l[:, 2] >> 8)
    ml.set("is_real", 1)
    tr.tables["MemoryLocal"] = ml
    machine = {n: R.chip(n) for n in tr.tables}
    # Global: MemoryLocal's events (initial = receive, final = send per row), then the syscall receives
    (_, recv, _), (_, send, _) = RT.eval_interactions(R.chip("MemoryLocal")[1], ml.main[:ml.n], None, kinds=(R.GLOBAL,))
    ev = [torch.stack([recv, send], dim=1).reshape(-1, 11)]
    ev += [v for _, v, _ in RT.eval_interactions(R.chip("SyscallPrecompile")[1], st.main[:st.n], None, kinds=(R.GLOBAL,))]
    tr.global_events = torch.cat(ev)
    pv = PVM.no_memory_events(PVM.initialized_state(ctx.pc_start))
    PVM.set_global(pv, *tr.global_chip(machine, tr.global_events))
    tr.tables["Program"], machine["Program"] = ctx.program_table(dev)
    tr.byte_range_tables(machine, pv)
    tr.fill_cluster(machine, RT.smallest_cluster(machine))
    names = sorted(machine)
    return [machine[n] for n in names], {n: (tr.tables[n].prep, tr.tables[n].main) for n in names}, PVM.to_tensor(pv), tr.global_events


def _syscall_addr_t(tb, prefix, addr):
    """SyscallAddrOperation::populate (operations/syscall_addr.rs:L27-L46); returns the address limbs."""
    al = _limbs_t(addr)
    tb.set(prefix + ".addr", al[:, :3])
    top = al[:, 1] + al[:, 2]
    tb.set(prefix + ".top_two_limb_min", RT.finv(top))
    dmax = (top - 2 * MASK16) % P
    tb.set(prefix + ".top_two_limb_max.inverse", torch.where(dmax == 0, torch.zeros_like(dmax), RT.finv(dmax)))
    tb.set(prefix + ".top_two_limb_max.result", (dmax == 0).to(I64))
    return al


def poseidon2_shard_from(clk, ptr, pre, t_prev, post, device="cpu", ctx=None):
    """The POSEIDON2 precompile shard of the syscalls (clk [n], pointer ptr [n], the eight words read pre [n, 8] with their previous
    timestamps t_prev [n, 8], the eight words written post [n, 8] — at clk, in place): the Poseidon2 chip's rows
    (`generate_trace_into`, syscall/precompiles/poseidon2/air.rs:L107-L290: the permutation's 179 columns are RECOMPUTED from the
    words read, so a wrong `post` fails the output constraints), SyscallPrecompile, MemoryLocal, Global, Byte, Range."""
    from . import septic as SE
    dev = torch.device(device)
    n = int(clk.shape[0])
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    air, _ = R.chip("Poseidon2")
    tb = RT.Table(air, n, dev)
    tb.set("clk_high", clk >> 24); tb.set("clk_low", clk & 0xFFFFFF); tb.set("is_real", 1)
    al = _syscall_addr_t(tb, "ptr", ptr)
    top = (P - 1) >> 16
    lo32, hi32 = lambda v: v & 0xFFFFFFFF, lambda v: (v >> 32) & 0xFFFFFFFF
    for i in range(8):
        tb.set("addrs.%d.value" % i, _limbs_t(ptr + 8 * i)[:, :3])
        _mem_access_t(tb, "memory.%d" % i, pre[:, i], t_prev[:, i], clk)
        hl, pl = _limbs_t(post[:, i]), _limbs_t(pre[:, i])
        tb.set("hash_result.%d" % i, hl)
        for name, l in (("hash_result_range_checkers", hl), ("input_range_checkers", pl)):   # SP1FieldWordRangeChecker::populate
            tb.set(name, (l[:, 1] < top).to(I64), off=2 * i)
            tb.set(name, (l[:, 3] < top).to(I64), off=2 * i + 1)
    state = torch.stack([f(pre[:, i]) for i in range(8) for f in (lo32, hi32)], dim=1)
    pcol = tb.L["permutation"]
    tb.main[:n, pcol:pcol + SE.P2_WIDTH] = SE.poseidon2_rows(state)
    if tb.main.shape[0] > n:                                   # padding rows: the permutation of the zero state (air.rs:L283-L290)
        tb.main[n:, pcol:pcol + SE.P2_WIDTH] = SE.poseidon2_rows(torch.zeros((1, 16), dtype=I64, device=dev))[0]
    tr.tables["Poseidon2"] = tb
    wa = (ptr[:, None] + 8 * torch.arange(8, device=dev)[None, :]).reshape(-1)
    return _close_precompile_shard(tr, M.SYS_POSEIDON2, clk, al, wa, t_prev.reshape(-1), clk[:, None].expand(-1, 8).reshape(-1),
                                   pre.reshape(-1), post.reshape(-1), ctx=ctx)


def memory_shard(n_words, seed=0, device="cpu", with_zero=True):
    """Global memory initialisation and finalisation of `n_words` random addresses (see memory_shard_from)."""
    rng = np.random.default_rng(seed)
    addrs = np.sort(rng.choice(np.arange(1, 1 << 20), size=n_words - int(with_zero), replace=False)) * 8 + (1 << 16)       # all > 2^16
    if with_zero:
        addrs = np.concatenate([[0], addrs])                # register x0: address 0 with value 0 (the `is_comp = 0` row)
    word = lambda a: 0 if a == 0 else int(rng.integers(0, 1 << 63, dtype=np.int64)) * 2 + int(rng.integers(2))
    init = [(word(int(a)), int(rng.integers(1, 1 << 40))) for a in addrs]
    fin = [(word(int(a)), int(rng.integers(1, 1 << 40))) for a in addrs]
    # the first memory shard starts at previous_addr = 0 and must initialise address 0 (register x0) with 0; a later one
    # continues from the previous shard's last address (public values previous_init_addr / previous_finalize_addr)
    return memory_shard_from([int(a) for a in addrs], init, fin, device, previous_addr=0 if with_zero else 1 << 16)[:3]


def memory_shard_from(addrs, init, fin, device="cpu", previous_addr=0, ctx=None, fin_addrs=None, previous_fin_addr=None):
    """A memory shard of a run (memory/global.rs generate_trace_into): MemoryGlobalInit rows for the strictly increasing addresses
    `addrs` and MemoryGlobalFinalize rows for `fin_addrs` (default: the same addresses; the reference finalises every word of the
    program image but initialises none of them, so the two lists differ there) — init[i] / fin[i] = (value, timestamp) of address i
    (the Init chip's Global message carries timestamp 0 whatever its clk columns hold) —, the chains of (index, prev_addr, validity)
    control messages whose two ends are the shard's public values (previous_*_addr / last_*_addr / global_*_count:
    eval_global_memory_init / finalize; a chip without events is at height zero and its chain stands still), one Global event per
    row, the Global chip, the run's Program table and the byte tables; the public values are the program's FINAL state
    (`update_finalized_state`, ctx.final). Returns (machine, tables, publics, global events)."""
    dev = torch.device(device)
    ctx = ctx if ctx is not None else RT.RunContext()
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    fin_addrs = addrs if fin_addrs is None else fin_addrs
    previous_fin_addr = previous_addr if previous_fin_addr is None else previous_fin_addr
    pv = PVM.finalized_state(*ctx.final)
    machine, ev = {}, []
    for name, which, a_list, recs, previous in (("MemoryGlobalInit", "init", addrs, init, previous_addr),
                                                ("MemoryGlobalFinalize", "finalize", fin_addrs, fin, previous_fin_addr)):
        n = len(a_list)
        PVM.put(pv, "previous_%s_addr" % which, PVM.addr_limbs(previous))
        PVM.put(pv, "last_%s_addr" % which, PVM.addr_limbs(int(a_list[-1]) if n else previous))
        PVM.put(pv, "global_%s_count" % which, n)
        if n == 0:
            continue
        air, it = R.chip(name)
        L = air.layout
        tb = RT.Table(air, n, dev)
        rows = np.zeros((tb.main.shape[0], air.main_width), dtype=np.int64)   # all rows at once: a large program touches millions of words
        addr_np = np.asarray(a_list, dtype=np.uint64)
        rec = recs.astype(np.uint64) if isinstance(recs, np.ndarray) else np.array([[int(x) & ((1 << 64) - 1) for x in r] for r in recs], dtype=np.uint64)
        v, t = rec[:, 0], rec[:, 1].astype(np.int64)
        prev = np.concatenate([np.array([previous], dtype=np.uint64), addr_np[:-1]])
        idx = np.arange(n, dtype=np.int64)
        xl, yl, vl = _limbs_np(prev), _limbs_np(addr_np), _limbs_np(v)
        rows[:n, L["clk_high"]], rows[:n, L["clk_low"]] = t >> 24, t & 0xFFFFFF
        rows[:n, L["index"]] = idx
        rows[:n, L["prev_addr"]:L["prev_addr"] + 3] = xl[:, :3]
        rows[:n, L["addr"]:L["addr"] + 3] = yl[:, :3]
        rows[:n, L["value"]:L["value"] + 4] = vl
        rows[:n, L["value_lower"]], rows[:n, L["value_upper"]] = vl[:, 2] & 0xFF, vl[:, 2] >> 8
        rows[:n, L["is_real"]] = 1
        rows[:n, L["prev_valid"]] = 1 - ((prev == 0) & (idx != 0))
        s_ = xl[:, :3].sum(axis=1)
        rows[:n, L["is_prev_addr_zero.inverse"]], rows[:n, L["is_prev_addr_zero.result"]] = _inv_np(s_), s_ == 0
        rows[:n, L["is_index_zero.inverse"]], rows[:n, L["is_index_zero.result"]] = _inv_np(idx), idx == 0
        comp = (prev != 0) | (idx != 0)
        assert bool((addr_np[comp] > prev[comp]).all()), "addresses must increase strictly"
        j = 3 - np.argmax((xl != yl)[:, ::-1], axis=1)                   # the most significant limb that differs
        xj, yj = xl[idx, j], yl[idx, j]
        rows[:n, L["is_comp"]] = comp
        flags = np.zeros((n, 4), dtype=np.int64)
        flags[idx, j] = comp
        rows[:n, L["lt_cols.u16_flags"]:L["lt_cols.u16_flags"] + 4] = flags
        rows[:n, L["lt_cols.comparison_limbs"]], rows[:n, L["lt_cols.comparison_limbs"] + 1] = xj * comp, yj * comp
        rows[:n, L["lt_cols.not_eq_inv"]] = _inv_np((xj - yj) % P) * comp
        rows[:n, L["lt_cols.bit"]] = (xj < yj) & comp
        # the public values receive (count, last_addr, 1): the chain cannot end on the uncompared row of address 0 alone
        assert bool(comp[-1]), "a memory shard whose only %s event is address 0 does not close its chain" % which
        tb.main[:] = torch.as_tensor(rows % P, device=dev)
        tr.tables[name], machine[name] = tb, (air, it)
        ev += [v_ for _, v_, _ in RT.eval_interactions(it, tb.main[:tb.n], None, kinds=(R.GLOBAL,))]
    tr.global_events = torch.cat(ev)
    PVM.set_global(pv, *tr.global_chip(machine, tr.global_events))
    tr.tables["Program"], machine["Program"] = ctx.program_table(dev)
    tr.byte_range_tables(machine, pv)
    tr.fill_cluster(machine, RT.MEMORY_CLUSTER)
    names = sorted(machine)
    return [machine[n] for n in names], {n: (tr.tables[n].prep, tr.tables[n].main) for n in names}, PVM.to_tensor(pv), tr.global_events


# ---------------------------------------------------------------------------------------------------------------------
# SHA-256 precompile shards (syscall/precompiles/sha256/{extend,compress}/trace.rs), vectorised over rows
M32 = 0xFFFFFFFF


def _half(v):
    return torch.stack([v & MASK16, (v >> 16) & MASK16], dim=-1)


def _set_fixed(tb, prefix, x, r, shift=False):
    """FixedRotateRightOperation / FixedShiftRightOperation::populate (fixed_rotate_right.rs:L37-L61, fixed_shift_right.rs:L36-L64)."""
    nl, nb = r // 16, r % 16
    out = (x >> r) if shift else (((x >> r) | (x << (32 - r))) & M32)
    limbs = _half(x)
    zero = torch.zeros_like(x)
    rot = [limbs[:, i + nl] if i + nl < 2 else zero for i in range(2)] if shift else [limbs[:, nl % 2], limbs[:, (1 + nl) % 2]]
    tb.set(prefix + ".value", _half(out))
    tb.set(prefix + ".higher_limb", torch.stack([rot[0] >> nb, rot[1] >> nb], dim=1))
    return out


def _set_byte_op(tb, prefix, x, y, op):
    """XorU32Operation / AndU32Operation::populate (xor_u32.rs:L26-L59)."""
    out = (x ^ y) if op == "xor" else (x & y)
    for nm, v in (("b_low_bytes", x), ("c_low_bytes", y)):
        tb.set(prefix + "." + nm + ".low_bytes", torch.stack([v & 0xFF, (v >> 16) & 0xFF], dim=1))
    tb.set(prefix + ".value", torch.stack([(out >> (8 * i)) & 0xFF for i in range(4)], dim=1))
    return out


def _set_sum(tb, prefix, *terms):
    out = terms[0]
    for t in terms[1:]:
        out = (out + t) & M32
    tb.set(prefix + ".value", _half(out))
    return out


def sha_extend_shard_from(events, device="cpu", ctx=None):
    """The SHA_EXTEND precompile shard of the executor's events ([n, 786] int64: riscv_exec.SHA_EXTEND_WORDS): ShaExtend (48 rows per
    call, `event_to_rows`, extend/trace.rs:L82-L190), ShaExtendControl, SyscallPrecompile, MemoryLocal, Global, Byte, Range."""
    dev = torch.device(device)
    ev = torch.as_tensor(events, device=dev)
    n = ev.shape[0]
    clk, w_ptr = ev[:, 0], ev[:, 1]
    steps = ev[:, 2:2 + 48 * 11].reshape(n, 48, 11)
    around = ev[:, 2 + 48 * 11:].reshape(n, 64, 4)
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    rows = 48 * n
    tb = RT.Table(R.chip("ShaExtend")[0], rows, dev)
    rep = lambda v: v[:, None].expand(-1, 48).reshape(-1)
    i = (16 + torch.arange(48, device=dev))[None, :].expand(n, -1).reshape(-1)
    c, wp = rep(clk), rep(w_ptr)
    row_low = (c & 0xFFFFFF) + 1                                                          # the controller hands over (clk_high, clk_low + 1)
    tb.set("clk_high", c >> 24); tb.set("clk_low", row_low); tb.set("i", i); tb.set("is_real", 1)
    nxt = row_low + (i - 16)
    over = (nxt >> 24) & 1
    tb.set("next_clk.is_overflow", over)
    tb.set("next_clk.next_clk_16_24", (nxt >> 16) & 0xFF)
    tb.set("next_clk.next_clk_0_16", nxt & MASK16)
    ts = c + 1 + (i - 16)
    tb.set("w_ptr", _limbs_t(wp)[:, :3])
    st = steps.reshape(rows, 11)
    vals = {}
    for k, (name, off) in enumerate((("w_i_minus_15", 15), ("w_i_minus_2", 2), ("w_i_minus_16", 16), ("w_i_minus_7", 7))):
        tb.set(name + "_ptr.value", _limbs_t(wp + 8 * (i - off))[:, :3])
        _mem_access_t(tb, name, st[:, 2 * k + 1], st[:, 2 * k], ts)
        vals[name] = st[:, 2 * k + 1] & M32
    tb.set("w_i_ptr.value", _limbs_t(wp + 8 * i)[:, :3])
    _mem_access_t(tb, "w_i", st[:, 9], st[:, 8], ts)
    x = vals["w_i_minus_15"]
    a7, a18, a3 = _set_fixed(tb, "w_i_minus_15_rr_7", x, 7), _set_fixed(tb, "w_i_minus_15_rr_18", x, 18), _set_fixed(tb, "w_i_minus_15_rs_3", x, 3, shift=True)
    s0 = _set_byte_op(tb, "s0", _set_byte_op(tb, "s0_intermediate", a7, a18, "xor"), a3, "xor")
    x = vals["w_i_minus_2"]
    b17, b19, b10 = _set_fixed(tb, "w_i_minus_2_rr_17", x, 17), _set_fixed(tb, "w_i_minus_2_rr_19", x, 19), _set_fixed(tb, "w_i_minus_2_rs_10", x, 10, shift=True)
    s1 = _set_byte_op(tb, "s1", _set_byte_op(tb, "s1_intermediate", b17, b19, "xor"), b10, "xor")
    s2 = _set_sum(tb, "s2", vals["w_i_minus_16"], s0, vals["w_i_minus_7"], s1)
    assert bool((s2 == (st[:, 10] & M32)).all()), "the executor's w[i] is not the extension of its inputs"
    tr.tables["ShaExtend"] = tb
    ct = RT.Table(R.chip("ShaExtendControl")[0], n, dev)
    ct.set("clk_high", clk >> 24); ct.set("clk_low", clk & 0xFFFFFF); ct.set("is_real", 1)
    al = _syscall_addr_t(ct, "w_ptr", w_ptr)
    for name, off in (("w_16th_addr", 15), ("w_17th_addr", 16), ("w_64th_addr", 63)):
        ct.set(name + ".value", _limbs_t(w_ptr + 8 * off)[:, :3])
    tr.tables["ShaExtendControl"] = ct
    wa = (w_ptr[:, None] + 8 * torch.arange(64, device=dev)[None, :]).reshape(-1)
    ar = around.reshape(-1, 4)
    return _close_precompile_shard(tr, M.SYS_SHA_EXTEND, clk, al, wa, ar[:, 0], ar[:, 2], ar[:, 1], ar[:, 3], ctx=ctx)


def sha_compress_shard_from(events, device="cpu", ctx=None):
    """The SHA_COMPRESS precompile shard of the executor's events ([n, 155] int64): ShaCompress (80 rows per call: 8 initialise, 64
    compress, 8 finalise; padding rows keep cycling the octet flags, compress/trace.rs:L78-L104, L120-L360), ShaCompressControl, ..."""
    dev = torch.device(device)
    ev = torch.as_tensor(events, device=dev)
    n = ev.shape[0]
    clk, w_ptr, h_ptr = ev[:, 0], ev[:, 1], ev[:, 2]
    h_rd = ev[:, 3:19].reshape(n, 8, 2)                                                    # (previous timestamp, value)
    w_rd = ev[:, 19:147].reshape(n, 64, 2)
    h_wr = ev[:, 147:155]
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    rows = 80 * n
    air = R.chip("ShaCompress")[0]
    tb = RT.Table(air, rows, dev)
    total = tb.main.shape[0]
    # flags, index and k on EVERY row, padding included
    idx = torch.arange(total, device=dev) % 80
    octet, onum = idx % 8, idx // 8
    ar = torch.arange(total, device=dev)
    tb.main[ar, tb.L["octet"] + octet] = 1
    tb.main[ar, tb.L["octet_num"] + onum] = 1
    tb.main[:, tb.L["index"]] = idx
    K = torch.tensor(M.SHA_K, dtype=I64, device=dev)
    kk = torch.where((onum >= 1) & (onum <= 8), K[(idx - 8).clamp(0, 63)], torch.zeros_like(idx))
    tb.main[:, tb.L["k"]:tb.L["k"] + 2] = _half(kk)
    # the 80 states of every call
    h0 = h_rd[:, :, 1] & M32                                                              # [n, 8]
    w = w_rd[:, :, 1] & M32
    rotr = lambda v, r: ((v >> r) | (v << (32 - r))) & M32
    state = torch.zeros((n, 80, 8), dtype=I64, device=dev)
    state[:, :8] = h0[:, None, :]
    v = h0.clone()
    for j in range(64):
        state[:, 8 + j] = v
        a_, b_, c_, d_, e_, f_, g_, hh = (v[:, q] for q in range(8))
        s1 = rotr(e_, 6) ^ rotr(e_, 11) ^ rotr(e_, 25)
        ch = (e_ & f_) ^ ((~e_ & M32) & g_)
        t1 = (hh + s1 + ch + int(M.SHA_K[j]) + w[:, j]) & M32
        s0 = rotr(a_, 2) ^ rotr(a_, 13) ^ rotr(a_, 22)
        mj = (a_ & b_) ^ (a_ & c_) ^ (b_ & c_)
        t2 = (s0 + mj) & M32
        v = torch.stack([(t1 + t2) & M32, a_, b_, c_, (d_ + t1) & M32, e_, f_, g_], dim=1)
    state[:, 72:] = v[:, None, :]
    assert bool((((h0 + v) & M32) == (h_wr & M32)).all()), "the executor's digest words are not the compression of its inputs"
    rep = lambda t: t[:, None].expand(-1, 80).reshape(-1)
    c = rep(clk)
    tb.set("clk_high", c >> 24); tb.set("clk_low", c & 0xFFFFFF); tb.set("is_real", 1)
    tb.set("w_ptr", _limbs_t(rep(w_ptr))[:, :3]); tb.set("h_ptr", _limbs_t(rep(h_ptr))[:, :3])
    S = state.reshape(rows, 8)
    for q, nm in enumerate("abcdefgh"):
        tb.set(nm, _half(S[:, q]))
    j = idx[:rows]
    phase = torch.where(j < 8, 0, torch.where(j < 72, 1, 2))
    tb.set("is_initialize", (phase == 0).to(I64)); tb.set("is_compression", (phase == 1).to(I64)); tb.set("is_finalize", (phase == 2).to(I64))
    # the memory access of the row: h[j] read at clk, w[j - 8] read at clk + 1, h[j - 72] written at clk + 2
    hq, wq = (j % 8), (j - 8).clamp(0, 63)
    ev_i = torch.arange(n, device=dev)[:, None].expand(-1, 80).reshape(-1)
    addr = torch.where(phase == 1, rep(w_ptr) + 8 * wq, rep(h_ptr) + 8 * hq)
    t_prev = torch.where(phase == 0, h_rd[ev_i, hq, 0], torch.where(phase == 1, w_rd[ev_i, wq, 0], c))
    prev_v = torch.where(phase == 1, w_rd[ev_i, wq, 1], h_rd[ev_i, hq, 1])
    new_v = torch.where(phase == 2, h_wr[ev_i, hq], prev_v)
    _mem_access_t(tb, "mem", prev_v, t_prev, c + phase)
    tb.set("mem_value", _half(new_v & M32))
    al = _limbs_t(addr)[:, :3]
    tb.set("mem_addr", al)
    for ph, nm in ((0, "mem_addr_init"), (1, "mem_addr_compress"), (2, "mem_addr_finalize")):
        tb.set(nm + ".value", al * (phase == ph).to(I64)[:, None])
    # compression rows
    cm = (phase == 1).to(I64)
    a_, b_, c_, d_, e_, f_, g_, hh = (S[:, q] * cm for q in range(8))
    wv = (prev_v & M32) * cm
    kq = kk[:rows] * cm
    e6, e11, e25 = _set_fixed(tb, "e_rr_6", e_, 6), _set_fixed(tb, "e_rr_11", e_, 11), _set_fixed(tb, "e_rr_25", e_, 25)
    s1 = _set_byte_op(tb, "s1", _set_byte_op(tb, "s1_intermediate", e6, e11, "xor"), e25, "xor")
    eaf = _set_byte_op(tb, "e_and_f", e_, f_, "and")
    en = (~e_) & M32
    tb.set("e_not.value", _half(en))
    eng = _set_byte_op(tb, "e_not_and_g", en, g_, "and")
    ch = _set_byte_op(tb, "ch", eaf, eng, "xor")
    t1 = _set_sum(tb, "temp1", hh, s1, ch, kq, wv)
    a2, a13, a22 = _set_fixed(tb, "a_rr_2", a_, 2), _set_fixed(tb, "a_rr_13", a_, 13), _set_fixed(tb, "a_rr_22", a_, 22)
    s0 = _set_byte_op(tb, "s0", _set_byte_op(tb, "s0_intermediate", a2, a13, "xor"), a22, "xor")
    ab, ac, bc = _set_byte_op(tb, "a_and_b", a_, b_, "and"), _set_byte_op(tb, "a_and_c", a_, c_, "and"), _set_byte_op(tb, "b_and_c", b_, c_, "and")
    mj = _set_byte_op(tb, "maj", _set_byte_op(tb, "maj_intermediate", ab, ac, "xor"), bc, "xor")
    t2 = _set_sum(tb, "temp2", s0, mj)
    _set_sum(tb, "d_add_temp1", d_, t1)
    _set_sum(tb, "temp1_add_temp2", t1, t2)
    # every column group above was written on all rows with the compression mask applied to its INPUTS: a non-compression row holds
    # the operations' values on zero inputs — zero, except e_not = 0xffff 0xffff — which its constraints do not read; the reference
    # leaves those rows zero, so clear them
    first, last = tb.L["e_rr_6.value"], tb.L["temp1_add_temp2.value"] + 2
    tb.main[:rows, first:last] *= cm[:, None]
    # finalise rows
    fm = (phase == 2).to(I64)
    operand = S[torch.arange(rows, device=dev), hq] * fm
    tb.set("finalized_operand", _half(operand))
    tb.set("finalize_add.value", _half(((prev_v & M32) + operand) & M32) * fm[:, None])
    tr.tables["ShaCompress"] = tb
    ct = RT.Table(R.chip("ShaCompressControl")[0], n, dev)
    ct.set("clk_high", clk >> 24); ct.set("clk_low", clk & 0xFFFFFF); ct.set("is_real", 1)
    wl = _syscall_addr_t(ct, "w_ptr", w_ptr)
    hl = _syscall_addr_t(ct, "h_ptr", h_ptr)
    ct.set("w_slice_end.value", _limbs_t(w_ptr + 63 * 8)[:, :3]); ct.set("h_slice_end.value", _limbs_t(h_ptr + 7 * 8)[:, :3])
    ct.set("initial_state", _half(h0).reshape(n, 16)); ct.set("final_state", _half(v).reshape(n, 16))
    tr.tables["ShaCompressControl"] = ct
    # local memory: the eight h words (read at clk, written at clk + 2), the 64 w words (read at clk + 1)
    eight, sixty4 = torch.arange(8, device=dev)[None, :], torch.arange(64, device=dev)[None, :]
    wa = torch.cat([(h_ptr[:, None] + 8 * eight).reshape(-1), (w_ptr[:, None] + 8 * sixty4).reshape(-1)])
    t_i = torch.cat([h_rd[:, :, 0].reshape(-1), w_rd[:, :, 0].reshape(-1)])
    v_i = torch.cat([h_rd[:, :, 1].reshape(-1), w_rd[:, :, 1].reshape(-1)])
    t_f = torch.cat([(clk[:, None] + 2).expand(-1, 8).reshape(-1), (clk[:, None] + 1).expand(-1, 64).reshape(-1)])
    v_f = torch.cat([h_wr.reshape(-1), w_rd[:, :, 1].reshape(-1)])
    return _close_precompile_shard(tr, M.SYS_SHA_COMPRESS, clk, wl, wa, t_i, t_f, v_i, v_f, arg2_limbs=hl, ctx=ctx)   # arg2 = h_ptr


# ---------------------------------------------------------------------------------------------------------------------
# field operations on byte limbs (operations/field/field_op.rs populate_*; Python integers: precompile calls are few)
def field_op_columns(a, b, modulus, n_limbs, n_witness, op="mul", n_modulus_limbs=None):
    """FieldOpCols::populate_with_modulus (field_op.rs:L224-L300): (result, carry, witness) byte / u16 lists and the integer
    result, for result = a op b mod modulus with op in {"add", "mul"} (sub / div are the same identities with a and result swapped)."""
    n_mod = n_modulus_limbs or n_limbs
    val = a + b if op == "add" else a * b
    result = val % modulus
    carry = (val - result) // modulus
    by = lambda v, n: [(v >> (8 * i)) & 0xFF for i in range(n)]
    pa, pb, pr, pc, pm = by(a, n_limbs), by(b, n_limbs), by(result, n_limbs), by(carry, n_limbs), by(modulus, n_mod)
    van = [0] * (n_witness + 1)
    if op == "add":
        for i in range(n_limbs):
            van[i] += pa[i] + pb[i]
    else:
        for i in range(n_limbs):
            for j in range(n_limbs):
                van[i + j] += pa[i] * pb[j]
    for i in range(n_limbs):
        van[i] -= pr[i]
    for i in range(n_limbs):
        for j in range(n_mod):
            van[i + j] -= pc[i] * pm[j]
    # divide by (x - 256): synthetic division from the top (field_op.rs:L71-L78)
    w = [0] * n_witness
    acc = van[n_witness]
    for i in range(n_witness - 1, -1, -1):
        w[i] = acc
        acc = van[i] + acc * 256
    assert acc == 0, "the vanishing polynomial does not vanish at 256"
    witness = [x + (1 << 14) for x in w]
    assert all(0 <= x < (1 << 16) for x in witness)
    return pr, pc, witness, result


def field_lt_columns(lhs, rhs, n_limbs):
    """FieldLtCols::populate (field/range.rs:L30-L61): byte flags and the two comparison bytes for lhs < rhs."""
    assert lhs < rhs
    flags, lb, rb = [0] * n_limbs, 0, 0
    for i in range(n_limbs - 1, -1, -1):
        x, y = (lhs >> (8 * i)) & 0xFF, (rhs >> (8 * i)) & 0xFF
        if x < y:
            flags[i], lb, rb = 1, x, y
            break
    return flags, lb, rb


def uint256_shard_from(events, device="cpu", ctx=None):
    """The UINT256_MUL precompile shard of the executor's events ([n, 31] int64): Uint256MulMod rows (`generate_trace_into`,
    syscall/precompiles/uint256/air.rs:L118-L290), SyscallPrecompile, MemoryLocal, Global, Byte, Range."""
    dev = torch.device(device)
    ev = np.asarray(events).astype(np.uint64)
    n = ev.shape[0]
    air = R.chip("Uint256MulMod")[0]
    L = air.layout
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    tb = RT.Table(air, n, dev)
    rows = np.zeros((tb.main.shape[0], air.main_width), dtype=np.int64)
    u = lambda v: int(v)
    word = lambda ws: sum(u(x) << (64 * i) for i, x in enumerate(ws))
    for r in range(n):
        e = ev[r]
        clk, xp, yp = u(e[0]), u(e[1]), u(e[2])
        xs, ys = e[3:11].reshape(4, 2), e[11:27].reshape(8, 2)
        x, y, m = word(xs[:, 1]), word(ys[:4, 1]), word(ys[4:, 1])
        res, car, wit, out = field_op_columns(x, y, m if m else 1 << 256, 32, 63, "mul", n_modulus_limbs=33)
        assert out == word(e[27:31]), "the executor's product is not x * y mod modulus"
        rows[r, L["output.result"]:L["output.result"] + 32] = res
        rows[r, L["output.carry"]:L["output.carry"] + 32] = car
        rows[r, L["output.witness"]:L["output.witness"] + 63] = wit
        s = sum((m >> (8 * i)) & 0xFF for i in range(32))
        rows[r, L["modulus_is_zero.inverse"]] = pow(s, P - 2, P) if s else 0
        rows[r, L["modulus_is_zero.result"]] = int(s == 0)
        rows[r, L["modulus_is_not_zero"]] = int(s != 0)
        if m:
            flags, lb, rb = field_lt_columns(out, m, 32)
            rows[r, L["output_range_check.byte_flags"]:L["output_range_check.byte_flags"] + 32] = flags
            rows[r, L["output_range_check.lhs_comparison_byte"]], rows[r, L["output_range_check.rhs_comparison_byte"]] = lb, rb
    rows[n:, L["output.witness"]:L["output.witness"] + 63] = 1 << 14       # padding rows: the field operation on zero operands (air.rs:L266-L280)
    tb.main[:] = torch.as_tensor(rows, device=dev)
    t = torch.as_tensor(ev.astype(np.int64), device=dev)
    clk, xp, yp = t[:, 0], t[:, 1], t[:, 2]
    tb.set("clk_high", clk >> 24); tb.set("clk_low", clk & 0xFFFFFF); tb.set("is_real", 1)
    xl = _syscall_addr_t(tb, "x_ptr", xp)
    yl = _syscall_addr_t(tb, "y_ptr", yp)
    xs, ys = t[:, 3:11].reshape(n, 4, 2), t[:, 11:27].reshape(n, 8, 2)
    low_bytes = lambda v: torch.stack([(v >> (16 * k)) & 0xFF for k in range(4)], dim=1)
    for i in range(4):
        tb.set("x_addrs.%d.value" % i, _limbs_t(xp + 8 * i)[:, :3])
        _mem_access_t(tb, "x_memory.%d.memory_access" % i, xs[:, i, 1], xs[:, i, 0], clk + 1)
        tb.set("x_memory.%d.prev_value_u8.low_bytes" % i, low_bytes(xs[:, i, 1]))
    for i in range(8):
        tb.set("y_and_modulus_addrs.%d.value" % i, _limbs_t(yp + 8 * i)[:, :3])
        name = ("y_memory.%d" % i) if i < 4 else ("modulus_memory.%d" % (i - 4))
        _mem_access_t(tb, name + ".memory_access", ys[:, i, 1], ys[:, i, 0], clk)
        tb.set(name + ".prev_value_u8.low_bytes", low_bytes(ys[:, i, 1]))
    tr.tables["Uint256MulMod"] = tb
    four, eight = torch.arange(4, device=dev)[None, :], torch.arange(8, device=dev)[None, :]
    wa = torch.cat([(xp[:, None] + 8 * four).reshape(-1), (yp[:, None] + 8 * eight).reshape(-1)])
    t_i = torch.cat([xs[:, :, 0].reshape(-1), ys[:, :, 0].reshape(-1)])
    v_i = torch.cat([xs[:, :, 1].reshape(-1), ys[:, :, 1].reshape(-1)])
    t_f = torch.cat([(clk[:, None] + 1).expand(-1, 4).reshape(-1), clk[:, None].expand(-1, 8).reshape(-1)])
    v_f = torch.cat([t[:, 27:31].reshape(-1), ys[:, :, 1].reshape(-1)])
    return _close_precompile_shard(tr, M.SYS_UINT256_MUL, clk, xl, wa, t_i, t_f, v_i, v_f, arg2_limbs=yl, ctx=ctx)


# ---------------------------------------------------------------------------------------------------------------------
# secp256k1 point addition / doubling (syscall/precompiles/weierstrass/weierstrass_{add,double}.rs)
def _le_bytes(vals, n):
    return np.frombuffer(b"".join(int(v).to_bytes(n, "little") for v in vals), dtype=np.uint8).reshape(len(vals), n).astype(np.int64)


def field_op_columns_batch(A, B, modulus, n_limbs, n_witness, op, offset=1 << 14):
    """`field_op_columns` for lists of operands: the integer arithmetic stays with Python (a few microseconds per row), the
    byte-limb polynomial identity and its division by (x - 256) run over all rows at once."""
    vals = [a + b for a, b in zip(A, B)] if op == "add" else [a * b for a, b in zip(A, B)]
    res = [v % modulus for v in vals]
    pa, pb, pr = _le_bytes(A, n_limbs), _le_bytes(B, n_limbs), _le_bytes(res, n_limbs)
    pc = _le_bytes([v // modulus for v in vals], n_limbs)
    pm = [(modulus >> (8 * j)) & 0xFF for j in range(n_limbs)]
    van = np.zeros((len(A), n_witness + 1), dtype=np.int64)
    if op == "add":
        van[:, :n_limbs] += pa + pb
    else:
        for i in range(n_limbs):
            van[:, i:i + n_limbs] += pa[:, i:i + 1] * pb
    van[:, :n_limbs] -= pr
    for j in range(n_limbs):
        if pm[j]:
            van[:, j:j + n_limbs] -= pc * pm[j]
    wit = np.zeros((len(A), n_witness), dtype=np.int64)
    acc = van[:, n_witness].copy()
    for i in range(n_witness - 1, -1, -1):                              # synthetic division from the top (field_op.rs:L71-L78)
        wit[:, i] = acc
        acc = van[:, i] + acc * 256
    assert not acc.any(), "the vanishing polynomial does not vanish at 256"
    wit += offset
    assert wit.min() >= 0 and wit.max() < (1 << 16)
    return pr, pc, wit, res


def _set_field_op(rows, L, prefix, A, B, op, modulus=M.SECP256K1_P, n_limbs=32, n_witness=62, offset=1 << 14):
    """FieldOpCols::populate_with_modulus (field_op.rs:L286-L345) into rows[:len(A)]: sub / div are stated as result + b = a and
    result * b = a, with the `result` columns holding the difference / quotient."""
    n = len(A)
    if op == "sub":
        result = [(modulus + a - b) % modulus for a, b in zip(A, B)]
        _, car, wit, back = field_op_columns_batch(result, B, modulus, n_limbs, n_witness, "add", offset)
        assert back == [a % modulus for a in A]
        res = _le_bytes(result, n_limbs)
    elif op == "div":
        assert all(b % modulus for b in B), "division by zero is not allowed"
        result = [a * pow(b, modulus - 2, modulus) % modulus for a, b in zip(A, B)]
        _, car, wit, back = field_op_columns_batch(result, B, modulus, n_limbs, n_witness, "mul", offset)
        assert back == [a % modulus for a in A]
        res = _le_bytes(result, n_limbs)
    else:
        res, car, wit, result = field_op_columns_batch(A, B, modulus, n_limbs, n_witness, op, offset)
    rows[:n, L[prefix + ".result"]:L[prefix + ".result"] + n_limbs] = res
    rows[:n, L[prefix + ".carry"]:L[prefix + ".carry"] + n_limbs] = car
    rows[:n, L[prefix + ".witness"]:L[prefix + ".witness"] + n_witness] = wit
    return result


def _set_field_lt(rows, L, prefix, lhs, rhs=M.SECP256K1_P, n_limbs=32):
    """FieldLtCols::populate (field/range.rs:L30-L61) into rows[:len(lhs)]: the flag at the most significant differing byte."""
    n = len(lhs)
    assert all(x < rhs for x in lhs)
    x = _le_bytes(lhs, n_limbs)
    y = np.array([(rhs >> (8 * j)) & 0xFF for j in range(n_limbs)], dtype=np.int64)[None, :]
    at = n_limbs - 1 - np.argmax((x != y)[:, ::-1], axis=1)
    r = np.arange(n)
    flags = np.zeros((n, n_limbs), dtype=np.int64)
    flags[r, at] = 1
    rows[:n, L[prefix + ".byte_flags"]:L[prefix + ".byte_flags"] + n_limbs] = flags
    rows[:n, L[prefix + ".lhs_comparison_byte"]] = x[r, at]
    rows[:n, L[prefix + ".rhs_comparison_byte"]] = y[0, at]


def _curve_ops(row, L, curve):
    modulus, _, nl, nw, off = M.CURVES[curve]
    return (lambda prefix, A, B, op: _set_field_op(row, L, prefix, A, B, op, modulus, nl, nw, off),
            lambda prefix, lhs: _set_field_lt(row, L, prefix, lhs, modulus, nl))


def _secp_add_field_ops(row, L, px, py, qx, qy, curve="Secp256k1"):      # populate_field_ops, weierstrass_add.rs:L95-L165
    fo, lt = _curve_ops(row, L, curve)
    num = fo("slope_numerator", qy, py, "sub")
    den = fo("slope_denominator", qx, px, "sub")
    fo("inverse_check", [1] * len(den), den, "div")
    slope = fo("slope", num, den, "div")
    sq = fo("slope_squared", slope, slope, "mul")
    pq = fo("p_x_plus_q_x", px, qx, "add")
    x3 = fo("x3_ins", sq, pq, "sub")
    lt("x3_range", x3)
    d = fo("p_x_minus_x", px, x3, "sub")
    sd = fo("slope_times_p_x_minus_x", slope, d, "mul")
    y3 = fo("y3_ins", sd, py, "sub")
    lt("y3_range", y3)
    return x3, y3


def _secp_double_field_ops(row, L, px, py, curve="Secp256k1"):           # populate_field_ops, weierstrass_double.rs:L89-L160
    fo, lt = _curve_ops(row, L, curve)
    a_coeff = M.CURVES[curve][1]
    xx = fo("p_x_squared", px, px, "mul")
    xx3 = fo("p_x_squared_times_3", xx, [3] * len(px), "mul")
    num = fo("slope_numerator", [a_coeff] * len(px), xx3, "add")
    den = fo("slope_denominator", [2] * len(px), py, "mul")
    slope = fo("slope", num, den, "div")
    sq = fo("slope_squared", slope, slope, "mul")
    pp = fo("p_x_plus_p_x", px, px, "add")
    x3 = fo("x3_ins", sq, pp, "sub")
    lt("x3_range", x3)
    d = fo("p_x_minus_x", px, x3, "sub")
    sd = fo("slope_times_p_x_minus_x", slope, d, "mul")
    y3 = fo("y3_ins", sd, py, "sub")
    lt("y3_range", y3)
    return x3, y3


def _dummy_access(row, L, prefix):
    """A MemoryAccessColsU8 populated from the reference's dummy record {value 1, timestamp 1, prev_timestamp 0}: the padding rows'
    operands are small non-zero constants so that the (unconditional) field operations have an inverse to work with."""
    row[L[prefix + ".memory_access.prev_value"]] = 1
    row[L[prefix + ".memory_access.compare_low"]] = 1
    row[L[prefix + ".prev_value_u8.low_bytes"]] = 1


_word256 = lambda ws: sum(int(x) << (64 * i) for i, x in enumerate(ws))
_low_bytes = lambda v: torch.stack([(v >> (16 * k)) & 0xFF for k in range(4)], dim=1)


def secp256k1_add_shard_from(events, device="cpu", ctx=None):
    """The SECP256K1_ADD precompile shard of the executor's events ([n, 43] int64): Secp256k1AddAssign rows (`populate_row` and
    the dummy row of `generate_trace_into`, weierstrass_add.rs:L246-L330, L612-L680), SyscallPrecompile, MemoryLocal, Global, Byte, Range."""
    dev = torch.device(device)
    ev = np.asarray(events).astype(np.uint64)
    n = ev.shape[0]
    air = R.chip("Secp256k1AddAssign")[0]
    L = air.layout
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    tb = RT.Table(air, n, dev)
    rows = np.zeros((tb.main.shape[0], air.main_width), dtype=np.int64)
    coord = lambda cols: [_word256(w) for w in ev[:, cols]]
    if n:
        x3, y3 = _secp_add_field_ops(rows, L, coord([4, 6, 8, 10]), coord([12, 14, 16, 18]), coord([20, 22, 24, 26]), coord([28, 30, 32, 34]))
        assert x3 == coord([35, 36, 37, 38]) and y3 == coord([39, 40, 41, 42]), "the executor's sum is not p + q"
    if rows.shape[0] > n:
        _secp_add_field_ops(rows[n:n + 1], L, [0], [0], [1], [1])
        _dummy_access(rows[n], L, "q_access.0")
        _dummy_access(rows[n], L, "q_access.4")
        rows[n + 1:] = rows[n]
    tb.main[:] = torch.as_tensor(rows, device=dev)
    t = torch.as_tensor(ev.astype(np.int64), device=dev)
    clk, pp, qp = t[:, 0], t[:, 1], t[:, 2]
    tb.set("clk_high", clk >> 24); tb.set("clk_low", clk & 0xFFFFFF); tb.set("is_real", 1)
    pl = _syscall_addr_t(tb, "p_ptr", pp)
    ql = _syscall_addr_t(tb, "q_ptr", qp)
    ps, qs = t[:, 3:19].reshape(n, 8, 2), t[:, 19:35].reshape(n, 8, 2)
    for i in range(8):
        tb.set("p_addrs.%d.value" % i, _limbs_t(pp + 8 * i)[:, :3])
        tb.set("q_addrs.%d.value" % i, _limbs_t(qp + 8 * i)[:, :3])
        _mem_access_t(tb, "p_access.%d.memory_access" % i, ps[:, i, 1], ps[:, i, 0], clk + 1)
        tb.set("p_access.%d.prev_value_u8.low_bytes" % i, _low_bytes(ps[:, i, 1]))
        _mem_access_t(tb, "q_access.%d.memory_access" % i, qs[:, i, 1], qs[:, i, 0], clk)
        tb.set("q_access.%d.prev_value_u8.low_bytes" % i, _low_bytes(qs[:, i, 1]))
    tr.tables["Secp256k1AddAssign"] = tb
    eight = torch.arange(8, device=dev)[None, :]
    wa = torch.cat([(pp[:, None] + 8 * eight).reshape(-1), (qp[:, None] + 8 * eight).reshape(-1)])
    t_i = torch.cat([ps[:, :, 0].reshape(-1), qs[:, :, 0].reshape(-1)])
    v_i = torch.cat([ps[:, :, 1].reshape(-1), qs[:, :, 1].reshape(-1)])
    t_f = torch.cat([(clk[:, None] + 1).expand(-1, 8).reshape(-1), clk[:, None].expand(-1, 8).reshape(-1)])
    v_f = torch.cat([t[:, 35:43].reshape(-1), qs[:, :, 1].reshape(-1)])
    return _close_precompile_shard(tr, M.SYS_SECP256K1_ADD, clk, pl, wa, t_i, t_f, v_i, v_f, arg2_limbs=ql, ctx=ctx)


def secp256k1_double_shard_from(events, device="cpu", ctx=None):
    """The SECP256K1_DOUBLE precompile shard of the executor's events ([n, 26] int64): Secp256k1DoubleAssign rows
    (weierstrass_double.rs:L262-L380: the point is rewritten in place at clk), SyscallPrecompile, MemoryLocal, Global, Byte, Range."""
    dev = torch.device(device)
    ev = np.asarray(events).astype(np.uint64)
    n = ev.shape[0]
    air = R.chip("Secp256k1DoubleAssign")[0]
    L = air.layout
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    tb = RT.Table(air, n, dev)
    rows = np.zeros((tb.main.shape[0], air.main_width), dtype=np.int64)
    coord = lambda cols: [_word256(w) for w in ev[:, cols]]
    if n:
        x3, y3 = _secp_double_field_ops(rows, L, coord([3, 5, 7, 9]), coord([11, 13, 15, 17]))
        assert x3 == coord([18, 19, 20, 21]) and y3 == coord([22, 23, 24, 25]), "the executor's point is not 2 p"
    if rows.shape[0] > n:
        _secp_double_field_ops(rows[n:n + 1], L, [0], [1])
        _dummy_access(rows[n], L, "p_access.4")
        rows[n + 1:] = rows[n]
    tb.main[:] = torch.as_tensor(rows, device=dev)
    t = torch.as_tensor(ev.astype(np.int64), device=dev)
    clk, pp = t[:, 0], t[:, 1]
    tb.set("clk_high", clk >> 24); tb.set("clk_low", clk & 0xFFFFFF); tb.set("is_real", 1)
    pl = _syscall_addr_t(tb, "p_ptr", pp)
    ps = t[:, 2:18].reshape(n, 8, 2)
    for i in range(8):
        tb.set("p_addrs.%d.value" % i, _limbs_t(pp + 8 * i)[:, :3])
        _mem_access_t(tb, "p_access.%d.memory_access" % i, ps[:, i, 1], ps[:, i, 0], clk)
        tb.set("p_access.%d.prev_value_u8.low_bytes" % i, _low_bytes(ps[:, i, 1]))
    tr.tables["Secp256k1DoubleAssign"] = tb
    eight = torch.arange(8, device=dev)[None, :]
    wa = (pp[:, None] + 8 * eight).reshape(-1)
    return _close_precompile_shard(tr, M.SYS_SECP256K1_DOUBLE, clk, pl, wa, ps[:, :, 0].reshape(-1), clk[:, None].expand(-1, 8).reshape(-1),
                                   ps[:, :, 1].reshape(-1), t[:, 18:26].reshape(-1), ctx=ctx)


# ---------------------------------------------------------------------------------------------------------------------
# The other field / curve precompiles (sp1hip_rv64_precompile_events): one builder over the shapes they share
_int_of = lambda ws: sum(int(x) << (64 * i) for i, x in enumerate(ws))
# kind: (chip, shape, column prefixes of the two operands, u64 words per operand)
FAMILY_SHAPES = {
    **{cv.lower() + "_add": (cv + "AddAssign", "curve_add", ("p", "q"), M.CURVES[cv][2] // 4) for cv in ("Secp256r1", "Bn254", "Bls12381")},
    **{cv.lower() + "_double": (cv + "DoubleAssign", "curve_double", ("p",), M.CURVES[cv][2] // 4) for cv in ("Secp256r1", "Bn254", "Bls12381")},
    **{f.lower() + "_fp": (f + "FpOpAssign", "fp", ("x", "y"), M.FP_FIELDS[f][1] // 8) for f in M.FP_FIELDS},
    **{f.lower() + "_fp2_addsub": (f + "Fp2AddSubAssign", "fp2_addsub", ("x", "y"), M.FP_FIELDS[f][1] // 4) for f in M.FP_FIELDS},
    **{f.lower() + "_fp2_mul": (f + "Fp2MulAssign", "fp2_mul", ("x", "y"), M.FP_FIELDS[f][1] // 4) for f in M.FP_FIELDS},
    "ed_add": ("EdAddAssign", "ed_add", ("x", "y"), 8), "ed_decompress": ("EdDecompress", "ed_decompress", (), 4),
    "uint256_ops": ("Uint256Ops", "uint256_ops", (), 4),
}


def _fp_field_ops(rows, L, chip_name, shape, code, xs, ys):
    """The FieldOpCols of the tower chips (populate_field_ops of fptower/{fp,fp2_addsub,fp2_mul}.rs) for operand lists xs / ys
    (one or two components each, lists of integers per row) and the rows' system-call codes; returns the result components."""
    field = "Bn254" if chip_name.startswith("Bn254") else "Bls12381"
    modulus, nl, nw, off = M.FP_FIELDS[field]
    fo = lambda prefix, A, B, op, r=rows: _set_field_op(r, L, prefix, A, B, op, modulus, nl, nw, off)
    lt = lambda prefix, lhs, r=rows: _set_field_lt(r, L, prefix, lhs, modulus, nl)
    n = len(xs[0])
    k = np.asarray([int(c) & 0xFF for c in code]) - M.FP_SYSCALLS[field][0]         # 0..5 as in FP_SYSCALLS
    if shape == "fp2_mul":
        a0b0, a1b1 = fo("a0_mul_b0", xs[0], ys[0], "mul"), fo("a1_mul_b1", xs[1], ys[1], "mul")
        a0b1, a1b0 = fo("a0_mul_b1", xs[0], ys[1], "mul"), fo("a1_mul_b0", xs[1], ys[0], "mul")
        out = [fo("c0", a0b0, a1b1, "sub"), fo("c1", a0b1, a1b0, "add")]
        lt("c0_range", out[0]); lt("c1_range", out[1])
        return out
    names = ("output",) if shape == "fp" else ("c0", "c1")
    base = 0 if shape == "fp" else 3
    out = [[0] * n for _ in names]
    for op_i, op in enumerate(("add", "sub", "mul")[:3 if shape == "fp" else 2]):     # rows of one operation together
        at = np.nonzero(k == base + op_i)[0]
        if at.size == 0:
            continue
        sub = rows[at]
        for c, name in enumerate(names):
            res = _set_field_op(sub, L, name, [xs[c][i] for i in at], [ys[c][i] for i in at], op, modulus, nl, nw, off)
            _set_field_lt(sub, L, name + "_range", res, modulus, nl)
            for i, v in zip(at, res):
                out[c][i] = v
        if shape == "fp":
            sub[:, L["is_" + op]] = 1
        else:
            sub[:, L["is_add"]] = int(op == "add")
        rows[at] = sub
    return out


def _ed_field_ops(rows, L, x1, y1, x2, y2):                               # populate_field_ops, edwards/ed_add.rs:L94-L130
    Pm, D = M.ED25519_P, M.ED25519_D
    n = len(x1)
    fo = lambda prefix, A, B, op: _set_field_op(rows, L, prefix, A, B, op, Pm)

    def witness(prefix, van_terms, result, carry):
        """result / carry / witness columns of the identity sum(a_i * b_i) (+ linear terms) = result-side + carry * p, given as the
        integer polynomial `van` per row (FieldInnerProductCols / FieldDenCols populate: division of the vanishing polynomial by x - 256)."""
        van = van_terms
        pc = _le_bytes(carry, 32)
        pm = [(Pm >> (8 * j)) & 0xFF for j in range(32)]
        for j in range(32):
            if pm[j]:
                van[:, j:j + 32] -= pc * pm[j]
        wit = np.zeros((n, 62), dtype=np.int64)
        acc = van[:, 62].copy()
        for i in range(61, -1, -1):
            wit[:, i] = acc
            acc = van[:, i] + acc * 256
        assert not acc.any(), "the vanishing polynomial does not vanish at 256"
        wit += 1 << 14
        assert wit.min() >= 0 and wit.max() < (1 << 16)
        rows[:n, L[prefix + ".result"]:L[prefix + ".result"] + 32] = _le_bytes(result, 32)
        rows[:n, L[prefix + ".carry"]:L[prefix + ".carry"] + 32] = pc
        rows[:n, L[prefix + ".witness"]:L[prefix + ".witness"] + 62] = wit

    def conv(A, B):
        pa, pb = _le_bytes(A, 32), _le_bytes(B, 32)
        out = np.zeros((n, 63), dtype=np.int64)
        for i in range(32):
            out[:, i:i + 32] += pa[:, i:i + 1] * pb
        return out

    def inner(prefix, A, B):                                              # FieldInnerProductCols::populate (field_inner_product.rs:L29-L103)
        ip = [a0 * b0 + a1 * b1 for a0, b0, a1, b1 in zip(A[0], B[0], A[1], B[1])]
        res = [v % Pm for v in ip]
        van = conv(A[0], B[0]) + conv(A[1], B[1])
        van[:, :32] -= _le_bytes(res, 32)
        witness(prefix, van, res, [v // Pm for v in ip])
        return res

    def den(prefix, a, b, sign):                                          # FieldDenCols::populate (field_den.rs:L29-L110)
        dens = [((bv if sign else Pm - bv) + 1) % Pm for bv in b]
        assert all(dens), "a denominator vanishes"
        res = [av * pow(dv, Pm - 2, Pm) % Pm for av, dv in zip(a, dens)]
        lhs = [bv * r + (r if sign else av) for av, bv, r in zip(a, b, res)]
        rhs = a if sign else res
        car = [(lv - rv) // Pm for lv, rv in zip(lhs, rhs)]
        assert all((lv - rv) % Pm == 0 for lv, rv in zip(lhs, rhs))
        van = conv(b, res)
        van[:, :32] += (_le_bytes(res, 32) - _le_bytes(a, 32)) * (1 if sign else -1)
        witness(prefix, van, res, car)
        return res

    xn = inner("x3_numerator", [x1, x2], [y2, y1])
    yn = inner("y3_numerator", [y1, x1], [y2, x2])
    x1y1, x2y2 = fo("x1_mul_y1", x1, y1, "mul"), fo("x2_mul_y2", x2, y2, "mul")
    f = fo("f", x1y1, x2y2, "mul")
    df = fo("d_mul_f", f, [D] * n, "mul")
    x3 = den("x3_ins", xn, df, True)
    y3 = den("y3_ins", yn, df, False)
    _set_field_lt(rows, L, "x3_range", x3, Pm)
    _set_field_lt(rows, L, "y3_range", y3, Pm)
    return x3, y3


def ed25519_even_sqrt(a):
    """`ed25519_sqrt` (curves/src/edwards/ed25519.rs:L75-L120): the even square root of a, or None."""
    Pm = M.ED25519_P
    beta = pow(a, (Pm + 3) // 8, Pm)
    if beta * beta % Pm == (Pm - a) % Pm:
        beta = beta * 19681161376707505956807079304988542015446066515923890162744021073123829784752 % Pm
    if beta * beta % Pm != a % Pm:
        return None
    return Pm - beta if beta & 1 else beta


def _ed_decompress_field_ops(rows, L, ys):                                # populate_field_ops, edwards/ed_decompress.rs:L157-L180
    Pm, D = M.ED25519_P, M.ED25519_D
    n = len(ys)
    fo = lambda prefix, A, B, op: _set_field_op(rows, L, prefix, A, B, op, Pm)
    _set_field_lt(rows, L, "y_range", ys, Pm)
    yy = fo("yy", ys, ys, "mul")
    u = fo("u", yy, [1] * n, "sub")
    dyy = fo("dyy", [D] * n, yy, "mul")
    v = fo("v", [1] * n, dyy, "add")
    uv = fo("u_div_v", u, v, "div")
    xs = [ed25519_even_sqrt(a) for a in uv]
    assert all(x is not None for x in xs), "u / v is not a square"
    sq = fo("x.multiplication", xs, xs, "mul")                            # FieldSqrtCols::populate (field_sqrt.rs:L29-L70)
    assert sq == uv
    rows[:n, L["x.multiplication.result"]:L["x.multiplication.result"] + 32] = _le_bytes(xs, 32)
    _set_field_lt(rows, L, "x.range", xs, Pm)
    rows[:n, L["x.lsb"]] = 0
    neg = fo("neg_x", [0] * n, xs, "sub")
    _set_field_lt(rows, L, "neg_x_range", neg, Pm)
    return xs, neg


def family_shard_from(kind, events, device="cpu", ctx=None):
    """The precompile shard of one family's events ([n, words] int64, the layouts of include/sp1hip.h): the chip's rows (the
    reference's `populate_field_ops` / padding rows, cited at each filler), SyscallPrecompile, MemoryLocal, Global, Byte, Range.
    Every result the executor wrote is recomputed here from the operands it read and compared."""
    chip_name, shape, names, nw = FAMILY_SHAPES[kind]
    dev = torch.device(device)
    ev = np.asarray(events).astype(np.uint64)
    n = ev.shape[0]
    air = R.chip(chip_name)[0]
    L = air.layout
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    tb = RT.Table(air, n, dev)
    rows = np.zeros((tb.main.shape[0], air.main_width), dtype=np.int64)
    t = torch.as_tensor(ev.astype(np.int64), device=dev)
    clk, a1, a2, code = t[:, 0], t[:, 1], t[:, 2], t[:, 3] & 0xFF
    pad = rows.shape[0] > n
    ints = lambda block, lo, hi: [_int_of(w) for w in block[:, lo:hi]]    # block [n, words] of u64 values -> integers
    arange = lambda k: torch.arange(k, device=dev)[None, :]
    if shape in ("curve_add", "curve_double", "fp", "fp2_addsub", "fp2_mul", "ed_add"):
        two = len(names) == 2
        xs = ev[:, 4:4 + 2 * nw].reshape(n, nw, 2)
        ys = ev[:, 4 + 2 * nw:4 + 4 * nw].reshape(n, nw, 2) if two else None
        out = ev[:, 4 + (4 if two else 2) * nw:]
        h = nw // 2
        if shape.startswith("curve"):
            curve = chip_name[:chip_name.index("Add" if two else "Double")]
            if n:
                got = (_secp_add_field_ops(rows, L, ints(xs[:, :, 1], 0, h), ints(xs[:, :, 1], h, nw), ints(ys[:, :, 1], 0, h), ints(ys[:, :, 1], h, nw), curve)
                       if two else _secp_double_field_ops(rows, L, ints(xs[:, :, 1], 0, h), ints(xs[:, :, 1], h, nw), curve))
            if pad:                                                      # the dummy row of generate_trace_into (weierstrass_add.rs:L290-L330, _double.rs:L300-L340)
                if two:
                    _secp_add_field_ops(rows[n:n + 1], L, [0], [0], [1], [1], curve)
                    _dummy_access(rows[n], L, "q_access.0"); _dummy_access(rows[n], L, "q_access.%d" % h)
                else:
                    _secp_double_field_ops(rows[n:n + 1], L, [0], [1], curve)
                    _dummy_access(rows[n], L, "p_access.%d" % h)
        elif shape == "ed_add":
            if n:
                got = _ed_field_ops(rows, L, ints(xs[:, :, 1], 0, 4), ints(xs[:, :, 1], 4, 8), ints(ys[:, :, 1], 0, 4), ints(ys[:, :, 1], 4, 8))
            if pad:
                _ed_field_ops(rows[n:n + 1], L, [0], [0], [0], [0])
        else:
            comps = 1 if shape == "fp" else 2
            w = nw // comps
            if n:
                got = _fp_field_ops(rows, L, chip_name, shape, ev[:, 3], [ints(xs[:, :, 1], c * w, (c + 1) * w) for c in range(comps)],
                                    [ints(ys[:, :, 1], c * w, (c + 1) * w) for c in range(comps)])
            if pad:                                                      # the field operations on zero operands, as an addition (fp.rs:L238-L248)
                field = "Bn254" if chip_name.startswith("Bn254") else "Bls12381"
                pad_code = [M.FP_SYSCALLS[field][{"fp": 0, "fp2_addsub": 3, "fp2_mul": 5}[shape]]]
                _fp_field_ops(rows[n:n + 1], L, chip_name, shape, pad_code, [[0]] * comps, [[0]] * comps)
        if n:
            wr = len(got)
            assert all(got[c] == ints(out, c * nw // wr, (c + 1) * nw // wr) for c in range(wr)), "the executor's result is not what the operands give"
        if pad:
            rows[n + 1:] = rows[n]
        tb.main[:] = torch.as_tensor(rows, device=dev)
        tb.set("clk_high", clk >> 24); tb.set("clk_low", clk & 0xFFFFFF); tb.set("is_real", 1)
        xa = names[0]
        xl = _syscall_addr_t(tb, xa + "_ptr", a1)
        xst, yst = t[:, 4:4 + 2 * nw].reshape(n, nw, 2), (t[:, 4 + 2 * nw:4 + 4 * nw].reshape(n, nw, 2) if two else None)
        x_ts = clk + 1 if two else clk
        for i in range(nw):
            tb.set("%s_addrs.%d.value" % (xa, i), _limbs_t(a1 + 8 * i)[:, :3])
            _mem_access_t(tb, "%s_access.%d.memory_access" % (xa, i), xst[:, i, 1], xst[:, i, 0], x_ts)
            tb.set("%s_access.%d.prev_value_u8.low_bytes" % (xa, i), _low_bytes(xst[:, i, 1]))
        if two:
            yl = _syscall_addr_t(tb, names[1] + "_ptr", a2)
            for i in range(nw):
                tb.set("%s_addrs.%d.value" % (names[1], i), _limbs_t(a2 + 8 * i)[:, :3])
                _mem_access_t(tb, "%s_access.%d.memory_access" % (names[1], i), yst[:, i, 1], yst[:, i, 0], clk)
                tb.set("%s_access.%d.prev_value_u8.low_bytes" % (names[1], i), _low_bytes(yst[:, i, 1]))
        tr.tables[chip_name] = tb
        outs = t[:, 4 + (4 if two else 2) * nw:]
        wa = (a1[:, None] + 8 * arange(nw)).reshape(-1)
        t_i, v_i, t_f, v_f = xst[:, :, 0].reshape(-1), xst[:, :, 1].reshape(-1), x_ts[:, None].expand(-1, nw).reshape(-1), outs.reshape(-1)
        if two:
            wa = torch.cat([wa, (a2[:, None] + 8 * arange(nw)).reshape(-1)])
            t_i, v_i = torch.cat([t_i, yst[:, :, 0].reshape(-1)]), torch.cat([v_i, yst[:, :, 1].reshape(-1)])
            t_f, v_f = torch.cat([t_f, clk[:, None].expand(-1, nw).reshape(-1)]), torch.cat([v_f, yst[:, :, 1].reshape(-1)])
        return _close_precompile_shard(tr, code, clk, xl, wa, t_i, t_f, v_i, v_f, arg2_limbs=yl if two else None, ctx=ctx)
    if shape == "ed_decompress":
        xs, ys, out = ev[:, 4:12].reshape(n, 4, 2), ev[:, 12:20].reshape(n, 4, 2), ev[:, 20:24]
        if n:
            roots, negs = _ed_decompress_field_ops(rows, L, ints(ys[:, :, 1], 0, 4))
            want = [ng if int(sg) else rt for rt, ng, sg in zip(roots, negs, ev[:, 2])]
            assert want == ints(out, 0, 4), "the executor's x is not the root with the requested sign"
        if pad:
            _ed_decompress_field_ops(rows[n:n + 1], L, [0])
            rows[n + 1:] = rows[n]
        tb.main[:] = torch.as_tensor(rows, device=dev)
        tb.set("clk_high", clk >> 24); tb.set("clk_low", clk & 0xFFFFFF); tb.set("is_real", 1); tb.set("sign", a2)
        pl = _syscall_addr_t(tb, "ptr", a1)
        xst, yst, outs = t[:, 4:12].reshape(n, 4, 2), t[:, 12:20].reshape(n, 4, 2), t[:, 20:24]
        for i in range(4):
            tb.set("addrs.%d.value" % i, _limbs_t(a1 + 8 * i)[:, :3])
            tb.set("read_ptrs.%d.value" % i, _limbs_t(a1 + 32 + 8 * i)[:, :3])
            _mem_access_t(tb, "x_access.%d" % i, xst[:, i, 1], xst[:, i, 0], clk + 1)
            tb.set("x_value.%d" % i, _limbs_t(outs[:, i]))
            _mem_access_t(tb, "y_access.%d.memory_access" % i, yst[:, i, 1], yst[:, i, 0], clk)
            tb.set("y_access.%d.prev_value_u8.low_bytes" % i, _low_bytes(yst[:, i, 1]))
        tr.tables[chip_name] = tb
        wa = torch.cat([(a1[:, None] + 8 * arange(4)).reshape(-1), (a1[:, None] + 32 + 8 * arange(4)).reshape(-1)])
        t_i, v_i = torch.cat([xst[:, :, 0].reshape(-1), yst[:, :, 0].reshape(-1)]), torch.cat([xst[:, :, 1].reshape(-1), yst[:, :, 1].reshape(-1)])
        t_f = torch.cat([(clk[:, None] + 1).expand(-1, 4).reshape(-1), clk[:, None].expand(-1, 4).reshape(-1)])
        v_f = torch.cat([outs.reshape(-1), yst[:, :, 1].reshape(-1)])
        return _close_precompile_shard(tr, code, clk, pl, wa, t_i, t_f, v_i, v_f, arg2_limbs=_limbs_t(a2), ctx=ctx)
    assert shape == "uint256_ops"
    blocks = ev[:, 10:50].reshape(n, 5, 4, 2)                             # a, b, c, d, e: (previous timestamp, word before)
    is_mul = (ev[:, 3] & np.uint64(0xFF)) == M.SYS_UINT256_MUL_CARRY
    if n:                                                                 # populate_conditional_op_and_carry (field_op.rs:L100-L200), modulus 2^256
        a, b, c = (ints(blocks[:, k, :, 1], 0, 4) for k in range(3))
        full = [(x * y if m else x + y) + z for x, y, z, m in zip(a, b, c, is_mul)]
        assert [v & ((1 << 256) - 1) for v in full] == ints(ev[:, 50:54], 0, 4) and [v >> 256 for v in full] == ints(ev[:, 54:58], 0, 4), "d, e are not a op b + c"
    def fill(rs, a, b, c, mul):
        k = len(a)
        pa, pb, pc = _le_bytes(a, 32), _le_bytes(b, 32), _le_bytes(c, 32)
        full = [(x * y if m else x + y) + z for x, y, z, m in zip(a, b, c, mul)]
        res, car = _le_bytes([v & ((1 << 256) - 1) for v in full], 32), _le_bytes([v >> 256 for v in full], 32)
        van = np.zeros((k, 64), dtype=np.int64)
        mk = np.asarray(mul, dtype=np.int64)[:, None]
        for i in range(32):
            van[:, i:i + 32] += pa[:, i:i + 1] * pb * mk
        van[:, :32] += (pa + pb) * (1 - mk) + pc - res
        van[:, 32:64] -= car
        wit = np.zeros((k, 63), dtype=np.int64)
        acc = van[:, 63].copy()
        for i in range(62, -1, -1):
            wit[:, i] = acc
            acc = van[:, i] + acc * 256
        assert not acc.any()
        wit += 1 << 14
        assert wit.min() >= 0 and wit.max() < (1 << 16)
        rs[:k, L["field_op.result"]:L["field_op.result"] + 32] = res
        rs[:k, L["field_op.carry"]:L["field_op.carry"] + 32] = car
        rs[:k, L["field_op.witness"]:L["field_op.witness"] + 63] = wit
    if n:
        fill(rows, a, b, c, list(is_mul))
    if pad:
        fill(rows[n:n + 1], [0], [0], [0], [False])
        rows[n + 1:] = rows[n]
    tb.main[:] = torch.as_tensor(rows, device=dev)
    tb.set("clk_high", clk >> 24); tb.set("clk_low", clk & 0xFFFFFF); tb.set("is_real", 1)
    tb.set("is_mul", torch.as_tensor(is_mul.astype(np.int64), device=dev)); tb.set("is_add", torch.as_tensor((~is_mul).astype(np.int64), device=dev))
    ptrs = {"a": a1, "b": a2, "c": t[:, 4], "d": t[:, 5], "e": t[:, 6]}
    limbs = {k: _syscall_addr_t(tb, k + "_ptr", v) for k, v in ptrs.items()}
    for j, k in enumerate("cde"):
        _mem_access_t(tb, k + "_ptr_memory", ptrs[k], t[:, 7 + j], clk)
    bt = t[:, 10:50].reshape(n, 5, 4, 2)
    outs = {"d": t[:, 50:54], "e": t[:, 54:58]}
    for j, k in enumerate("abcde"):
        for i in range(4):
            tb.set("%s_addrs.%d.value" % (k, i), _limbs_t(ptrs[k] + 8 * i)[:, :3])
            if j < 3:
                _mem_access_t(tb, "%s_memory.%d.memory_access" % (k, i), bt[:, j, i, 1], bt[:, j, i, 0], clk + j)
                tb.set("%s_memory.%d.prev_value_u8.low_bytes" % (k, i), _low_bytes(bt[:, j, i, 1]))
            else:
                _mem_access_t(tb, "%s_memory.%d" % (k, i), bt[:, j, i, 1], bt[:, j, i, 0], clk + j)
    tr.tables[chip_name] = tb
    regs = torch.tensor([12, 13, 14], dtype=I64, device=dev)[None, :].expand(n, -1)
    wa = torch.cat([regs.reshape(-1)] + [(ptrs[k][:, None] + 8 * arange(4)).reshape(-1) for k in "abcde"])
    t_i = torch.cat([t[:, 7:10].reshape(-1)] + [bt[:, j, :, 0].reshape(-1) for j in range(5)])
    v_i = torch.cat([t[:, 4:7].reshape(-1)] + [bt[:, j, :, 1].reshape(-1) for j in range(5)])
    t_f = torch.cat([clk[:, None].expand(-1, 3).reshape(-1)] + [(clk[:, None] + j).expand(-1, 4).reshape(-1) for j in range(5)])
    v_f = torch.cat([t[:, 4:7].reshape(-1)] + [bt[:, j, :, 1].reshape(-1) for j in range(3)] + [outs["d"].reshape(-1), outs["e"].reshape(-1)])
    return _close_precompile_shard(tr, code, clk, limbs["a"], wa, t_i, t_f, v_i, v_f, arg2_limbs=limbs["b"], ctx=ctx)
