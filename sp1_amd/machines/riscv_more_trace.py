"""Satisfying traces for the shards that are NOT made of CPU instructions (riscv_more.py's chips):

  * `precompile_shard` — a KECCAK_PERMUTE precompile shard as the reference builds it (crates/core/executor/src/record.rs splits
    precompile events into their own shards): KeccakPermuteControl (one row per syscall: syscall receive, 25 reads + 25 writes of
    the state words), KeccakPermute (24 rows per syscall: one Keccak-f round each), SyscallPrecompile (the Global receive of the
    syscall), MemoryLocal (one row per touched word), Global (every global interaction, septic-curve digest), Byte, Range — all
    REAL chips — plus the 2-row `GlobalAccBoundary` closing chip where the reference has `eval_public_values`.
    The permutation is COMPUTED here (keccak_f_rows: theta / rho / pi / chi / iota on 64-bit lanes, every intermediate the AIR
    names) and checked against hashlib's SHA3-256 in the tests.
  * `memory_shard` — global memory initialisation / finalisation (MemoryGlobalInit, MemoryGlobalFinalize, Global, Byte, Range +
    closing chips for the two control chains and the accumulation chain).

As in riscv_trace.py nothing is fitted: tests/machine_check.py requires every constraint of every chip to vanish on every row and
every bus to balance. References per function.
"""
import numpy as np
import torch

from ..air import AirProgram, InteractionProgram, P, VCol
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


def keccak_permute_table(events, dev):
    """KeccakPermuteChip::generate_trace_into (keccak256/trace.rs:L63-L162): 24 rows per event, padding rows = the rows of a
    permutation of the zero state with is_real = 0. events: [(clk, state_addr, pre_state[25])]."""
    air, _ = R.chip("KeccakPermute")
    L = air.layout
    n = 24 * len(events)
    tb = RT.Table(air, n, dev)
    rows = np.zeros((tb.main.shape[0], air.main_width), dtype=np.int64)

    def fill(r0, kr, preimage, real, clk, addr):
        for rnd, kv in enumerate(kr):
            r = r0 + rnd
            if r >= rows.shape[0]:
                return
            rows[r, L["keccak.step_flags"] + rnd] = 1
            rows[r, L["keccak.export"]] = int(rnd == 23)
            for y in range(5):
                for x in range(5):
                    for nm, grid in (("preimage", preimage), ("a", kv["a"]), ("a_prime_prime", kv["a_prime_prime"])):
                        c0 = L["keccak.%s.%d.%d" % (nm, y, x)]
                        rows[r, c0:c0 + 4] = _limbs(grid[y][x])
                    c0 = L["keccak.a_prime.%d.%d" % (y, x)]
                    rows[r, c0:c0 + 64] = _bits(kv["a_prime"][y][x])
            for x in range(5):
                rows[r, L["keccak.c.%d" % x]:L["keccak.c.%d" % x] + 64] = _bits(kv["c"][x])
                rows[r, L["keccak.c_prime.%d" % x]:L["keccak.c_prime.%d" % x] + 64] = _bits(kv["c_prime"][x])
            c0 = L["keccak.a_prime_prime_0_0_bits"]
            rows[r, c0:c0 + 64] = _bits(kv["a_prime_prime"][0][0])
            c0 = L["keccak.a_prime_prime_prime_0_0_limbs"]
            rows[r, c0:c0 + 4] = _limbs(kv["appp00"])
            if real:
                rows[r, L["clk_high"]], rows[r, L["clk_low"]] = clk >> 24, clk & 0xFFFFFF
                rows[r, L["state_addr"]:L["state_addr"] + 3] = _limbs(addr)[:3]
                rows[r, L["index"]], rows[r, L["is_real"]] = rnd, 1
    posts = []
    for e, (clk, addr, pre) in enumerate(events):
        kr, post = keccak_f_rows(pre)
        fill(24 * e, kr, [[pre[x + 5 * y] for x in range(5)] for y in range(5)], True, clk, addr)
        posts.append(post)
    if rows.shape[0] > n:
        kr0, _ = keccak_f_rows([0] * 25)
        fill(n, kr0, [[0] * 5 for _ in range(5)], False, 0, 0)
    tb.main[:] = torch.as_tensor(rows, device=dev)
    return tb, posts


def _mem_access(rows, L, r, prefix, prev_val, t_prev, t_cur):
    """MemoryAccessCols::populate (memory/consistency/trace.rs:L36-L101)."""
    ph, pl, ch, cl = t_prev >> 24, t_prev & 0xFFFFFF, t_cur >> 24, t_cur & 0xFFFFFF
    same = int(ph == ch)
    d = (cl - pl if same else ch - ph) - 1
    assert d >= 0
    c0 = L[prefix + ".prev_value"]
    rows[r, c0:c0 + 4] = _limbs(prev_val)
    rows[r, L[prefix + ".prev_high"]], rows[r, L[prefix + ".prev_low"]] = ph, pl
    rows[r, L[prefix + ".compare_low"]] = same
    rows[r, L[prefix + ".diff_low_limb"]], rows[r, L[prefix + ".diff_high_limb"]] = d & MASK16, d >> 16


def precompile_shard(n_events, seed=0, device="cpu", clk0=(5 << 24) + 1001):
    """A KECCAK_PERMUTE precompile shard of `n_events` syscalls: (machine, tables, publics) like riscv_trace.generate."""
    dev = torch.device(device)
    rng = np.random.default_rng(seed)
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    # events: distinct, 8-aligned, non-overlapping state addresses >= 2^16; increasing clocks (≡ 1 mod 8 like every instruction)
    base = 0x20_0000
    slots = rng.permutation(4 * n_events + 4)[:n_events]
    events, prev_t = [], {}
    for e in range(n_events):
        addr = base + 256 * int(slots[e])                                      # 200-byte states in 256-byte slots
        clk = clk0 + 8 * 40 * e
        pre = [int(v) for v in rng.integers(0, 1 << 63, 25, dtype=np.int64)]
        pre = [(v << 1 | int(rng.integers(2))) & U64 for v in pre]
        events.append((clk, addr, pre))
    # KeccakPermute
    kp, posts = keccak_permute_table(events, dev)
    tr.tables["KeccakPermute"] = kp
    # KeccakPermuteControl (controller.rs:L155-L237)
    air, _ = R.chip("KeccakPermuteControl")
    L = air.layout
    ct = RT.Table(air, n_events, dev)
    rows = np.zeros((ct.main.shape[0], air.main_width), dtype=np.int64)
    inv = lambda v: pow(v % P, P - 2, P) if v % P else 0
    words = []                                                                  # MemoryLocal rows: (addr, t_init, v_init, t_final, v_final)
    for r, ((clk, addr, pre), post) in enumerate(zip(events, posts)):
        rows[r, L["clk_high"]], rows[r, L["clk_low"]], rows[r, L["is_real"]] = clk >> 24, clk & 0xFFFFFF, 1
        al = _limbs(addr)
        rows[r, L["state_addr.addr"]:L["state_addr.addr"] + 3] = al[:3]            # SyscallAddrOperation::populate (syscall_addr.rs:L27-L46)
        top = al[1] + al[2]
        rows[r, L["state_addr.top_two_limb_min"]] = inv(top)
        dmax = top - 2 * MASK16
        rows[r, L["state_addr.top_two_limb_max.inverse"]], rows[r, L["state_addr.top_two_limb_max.result"]] = inv(dmax), int(dmax % P == 0)
        for i in range(25):
            wa = addr + 8 * i
            c0 = L["addrs.%d.value" % i]
            rows[r, c0:c0 + 3] = _limbs(wa)[:3]
            t_prev = int(rng.integers(1, clk0 - 8))                                 # the word's last access, in an earlier shard
            _mem_access(rows, L, r, "initial_memory_access.%d" % i, pre[i], t_prev, clk)
            _mem_access(rows, L, r, "final_memory_access.%d" % i, pre[i], clk, clk + 1)
            c0 = L["final_value.%d" % i]
            rows[r, c0:c0 + 4] = _limbs(post[i])
            words.append((wa, t_prev, RT._S64(pre[i]), clk + 1, RT._S64(post[i])))
    ct.main[:] = torch.as_tensor(rows, device=dev)
    tr.tables["KeccakPermuteControl"] = ct
    # SyscallPrecompile (syscall/chip.rs:L196-L254)
    air, _ = R.chip("SyscallPrecompile")
    st = RT.Table(air, n_events, dev)
    for r, (clk, addr, _) in enumerate(events):
        st.main[r] = torch.tensor([clk >> 24, clk & 0xFFFFFF, M.SYS_KECCAK_PERMUTE] + _limbs(addr)[:3] + [0, 0, 0, 1], device=dev)
    tr.tables["SyscallPrecompile"] = st
    # MemoryLocal (memory/local.rs:L98-L250): one row per touched word
    air, _ = R.chip("MemoryLocal")
    ml = RT.Table(air, len(words), dev)
    wt = torch.tensor(words, dtype=I64, device=dev)
    ml.set("addr", RT.limbs16(wt[:, 0])[:, :3])
    ml.set("initial_clk_high", wt[:, 1] >> 24); ml.set("initial_clk_low", wt[:, 1] & 0xFFFFFF)
    ml.set("final_clk_high", wt[:, 3] >> 24); ml.set("final_clk_low", wt[:, 3] & 0xFFFFFF)
    for tag, col in (("initial", 2), ("final", 4)):
        l = RT.limbs16(wt[:, col])
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
    tr.global_chip(machine, torch.cat(ev))
    tr.byte_range_tables(machine)
    names = sorted(machine)
    return [machine[n] for n in names], {n: (tr.tables[n].prep, tr.tables[n].main) for n in names}, torch.zeros(M.PV_NUM_ELTS, dtype=I64)


# ---------------------------------------------------------------------------------------------------------------------
def control_boundary_chip(name, kind):
    """Closing chip for a MemoryGlobal{Init, Finalize}Control chain, where the reference has eval_public_values
    (riscv/mod.rs: the chain starts at (0, previous_addr, 1) and ends at (count, last_addr, 1)): 2 rows of
    [index, addr[3], flag, is_send, is_recv]."""
    air = AirProgram(name, 7, 0)
    it = InteractionProgram(name, 7, 0)
    col = lambda i: VCol([("main", i, 1)], 0)
    it.send(kind, [col(i) for i in range(5)], col(5))
    it.receive(kind, [col(i) for i in range(5)], col(6))
    return air, it


def memory_shard(n_words, seed=0, device="cpu", with_zero=True):
    """Global memory initialisation and finalisation of `n_words` addresses (memory/global.rs generate_trace_into: events sorted
    by address, the chain of (index, prev_addr, validity) control messages, one Global event per word)."""
    dev = torch.device(device)
    rng = np.random.default_rng(seed)
    tr = RT.Tracer.__new__(RT.Tracer)
    tr.dev, tr.tables = dev, {}
    addrs = np.sort(rng.choice(np.arange(1, 1 << 20), size=n_words - int(with_zero), replace=False)) * 8 + (1 << 16)       # all > 2^16
    if with_zero:
        addrs = np.concatenate([[0], addrs])                # register x0: address 0 with value 0 (the `is_comp = 0` row)
    inv = lambda v: pow(int(v) % P, P - 2, P) if int(v) % P else 0
    machine, ev = {}, []
    for name, kind in (("MemoryGlobalInit", M.MEMORY_GLOBAL_INIT_CONTROL), ("MemoryGlobalFinalize", M.MEMORY_GLOBAL_FINALIZE_CONTROL)):
        air, it = R.chip(name)
        L = air.layout
        tb = RT.Table(air, n_words, dev)
        rows = np.zeros((tb.main.shape[0], air.main_width), dtype=np.int64)
        # the first memory shard starts at previous_addr = 0 and must initialise address 0 (register x0) with 0; a later one
        # continues from the previous shard's last address (public values previous_init_addr / previous_finalize_addr)
        prev = previous_addr = 0 if with_zero else 1 << 16
        for i, a in enumerate(int(x) for x in addrs):
            v = 0 if a == 0 else int(rng.integers(0, 1 << 63, dtype=np.int64)) * 2 + int(rng.integers(2))
            t = int(rng.integers(1, 1 << 40))
            r = i
            rows[r, L["clk_high"]], rows[r, L["clk_low"]] = t >> 24, t & 0xFFFFFF
            rows[r, L["index"]] = i
            rows[r, L["prev_addr"]:L["prev_addr"] + 3] = _limbs(prev)[:3]
            rows[r, L["addr"]:L["addr"] + 3] = _limbs(a)[:3]
            vl = _limbs(v)
            rows[r, L["value"]:L["value"] + 4] = vl
            rows[r, L["value_lower"]], rows[r, L["value_upper"]] = vl[2] & 0xFF, vl[2] >> 8
            rows[r, L["is_real"]] = 1
            rows[r, L["prev_valid"]] = 0 if (prev == 0 and i != 0) else 1
            s = sum(_limbs(prev)[:3])
            rows[r, L["is_prev_addr_zero.inverse"]], rows[r, L["is_prev_addr_zero.result"]] = inv(s), int(s == 0)
            rows[r, L["is_index_zero.inverse"]], rows[r, L["is_index_zero.result"]] = inv(i), int(i == 0)
            if prev != 0 or i != 0:
                rows[r, L["is_comp"]] = 1
                xl, yl = _limbs(prev), _limbs(a)
                j = [q for q in (3, 2, 1, 0) if xl[q] != yl[q]][0]
                rows[r, L["lt_cols.u16_flags"] + j] = 1
                rows[r, L["lt_cols.comparison_limbs"]], rows[r, L["lt_cols.comparison_limbs"] + 1] = xl[j], yl[j]
                rows[r, L["lt_cols.not_eq_inv"]] = inv(xl[j] - yl[j])
                rows[r, L["lt_cols.bit"]] = int(xl[j] < yl[j])
            prev = a
        tb.main[:] = torch.as_tensor(rows % P, device=dev)
        tr.tables[name], machine[name] = tb, (air, it)
        ev += [v_ for _, v_, _ in RT.eval_interactions(it, tb.main[:tb.n], None, kinds=(R.GLOBAL,))]
        # the two ends of the control chain: (0, previous_addr = 0, prev_valid = 1) is sent, (n, last_addr, is_comp of the last row) received
        bair, bit = control_boundary_chip(name + "Boundary", kind)
        bt = RT.Table(bair, 2, dev)
        bt.main[0] = torch.tensor([0] + _limbs(previous_addr)[:3] + [1, 1, 0], device=dev)
        bt.main[1] = torch.tensor([n_words] + _limbs(int(addrs[-1]))[:3] + [int(rows[n_words - 1, L["is_comp"]]), 0, 1], device=dev)
        tr.tables[bair.name], machine[bair.name] = bt, (bair, bit)
    tr.global_chip(machine, torch.cat(ev))
    tr.byte_range_tables(machine)
    names = sorted(machine)
    return [machine[n] for n in names], {n: (tr.tables[n].prep, tr.tables[n].main) for n in names}, torch.zeros(M.PV_NUM_ELTS, dtype=I64)
