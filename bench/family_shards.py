"""One FULL shard of every field / curve precompile kind, proved on the GPU with production parameters and timed: the fifteen chip
families of /root/reference/crates/core/machine/src/syscall/precompiles/{weierstrass,fptower,edwards,uint256_ops} that the rsp
block does not call (it has secp256k1 add / double, Keccak and SHA-256 only), plus UINT256_MUL.

    python bench/family_shards.py [--kinds bn254_fp,ed_add] [--events 0] [--verify] [--out file]

Per kind: a hand-assembled rv64im loop makes as many calls of the system call as `SplitOpts` puts into one shard of it
(sp1_amd.machines.riscv_exec.split_thresholds: the trace area beside the fixed tables over the cost of one event; --events overrides),
the executor (C++) records the events, `program_shards` builds the shard's tables (the family chip, SyscallPrecompile, MemoryLocal,
Global, Program / Byte / Range and the zero-height chips of its cluster), `sp1hip_prove_shard` proves it twice — the first call
also plans the chips' constraint programs (a per-process cost, listed apart) — and the second proof is timed, with the library's
stage clocks. ONE JSON line: per kind rows, cells, first_proof_ms, prove_ms, stage_ms. `--verify`: the pinned verifier (oracle/, the
checker) on every timed proof."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bench"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

DATA = 0x78100000
M64 = (1 << 64) - 1
SECP256R1_G = (0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296, 0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)
BN254_G = (1, 2)
BLS12381_G = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
              0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
ED_B = (15112221349535400772501151409588531511454012693041857206046113283949847762202, 46316835694926478169428394003475163141307993866256225615783033603165251855960)


def words(v, n):
    import struct
    return b"".join(struct.pack("<Q", (v >> (64 * i)) & M64) for i in range(n))


def looped(A, body, n):
    """x29 counts n passes of `body` (a list of instruction words that leaves x28, x29 alone)."""
    return A.li(28, DATA) + A.li(29, n) + body + [A.enc("addi", 29, 29, -1), A.enc("bne", 29, 0, -4 * (len(body) + 1))]


def programs(A, M):
    """kind -> (loop body, data): every pass is one valid call whatever came before it."""
    call = lambda code, a0, a1: [A.enc("addi", 10, 28, a0), A.enc("addi", 11, 28, a1) if a1 is not None else A.enc("addi", 11, 0, 0)] + A.li(5, code) + [A.enc("ecall")]
    out = {}
    for curve, G, add_code, double_code in (("Secp256r1", SECP256R1_G, 0x0001012C, 0x0000012D), ("Bn254", BN254_G, 0x0001010E, 0x0000010F),
                                            ("Bls12381", BLS12381_G, 0x0001011E, 0x0000011F)):
        Pm, a, nl = M.CURVES[curve][:3]
        n = nl // 8
        lam = (3 * G[0] * G[0] + a) * pow(2 * G[1], Pm - 2, Pm) % Pm
        x2 = (lam * lam - 2 * G[0]) % Pm
        g2 = (x2, (lam * (G[0] - x2) - G[1]) % Pm)
        size = 2 * 8 * n
        # q <- q + p with p = 2 G fixed and q = G at first: G, 3 G, 5 G, ... never meets p or -p
        out[curve.lower() + "_add"] = (call(add_code, size, 0), words(g2[0], n) + words(g2[1], n) + words(G[0], n) + words(G[1], n))
        out[curve.lower() + "_double"] = (call(double_code, 0, None), words(G[0], n) + words(G[1], n))
    for field in ("Bn254", "Bls12381"):
        Pm, nl = M.FP_FIELDS[field][:2]
        n = nl // 8
        codes = [0x00010100 | c for c in M.FP_SYSCALLS[field]]
        slot = 8 * n
        a, b = (3 ** 200 + 12345) % Pm, Pm - 7
        c0, c1, d0, d1 = 5 ** 150 % Pm, Pm - 1, 7 ** 130 % Pm, 11 ** 100 % Pm
        f = field.lower()
        out[f + "_fp"] = (call(codes[2], 0, slot), words(a, n) + words(b, n))                                  # x <- x * y
        out[f + "_fp2_addsub"] = (call(codes[3], 0, 2 * slot), words(c0, n) + words(c1, n) + words(d0, n) + words(d1, n))
        out[f + "_fp2_mul"] = (call(codes[5], 0, 2 * slot), words(c0, n) + words(c1, n) + words(d0, n) + words(d1, n))
    Pm, D = M.ED25519_P, M.ED25519_D
    f = D * ED_B[0] * ED_B[0] * ED_B[1] * ED_B[1] % Pm
    b2 = ((2 * ED_B[0] * ED_B[1]) * pow(1 + f, Pm - 2, Pm) % Pm, (ED_B[1] * ED_B[1] + ED_B[0] * ED_B[0]) * pow(1 - f, Pm - 2, Pm) % Pm)
    pt = words(ED_B[0], 4) + words(ED_B[1], 4)
    out["ed_add"] = (call(0x00010107, 0, 64), pt + pt)                                                        # p <- p + B (complete law)
    out["ed_decompress"] = ([A.enc("addi", 10, 28, 0), A.enc("addi", 11, 0, b2[0] & 1)] + A.li(5, 0x00000108) + [A.enc("ecall")], bytes(32) + words(b2[1], 4))
    top = (1 << 256) - 1
    out["uint256_ops"] = ([A.enc("addi", 12, 28, 64), A.enc("addi", 13, 28, 96), A.enc("addi", 14, 28, 128)] + call(0x00010131, 0, 32),
                          words(top, 4) + words(top - 5, 4) + words(top, 4) + bytes(64))                      # d, e <- a * b + c
    out["uint256"] = (call(0x0001011D, 0, 32), words(3 ** 150, 4) + words(5 ** 100, 4) + words(top - 188, 4))   # x <- x * y mod m (UINT256_MUL: y, then m)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kinds", default="")
    ap.add_argument("--events", type=int, default=0, help="calls per kind (0 = one full shard: the SplitOpts threshold of the kind)")
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="build every program and its shards on the CPU (a few calls each), prove nothing")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import numpy as np
    import torch
    import rv_asm as A
    from sp1_amd.machines import public_values as PVM, riscv_exec as X, riscv_more as M, riscv_trace as RT
    device = "cpu" if args.dry_run else "cuda"
    if not args.dry_run:
        from core_real import to_col_major
        from sp1_amd import api
        torch.cuda.set_device(0)
        lib = api._L()
        api.check(lib.sp1hip_timers_enable(1))
    L, lsh = 22, 21
    progs = programs(A, M)
    kinds = [k for k in args.kinds.split(",") if k] or list(progs)
    rows = []
    for kind in kinds:
        body, data = progs[kind]
        limit = X.split_thresholds(len(body) + 16)[kind]
        n = args.events or (4 if args.dry_run else limit)
        t0 = time.perf_counter()
        ex = X.Executor(A.elf(looped(A, body, n) + A.halt(0), data=data + bytes(32)), stdin=[])
        ex.cut_by_area()
        pk = None
        for k, machine, tabs, publics, gev, sh in X.program_shards(ex, 1 << 40, device=device):
            if args.dry_run:
                if k == kind:
                    rows.append({"kind": kind, "events": n, "full_shard_events": limit, "chips": len(machine)})
                continue
            if pk is None:                                       # the program's proving key from the first shard's preprocessed tables
                pk_prep = {a.name: to_col_major(tabs[a.name][0]) for a, _ in machine if tabs[a.name][0] is not None}
                vk_words = RT.to_monty_np(torch.tensor(X.verifying_key_words(ex, sh.pc_start, "cuda")))
                pk = api.ProvingKey([pk_prep[nm] for nm in sorted(pk_prep)], L, lsh, 32, pc_start=vk_words[:3], initial_global_cumulative_sum=vk_words[3:])
            if k != kind:
                continue
            build_s = time.perf_counter() - t0
            area = sum(int(tabs[a.name][1].shape[0]) * (a.main_width + a.prep_width) for a, _ in machine)
            heights = {a.name: int(tabs[a.name][1].shape[0]) for a, _ in machine if a.name not in ("Byte", "Range", "Program") and tabs[a.name][1].shape[0]}
            chips = [(a, i, to_col_major(tabs[a.name][1]), pk_prep.get(a.name)) for a, i in machine]
            tabs.clear()
            pv = RT.to_monty_np(publics)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            first = pk.prove_shard(chips, pv)
            torch.cuda.synchronize()
            first_ms = 1e3 * (time.perf_counter() - t1)
            api.check(lib.sp1hip_timers_reset())
            t1 = time.perf_counter()
            proof = pk.prove_shard(chips, pv)
            torch.cuda.synchronize()
            prove_ms = 1e3 * (time.perf_counter() - t1)
            assert proof == first
            row = {"kind": kind, "events": n, "full_shard_events": limit, "rows": heights, "cells": area, "build_s": round(build_s, 2),
                   "first_proof_ms": round(first_ms, 2), "prove_ms": round(prove_ms, 2), "stage_ms": {}, "proof_bytes": len(proof)}
            for name in ("stage_commit", "stage_logup_gkr", "stage_zerocheck", "stage_evaluation_proof"):
                n_, ms_ = C.c_uint64(), C.c_double()
                api.check(lib.sp1hip_timers_read(name.encode(), C.byref(n_), C.byref(ms_)))
                row["stage_ms"][name[6:]] = round(ms_.value, 2)
            if args.verify:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import pyoracle as orc                           # the checker, after everything timed for this kind
                shapes = [(a, i, np.zeros((0, a.main_width), np.uint32), np.zeros((0, a.prep_width), np.uint32) if a.prep_width else None) for a, i in machine]
                commit = np.asarray(pk.preprocessed_commit).copy()
                v_ch = orc.Challenger()
                v_ch.observe(np.concatenate([commit, vk_words, np.zeros(7, np.uint32)]))
                row["verified"] = int(orc.shard_verify(shapes, commit, proof, L, lsh, v_ch, 2, 124, 16, pv_program=PVM.verifier_program())) == 0
            rows.append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
            del chips
            torch.cuda.empty_cache()
        del ex
    line = json.dumps({"parameters": "max_log_row_count 22, stack 2^21, blowup 4, 124 queries, 16-bit PoW", "kinds": rows})
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
