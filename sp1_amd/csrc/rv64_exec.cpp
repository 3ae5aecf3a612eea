// rv64im executor for SP1 guest programs: runs an ELF the way the reference's tracing VM does and records, per shard, the
// events its chips' trace generation starts from. Host code only (no device pass); the tables are filled from these events by
// sp1_amd/machines/riscv_exec.py and proved by sp1hip_prove_shard.
//
// What it follows (semantics, not code):
//   ELF image / instruction list     /root/reference/crates/core/executor/src/disassembler/elf.rs:L63-L318
//   instruction transpilation       /root/reference/crates/core/executor/src/disassembler/rrs.rs:L9-L437
//   one cycle                        /root/reference/crates/core/executor/src/vm.rs:L139-L431 (ALU / jump / branch / U-type / ECALL),
//                                    L433-L518 (load / store values), L586-L627 (supervisor load / store), L782-L857 (rr / rw)
//   timestamps                       clk + MemoryAccessPosition {Memory 1, C 2, B 3, A 4} (events/memory.rs:L63-L74); CLK_INC 8,
//                                    an ECALL adds 256 (vm.rs:L428)
//   system calls                     minimal/ecall.rs:L75-L189, minimal/write.rs:L86-L149, minimal/hint.rs:L5-L67 (hinted words are
//                                    initial memory: timestamp 0), vm/syscall/{halt,commit,deferred}.rs; unconstrained blocks
//                                    run without a trace and are rolled back (minimal/arch/portable/mod.rs:L258-L280)
//   Keccak precompile                vm/syscall/precompiles/keccak256/permute.rs (reads at clk, writes at clk + 1)
//   SHA-256 precompiles              vm/syscall/precompiles/sha256/{extend,compress}.rs (extend: step i at clk + 1 + (i - 16); compress:
//                                    h read at clk, w at clk + 1, h written at clk + 2)
//   Poseidon2 precompile             vm/syscall/poseidon2.rs, minimal/precompiles/poseidon2.rs (eight words rewritten at clk)
//   per-shard local memory events    tracing.rs:L548-L577, L1490-L1515 (first / last access of every address touched in the shard)
// User mode (page protection, untrusted programs), the trap context and the other precompiles are not implemented: an ELF that
// needs them stops with an error naming the system call.
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/sp1hip.h"
#include "common.hpp"
#include "kb31.hpp"
#include "rv64_bigmod.hpp"

namespace {

enum Op : uint32_t {
    ADD, ADDI, SUB, XOR, OR, AND, SLL, SRL, SRA, SLT, SLTU, MUL, MULH, MULHU, MULHSU, DIV, DIVU, REM, REMU, ADDW, SUBW, SLLW, SRLW,
    SRAW, MULW, DIVW, DIVUW, REMW, REMUW, LB, LH, LW, LBU, LHU, LWU, LD, SB, SH, SW, SD, BEQ, BNE, BLT, BGE, BLTU, BGEU, JAL, JALR,
    AUIPC, LUI, ECALL, EBREAK, UNIMP
};

struct Instr { uint32_t op; uint32_t a; uint64_t b, c; bool imm_b, imm_c; };

constexpr uint64_t HALT_PC = 1, CLK_INC = 8, ECALL_EXTRA = 256;
constexpr uint64_t SYS_HALT = 0x00, SYS_WRITE = 0x02, SYS_ENTER_UNC = 0x03, SYS_EXIT_UNC = 0x04, SYS_KECCAK = 0x00010109,
                   SYS_POSEIDON2 = 0x00000133, SYS_UINT256_MUL = 0x0001011D, SYS_SECP256K1_ADD = 0x0001010A, SYS_SECP256K1_DOUBLE = 0x0000010B, SYS_SHA_EXTEND = 0x00300105, SYS_SHA_COMPRESS = 0x00010106, SYS_COMMIT = 0x10, SYS_COMMIT_DEFERRED = 0x1A, SYS_VERIFY_PROOF = 0x1B, SYS_HINT_LEN = 0xF0, SYS_HINT_READ = 0xF1;
// The field / curve precompiles behind sp1hip_rv64_precompile_events (syscall_code.rs:L65-L168): family = the chip that proves them
constexpr uint64_t SYS_ED_ADD = 0x00010107, SYS_ED_DECOMPRESS = 0x00000108, SYS_BN254_ADD = 0x0001010E, SYS_BN254_DOUBLE = 0x0000010F,
                   SYS_BLS12381_ADD = 0x0001011E, SYS_BLS12381_DOUBLE = 0x0000011F, SYS_BLS12381_FP_ADD = 0x00010120, SYS_BLS12381_FP2_MUL = 0x00010125,
                   SYS_BN254_FP_ADD = 0x00010126, SYS_BN254_FP2_MUL = 0x0001012B, SYS_SECP256R1_ADD = 0x0001012C, SYS_SECP256R1_DOUBLE = 0x0000012D,
                   SYS_UINT256_ADD_CARRY = 0x00010130, SYS_UINT256_MUL_CARRY = 0x00010131;
enum Family { F_SECP256R1_ADD, F_SECP256R1_DOUBLE, F_BN254_ADD, F_BN254_DOUBLE, F_BLS12381_ADD, F_BLS12381_DOUBLE, F_BN254_FP, F_BLS12381_FP,
              F_BN254_FP2_ADDSUB, F_BLS12381_FP2_ADDSUB, F_BN254_FP2_MUL, F_BLS12381_FP2_MUL, F_ED_ADD, F_ED_DECOMPRESS, F_UINT256_OPS, N_FAMILIES };
static_assert(N_FAMILIES == SP1HIP_RV64_FAMILIES, "include/sp1hip.h lists the families");
// u64 words per event: [clk, arg1, arg2, syscall code] + two-operand calls n x (ts, x word), n x (ts, y word), n x words written;
// doublings n x (ts, word), n written; ED_DECOMPRESS 4 x (ts, x word before), 4 x (ts, y word), 4 x words written; UINT256 add /
// mul with carry: c, d, e pointers, the timestamps of x12 / x13 / x14 before, 5 x 4 x (ts, word before) for a, b, c, d, e, 4 d and 4 e words written
constexpr uint32_t FAMILY_WORDS[N_FAMILIES] = {44, 28, 44, 28, 64, 40, 24, 34, 44, 64, 44, 64, 44, 24, 58};
constexpr uint64_t FD_PUBLIC_VALUES = 13, FD_HINT = 14, FD_FP_SQRT = 20, FD_FP_INV = 21;

Instr decode(uint32_t w) {
    auto R = [&](uint32_t op) { return Instr{op, (w >> 7) & 31, (w >> 15) & 31, (w >> 20) & 31, false, false}; };
    auto I = [&](uint32_t op) { return Instr{op, (w >> 7) & 31, (w >> 15) & 31, (uint64_t)(int64_t)((int32_t)w >> 20), false, true}; };
    auto SH6 = [&](uint32_t op) { return Instr{op, (w >> 7) & 31, (w >> 15) & 31, (w >> 20) & 63, false, true}; };
    auto SH5 = [&](uint32_t op) { return Instr{op, (w >> 7) & 31, (w >> 15) & 31, (w >> 20) & 31, false, true}; };
    const Instr unimp{UNIMP, 0, 0, 0, true, true};
    const uint32_t f3 = (w >> 12) & 7, f7 = w >> 25;
    switch (w & 0x7F) {
    case 0x33:
        if (f7 == 0x00) { const uint32_t t[8] = {ADD, SLL, SLT, SLTU, XOR, SRL, OR, AND}; return R(t[f3]); }
        if (f7 == 0x20) { if (f3 == 0) return R(SUB); if (f3 == 5) return R(SRA); return unimp; }
        if (f7 == 0x01) { const uint32_t t[8] = {MUL, MULH, MULHSU, MULHU, DIV, DIVU, REM, REMU}; return R(t[f3]); }
        return unimp;
    case 0x3B:
        if (f7 == 0x00) { if (f3 == 0) return R(ADDW); if (f3 == 1) return R(SLLW); if (f3 == 5) return R(SRLW); return unimp; }
        if (f7 == 0x20) { if (f3 == 0) return R(SUBW); if (f3 == 5) return R(SRAW); return unimp; }
        if (f7 == 0x01) { const uint32_t t[8] = {MULW, UNIMP, UNIMP, UNIMP, DIVW, DIVUW, REMW, REMUW}; return t[f3] == UNIMP ? unimp : R(t[f3]); }
        return unimp;
    case 0x13:
        switch (f3) {
        case 0: return I(ADDI); case 2: return I(SLT); case 3: return I(SLTU); case 4: return I(XOR); case 6: return I(OR); case 7: return I(AND);
        case 1: return (w >> 26) == 0 ? SH6(SLL) : unimp;
        default: return (w >> 26) == 0 ? SH6(SRL) : (w >> 26) == 0x10 ? SH6(SRA) : unimp;
        }
    case 0x1B:
        if (f3 == 0) return I(ADDW);
        if (f3 == 1) return f7 == 0 ? SH5(SLLW) : unimp;
        if (f3 == 5) return f7 == 0 ? SH5(SRLW) : f7 == 0x20 ? SH5(SRAW) : unimp;
        return unimp;
    case 0x03: { const uint32_t t[8] = {LB, LH, LW, LD, LBU, LHU, LWU, UNIMP}; return t[f3] == UNIMP ? unimp : I(t[f3]); }
    case 0x23: {
        if (f3 > 3) return unimp;
        const uint32_t t[4] = {SB, SH, SW, SD};
        const int64_t imm = (int64_t)((int32_t)(w & 0xFE000000) >> 20) | ((w >> 7) & 31);
        return Instr{t[f3], (w >> 20) & 31, (w >> 15) & 31, (uint64_t)imm, false, true};            // op_a = rs2, op_b = rs1
    }
    case 0x63: {
        const uint32_t t[8] = {BEQ, BNE, UNIMP, UNIMP, BLT, BGE, BLTU, BGEU};
        if (t[f3] == UNIMP) return unimp;
        const int64_t imm = (int64_t)((int32_t)(w & 0x80000000) >> 19) | ((w & 0x80) << 4) | ((w >> 20) & 0x7E0) | ((w >> 7) & 0x1E);
        return Instr{t[f3], (w >> 15) & 31, (w >> 20) & 31, (uint64_t)imm, false, true};            // op_a = rs1, op_b = rs2
    }
    case 0x6F: {
        const int64_t imm = (int64_t)((int32_t)(w & 0x80000000) >> 11) | (w & 0xFF000) | ((w >> 9) & 0x800) | ((w >> 20) & 0x7FE);
        return Instr{JAL, (w >> 7) & 31, (uint64_t)imm, 0, true, true};
    }
    case 0x67: return f3 == 0 ? I(JALR) : unimp;
    case 0x37: { const uint64_t imm = (uint64_t)(int64_t)(int32_t)(w & 0xFFFFF000); return Instr{LUI, (w >> 7) & 31, imm, imm, true, true}; }
    case 0x17: { const uint64_t imm = (uint64_t)(int64_t)(int32_t)(w & 0xFFFFF000); return Instr{AUIPC, (w >> 7) & 31, imm, imm, true, true}; }
    case 0x73:
        if (w == 0x00000073) return Instr{ECALL, 5, 10, 11, false, false};
        if (w == 0x00100073) return Instr{EBREAK, 0, 0, 0, false, false};
        return unimp;
    default: return unimp;
    }
}

struct Cell { uint64_t val = 0, ts = 0; uint32_t shard = 0, open = 0; bool ever = false; };   // open: this shard's local entry of the address
constexpr int PAGE_WORDS = 512;
struct Page { Cell w[PAGE_WORDS]; };

constexpr int EV = SP1HIP_RV64_EVENT_WORDS;
enum { E_PC, E_CLK, E_OP, E_OPA, E_OPB, E_OPC, E_FLAGS, E_A, E_B, E_C, E_A_PREV, E_A_PTS, E_B_PTS, E_C_PTS, E_MADDR, E_M_PTS, E_M_PREV,
       E_M_NEW, E_NEXT_PC, E_SPARE };

const uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull, 0x0000000080000001ull,
    0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull,
    0x000000000000800aull, 0x800000008000000aull, 0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

// (x * y) mod m on 256-bit little-endian limbs; m = 0 means 2^256 (vm/syscall/uint256.rs, minimal/precompiles/uint256.rs)
void uint256_mulmod(const uint64_t x[4], const uint64_t y[4], const uint64_t m[4], uint64_t out[4]) {
    uint64_t prod[8] = {0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 carry = 0;
        for (int j = 0; j < 4; ++j) {
            const unsigned __int128 t = (unsigned __int128)x[i] * y[j] + prod[i + j] + carry;
            prod[i + j] = (uint64_t)t; carry = t >> 64;
        }
        prod[i + 4] = (uint64_t)carry;
    }
    if (!(m[0] | m[1] | m[2] | m[3])) { memcpy(out, prod, 32); return; }
    uint64_t rem[5] = {0};                                             // one bit wider than m while shifting
    for (int bit = 511; bit >= 0; --bit) {
        for (int k = 4; k > 0; --k) rem[k] = (rem[k] << 1) | (rem[k - 1] >> 63);
        rem[0] = (rem[0] << 1) | ((prod[bit >> 6] >> (bit & 63)) & 1);
        bool ge = rem[4] != 0;
        if (!ge) { ge = true; for (int k = 3; k >= 0; --k) if (rem[k] != m[k]) { ge = rem[k] > m[k]; break; } }
        if (ge) {
            unsigned __int128 borrow = 0;
            for (int k = 0; k < 4; ++k) { const unsigned __int128 t = (unsigned __int128)rem[k] - m[k] - borrow; rem[k] = (uint64_t)t; borrow = (t >> 64) & 1; }
            rem[4] -= (uint64_t)borrow;
        }
    }
    memcpy(out, rem, 32);
}

// secp256k1's base field on 256-bit little-endian limbs: p = 2^256 - 2^32 - 977 (curves/src/weierstrass/secp256k1.rs)
struct U256 { uint64_t w[4]; };
const U256 SECP_P = {{0xFFFFFFFEFFFFFC2Full, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull}};
bool u256_ge(const U256& a, const U256& b) { for (int k = 3; k >= 0; --k) if (a.w[k] != b.w[k]) return a.w[k] > b.w[k]; return true; }
U256 fp_mul(const U256& a, const U256& b) {                           // 2^256 = 2^32 + 977 (mod p): fold the high half twice
    uint64_t prod[8] = {0};
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 carry = 0;
        for (int j = 0; j < 4; ++j) { const unsigned __int128 t = (unsigned __int128)a.w[i] * b.w[j] + prod[i + j] + carry; prod[i + j] = (uint64_t)t; carry = t >> 64; }
        prod[i + 4] = (uint64_t)carry;
    }
    const uint64_t C = 0x1000003D1ull;
    uint64_t lo[5];
    unsigned __int128 c = 0;
    for (int k = 0; k < 4; ++k) { c += (unsigned __int128)prod[4 + k] * C + prod[k]; lo[k] = (uint64_t)c; c >>= 64; }
    lo[4] = (uint64_t)c;                                                // < 2^34
    c = (unsigned __int128)lo[4] * C;
    U256 r;
    for (int k = 0; k < 4; ++k) { c += lo[k]; r.w[k] = (uint64_t)c; c >>= 64; }
    if (c) { c = C; for (int k = 0; k < 4; ++k) { c += r.w[k]; r.w[k] = (uint64_t)c; c >>= 64; } }   // one more 2^256: cannot carry again
    if (u256_ge(r, SECP_P)) { unsigned __int128 borrow = 0; for (int k = 0; k < 4; ++k) { const unsigned __int128 t = (unsigned __int128)r.w[k] - SECP_P.w[k] - borrow; r.w[k] = (uint64_t)t; borrow = (t >> 64) & 1; } }
    return r;
}
U256 fp_sub(const U256& a, const U256& b) {                            // a, b < p
    U256 r; unsigned __int128 borrow = 0;
    for (int k = 0; k < 4; ++k) { const unsigned __int128 t = (unsigned __int128)a.w[k] - b.w[k] - borrow; r.w[k] = (uint64_t)t; borrow = (t >> 64) & 1; }
    if (borrow) { unsigned __int128 c = 0; for (int k = 0; k < 4; ++k) { c += (unsigned __int128)r.w[k] + SECP_P.w[k]; r.w[k] = (uint64_t)c; c >>= 64; } }
    return r;
}
U256 fp_add(const U256& a, const U256& b) {
    U256 r; unsigned __int128 c = 0;
    for (int k = 0; k < 4; ++k) { c += (unsigned __int128)a.w[k] + b.w[k]; r.w[k] = (uint64_t)c; c >>= 64; }
    if (c || u256_ge(r, SECP_P)) { unsigned __int128 borrow = 0; for (int k = 0; k < 4; ++k) { const unsigned __int128 t = (unsigned __int128)r.w[k] - SECP_P.w[k] - borrow; r.w[k] = (uint64_t)t; borrow = (t >> 64) & 1; } }
    return r;
}
U256 fp_inv(const U256& a) {                                           // a^(p - 2)
    U256 e = SECP_P; e.w[0] -= 2;
    U256 r = {{1, 0, 0, 0}}, base = a;
    for (int bit = 0; bit < 256; ++bit) { if ((e.w[bit >> 6] >> (bit & 63)) & 1) r = fp_mul(r, base); base = fp_mul(base, base); }
    return r;
}

void keccak_f(uint64_t s[25]) {                                        // FIPS 202, state s[x + 5 y]
    static const int rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    auto rl = [](uint64_t v, int r) { return r ? (v << r) | (v >> (64 - r)) : v; };
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) { const uint64_t d = c[(x + 4) % 5] ^ rl(c[(x + 1) % 5], 1); for (int y = 0; y < 5; ++y) s[x + 5 * y] ^= d; }
        for (int x = 0; x < 5; ++x) for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rl(s[x + 5 * y], rot[x + 5 * y]);
        for (int x = 0; x < 5; ++x) for (int y = 0; y < 5; ++y) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= KECCAK_RC[round];
    }
}

// The base fields of the curve / tower precompiles (curves/src/weierstrass/{secp256r1,bn254,bls12_381}.rs, edwards/ed25519.rs)
const uint64_t SECP256R1_P_W[4] = {0xFFFFFFFFFFFFFFFFull, 0x00000000FFFFFFFFull, 0x0000000000000000ull, 0xFFFFFFFF00000001ull};
const uint64_t BN254_P_W[4] = {0x3C208C16D87CFD47ull, 0x97816A916871CA8Dull, 0xB85045B68181585Dull, 0x30644E72E131A029ull};
const uint64_t BLS12381_P_W[6] = {0xB9FEFFFFFFFFAAABull, 0x1EABFFFEB153FFFFull, 0x6730D2A0F6B0F624ull, 0x64774B84F38512BFull, 0x4B1BA7B6434BACD7ull, 0x1A0111EA397FE69Aull};
const uint64_t ED25519_P_W[4] = {0xFFFFFFFFFFFFFFEDull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0x7FFFFFFFFFFFFFFFull};
const uint64_t ED25519_D_W[4] = {0x75EB4DCA135978A3ull, 0x00700A4D4141D8ABull, 0x8CC740797779E898ull, 0x52036CEE2B6FFE73ull};
const uint64_t ED25519_SQRT_M1_W[4] = {0xC4EE1B274A0EA0B0ull, 0x2F431806AD2FE478ull, 0x2B4D00993DFBD7A7ull, 0x2B8324804FC1DF0Bull};
const uint64_t ED25519_P_PLUS_3_OVER_8_W[4] = {0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0x0FFFFFFFFFFFFFFFull};
const bigmod::Field& field_secp256r1() { static const bigmod::Field f(SECP256R1_P_W, 4); return f; }
const bigmod::Field& field_bn254() { static const bigmod::Field f(BN254_P_W, 4); return f; }
const bigmod::Field& field_bls12381() { static const bigmod::Field f(BLS12381_P_W, 6); return f; }
const bigmod::Field& field_ed25519() { static const bigmod::Field f(ED25519_P_W, 4); return f; }

struct Vm {
    std::vector<Instr> program;
    std::vector<uint32_t> words;
    uint64_t pc_base = 0, pc_start = 0;
    std::unordered_map<uint64_t, Page*> pages;
    uint64_t last_page_key = ~0ull; Page* last_page = nullptr;
    Cell regs[32];
    uint64_t pc = 0, clk = 1, cycles = 0;
    uint32_t shard = 0;                                               // index of the shard being recorded (cells carry shard + 1)
    bool halted = false; uint32_t exit_code = 0;
    std::deque<std::vector<uint8_t>> input;
    std::vector<uint8_t> public_values, out;
    uint32_t committed_digest[8] = {0}, deferred_digest[8] = {0};
    bool commit_syscall = false, commit_deferred_syscall = false;
    // unconstrained block
    bool unc = false; uint64_t unc_regs[32], unc_pc = 0, unc_clk = 0; std::unordered_map<uint64_t, uint64_t> unc_mem;
    // the shard being recorded
    std::vector<uint64_t> events;                                      // [n][EV]
    std::vector<uint64_t> local;                                       // [m][5]: addr, initial ts, initial value, (final ts, final value)
    std::vector<uint8_t> local_closed;
    std::vector<uint64_t> sha_extend;                                  // [k][clk, w_ptr, 48 x (4 x (previous timestamp, word read), previous timestamp and value of w[i], w[i] written), 64 x (timestamp, value before; after)]
    std::vector<uint64_t> sha_compress;                                // [k][clk, w_ptr, h_ptr, 8 x (previous timestamp, h word), 64 x (previous timestamp, w word), 8 h words written]
    std::vector<uint64_t> secp_add, secp_double;                       // [k][clk, p_ptr, q_ptr, 8 x (ts, p word), 8 x (ts, q word), 8 p words written] / [k][clk, p_ptr, 8 x (ts, p word), 8 written]
    std::vector<uint64_t> family[N_FAMILIES];                          // [k][FAMILY_WORDS[f]]: the field / curve precompiles (layouts above)
    std::vector<uint64_t> uint256;                                     // [k][clk, x_ptr, y_ptr, 4 x (previous timestamp, x word), 8 x (previous timestamp, y / modulus word), 4 x words written]
    std::vector<uint64_t> poseidon2;                                   // POSEIDON2 events: [k][clk, pointer, 8 x (previous timestamp, word read), 8 words written]
    std::vector<uint64_t> precompile;                                  // Keccak events: [k][clk, pointer, 25 x (previous timestamp, word read), 25 words written]
    // the whole run
    std::vector<uint64_t> touched;                                     // [t][2]: addr, initial value (then final value, final ts)
    std::vector<uint64_t> global_out;
    std::vector<uint64_t> image;                                       // [m][2]: addr, value of every word of the ELF's segments, ascending
    std::string error;

    ~Vm() { for (auto& kv : pages) delete kv.second; }

    Cell& cell(uint64_t addr) {                                        // addr: 8-byte aligned
        const uint64_t key = addr >> 12;
        if (key != last_page_key) {
            auto it = pages.find(key);
            if (it == pages.end()) it = pages.emplace(key, new Page()).first;
            last_page_key = key; last_page = it->second;
        }
        return last_page->w[(addr >> 3) & (PAGE_WORDS - 1)];
    }
    void touch(Cell& c, uint64_t addr) {                               // first access of `addr` in this shard / in the run
        if (c.shard != shard + 1) {
            c.shard = shard + 1;
            c.open = (uint32_t)(local.size() / 5);
            local.insert(local.end(), {addr, c.ts, c.val, 0, 0});
            local_closed.push_back(0);
            if (addr >= 32) ++est_new_local;
            if (!c.ever) { c.ever = true; touched.insert(touched.end(), {addr, c.val}); }
        }
    }
    // a precompile's access: its shard carries its own local memory events (tracing.rs:L1619-L1630), so the CPU's entry of the
    // address, if one is open, ends here; the next CPU access opens another
    void touch_precompile(Cell& c, uint64_t addr) {
        if (c.shard == shard + 1) {
            local[5 * (size_t)c.open + 3] = c.ts; local[5 * (size_t)c.open + 4] = c.val;
            local_closed[c.open] = 1;
            c.shard = 0;
        }
        if (!c.ever) { c.ever = true; touched.insert(touched.end(), {addr, c.val}); }
    }
    // register accesses (addresses 0..31)
    uint64_t rr(uint32_t r, uint64_t pos, uint64_t& prev_ts) {
        Cell& c = regs[r];
        touch(c, r);
        prev_ts = c.ts; c.ts = clk + pos;
        return c.val;
    }
    void rw(uint32_t r, uint64_t v, uint64_t& prev_ts, uint64_t& prev_val) {
        Cell& c = regs[r];
        touch(c, r);
        prev_ts = c.ts; prev_val = c.val; c.ts = clk + 4; c.val = r ? v : 0;
    }
    uint64_t peek(uint64_t addr) {                                     // no trace (WRITE's buffer, unconstrained blocks)
        if (unc) { auto it = unc_mem.find(addr); if (it != unc_mem.end()) return it->second; }
        return cell(addr).val;
    }
    bool fail(const char* fmt, ...) {
        char buf[256]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        char where[96]; snprintf(where, sizeof where, " (pc 0x%llx, clk %llu)", (unsigned long long)pc, (unsigned long long)clk);
        error = std::string(buf) + where;
        return false;
    }

    static bool load_value(uint32_t op, uint64_t addr, uint64_t word, uint64_t& out) {
        const uint64_t sh = (addr & 7) * 8;
        switch (op) {
        case LB: out = (uint64_t)(int64_t)(int8_t)(word >> sh); return true;
        case LBU: out = (word >> sh) & 0xFF; return true;
        case LH: out = (uint64_t)(int64_t)(int16_t)(word >> sh); return !(addr & 1);
        case LHU: out = (word >> sh) & 0xFFFF; return !(addr & 1);
        case LW: out = (uint64_t)(int64_t)(int32_t)(word >> sh); return !(addr & 3);
        case LWU: out = (word >> sh) & 0xFFFFFFFFull; return !(addr & 3);
        default: out = word; return !(addr & 7);
        }
    }
    static bool store_value(uint32_t op, uint64_t src, uint64_t addr, uint64_t word, uint64_t& out) {
        const uint64_t sh = (addr & 7) * 8;
        switch (op) {
        case SB: out = (word & ~(0xFFull << sh)) | ((src & 0xFF) << sh); return true;
        case SH: out = (word & ~(0xFFFFull << sh)) | ((src & 0xFFFF) << sh); return !(addr & 1);
        case SW: out = (word & ~(0xFFFFFFFFull << sh)) | ((src & 0xFFFFFFFFull) << sh); return !(addr & 3);
        default: out = src; return !(addr & 7);
        }
    }
    static uint64_t alu(uint32_t op, uint64_t b, uint64_t c) {
        const int64_t sb = (int64_t)b, sc = (int64_t)c;
        const int32_t wb = (int32_t)b, wc = (int32_t)c;
        auto sx = [](int32_t v) { return (uint64_t)(int64_t)v; };
        switch (op) {
        case ADD: case ADDI: return b + c;
        case SUB: return b - c;
        case XOR: return b ^ c; case OR: return b | c; case AND: return b & c;
        case SLL: return b << (c & 63); case SRL: return b >> (c & 63); case SRA: return (uint64_t)(sb >> (c & 63));
        case SLT: return sb < sc; case SLTU: return b < c;
        case MUL: return b * c;
        case MULH: return (uint64_t)(((__int128)sb * (__int128)sc) >> 64);
        case MULHU: return (uint64_t)(((unsigned __int128)b * (unsigned __int128)c) >> 64);
        case MULHSU: return (uint64_t)(((__int128)sb * (__int128)(unsigned __int128)c) >> 64);
        case DIV: return c == 0 ? ~0ull : (sb == INT64_MIN && sc == -1) ? b : (uint64_t)(sb / sc);
        case DIVU: return c == 0 ? ~0ull : b / c;
        case REM: return c == 0 ? b : (sb == INT64_MIN && sc == -1) ? 0 : (uint64_t)(sb % sc);
        case REMU: return c == 0 ? b : b % c;
        case ADDW: return sx((int32_t)((uint32_t)wb + (uint32_t)wc));
        case SUBW: return sx((int32_t)((uint32_t)wb - (uint32_t)wc));
        case MULW: return sx((int32_t)((uint32_t)wb * (uint32_t)wc));
        case DIVW: return wc == 0 ? ~0ull : (wb == INT32_MIN && wc == -1) ? sx(wb) : sx(wb / wc);
        case DIVUW: return wc == 0 ? ~0ull : sx((int32_t)((uint32_t)b / (uint32_t)c));
        case REMW: return wc == 0 ? sx(wb) : (wb == INT32_MIN && wc == -1) ? 0 : sx(wb % wc);
        case REMUW: return (uint32_t)c == 0 ? sx(wb) : sx((int32_t)((uint32_t)b % (uint32_t)c));
        case SLLW: return sx((int32_t)((uint32_t)b << (c & 31)));
        case SRLW: return sx((int32_t)((uint32_t)b >> (c & 31)));
        default: return sx(wb >> (c & 31));                            // SRAW
        }
    }

    // what the memory instructions' AddressOperation constrains (operations/address.rs:L47-L98; the executor's InvalidMemoryAccess):
    // 2^16 <= addr < 2^48. Below 2^16 an access would alias the register file (addresses < 32 ARE the registers in the memory argument)
    static bool addr_ok(uint64_t addr) { return addr >= (1ull << 16) && addr < (1ull << 48); }

    // an unconstrained block: registers, pc, clock and memory are rolled back at EXIT_UNCONSTRAINED; only WRITE escapes. Its
    // instructions do not count as cycles, so a block that never exits is bounded separately (UNC_STEP_LIMIT)
    static constexpr uint64_t UNC_STEP_LIMIT = 1ull << 32;
    uint64_t unc_steps = 0;
    bool step_unconstrained(const Instr& in) {
        if (++unc_steps > UNC_STEP_LIMIT) return fail("an unconstrained block ran %llu instructions without EXIT_UNCONSTRAINED", (unsigned long long)UNC_STEP_LIMIT);
        auto reg = [&](uint64_t r) { return r ? regs[r].val : 0ull; };
        auto setreg = [&](uint32_t r, uint64_t v) { if (r) regs[r].val = v; };
        uint64_t next_pc = pc + 4;
        if (in.op <= REMUW) setreg(in.a, alu(in.op, reg(in.b), in.imm_c ? in.c : reg(in.c)));
        else if (in.op <= LD) {
            const uint64_t addr = reg(in.b) + in.c; uint64_t v;
            if (!addr_ok(addr)) return fail("invalid memory access in an unconstrained block: load at 0x%llx", (unsigned long long)addr);
            if (!load_value(in.op, addr, peek(addr & ~7ull), v)) return fail("misaligned load in an unconstrained block");
            setreg(in.a, v);
        } else if (in.op <= SD) {
            const uint64_t addr = reg(in.b) + in.c; uint64_t v;
            if (!addr_ok(addr)) return fail("invalid memory access in an unconstrained block: store at 0x%llx", (unsigned long long)addr);
            if (!store_value(in.op, reg(in.a), addr, peek(addr & ~7ull), v)) return fail("misaligned store in an unconstrained block");
            unc_mem[addr & ~7ull] = v;
        } else if (in.op <= BGEU) {
            const uint64_t a = reg(in.a), b = reg(in.b);
            const bool t = in.op == BEQ ? a == b : in.op == BNE ? a != b : in.op == BLT ? (int64_t)a < (int64_t)b
                         : in.op == BGE ? (int64_t)a >= (int64_t)b : in.op == BLTU ? a < b : a >= b;
            if (t) next_pc = pc + in.c;
        } else if (in.op == JAL) { setreg(in.a, pc + 4); next_pc = pc + in.b; }
        else if (in.op == JALR) { const uint64_t t = (reg(in.b) + in.c) & ~1ull; setreg(in.a, pc + 4); next_pc = t; }
        else if (in.op == AUIPC) setreg(in.a, pc + in.b);
        else if (in.op == LUI) setreg(in.a, in.b);
        else if (in.op == ECALL) {
            const uint64_t code = regs[5].val;
            if (code == SYS_WRITE) { if (!sys_write(reg(10), reg(11))) return false; }
            else if (code == SYS_EXIT_UNC) {
                for (int r = 0; r < 32; ++r) regs[r].val = unc_regs[r];
                unc = false; unc_steps = 0; unc_mem.clear(); pc = unc_pc; clk = unc_clk;
                return true;                                           // back at the ENTER_UNCONSTRAINED ecall, now traced with a = 0
            } else return fail("system call 0x%llx inside an unconstrained block", (unsigned long long)code);
        } else return fail("unimplemented instruction in an unconstrained block");
        pc = next_pc;
        return true;
    }

    // ---- the field / curve precompiles (one generic modular arithmetic, rv64_bigmod.hpp). Common shape (minimal/precompiles/ec.rs,
    // fptower/*.rs): the second operand is read at clk, the first is read and rewritten at clk + 1; operands must be reduced
    // (the chips' carries only fit then) and the affine formulas have no special cases, as in the reference's.
    using BI = bigmod::Int;
    // what SyscallAddrOperation constrains (operations/syscall_addr.rs:L51-L93): 8-aligned, above the registers' 2^16, the slice below 2^48
    static bool slice_ok(uint64_t ptr, int n_words) { return !(ptr & 7) && ptr >= (1ull << 16) && ptr < (1ull << 48) && ptr + 8 * (uint64_t)n_words <= (1ull << 48); }
    // reads `n` words at `ptr` for a precompile: (previous timestamp, value) pairs appended to rec, the words returned; the cells'
    // timestamps become `ts` when `stamp` (a read), stay for the caller to set otherwise (a slice that is rewritten)
    void read_words(uint64_t ptr, int n, uint64_t ts, bool stamp, std::vector<uint64_t>& rec, uint64_t* out) {
        for (int i = 0; i < n; ++i) { Cell& m = cell(ptr + 8 * i); touch_precompile(m, ptr + 8 * i); rec.push_back(m.ts); rec.push_back(m.val); out[i] = m.val; if (stamp) m.ts = ts; }
    }
    void write_words(uint64_t ptr, int n, uint64_t ts, const uint64_t* v, std::vector<uint64_t>& rec) {
        for (int i = 0; i < n; ++i) { Cell& m = cell(ptr + 8 * i); m.val = v[i]; m.ts = ts; rec.push_back(v[i]); }
    }
    bool sys_weierstrass(uint64_t code, uint64_t p_ptr, uint64_t q_ptr) {   // minimal/precompiles/ec.rs ec_add / ec_double, curves/src/weierstrass/mod.rs
        const bool is_add = code == SYS_SECP256R1_ADD || code == SYS_BN254_ADD || code == SYS_BLS12381_ADD;
        const bool r1 = code == SYS_SECP256R1_ADD || code == SYS_SECP256R1_DOUBLE, bn = code == SYS_BN254_ADD || code == SYS_BN254_DOUBLE;
        const bigmod::Field& F = r1 ? field_secp256r1() : bn ? field_bn254() : field_bls12381();
        const int fam = (r1 ? F_SECP256R1_ADD : bn ? F_BN254_ADD : F_BLS12381_ADD) + (is_add ? 0 : 1), n = F.n, nw = 2 * n;
        if (!slice_ok(p_ptr, nw) || (is_add && !slice_ok(q_ptr, nw)) || (!is_add && q_ptr != 0)) return fail("curve point arguments (system call 0x%llx)", (unsigned long long)code);
        std::vector<uint64_t> rec = {clk, p_ptr, q_ptr, code};
        uint64_t pw[12], qw[12] = {0};
        std::vector<uint64_t> q_rec;                                      // q is read first (at clk): a word shared with p then carries clk into p's access
        if (is_add) read_words(q_ptr, nw, clk, true, q_rec, qw);
        read_words(p_ptr, nw, 0, false, rec, pw);
        rec.insert(rec.end(), q_rec.begin(), q_rec.end());
        const BI px = bigmod::from_words(pw, n), py = bigmod::from_words(pw + n, n), qx = bigmod::from_words(qw, n), qy = bigmod::from_words(qw + n, n);
        if (!F.reduced(px) || !F.reduced(py) || !F.reduced(qx) || !F.reduced(qy)) return fail("curve coordinate is not reduced (system call 0x%llx)", (unsigned long long)code);
        BI slope;
        if (is_add) {
            const BI den = F.sub(qx, px);
            if (bigmod::is_zero(den)) return fail("curve addition of points with equal x (system call 0x%llx)", (unsigned long long)code);
            slope = F.mul(F.sub(qy, py), F.inv(den));
        } else {
            const BI den = F.dbl(py);
            if (bigmod::is_zero(den)) return fail("curve doubling of a point with y = 0 (system call 0x%llx)", (unsigned long long)code);
            const BI xx = F.mul(px, px);
            BI num = F.add(F.dbl(xx), xx);
            if (r1) num = F.sub(num, bigmod::small(3));                 // a = -3 on secp256r1, 0 on the other curves
            slope = F.mul(num, F.inv(den));
        }
        const BI x3 = F.sub(F.mul(slope, slope), is_add ? F.add(px, qx) : F.dbl(px));
        const BI y3 = F.sub(F.mul(slope, F.sub(px, x3)), py);
        uint64_t out[12];
        memcpy(out, x3.w, 8 * n); memcpy(out + n, y3.w, 8 * n);
        write_words(p_ptr, nw, is_add ? clk + 1 : clk, out, rec);
        family[fam].insert(family[fam].end(), rec.begin(), rec.end());
        return true;
    }
    bool sys_fptower(uint64_t code, uint64_t x_ptr, uint64_t y_ptr) {    // minimal/precompiles/fptower/{fp,fp2_addsub,fp2_mul}.rs
        const bool bls = code < SYS_BN254_FP_ADD;
        const int k = (int)(code - (bls ? SYS_BLS12381_FP_ADD : SYS_BN254_FP_ADD));   // 0..2: Fp add / sub / mul; 3, 4: Fp2 add / sub; 5: Fp2 mul
        const bigmod::Field& F = bls ? field_bls12381() : field_bn254();
        const int fam = (k < 3 ? F_BN254_FP : k < 5 ? F_BN254_FP2_ADDSUB : F_BN254_FP2_MUL) + (bls ? 1 : 0), n = F.n, nw = k < 3 ? n : 2 * n;
        if (!slice_ok(x_ptr, nw) || !slice_ok(y_ptr, nw)) return fail("field operand pointers (system call 0x%llx)", (unsigned long long)code);
        std::vector<uint64_t> rec = {clk, x_ptr, y_ptr, code};
        uint64_t xw[12], yw[12];
        std::vector<uint64_t> y_rec;                                      // y is read first (at clk): x <- x op x is legal, its words then carry clk into x's access
        read_words(y_ptr, nw, clk, true, y_rec, yw);
        read_words(x_ptr, nw, 0, false, rec, xw);
        rec.insert(rec.end(), y_rec.begin(), y_rec.end());
        BI a[2], bb[2], r[2];
        for (int h = 0; h < nw / n; ++h) {
            a[h] = bigmod::from_words(xw + h * n, n); bb[h] = bigmod::from_words(yw + h * n, n);
            if (!F.reduced(a[h]) || !F.reduced(bb[h])) return fail("field operand is not reduced (system call 0x%llx)", (unsigned long long)code);
        }
        if (k == 0 || k == 3) for (int h = 0; h < nw / n; ++h) r[h] = F.add(a[h], bb[h]);
        else if (k == 1 || k == 4) for (int h = 0; h < nw / n; ++h) r[h] = F.sub(a[h], bb[h]);
        else if (k == 2) r[0] = F.mul(a[0], bb[0]);
        else { r[0] = F.sub(F.mul(a[0], bb[0]), F.mul(a[1], bb[1])); r[1] = F.add(F.mul(a[0], bb[1]), F.mul(a[1], bb[0])); }   // u^2 = -1
        uint64_t out[12];
        for (int h = 0; h < nw / n; ++h) memcpy(out + h * n, r[h].w, 8 * n);
        write_words(x_ptr, nw, clk + 1, out, rec);
        family[fam].insert(family[fam].end(), rec.begin(), rec.end());
        return true;
    }
    bool sys_ed_add(uint64_t p_ptr, uint64_t q_ptr) {                    // twisted Edwards, a = -1 (curves/src/edwards/mod.rs ed_add)
        const bigmod::Field& F = field_ed25519();
        if (!slice_ok(p_ptr, 8) || !slice_ok(q_ptr, 8)) return fail("ED_ADD arguments");
        std::vector<uint64_t> rec = {clk, p_ptr, q_ptr, SYS_ED_ADD};
        uint64_t pw[8], qw[8];
        std::vector<uint64_t> q_rec;
        read_words(q_ptr, 8, clk, true, q_rec, qw);
        read_words(p_ptr, 8, 0, false, rec, pw);
        rec.insert(rec.end(), q_rec.begin(), q_rec.end());
        const BI x1 = bigmod::from_words(pw, 4), y1 = bigmod::from_words(pw + 4, 4), x2 = bigmod::from_words(qw, 4), y2 = bigmod::from_words(qw + 4, 4);
        if (!F.reduced(x1) || !F.reduced(y1) || !F.reduced(x2) || !F.reduced(y2)) return fail("ED_ADD coordinate is not reduced");
        const BI xn = F.add(F.mul(x1, y2), F.mul(x2, y1)), yn = F.add(F.mul(y1, y2), F.mul(x1, x2));
        const BI df = F.mul(bigmod::from_words(ED25519_D_W, 4), F.mul(F.mul(x1, y1), F.mul(x2, y2)));
        const BI dx = F.add(bigmod::small(1), df), dy = F.sub(bigmod::small(1), df);
        if (bigmod::is_zero(dx) || bigmod::is_zero(dy)) return fail("ED_ADD: a denominator vanishes (a point is not on the curve)");
        const BI x3 = F.mul(xn, F.inv(dx)), y3 = F.mul(yn, F.inv(dy));
        uint64_t out[8];
        memcpy(out, x3.w, 32); memcpy(out + 4, y3.w, 32);
        write_words(p_ptr, 8, clk + 1, out, rec);
        family[F_ED_ADD].insert(family[F_ED_ADD].end(), rec.begin(), rec.end());
        return true;
    }
    bool sys_ed_decompress(uint64_t ptr, uint64_t sign) {                // minimal/precompiles/edwards/decompress.rs; curves/src/edwards/ed25519.rs:L75-L145
        const bigmod::Field& F = field_ed25519();
        if (!slice_ok(ptr, 8) || sign > 1) return fail("ED_DECOMPRESS arguments");
        std::vector<uint64_t> rec = {clk, ptr, sign, SYS_ED_DECOMPRESS};
        uint64_t xw[4], yw[4];
        read_words(ptr, 4, 0, false, rec, xw);                           // disjoint from the y half: the order does not matter
        read_words(ptr + 32, 4, clk, true, rec, yw);
        const BI y = bigmod::from_words(yw, 4);
        if (!F.reduced(y)) return fail("ED_DECOMPRESS: y is not a reduced field element (the sign bit travels in the second argument)");
        const BI yy = F.mul(y, y), u = F.sub(yy, bigmod::small(1)), v = F.add(F.mul(bigmod::from_words(ED25519_D_W, 4), yy), bigmod::small(1));
        if (bigmod::is_zero(v)) return fail("ED_DECOMPRESS: d y^2 + 1 vanishes");
        const BI a = F.mul(u, F.inv(v));
        BI beta = F.pow(a, bigmod::from_words(ED25519_P_PLUS_3_OVER_8_W, 4));
        const BI bsq = F.mul(beta, beta);
        if (bigmod::eq(bsq, F.neg(a)) && !bigmod::is_zero(a)) beta = F.mul(beta, bigmod::from_words(ED25519_SQRT_M1_W, 4));
        else if (!bigmod::eq(bsq, a)) return fail("ED_DECOMPRESS: not a point of the curve (u / v is not a square)");
        if (beta.w[0] & 1) beta = F.neg(beta);                          // the even root; the chip negates it when sign = 1
        const BI x = sign ? F.neg(beta) : beta;
        write_words(ptr, 4, clk + 1, x.w, rec);
        family[F_ED_DECOMPRESS].insert(family[F_ED_DECOMPRESS].end(), rec.begin(), rec.end());
        return true;
    }
    bool sys_uint256_ops(uint64_t code, uint64_t a_ptr, uint64_t b_ptr) { // vm/syscall/uint256_ops.rs: d, e <- low, high words of a + b + c | a * b + c
        std::vector<uint64_t> rec = {clk, a_ptr, b_ptr, code};
        uint64_t ptrs[3], reg_ts[3];
        for (int i = 0; i < 3; ++i) { Cell& r = regs[12 + i]; touch_precompile(r, 12 + i); ptrs[i] = r.val; reg_ts[i] = r.ts; r.ts = clk; }
        rec.insert(rec.end(), ptrs, ptrs + 3); rec.insert(rec.end(), reg_ts, reg_ts + 3);
        const uint64_t c_ptr = ptrs[0], d_ptr = ptrs[1], e_ptr = ptrs[2];
        if (!slice_ok(a_ptr, 4) || !slice_ok(b_ptr, 4) || !slice_ok(c_ptr, 4) || !slice_ok(d_ptr, 4) || !slice_ok(e_ptr, 4)) return fail("UINT256 add / mul with carry: pointer arguments");
        uint64_t a[4], bw[4], c[4], scratch[4];
        read_words(a_ptr, 4, clk, true, rec, a);
        read_words(b_ptr, 4, clk + 1, true, rec, bw);
        read_words(c_ptr, 4, clk + 2, true, rec, c);
        uint64_t res[9] = {0};
        if (code == SYS_UINT256_ADD_CARRY) {
            unsigned __int128 carry = 0;
            for (int k = 0; k < 4; ++k) { carry += (unsigned __int128)a[k] + bw[k] + c[k]; res[k] = (uint64_t)carry; carry >>= 64; }
            res[4] = (uint64_t)carry;
        } else {
            for (int i = 0; i < 4; ++i) {
                unsigned __int128 carry = 0;
                for (int j = 0; j < 4; ++j) { carry += (unsigned __int128)a[i] * bw[j] + res[i + j]; res[i + j] = (uint64_t)carry; carry >>= 64; }
                res[i + 4] = (uint64_t)carry;
            }
            unsigned __int128 carry = 0;
            for (int k = 0; k < 8; ++k) { carry += (unsigned __int128)res[k] + (k < 4 ? c[k] : 0); res[k] = (uint64_t)carry; carry >>= 64; }
        }
        read_words(d_ptr, 4, 0, false, rec, scratch);                   // the words d and e held before
        for (int i = 0; i < 4; ++i) { Cell& m = cell(d_ptr + 8 * i); m.val = res[i]; m.ts = clk + 3; }
        read_words(e_ptr, 4, 0, false, rec, scratch);
        for (int i = 0; i < 4; ++i) { Cell& m = cell(e_ptr + 8 * i); m.val = res[4 + i]; m.ts = clk + 4; }
        rec.insert(rec.end(), res, res + 8);
        family[F_UINT256_OPS].insert(family[F_UINT256_OPS].end(), rec.begin(), rec.end());
        return true;
    }

    bool sys_write(uint64_t fd, uint64_t buf) {                        // minimal/write.rs:L86-L149 (no memory events)
        const uint64_t n = regs[12].val, start = buf & ~7ull, head = buf & 7;
        if (n > (1ull << 28)) return fail("WRITE of %llu bytes", (unsigned long long)n);
        std::vector<uint8_t> bytes(n);
        for (uint64_t i = 0; i < n; ++i) { const uint64_t o = head + i; bytes[i] = (uint8_t)(peek(start + (o & ~7ull)) >> (8 * (o & 7))); }
        if (fd == 1 || fd == 2) out.insert(out.end(), bytes.begin(), bytes.end());
        else if (fd == FD_PUBLIC_VALUES) public_values.insert(public_values.end(), bytes.begin(), bytes.end());
        else if (fd == FD_HINT) input.push_front(std::move(bytes));
        else if (fd == FD_FP_SQRT || fd == FD_FP_INV) return hook_fp(fd, bytes);
        else return fail("WRITE to file descriptor %llu (a hook) is not implemented", (unsigned long long)fd);
        return true;
    }

    // The field hooks (executor/src/hook.rs:L228-L306): hints computed outside the VM and pushed to the FRONT of the input stream,
    // which the guest then reads and CHECKS inside the VM (one multiplication), so nothing here is trusted. [len: u32 BE | element |
    // modulus | (sqrt only) a non-residue], all big endian, `len` bytes each; 256-bit fields only, square roots for p = 3 (mod 4).
    bool hook_fp(uint64_t fd, const std::vector<uint8_t>& buf) {
        const bool is_sqrt = fd == FD_FP_SQRT;
        if (buf.size() < 4) return fail("field hook: short buffer");
        const uint64_t len = ((uint64_t)buf[0] << 24) | ((uint64_t)buf[1] << 16) | ((uint64_t)buf[2] << 8) | buf[3];
        if (buf.size() != 4 + (is_sqrt ? 3 : 2) * len) return fail("field hook: invalid buffer length");
        if (len != 32) return fail("field hook for %llu-byte fields is not implemented (fd %llu)", (unsigned long long)len, (unsigned long long)fd);
        auto be = [&](uint64_t at) { U256 v = {}; for (int i = 0; i < 32; ++i) v.w[(31 - i) >> 3] |= (uint64_t)buf[at + i] << (8 * ((31 - i) & 7)); return v; };
        auto to_be = [](const U256& v) { std::vector<uint8_t> o(32); for (int i = 0; i < 32; ++i) o[i] = (uint8_t)(v.w[(31 - i) >> 3] >> (8 * ((31 - i) & 7))); return o; };
        const U256 x = be(4), m = be(36);
        const U256 zero = {};
        auto is_zero = [](const U256& v) { return !(v.w[0] | v.w[1] | v.w[2] | v.w[3]); };
        auto mul = [&](const U256& a, const U256& b) { U256 r; uint256_mulmod(a.w, b.w, m.w, r.w); return r; };
        auto pow = [&](const U256& a, const U256& e) {
            U256 r = {{1, 0, 0, 0}}, base = a;
            for (int bit = 0; bit < 256; ++bit) { if ((e.w[bit >> 6] >> (bit & 63)) & 1) r = mul(r, base); base = mul(base, base); }
            return r;
        };
        if (is_zero(m) || !u256_ge(m, x) || u256_ge(x, m)) return fail("field hook: the element is not reduced");
        if (!is_sqrt) {
            if (is_zero(x)) return fail("field hook: inverse of zero");
            U256 e = m; if (e.w[0] < 2) return fail("field hook: modulus"); e.w[0] -= 2;
            input.push_front(to_be(pow(x, e)));
            return true;
        }
        const U256 nqr = be(68);
        if (u256_ge(nqr, m)) return fail("field hook: the non-residue is not reduced");
        if ((m.w[0] & 3) != 3) return fail("field hook: square roots need p = 3 (mod 4) here (Tonelli-Shanks is not implemented)");
        U256 e = m;                                                    // (p + 1) / 4 = (p >> 2) + 1 for p = 3 (mod 4)
        for (int k = 0; k < 4; ++k) e.w[k] = (e.w[k] >> 2) | (k < 3 ? e.w[k + 1] << 62 : 0);
        for (int k = 0; k < 4 && ++e.w[k] == 0; ++k) {}
        uint8_t status = 1;
        U256 root = zero;
        if (!is_zero(x)) {
            root = pow(x, e);
            const U256 sq = mul(root, root);
            if (memcmp(sq.w, x.w, 32) != 0) { status = 0; root = pow(mul(nqr, x), e); }   // not a square: the root of nqr * x proves it
        }
        input.push_front(to_be(root));
        input.push_front(std::vector<uint8_t>{status});
        return true;
    }

    bool step() {
        const uint64_t idx = (pc - pc_base) >> 2;
        if (pc < pc_base || (pc & 3) || idx >= program.size()) return fail("pc outside the program");
        const Instr& in = program[idx];
        if (unc) return step_unconstrained(in);
        uint64_t e[EV] = {0};
        e[E_PC] = pc; e[E_CLK] = clk; e[E_OP] = in.op; e[E_OPA] = in.a; e[E_OPB] = in.b; e[E_OPC] = in.c;
        e[E_FLAGS] = (in.imm_b ? 1 : 0) | (in.imm_c ? 2 : 0);
        uint64_t next_pc = pc + 4, next_clk = clk + CLK_INC, a = 0, b = 0, c = 0;
        if (in.op <= REMUW) {
            if (!in.imm_c) c = rr((uint32_t)in.c, 2, e[E_C_PTS]); else c = in.c;
            b = rr((uint32_t)in.b, 3, e[E_B_PTS]);
            a = alu(in.op, b, c);
            rw(in.a, a, e[E_A_PTS], e[E_A_PREV]);
            if (in.a == 0) a = 0;
        } else if (in.op <= LD) {
            b = rr((uint32_t)in.b, 3, e[E_B_PTS]); c = in.c;
            const uint64_t addr = b + c, al = addr & ~7ull;
            if (!addr_ok(addr)) return fail("invalid memory access: load at 0x%llx", (unsigned long long)addr);
            Cell& m = cell(al);
            touch(m, al);
            if (!load_value(in.op, addr, m.val, a)) return fail("misaligned load at 0x%llx", (unsigned long long)addr);
            e[E_MADDR] = addr; e[E_M_PTS] = m.ts; e[E_M_PREV] = m.val; e[E_M_NEW] = m.val; e[E_FLAGS] |= 4;
            m.ts = clk + 1;
            rw(in.a, a, e[E_A_PTS], e[E_A_PREV]);
            if (in.a == 0) a = 0;
        } else if (in.op <= SD) {
            b = rr((uint32_t)in.b, 3, e[E_B_PTS]); c = in.c;
            a = rr(in.a, 4, e[E_A_PTS]); e[E_A_PREV] = a;
            const uint64_t addr = b + c, al = addr & ~7ull;
            if (!addr_ok(addr)) return fail("invalid memory access: store at 0x%llx", (unsigned long long)addr);
            Cell& m = cell(al);
            touch(m, al);
            uint64_t nv;
            if (!store_value(in.op, a, addr, m.val, nv)) return fail("misaligned store at 0x%llx", (unsigned long long)addr);
            e[E_MADDR] = addr; e[E_M_PTS] = m.ts; e[E_M_PREV] = m.val; e[E_M_NEW] = nv; e[E_FLAGS] |= 4;
            m.ts = clk + 1; m.val = nv;
        } else if (in.op <= BGEU) {
            b = rr((uint32_t)in.b, 3, e[E_B_PTS]); c = in.c;
            a = rr(in.a, 4, e[E_A_PTS]); e[E_A_PREV] = a;
            const bool t = in.op == BEQ ? a == b : in.op == BNE ? a != b : in.op == BLT ? (int64_t)a < (int64_t)b
                         : in.op == BGE ? (int64_t)a >= (int64_t)b : in.op == BLTU ? a < b : a >= b;
            if (t) next_pc = pc + c;
        } else if (in.op == JAL) {
            a = pc + 4; b = in.b; c = 0; next_pc = pc + in.b;
            rw(in.a, a, e[E_A_PTS], e[E_A_PREV]);
            if (in.a == 0) a = 0;
        } else if (in.op == JALR) {
            b = rr((uint32_t)in.b, 3, e[E_B_PTS]); c = in.c;
            a = pc + 4; next_pc = (b + c) & ~1ull;
            rw(in.a, a, e[E_A_PTS], e[E_A_PREV]);
            if (in.a == 0) a = 0;
        } else if (in.op == AUIPC || in.op == LUI) {
            b = c = in.b; a = in.op == AUIPC ? pc + in.b : in.b;
            rw(in.a, a, e[E_A_PTS], e[E_A_PREV]);
            if (in.a == 0) a = 0;
        } else if (in.op == ECALL) {
            const uint64_t code = regs[5].val;
            if (code == SYS_ENTER_UNC && !resume_enter) {             // run the block untraced first; its EXIT comes back here
                for (int r = 0; r < 32; ++r) unc_regs[r] = regs[r].val;
                unc_pc = pc; unc_clk = clk; unc = true; resume_enter = true;
                regs[5].val = 1; pc += 4;
                return true;
            }
            c = rr(11, 2, e[E_C_PTS]); b = rr(10, 3, e[E_B_PTS]);
            a = code; regs_code_of_this_ecall = code;
            switch (code) {
            case SYS_ENTER_UNC: a = 0; resume_enter = false; break;
            case SYS_HALT: next_pc = HALT_PC; exit_code = (uint32_t)b; halted = true; break;
            case SYS_WRITE: if (!sys_write(b, c)) return false; break;
            case SYS_COMMIT:
                if (b >= 8 || (c >> 32)) return fail("COMMIT word %llu", (unsigned long long)b);
                committed_digest[b] = (uint32_t)c; commit_syscall = true; break;
            case SYS_COMMIT_DEFERRED:
                if (b >= 8) return fail("COMMIT_DEFERRED_PROOFS word %llu", (unsigned long long)b);
                deferred_digest[b] = (uint32_t)c; commit_deferred_syscall = true; break;
            case SYS_HINT_LEN: a = input.empty() ? ~0ull : input.front().size(); break;
            case SYS_HINT_READ: {
                if (input.empty()) return fail("hint input stream exhausted");
                std::vector<uint8_t> v = std::move(input.front()); input.pop_front();
                if (v.size() != c || !slice_ok(b, (int)std::min<uint64_t>(v.size() / 8 + 1, 1u << 30))) return fail("HINT_READ of %llu bytes at 0x%llx against an entry of %zu", (unsigned long long)c, (unsigned long long)b, v.size());
                for (uint64_t i = 0; i < (v.size() + 7) / 8; ++i) {    // whole words, then the tail word if there is one (minimal/postprocess.rs:L10-L36)
                    uint64_t w = 0;
                    for (uint64_t j = 0; j < 8 && 8 * i + j < v.size(); ++j) w |= (uint64_t)v[8 * i + j] << (8 * j);
                    Cell& m = cell(b + 8 * i);
                    if (m.ever || m.ts) return fail("hint written over touched memory at 0x%llx", (unsigned long long)(b + 8 * i));
                    m.val = w;
                    // a hinted word is initialised (with the hinted value) and finalised whether or not the program reads it
                    // (prover/src/worker/controller/global.rs:L133-L142: hint addresses join the touched set)
                    m.ever = true; touched.insert(touched.end(), {b + 8 * i, w});
                }
                break;
            }
            case SYS_KECCAK: {
                if (!slice_ok(b, 25) || c != 0) return fail("KECCAK_PERMUTE arguments");
                uint64_t st[25];
                std::vector<uint64_t> rec = {clk, b};
                for (int i = 0; i < 25; ++i) {                        // reads at clk
                    Cell& m = cell(b + 8 * i); touch_precompile(m, b + 8 * i);
                    st[i] = m.val; rec.push_back(m.ts); rec.push_back(m.val); m.ts = clk;
                }
                keccak_f(st);
                for (int i = 0; i < 25; ++i) { Cell& m = cell(b + 8 * i); m.ts = clk + 1; m.val = st[i]; rec.push_back(st[i]); }   // writes at clk + 1
                precompile.insert(precompile.end(), rec.begin(), rec.end());
                break;
            }
            case SYS_SHA_EXTEND: {                                     // vm/syscall/precompiles/sha256/extend.rs, minimal/.../sha256/extend.rs
                if (!slice_ok(b, 64) || c != 0) return fail("SHA_EXTEND arguments");
                std::vector<uint64_t> rec = {clk, b};
                uint64_t first[64][2];                                 // the 64 words' state before the call
                for (int i = 0; i < 64; ++i) { Cell& m = cell(b + 8 * i); touch_precompile(m, b + 8 * i); first[i][0] = m.ts; first[i][1] = m.val; }
                auto rotr = [](uint32_t v, int r) { return (v >> r) | (v << (32 - r)); };
                for (int i = 16; i < 64; ++i) {                        // step i at clk + 1 + (i - 16): four reads, one write
                    const uint64_t ts = clk + 1 + (uint64_t)(i - 16);
                    uint32_t v[4];
                    const int offs[4] = {15, 2, 16, 7};
                    for (int k = 0; k < 4; ++k) {
                        Cell& m = cell(b + 8 * (uint64_t)(i - offs[k]));
                        rec.push_back(m.ts); rec.push_back(m.val);
                        v[k] = (uint32_t)m.val; m.ts = ts;
                    }
                    const uint32_t s0 = rotr(v[0], 7) ^ rotr(v[0], 18) ^ (v[0] >> 3), s1 = rotr(v[1], 17) ^ rotr(v[1], 19) ^ (v[1] >> 10);
                    const uint32_t wi = s1 + v[2] + s0 + v[3];
                    Cell& m = cell(b + 8 * (uint64_t)i);
                    rec.push_back(m.ts); rec.push_back(m.val); rec.push_back(wi);
                    m.ts = ts; m.val = wi;
                }
                for (int i = 0; i < 64; ++i) { Cell& m = cell(b + 8 * i); rec.insert(rec.end(), {first[i][0], first[i][1], m.ts, m.val}); }
                sha_extend.insert(sha_extend.end(), rec.begin(), rec.end());
                break;
            }
            case SYS_SHA_COMPRESS: {                                   // vm/syscall/precompiles/sha256/compress.rs: h read at clk, w at clk + 1, h written at clk + 2
                if (!slice_ok(b, 64) || !slice_ok(c, 8)) return fail("SHA_COMPRESS arguments");
                static const uint32_t K[64] = {
                    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
                    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
                    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
                    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
                    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
                    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
                auto rotr = [](uint32_t v, int r) { return (v >> r) | (v << (32 - r)); };
                std::vector<uint64_t> rec = {clk, b, c};
                uint32_t h[8], w[64];
                for (int i = 0; i < 8; ++i) { Cell& m = cell(c + 8 * i); touch_precompile(m, c + 8 * i); rec.push_back(m.ts); rec.push_back(m.val); h[i] = (uint32_t)m.val; m.ts = clk; }
                for (int i = 0; i < 64; ++i) { Cell& m = cell(b + 8 * i); touch_precompile(m, b + 8 * i); rec.push_back(m.ts); rec.push_back(m.val); w[i] = (uint32_t)m.val; m.ts = clk + 1; }
                uint32_t v[8];
                memcpy(v, h, sizeof v);
                for (int i = 0; i < 64; ++i) {
                    const uint32_t s1 = rotr(v[4], 6) ^ rotr(v[4], 11) ^ rotr(v[4], 25), ch = (v[4] & v[5]) ^ (~v[4] & v[6]);
                    const uint32_t t1 = v[7] + s1 + ch + K[i] + w[i];
                    const uint32_t s0 = rotr(v[0], 2) ^ rotr(v[0], 13) ^ rotr(v[0], 22), maj = (v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]);
                    const uint32_t t2 = s0 + maj;
                    v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
                }
                for (int i = 0; i < 8; ++i) { Cell& m = cell(c + 8 * i); m.val = (uint32_t)(h[i] + v[i]); m.ts = clk + 2; rec.push_back(m.val); }
                sha_compress.insert(sha_compress.end(), rec.begin(), rec.end());
                break;
            }
            case SYS_UINT256_MUL: {                                    // y and the modulus (y_ptr + 32) read at clk, x rewritten at clk + 1
                if (!slice_ok(b, 4) || !slice_ok(c, 8)) return fail("UINT256_MUL arguments");
                std::vector<uint64_t> rec = {clk, b, c};
                uint64_t x[4], ym[8], r[4];
                for (int i = 0; i < 4; ++i) { Cell& m = cell(b + 8 * i); touch_precompile(m, b + 8 * i); rec.push_back(m.ts); rec.push_back(m.val); x[i] = m.val; }
                for (int i = 0; i < 8; ++i) { Cell& m = cell(c + 8 * i); touch_precompile(m, c + 8 * i); rec.push_back(m.ts); rec.push_back(m.val); ym[i] = m.val; m.ts = clk; }
                uint256_mulmod(x, ym, ym + 4, r);
                for (int i = 0; i < 4; ++i) { Cell& m = cell(b + 8 * i); m.val = r[i]; m.ts = clk + 1; rec.push_back(r[i]); }
                uint256.insert(uint256.end(), rec.begin(), rec.end());
                break;
            }
            case SYS_SECP256K1_ADD: case SYS_SECP256K1_DOUBLE: {       // vm/syscall/precompiles/weierstrass/{add,double}.rs: affine, no special cases
                const bool is_add = code == SYS_SECP256K1_ADD;
                if (!slice_ok(b, 8) || (is_add && !slice_ok(c, 8)) || (!is_add && c != 0)) return fail("SECP256K1 point arguments");
                std::vector<uint64_t> rec = {clk, b};
                if (is_add) rec.push_back(c);
                U256 px, py, qx = {}, qy = {};
                for (int i = 0; i < 8; ++i) { Cell& m = cell(b + 8 * i); touch_precompile(m, b + 8 * i); rec.push_back(m.ts); rec.push_back(m.val); (i < 4 ? px : py).w[i & 3] = m.val; }
                if (is_add)
                    for (int i = 0; i < 8; ++i) { Cell& m = cell(c + 8 * i); touch_precompile(m, c + 8 * i); rec.push_back(m.ts); rec.push_back(m.val); (i < 4 ? qx : qy).w[i & 3] = m.val; m.ts = clk; }
                if (!u256_ge(SECP_P, px) || u256_ge(px, SECP_P) || u256_ge(py, SECP_P) || (is_add && (u256_ge(qx, SECP_P) || u256_ge(qy, SECP_P)))) return fail("SECP256K1 coordinate is not reduced");
                U256 slope;
                if (is_add) {
                    const U256 den = fp_sub(qx, px);
                    if (!(den.w[0] | den.w[1] | den.w[2] | den.w[3])) return fail("SECP256K1_ADD of points with equal x");
                    slope = fp_mul(fp_sub(qy, py), fp_inv(den));
                } else {
                    const U256 three = {{3, 0, 0, 0}}, two = {{2, 0, 0, 0}};
                    const U256 den = fp_mul(two, py);
                    if (!(den.w[0] | den.w[1] | den.w[2] | den.w[3])) return fail("SECP256K1_DOUBLE of a point with y = 0");
                    slope = fp_mul(fp_mul(three, fp_mul(px, px)), fp_inv(den));
                }
                const U256 x3 = fp_sub(fp_mul(slope, slope), is_add ? fp_add(px, qx) : fp_add(px, px));
                const U256 y3 = fp_sub(fp_mul(slope, fp_sub(px, x3)), py);
                for (int i = 0; i < 8; ++i) { Cell& m = cell(b + 8 * i); m.val = (i < 4 ? x3 : y3).w[i & 3]; m.ts = is_add ? clk + 1 : clk; rec.push_back(m.val); }
                (is_add ? secp_add : secp_double).insert((is_add ? secp_add : secp_double).end(), rec.begin(), rec.end());
                break;
            }
            case SYS_SECP256R1_ADD: case SYS_BN254_ADD: case SYS_BLS12381_ADD: case SYS_SECP256R1_DOUBLE: case SYS_BN254_DOUBLE: case SYS_BLS12381_DOUBLE:
                if (!sys_weierstrass(code, b, c)) return false;
                break;
            case SYS_BLS12381_FP_ADD: case SYS_BLS12381_FP_ADD + 1: case SYS_BLS12381_FP_ADD + 2: case SYS_BLS12381_FP_ADD + 3: case SYS_BLS12381_FP_ADD + 4: case SYS_BLS12381_FP2_MUL:
            case SYS_BN254_FP_ADD: case SYS_BN254_FP_ADD + 1: case SYS_BN254_FP_ADD + 2: case SYS_BN254_FP_ADD + 3: case SYS_BN254_FP_ADD + 4: case SYS_BN254_FP2_MUL:
                if (!sys_fptower(code, b, c)) return false;
                break;
            case SYS_ED_ADD: if (!sys_ed_add(b, c)) return false; break;
            case SYS_ED_DECOMPRESS: if (!sys_ed_decompress(b, c)) return false; break;
            case SYS_UINT256_ADD_CARRY: case SYS_UINT256_MUL_CARRY: if (!sys_uint256_ops(code, b, c)) return false; break;
            case SYS_POSEIDON2: {                                      // vm/syscall/poseidon2.rs, minimal/precompiles/poseidon2.rs
                if (!slice_ok(b, 8) || c != 0) return fail("POSEIDON2 arguments");
                uint32_t st[16];
                std::vector<uint64_t> rec = {clk, b};
                for (int i = 0; i < 8; ++i) {                          // eight words = sixteen field elements, rewritten in place at clk
                    Cell& m = cell(b + 8 * i); touch_precompile(m, b + 8 * i);
                    const uint32_t lo = (uint32_t)m.val, hi = (uint32_t)(m.val >> 32);
                    if (lo >= kb::P || hi >= kb::P) return fail("POSEIDON2 input is not a field element");
                    st[2 * i] = kb::to_monty(lo); st[2 * i + 1] = kb::to_monty(hi);
                    rec.push_back(m.ts); rec.push_back(m.val);
                }
                if (sp1hip_poseidon2_permute_host(st, 1, 1) != SP1HIP_SUCCESS) return fail("POSEIDON2 permutation failed");
                for (int i = 0; i < 8; ++i) {
                    Cell& m = cell(b + 8 * i);
                    m.val = (uint64_t)kb::from_monty(st[2 * i]) | ((uint64_t)kb::from_monty(st[2 * i + 1]) << 32);
                    m.ts = clk;
                    rec.push_back(m.val);
                }
                poseidon2.insert(poseidon2.end(), rec.begin(), rec.end());
                break;
            }
            case SYS_EXIT_UNC: case SYS_VERIFY_PROOF: break;
            default: return fail("system call 0x%llx is not implemented", (unsigned long long)code);
            }
            rw(5, a, e[E_A_PTS], e[E_A_PREV]);
            next_clk += ECALL_EXTRA;
        } else return fail(in.op == EBREAK ? "ebreak" : "unimplemented instruction 0x%08x", words[idx]);
        e[E_A] = a; e[E_B] = b; e[E_C] = c; e[E_NEXT_PC] = next_pc;
        if (record) events.insert(events.end(), e, e + EV);
        if (limits.on) estimate(in, in.op == ECALL && ((regs_code_of_this_ecall >> 8) & 0xFF) == 1, (clk >> 24) != (next_clk >> 24));
        pc = next_pc; clk = next_clk; ++cycles;
        return true;
    }

    // The shard-cutting estimator (vm/shapes.rs:L27-L245, `ShapeChecker`): trace area and tallest table of the shard so far, kept
    // per executed instruction from the chips' costs — the row of the instruction's chip, a MemoryLocal row and two Global rows
    // per address first touched (the 32 registers are assumed touched up front), a SyscallCore and a Global row per call sent to
    // a precompile shard, 32 MemoryBump rows and a StateBump row when the clock's high limb moves. The shard ends when the area
    // reaches the element threshold or a table the height threshold (both less the HALT allowance, splicing.rs:L413-L428),
    // unless a COMMIT has begun (the commit rows and the HALT share a shard). Costs and thresholds come from the caller.
    struct Limits {
        bool on = false;
        uint64_t element_threshold = 0, height_threshold = 0, fixed_area = 0;
        uint64_t op_cost[64] = {0}; uint32_t op_chip[64] = {0};
        uint64_t alu_x0 = 0, load_x0 = 0, memory_local = 0, global = 0, syscall_core = 0, memory_bump = 0, state_bump = 0;
    } limits;
    enum { H_ALU_X0 = 64, H_LOAD_X0, H_LOCAL, H_GLOBAL, H_SYSCALL_CORE, H_MEMORY_BUMP, H_STATE_BUMP, H_COUNT };
    uint64_t est_area = 0, est_max_height = 0, est_new_local = 0, est_heights[H_COUNT] = {0};
    uint64_t regs_code_of_this_ecall = 0;
    void estimate_reset() {
        est_area = limits.fixed_area + (1ull << 18) + (1ull << 18);      // MAXIMUM_PADDING_AREA + MAXIMUM_CYCLE_AREA
        est_max_height = 0; est_new_local = 32;
        memset(est_heights, 0, sizeof est_heights);
    }
    void est_add(uint32_t chip, uint64_t rows, uint64_t cost) {
        est_heights[chip] += rows; est_area += rows * cost;
        if (est_heights[chip] > est_max_height) est_max_height = est_heights[chip];
    }
    void estimate(const Instr& in, bool sent, bool clk_high_moves) {
        if (in.op <= REMUW && in.a == 0) est_add(H_ALU_X0, 1, limits.alu_x0);
        else if (in.op >= LB && in.op <= LD && in.a == 0) est_add(H_LOAD_X0, 1, limits.load_x0);
        else est_add(limits.op_chip[in.op] & 63, 1, limits.op_cost[in.op]);
        const uint64_t n = est_new_local; est_new_local = 0;
        est_add(H_LOCAL, n, limits.memory_local);
        est_add(H_GLOBAL, 2 * n + (sent ? 1 : 0), limits.global);
        if (clk_high_moves) { est_add(H_MEMORY_BUMP, 32, limits.memory_bump); est_add(H_STATE_BUMP, 1, limits.state_bump); }
        if (sent) est_add(H_SYSCALL_CORE, 1, limits.syscall_core);
    }
    bool shard_limit_reached() const {
        return limits.on && !commit_syscall && !commit_deferred_syscall &&
               (est_area + (1ull << 18) >= limits.element_threshold || est_max_height + (1ull << 10) >= limits.height_threshold);
    }
    bool resume_enter = false;
    bool record = true;                                               // sp1hip_rv64_set_recording: instruction events are kept

    void finish_shard() {
        for (size_t i = 0; i < local.size(); i += 5) {
            if (local_closed[i / 5]) continue;
            const uint64_t addr = local[i];
            const Cell& c = addr < 32 ? regs[addr] : cell(addr);
            local[i + 3] = c.ts; local[i + 4] = c.val;
        }
    }
};

bool load_elf(Vm& vm, const uint8_t* p, size_t n) {
    auto rd = [&](size_t off, int bytes) { uint64_t v = 0; for (int i = 0; i < bytes; ++i) v |= (uint64_t)p[off + i] << (8 * i); return v; };
    if (n < 64 || memcmp(p, "\x7f" "ELF", 4) != 0 || p[4] != 2 || p[5] != 1) return vm.fail("not a little-endian ELF64");
    if (rd(16, 2) != 2 || rd(18, 2) != 243) return vm.fail("not a RISC-V executable");
    const uint64_t entry = rd(24, 8), phoff = rd(32, 8), phentsize = rd(54, 2), phnum = rd(56, 2);
    if (entry & 3) return vm.fail("entry point is not aligned");
    // header arithmetic without wrap-around: the table of 56-byte entries lies inside the file
    if (phentsize != 56 || phoff > n || phnum > (n - phoff) / 56) return vm.fail("program header table outside the file");
    constexpr uint64_t MAX_SEGMENT = 1ull << 32;                       // 4 GiB per segment: far above any guest, far below an allocation bomb
    bool have_base = false;
    std::vector<uint64_t> image_words;
    for (uint64_t i = 0; i < phnum; ++i) {
        const size_t ph = phoff + i * phentsize;
        if (rd(ph, 4) != 1) continue;                                  // PT_LOAD
        const uint64_t flags = rd(ph + 4, 4), off = rd(ph + 8, 8), vaddr = rd(ph + 16, 8), filesz = rd(ph + 32, 8), memsz = rd(ph + 40, 8);
        if ((vaddr & 3) || filesz > n || off > n - filesz) return vm.fail("segment is not aligned or outside the file");
        if (filesz > memsz || memsz > MAX_SEGMENT || vaddr >= (1ull << 48) || vaddr + memsz > (1ull << 48)) return vm.fail("segment size or address out of range");
        const bool exec = flags & 1;
        if (exec && !have_base) { vm.pc_base = vaddr; have_base = true; }
        else if (exec && vaddr != vm.pc_base + 4 * vm.program.size()) return vm.fail("executable segments are not contiguous");
        for (uint64_t addr = vaddr; addr < vaddr + memsz; addr += 4) {
            uint64_t w = 0;
            Cell& c = vm.cell(addr & ~7ull);
            if (image_words.empty() || image_words.back() != (addr & ~7ull)) image_words.push_back(addr & ~7ull);
            if (addr < vaddr + filesz) { const uint64_t m = filesz - (addr - vaddr); for (uint64_t j = 0; j < 4 && j < m; ++j) w |= (uint64_t)p[off + (addr - vaddr) + j] << (8 * j); }
            else { c.val = 0; continue; }                              // zero fill REPLACES the word (elf.rs:L321-L324: `image.insert(addr - addr % 8, 0)`)
            c.val += w << (8 * (addr & 4));
            if (exec) { vm.words.push_back((uint32_t)w); vm.program.push_back(decode((uint32_t)w)); }
        }
    }
    std::sort(image_words.begin(), image_words.end());
    image_words.erase(std::unique(image_words.begin(), image_words.end()), image_words.end());
    for (uint64_t a : image_words) { vm.image.push_back(a); vm.image.push_back(vm.cell(a).val); }
    if (!have_base || vm.program.empty()) return vm.fail("no executable segment");
    vm.pc_start = vm.pc = entry;
    return true;
}

}  // namespace

extern "C" {

int sp1hip_rv64_create(const uint8_t* elf, uint64_t elf_len, sp1hip_rv64_vm_t* out) {
    if (!elf || !out) { sp1hip::set_error("sp1hip_rv64_create: null argument"); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    Vm* vm = new Vm();
    if (!load_elf(*vm, elf, elf_len)) { sp1hip::set_error("sp1hip_rv64_create: %s", vm->error.c_str()); delete vm; return SP1HIP_ERROR_INVALID_ARGUMENT; }
    *out = vm;
    return SP1HIP_SUCCESS;
}

void sp1hip_rv64_destroy(sp1hip_rv64_vm_t vm) { delete (Vm*)vm; }

int sp1hip_rv64_write_stdin(sp1hip_rv64_vm_t h, const uint8_t* data, uint64_t len) {
    if (!h || (!data && len)) { sp1hip::set_error("sp1hip_rv64_write_stdin: null argument"); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    ((Vm*)h)->input.emplace_back(data, data + len);
    return SP1HIP_SUCCESS;
}

int sp1hip_rv64_run_shard(sp1hip_rv64_vm_t h, uint64_t max_cycles, sp1hip_rv64_shard_info_t* info) {
    if (!h || !info) { sp1hip::set_error("sp1hip_rv64_run_shard: null argument"); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    Vm& vm = *(Vm*)h;
    if (vm.halted) { sp1hip::set_error("sp1hip_rv64_run_shard: the program has halted"); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    vm.events.clear(); vm.local.clear(); vm.local_closed.clear(); vm.precompile.clear(); vm.poseidon2.clear(); vm.sha_extend.clear(); vm.sha_compress.clear(); vm.uint256.clear(); vm.secp_add.clear(); vm.secp_double.clear(); for (auto& f : vm.family) f.clear();
    if (vm.record) vm.events.reserve((size_t)std::min<uint64_t>(max_cycles, 1ull << 24) * EV);   // one allocation (at most 2.7 GB), not a doubling chain of copies
    info->pc_start = vm.pc; info->clk_start = vm.clk;
    const uint64_t c0 = vm.cycles;
    vm.estimate_reset();
    while (!vm.halted && (vm.unc || (vm.cycles - c0 < max_cycles && !vm.shard_limit_reached())))
        if (!vm.step()) { sp1hip::set_error("sp1hip_rv64_run_shard: %s", vm.error.c_str()); return SP1HIP_ERROR_RUNTIME; }
    vm.finish_shard();
    info->n_cycles = vm.cycles - c0; info->n_events = vm.events.size() / EV; info->n_local = vm.local.size() / 5; info->n_keccak = vm.precompile.size() / SP1HIP_RV64_KECCAK_WORDS;
    info->n_poseidon2 = vm.poseidon2.size() / SP1HIP_RV64_POSEIDON2_WORDS;
    info->n_sha_extend = vm.sha_extend.size() / SP1HIP_RV64_SHA_EXTEND_WORDS; info->n_sha_compress = vm.sha_compress.size() / SP1HIP_RV64_SHA_COMPRESS_WORDS;
    info->n_uint256 = vm.uint256.size() / SP1HIP_RV64_UINT256_WORDS;
    info->n_secp256k1_add = vm.secp_add.size() / SP1HIP_RV64_SECP_ADD_WORDS; info->n_secp256k1_double = vm.secp_double.size() / SP1HIP_RV64_SECP_DOUBLE_WORDS;
    info->estimated_area = vm.limits.on ? vm.est_area : 0; info->estimated_max_height = vm.limits.on ? vm.est_max_height : 0;
    info->next_pc = vm.pc; info->clk_end = vm.clk; info->halted = vm.halted; info->exit_code = vm.exit_code;
    info->shard = vm.shard++;
    info->commit_syscall = vm.commit_syscall; info->commit_deferred_syscall = vm.commit_deferred_syscall;
    memcpy(info->committed_value_digest, vm.committed_digest, sizeof vm.committed_digest);
    memcpy(info->deferred_proofs_digest, vm.deferred_digest, sizeof vm.deferred_digest);
    return SP1HIP_SUCCESS;
}

int sp1hip_rv64_set_shard_limits(sp1hip_rv64_vm_t h, const sp1hip_rv64_shard_limits_t* l) {
    if (!h) { sp1hip::set_error("sp1hip_rv64_set_shard_limits: null handle"); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    Vm& vm = *(Vm*)h;
    vm.limits = Vm::Limits();
    if (!l) return SP1HIP_SUCCESS;                                    // back to cycle counts only
    if (l->element_threshold < (1ull << 18) || l->height_threshold < (1ull << 10)) { sp1hip::set_error("sp1hip_rv64_set_shard_limits: thresholds below the HALT allowance"); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    vm.limits.on = true;
    vm.limits.element_threshold = l->element_threshold; vm.limits.height_threshold = l->height_threshold; vm.limits.fixed_area = l->fixed_area;
    for (int k = 0; k < 64; ++k) { vm.limits.op_cost[k] = l->opcode_cost[k]; vm.limits.op_chip[k] = l->opcode_chip[k]; }
    vm.limits.alu_x0 = l->alu_x0_cost; vm.limits.load_x0 = l->load_x0_cost; vm.limits.memory_local = l->memory_local_cost; vm.limits.global = l->global_cost;
    vm.limits.syscall_core = l->syscall_core_cost; vm.limits.memory_bump = l->memory_bump_cost; vm.limits.state_bump = l->state_bump_cost;
    return SP1HIP_SUCCESS;
}

int sp1hip_rv64_set_recording(sp1hip_rv64_vm_t h, int on) {
    if (!h) { sp1hip::set_error("sp1hip_rv64_set_recording: null handle"); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    ((Vm*)h)->record = on != 0;
    return SP1HIP_SUCCESS;
}

const uint64_t* sp1hip_rv64_events(sp1hip_rv64_vm_t h) { return ((Vm*)h)->events.data(); }
const uint64_t* sp1hip_rv64_local_memory(sp1hip_rv64_vm_t h) { return ((Vm*)h)->local.data(); }
const uint64_t* sp1hip_rv64_keccak_events(sp1hip_rv64_vm_t h) { return ((Vm*)h)->precompile.data(); }
const uint64_t* sp1hip_rv64_poseidon2_events(sp1hip_rv64_vm_t h) { return ((Vm*)h)->poseidon2.data(); }
const uint64_t* sp1hip_rv64_sha_extend_events(sp1hip_rv64_vm_t h) { return ((Vm*)h)->sha_extend.data(); }
const uint64_t* sp1hip_rv64_sha_compress_events(sp1hip_rv64_vm_t h) { return ((Vm*)h)->sha_compress.data(); }
const uint64_t* sp1hip_rv64_uint256_events(sp1hip_rv64_vm_t h) { return ((Vm*)h)->uint256.data(); }
const uint64_t* sp1hip_rv64_secp256k1_add_events(sp1hip_rv64_vm_t h) { return ((Vm*)h)->secp_add.data(); }
const uint64_t* sp1hip_rv64_secp256k1_double_events(sp1hip_rv64_vm_t h) { return ((Vm*)h)->secp_double.data(); }
int sp1hip_rv64_precompile_events(sp1hip_rv64_vm_t h, uint32_t family, uint64_t* n_events, uint64_t* words_per_event, const uint64_t** data) {
    if (!h || family >= N_FAMILIES || !n_events || !words_per_event || !data) { sp1hip::set_error("sp1hip_rv64_precompile_events: null argument or unknown family"); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    const std::vector<uint64_t>& v = ((Vm*)h)->family[family];
    *words_per_event = FAMILY_WORDS[family]; *n_events = v.size() / FAMILY_WORDS[family]; *data = v.data();
    return SP1HIP_SUCCESS;
}

int sp1hip_rv64_program(sp1hip_rv64_vm_t h, uint64_t* pc_base, uint64_t* n_instructions, const uint64_t** table) {
    if (!h) return SP1HIP_ERROR_INVALID_ARGUMENT;
    Vm& vm = *(Vm*)h;
    static thread_local std::vector<uint64_t> t;
    t.clear();
    for (const Instr& in : vm.program) t.insert(t.end(), {in.op, in.a, in.b, in.c, (uint64_t)in.imm_b, (uint64_t)in.imm_c});
    if (pc_base) *pc_base = vm.pc_base;
    if (n_instructions) *n_instructions = vm.program.size();
    if (table) *table = t.data();
    return SP1HIP_SUCCESS;
}

int sp1hip_rv64_global_memory(sp1hip_rv64_vm_t h, uint64_t* n, const uint64_t** table) {
    if (!h || !n || !table) return SP1HIP_ERROR_INVALID_ARGUMENT;
    Vm& vm = *(Vm*)h;
    vm.global_out.clear();
    for (size_t i = 0; i < vm.touched.size(); i += 2) {
        const uint64_t addr = vm.touched[i];
        const Cell& c = addr < 32 ? vm.regs[addr] : vm.cell(addr);
        vm.global_out.insert(vm.global_out.end(), {addr, vm.touched[i + 1], c.val, c.ts});
    }
    *n = vm.touched.size() / 2; *table = vm.global_out.data();
    return SP1HIP_SUCCESS;
}

int sp1hip_rv64_memory_image(sp1hip_rv64_vm_t h, uint64_t* n, const uint64_t** table) {
    if (!h || !n || !table) return SP1HIP_ERROR_INVALID_ARGUMENT;
    const Vm& vm = *(Vm*)h;
    *n = vm.image.size() / 2; *table = vm.image.data();
    return SP1HIP_SUCCESS;
}

int sp1hip_rv64_output(sp1hip_rv64_vm_t h, int which, const uint8_t** data, uint64_t* len) {
    if (!h || !data || !len) return SP1HIP_ERROR_INVALID_ARGUMENT;
    Vm& vm = *(Vm*)h;
    const std::vector<uint8_t>& v = which == 0 ? vm.public_values : vm.out;
    *data = v.data(); *len = v.size();
    return SP1HIP_SUCCESS;
}

}  // extern "C"
