// sp1_amd/csrc/tracegen_riscv.hip — device trace generation for the RISC-V instruction chips that make up a core shard
// (VERDICT r5 #6): Add, Addi, Sub, Addw, Subw, Mul, ShiftRight, Branch — 99.6 % of the rows of a shard
// of the reference's fibonacci guest — from compact EVENT records instead of host-made tables.
//
// The reference fills these tables on the host, row by row, from its executor's events and copies them to the device
// (`generate_trace_into` / `event_to_row` of each chip, then sp1-gpu's `device_main_tracegen`,
// /root/reference/sp1-gpu/crates/jagged_tracegen/src/lib.rs:L819-L835); its device tracegen covers the recursion chips and
// the Global chip. Here one lane fills one row: an event is 88 bytes, a row 120-328, and the fields are a few dozen integer
// operations each, so the kernels run at the rate the table can be written.
//
//   CPUState                 /root/reference/crates/core/machine/src/adapter/state.rs:L26-L69
//   register access columns  crates/core/machine/src/memory/consistency/trace.rs:L22-L33, L104-L127
//   R / I / ALU adapters     crates/core/machine/src/adapter/register/{r_type,i_type,alu_type}.rs (populate)
//   Add / Addi / Sub         crates/core/machine/src/alu/add_sub/{add,addi,sub}.rs event_to_row
//   Addw / Subw              crates/core/machine/src/alu/{addw,subw}/mod.rs
//   Mul                      crates/core/machine/src/alu/mul/mod.rs, operations/mul.rs:L54-L137 (MulOperation::populate)
//   ShiftRight               crates/core/machine/src/alu/sr/mod.rs:L239-L312 (padding rows L165-L171)
//   Branch                   crates/core/machine/src/control_flow/branch/{columns,trace}.rs, operations/slt.rs:L50-L174
//
// Column order = the reference's #[repr(C)] column structs (sp1_amd/machines/riscv.py transcribes the same structs; its
// layouts are what tests/test_gpu_tracegen_riscv.py compares the constants below with). Output: column-major [width][height]
// Montgomery words, rows >= n_events are the chip's padding rows.
#include "device_ctx.hpp"

namespace sp1hip {
namespace tg {

constexpr uint32_t M16 = 0xffffu;
// flags of sp1hip_rv64_alu_event_t.ops (bits 32..): operand b / c is an immediate
constexpr uint64_t F_IMM_C = 1ull << 33;                     // (bit 32: operand b is an immediate — none of these chips has one)
constexpr uint32_t POS_C = 2, POS_B = 3, POS_A = 4;         // MemoryAccessPosition (core/executor/src/events/memory.rs:L63-L74)

struct Ev { uint64_t pc, clk, ops, a, b, c, a_prev, a_pts, b_pts, c_pts, aux; };
static_assert(sizeof(Ev) == sizeof(sp1hip_rv64_alu_event_t), "event layout");

// Opcode numbers (core/executor/src/opcode.rs:L46-L153, sp1_amd/machines/riscv.py OPC)
enum : uint32_t { OP_SRL = 7, OP_SRA = 8, OP_MUL = 11, OP_MULH = 12, OP_MULHU = 13, OP_MULHSU = 14, OP_SRLW = 22, OP_SRAW = 23, OP_MULW = 24,
                  OP_BEQ = 40, OP_BNE = 41, OP_BLT = 42, OP_BGE = 43, OP_BLTU = 44, OP_BGEU = 45 };

template <int W> struct Row {
    uint32_t c[W];
    __device__ __forceinline__ void limbs4(int at, uint64_t v) { c[at] = v & M16; c[at + 1] = (v >> 16) & M16; c[at + 2] = (v >> 32) & M16; c[at + 3] = (v >> 48) & M16; }
    __device__ __forceinline__ void limbs3(int at, uint64_t v) { c[at] = v & M16; c[at + 1] = (v >> 16) & M16; c[at + 2] = (v >> 32) & M16; }
};

// CPUState at columns 0..5
template <int W> __device__ __forceinline__ void fill_state(Row<W>& r, const Ev& e) {
    r.c[0] = (uint32_t)(e.clk >> 24); r.c[1] = (uint32_t)(e.clk >> 16) & 0xff; r.c[2] = (uint32_t)e.clk & M16;
    r.limbs3(3, e.pc);
}
// RegisterAccessCols at `at`: prev_value[4], prev_low, diff_low_limb. A previous access on the other side of a 2^24 clock
// boundary is bridged by a MemoryBump row (host side, rare): the columns then compare against 0
template <int W> __device__ __forceinline__ void fill_access(Row<W>& r, int at, uint64_t value, uint64_t t_prev, uint64_t t_cur, bool live = true) {
    r.limbs4(at, value);
    const bool cross = (t_prev >> 24) != (t_cur >> 24);
    const uint32_t old = cross ? 0u : (uint32_t)t_prev & 0xffffffu;
    const uint32_t diff = ((uint32_t)t_cur & 0xffffffu) - old - 1u;
    r.c[at + 4] = live ? old : 0u;
    r.c[at + 5] = live ? (diff & M16) : 0u;
}
// the part every adapter shares: op_a, its access, op_a_0, op_b, its access (columns 6..20)
template <int W> __device__ __forceinline__ void fill_ab(Row<W>& r, const Ev& e) {
    const uint32_t ra = (uint32_t)(e.ops >> 8) & 0xff, rb = (uint32_t)(e.ops >> 16) & 0xff;
    r.c[6] = ra;
    fill_access(r, 7, e.a_prev, e.a_pts, e.clk + POS_A);
    r.c[13] = ra == 0;
    r.c[14] = rb;
    fill_access(r, 15, e.b, e.b_pts, e.clk + POS_B);
}
// RTypeReader: + op_c, its access (21..27)
template <int W> __device__ __forceinline__ void fill_r(Row<W>& r, const Ev& e) {
    fill_ab(r, e);
    r.c[21] = (uint32_t)(e.ops >> 24) & 0xff;
    fill_access(r, 22, e.c, e.c_pts, e.clk + POS_C);
}
// ITypeReader: + op_c_imm[4] (21..24)
template <int W> __device__ __forceinline__ void fill_i(Row<W>& r, const Ev& e) {
    fill_ab(r, e);
    r.limbs4(21, e.c);
}
// ALUTypeReader: + op_c[4] (a register number or the immediate's limbs), its access (no access for an immediate; prev_value
// holds the operand either way), imm_c (21..31)
template <int W> __device__ __forceinline__ void fill_alu(Row<W>& r, const Ev& e) {
    fill_ab(r, e);
    const bool imm = (e.ops & F_IMM_C) != 0;
    if (imm) r.limbs4(21, e.c); else { r.c[21] = (uint32_t)(e.ops >> 24) & 0xff; r.c[22] = r.c[23] = r.c[24] = 0; }
    fill_access(r, 25, e.c, e.c_pts, e.clk + POS_C, !imm);
    r.c[31] = imm;
}

// LtOperationSigned / Unsigned::populate at `at`: bit, u16_flags[4], not_eq_inv, comparison_limbs[2], b_msb, c_msb
template <int W> __device__ __forceinline__ void fill_lt(Row<W>& r, int at, uint64_t b, uint64_t c, bool is_signed) {
    uint32_t bl[4], cl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { bl[i] = (b >> (16 * i)) & M16; cl[i] = (c >> (16 * i)) & M16; }
    r.c[at + 8] = is_signed ? bl[3] >> 15 : 0u;
    r.c[at + 9] = is_signed ? cl[3] >> 15 : 0u;
    if (is_signed) { bl[3] ^= 0x8000u; cl[3] ^= 0x8000u; }
    int idx = -1;
#pragma unroll
    for (int i = 0; i < 4; i++) if (bl[i] != cl[i]) idx = i;               // the most significant limb that differs
    uint32_t bs = 0, cs = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { r.c[at + 1 + i] = idx == i; if (idx == i) { bs = bl[i]; cs = cl[i]; } }
    r.c[at + 6] = bs; r.c[at + 7] = cs;
    // not_eq_inv = (b_sel - c_sel)^-1 in the field (canonical; the store converts)
    r.c[at + 5] = idx < 0 ? 0u : kb::from_monty(kb::inv(kb::to_monty(bs >= cs ? bs - cs : kb::P - (cs - bs))));
    r.c[at] = bs < cs;
}

enum Chip : int { ADD = 0, ADDI = 1, SUB = 2, ADDW = 3, SUBW = 4, MUL = 5, SHIFT_RIGHT = 6, BRANCH = 7, N_CHIPS = 8 };
__host__ __device__ constexpr int width_of(int chip) {
    return chip == ADD || chip == SUB ? 33 : chip == ADDI ? 30 : chip == ADDW ? 36 : chip == SUBW ? 32 : chip == MUL ? 82 : chip == SHIFT_RIGHT ? 69 : 45;
}

template <int CHIP> __device__ __forceinline__ void fill_row(Row<width_of(CHIP)>& r, const Ev& e) {
    constexpr int W = width_of(CHIP);
    const uint32_t op = (uint32_t)e.ops & 0xff;
    fill_state(r, e);
    if constexpr (CHIP == ADD || CHIP == SUB) {               // state | RTypeReader | value[4] | is_real
        fill_r(r, e);
        r.limbs4(28, e.a);
        r.c[32] = 1;
    } else if constexpr (CHIP == ADDI) {                      // state | ITypeReader | value[4] | is_real
        fill_i(r, e);
        r.limbs4(25, e.a);
        r.c[29] = 1;
    } else if constexpr (CHIP == ADDW) {                      // state | ALUTypeReader | value[2] | msb | is_real
        fill_alu(r, e);
        r.c[32] = e.a & M16; r.c[33] = (e.a >> 16) & M16; r.c[34] = (uint32_t)(e.a >> 31) & 1; r.c[35] = 1;
    } else if constexpr (CHIP == SUBW) {                      // state | RTypeReader | value[2] | msb | is_real
        fill_r(r, e);
        r.c[28] = e.a & M16; r.c[29] = (e.a >> 16) & M16; r.c[30] = (uint32_t)(e.a >> 31) & 1; r.c[31] = 1;
    } else if constexpr (CHIP == MUL) {                       // state | RTypeReader | a[4] | MulOperation | is_mul .. is_mulw
        fill_r(r, e);
        r.limbs4(28, e.a);
        const bool mulh = op == OP_MULH, mulhsu = op == OP_MULHSU, mulw = op == OP_MULW;
        const uint32_t b_msb = (uint32_t)(e.b >> 63), c_msb = (uint32_t)(e.c >> 63);
        const uint32_t bse = (mulh || mulhsu) ? b_msb : 0u, cse = mulh ? c_msb : 0u;
        // 16 x 16 byte product (the operands sign-extended to 128 bits), low 16 bytes with their carries
        uint32_t prod[16];
#pragma unroll
        for (int i = 0; i < 16; i++) prod[i] = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t bi = i < 8 ? (uint32_t)(e.b >> (8 * i)) & 0xff : bse * 0xffu;
#pragma unroll
            for (int j = 0; j < 16 - i; j++) {
                const uint32_t cj = j < 8 ? (uint32_t)(e.c >> (8 * j)) & 0xff : cse * 0xffu;
                prod[i + j] += bi * cj;
            }
        }
        uint32_t carry = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t v = prod[i] + carry;
            carry = v >> 8;
            r.c[32 + i] = carry;                              // mul.carry[i]
            r.c[48 + i] = v & 0xff;                           // mul.product[i]
        }
#pragma unroll
        for (int i = 0; i < 4; i++) { r.c[64 + i] = (uint32_t)(e.b >> (16 * i)) & 0xff; r.c[68 + i] = (uint32_t)(e.c >> (16 * i)) & 0xff; }
        r.c[72] = b_msb; r.c[73] = c_msb;
        r.c[74] = mulw ? (uint32_t)(e.a >> 31) & 1 : 0u;      // product_msb
        r.c[75] = bse; r.c[76] = cse;
        r.c[77] = op == OP_MUL; r.c[78] = mulh; r.c[79] = op == OP_MULHU; r.c[80] = mulhsu; r.c[81] = mulw;
    } else if constexpr (CHIP == SHIFT_RIGHT) {               // alu/sr/mod.rs ShiftRightCols
        fill_alu(r, e);
        const bool sra = op == OP_SRA, srlw = op == OP_SRLW, sraw = op == OP_SRAW, w = srlw || sraw;
        r.limbs4(32, e.a);
        const uint32_t c16 = (uint32_t)e.c & M16;
#pragma unroll
        for (int i = 0; i < 6; i++) r.c[38 + i] = (c16 >> i) & 1;
        const uint32_t amount = ((c16 >> 4) & 1) + (w ? 0u : 2 * ((c16 >> 5) & 1)), s = c16 & 15;
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[60 + i] = amount == (uint32_t)i;
        const uint32_t v = 1u << (16 - s);
        r.c[47] = 1u << (4 - (s & 3)); r.c[46] = 1u << (8 - (s & 7)); r.c[45] = v;
        uint32_t bl[4];
#pragma unroll
        for (int i = 0; i < 4; i++) bl[i] = (uint32_t)(e.b >> (16 * i)) & M16;
        const uint32_t msb = sra ? bl[3] >> 15 : sraw ? bl[1] >> 15 : 0u;
        r.c[36] = msb; r.c[44] = msb * v;
        if (w) bl[2] = bl[3] = 0;
        r.c[37] = w ? (uint32_t)(e.a >> 31) & 1 : 0u;         // srw_msb
        uint32_t lower[4], higher[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { lower[i] = bl[i] & ((1u << s) - 1); higher[i] = bl[i] >> s; r.c[48 + i] = lower[i]; r.c[52 + i] = higher[i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) r.c[56 + i] = higher[i] + (i < 3 ? lower[i + 1] * v : 0u);
        r.c[64] = op == OP_SRL; r.c[65] = sra; r.c[66] = srlw; r.c[67] = sraw;
        r.c[68] = w && (e.ops & F_IMM_C);
    } else {                                                  // BRANCH: state | ITypeReader | next_pc[3] | is_beq .. is_bgeu | is_branching | cmp
        fill_i(r, e);
        r.limbs3(25, e.aux);                                  // next_pc: where execution goes on (pc + 4, or pc + offset when taken)
#pragma unroll
        for (int i = 0; i < 6; i++) r.c[28 + i] = op == OP_BEQ + i;
        const bool is_signed = op == OP_BLT || op == OP_BGE;
        const uint64_t av = e.a_prev, bv = e.b;
        const bool eq = av == bv, lt = is_signed ? (int64_t)av < (int64_t)bv : av < bv;
        r.c[34] = op == OP_BEQ ? eq : op == OP_BNE ? !eq : (op == OP_BLT || op == OP_BLTU) ? lt : !lt;
        fill_lt(r, 35, av, bv, is_signed);
    }
    (void)W;
}

template <int CHIP>
__global__ __launch_bounds__(256) void tracegen_alu_kernel(uint32_t* __restrict__ out, uint32_t height, const Ev* __restrict__ events, uint32_t n) {
    constexpr int W = width_of(CHIP);
    const uint32_t row = blockIdx.x * 256u + threadIdx.x;
    if (row >= height) return;
    Row<W> r;
#pragma unroll
    for (int c = 0; c < W; c++) r.c[c] = 0;
    if (row < n) fill_row<CHIP>(r, events[row]);
    else if (CHIP == SHIFT_RIGHT) { r.c[47] = 16; r.c[46] = 256; r.c[45] = 65536; }       // the padded row template (alu/sr/mod.rs:L165-L171)
#pragma unroll
    for (int c = 0; c < W; c++) gptr(out)[(size_t)c * height + row] = r.c[c] ? kb::to_monty(r.c[c]) : 0u;
}

}  // namespace tg
}  // namespace sp1hip

using namespace sp1hip;

extern "C" {

int sp1hip_tracegen_riscv_alu_width(int chip) { return chip >= 0 && chip < tg::N_CHIPS ? tg::width_of(chip) : -1; }

int sp1hip_tracegen_riscv_alu(int chip, uint32_t* d_table, uint32_t height, const sp1hip_rv64_alu_event_t* d_events, uint32_t n_events,
                              sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(chip >= 0 && chip < tg::N_CHIPS, "unknown chip");
    SP1HIP_REQUIRE(n_events <= height && (d_table || height == 0) && (d_events || n_events == 0), "bad argument");
    if (height == 0) return SP1HIP_SUCCESS;
    hipStream_t s = S(stream);
    const dim3 grid((height + 255) / 256), block(256);
    const tg::Ev* ev = reinterpret_cast<const tg::Ev*>(d_events);
    switch (chip) {
#define SP1HIP_TG(C) case tg::C: hipLaunchKernelGGL(tg::tracegen_alu_kernel<tg::C>, grid, block, 0, s, d_table, height, ev, n_events); break
        SP1HIP_TG(ADD); SP1HIP_TG(ADDI); SP1HIP_TG(SUB); SP1HIP_TG(ADDW); SP1HIP_TG(SUBW); SP1HIP_TG(MUL); SP1HIP_TG(SHIFT_RIGHT); SP1HIP_TG(BRANCH);
#undef SP1HIP_TG
    }
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

}  // extern "C"
