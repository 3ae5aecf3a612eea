// sp1_amd/csrc/zc_mul.hpp — the product constraints of `MulOperation` as a fused piece of the zerocheck (hint kind 6).
//
// Reference: /root/reference/crates/core/machine/src/operations/mul.rs:L196-L236 (inside `MulOperation::eval`): with b, c cut into
// 8 bytes each and sign-extended to 16 (`b_sign_extend * 0xff`, `c_sign_extend * 0xff`), m[k] = sum_{i + j = k} b[i] c[j] for
// k < 16, and for every k
//        is_real * (product[k] - (m[k] + carry[k - 1] - 256 carry[k])) = 0.
// These 16 constraints are 136 byte products per row (81 distinct ones) with all 16 bytes of both operands live throughout: in
// the interpreter they are 70 % of the Mul chip's program and the reason its register file (38 extension values per lane) leaves
// ONE wave per SIMD. Here the bytes live in VGPRs, the sign-extension terms collapse to two prefix sums
//        m[k] = sum_{i + j = k; i, j < 8} b[i] c[j] + 255 c_sign_extend (b[0] + .. + b[k - 8]) + 255 b_sign_extend (c[0] + .. + c[k - 8])
// (no (i, j) with both >= 8 has i + j < 16), and nothing is decoded. Everything else of the chip — booleans, the limbs of `a`,
// CPU state, the register adapter — stays with the interpreter, whose program for the chip shrinks from 380 to ~120 instructions.
//
// Columns, relative to the first column of the `MulOperation` struct (hint operand `base_col`): carry[16] 0, product[16] 16,
// b_lower_byte.low_bytes[4] 32, c_lower_byte.low_bytes[4] 36, b_msb 40, c_msb 41, product_msb 42, b_sign_extend 43,
// c_sign_extend 44, then the chip's five opcode flags 45..49 (is_real = their sum: alu/mul/mod.rs:L208-L213). The 16-bit limbs of
// b are four columns at `aux0` (the adapter's op_b value), those of c seven columns further (RTypeReader: r_type.rs:L33-L41).
// Piece q of 2 evaluates the constraints k = q (mod 2): 32 byte products and 8 sign-extension products each. The planner checks the hint against the caller's SSA
// on a pseudo-random row, so a chip with another layout is rejected, not mis-proved.
#pragma once
#include "zc_poseidon2.hpp"

namespace sp1hip {

constexpr uint32_t ZC_HINT_MUL = 6;
constexpr uint32_t ZC_MUL_CONSTRAINTS = 16, ZC_MUL_PIECES = 2, ZC_MUL_OWNED = 16;      // owned: the carry columns (nothing else reads them)
constexpr uint32_t MUL_CARRY = 0, MUL_PRODUCT = 16, MUL_B_LOW = 32, MUL_C_LOW = 36, MUL_B_SE = 43, MUL_C_SE = 44, MUL_FLAGS = 45, MUL_COLUMNS = 50;
constexpr uint32_t MUL_OPC_FROM_OPB = 7;
constexpr uint32_t MUL_INV256 = 0x7e810001u;                                           // 256^-1 mod p (canonical)

template <class F, class LD, class LDX, class SINK>
KB_HD void zc_mul_piece(uint32_t q, LD&& ld, LDX&& ldx, SINK&& sink) {
    using T = typename F::T;
    const uint32_t inv256 = kb::to_monty(MUL_INV256), c255 = kb::to_monty(255u), c256 = kb::to_monty(256u);
    T xb[8], yb[8];
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {                                                   // u16_to_u8_unsafe (u16_operation.rs:L42-L56)
        const T xl = ld(MUL_B_LOW + i, false), yl = ld(MUL_C_LOW + i, false);
        xb[2 * i] = xl; xb[2 * i + 1] = F::mulc(F::sub(ldx(i, false), xl), inv256);
        yb[2 * i] = yl; yb[2 * i + 1] = F::mulc(F::sub(ldx(MUL_OPC_FROM_OPB + i, false), yl), inv256);
    }
    const T se_b = F::mulc(ld(MUL_B_SE, false), c255), se_c = F::mulc(ld(MUL_C_SE, false), c255);
    T is_real = ld(MUL_FLAGS, false);
#pragma unroll
    for (uint32_t i = 1; i < 5; i++) is_real = F::add(is_real, ld(MUL_FLAGS + i, false));
    // prefix sums of the bytes for the sign-extension terms of k >= 8: px = b[0] + .. + b[k - 8], py likewise
    T px = xb[0], py = yb[0];
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
        if (k > 8) { px = F::add(px, xb[k - 8]); py = F::add(py, yb[k - 8]); }
        if ((k & 1u) != q) continue;
        // k < 8: the byte products b[0..k] c[k..0]; k >= 8: the sign-extension terms, then b[k - 7..7] c[7..k - 7] (none for k = 15)
        const uint32_t i0 = k > 7 ? k - 7 : 0, i1 = k < 7 ? k : 7;
        T m = k >= 8 ? F::add(F::mul(se_c, px), F::mul(se_b, py)) : F::mul(xb[0], yb[k]);
#pragma unroll
        for (uint32_t i = (k >= 8 ? i0 : 1u); i <= i1; i++) m = F::add(m, F::mul(xb[i], yb[k - i]));
        // product[k] - m - carry[k - 1] + 256 carry[k]
        T v = F::add(F::sub(ld(MUL_PRODUCT + k, false), m), F::mulc(ld(MUL_CARRY + k, true), c256));
        if (k > 0) v = F::sub(v, ld(MUL_CARRY + k - 1, false));
        sink(k, F::mul(is_real, v));
    }
}

}  // namespace sp1hip
