// sp1_amd/csrc/zerocheck.hip — zerocheck sumcheck over AIR constraints on gfx950 (SURVEY §8 a9–a12).
//
//   zc_sum_kernel            `ZerocheckCpuProver::sum_as_poly_in_last_variable` + `increment_y_values`
//                            /root/reference/crates/hypercube/src/prover/zerocheck/sum_as_poly.rs:L53-L181,L355-L440
//                            with `ConstraintSumcheckFolder::assert_zero` (/root/reference/crates/hypercube/src/folder.rs:L276-L323)
//   zc_fix_kernel            `zerocheck_fix_last_variable` -> `mle_fix_last_variable`
//                            /root/reference/crates/hypercube/src/prover/zerocheck/fix_last_variable.rs:L8-L62,
//                            /root/reference/slop/crates/multilinear/src/restrict.rs:L11-L58
//   host driver              `ShardProver::zerocheck` (/root/reference/crates/hypercube/src/prover/shard.rs:L474-L646),
//                            `reduce_sumcheck_to_evaluation` (/root/reference/slop/crates/sumcheck/src/prover.rs:L13-L96),
//                            univariate assembly sum_as_poly.rs:L187-L287, `VirtualGeq`
//                            (/root/reference/slop/crates/multilinear/src/virtual_geq.rs:L12-L99)
//
// Constraints are data: an SSA program per chip (include/sp1hip.h, sp1_amd/air.py). The host
// linear-scan allocates registers; the kernel is a register machine per row pair — one lane = one
// pair of adjacent rows, evaluated at the interpolation nodes t = 0, 2, 4 (leaf = row0 + t (row1 -
// row0)), `acc += alpha_pow[k] * reg` per assert, plus the GKR-opening batching term, times eq(zeta',
// pair), block-reduced to three extension sums. Traces are column-major, so every leaf load of a
// wave is one coalesced 256 B run; the program, alpha/gkr powers and publics are wave-uniform
// scalar loads. Round 0 works on base-field words, later rounds on extension words (4 sub-columns
// per column). The register file lives in per-lane scratch (runtime-indexed); an ahead-of-time
// specialised kernel per chip (registers in VGPRs, no decode) is the planned replacement.
#include <algorithm>
#include <array>
#include <cstring>
#include <memory>
#include <vector>

#include "device_ctx.hpp"

namespace sp1hip {

enum ZcOp : uint32_t { ZC_LOAD_MAIN = 0, ZC_LOAD_PREP = 1, ZC_CONST = 2, ZC_PUBLIC = 3, ZC_ADD = 4, ZC_SUB = 5, ZC_MUL = 6,
                       ZC_NEG = 7, ZC_ASSERT_ZERO = 8 };

struct ZcArgs {
    const uint32_t* prog;        // [n_instr][4]: op, dst, a, b (register-allocated)
    uint32_t n_instr;
    const uint32_t* main;        // column-major; FIRST: [rows x main_w] base, else [rows x 4 main_w]
    const uint32_t* prep;
    uint32_t main_w, prep_w;
    uint32_t rows;               // real rows in the current tables
    const uint32_t* eq;          // ext vector (SoA) of length eq_len
    uint32_t eq_len;
    const uint32_t* alpha_pows;  // [num_constraints][4]
    const uint32_t* gkr_pows;    // [main_w + prep_w][4]
    const uint32_t* publics;     // base words
    uint32_t* partial;           // [gridDim.x][12]
};

// ---- K = base word (round 0) or extension element (later rounds)
template <bool FIRST> struct KT;
template <> struct KT<true> {
    using T = uint32_t;
    static __device__ __forceinline__ T zero() { return 0u; }
    static __device__ __forceinline__ T from_f(uint32_t x) { return x; }
    static __device__ __forceinline__ T add(T a, T b) { return kb::add(a, b); }
    static __device__ __forceinline__ T sub(T a, T b) { return kb::sub(a, b); }
    static __device__ __forceinline__ T mul(T a, T b) { return kb::mul(a, b); }
    static __device__ __forceinline__ kb::Ext scale(const kb::Ext& e, T k) { return kb::ext_mul_base(e, k); }
    static __device__ __forceinline__ kb::Ext to_ext(T k) { return kb::ext_from_base(k); }
    static __device__ __forceinline__ T load(const uint32_t* tbl, uint32_t col, uint32_t rows, uint32_t r) {
        return tbl[(size_t)col * rows + r];
    }
};
template <> struct KT<false> {
    using T = kb::Ext;
    static __device__ __forceinline__ T zero() { return kb::ext_zero(); }
    static __device__ __forceinline__ T from_f(uint32_t x) { return kb::ext_from_base(x); }
    static __device__ __forceinline__ T add(const T& a, const T& b) { return kb::ext_add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return kb::ext_sub(a, b); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return kb::ext_mul(a, b); }
    static __device__ __forceinline__ kb::Ext scale(const kb::Ext& e, const T& k) { return kb::ext_mul(e, k); }
    static __device__ __forceinline__ kb::Ext to_ext(const T& k) { return k; }
    static __device__ __forceinline__ T load(const uint32_t* tbl, uint32_t col, uint32_t rows, uint32_t r) {
        T v;
#pragma unroll
        for (int k = 0; k < 4; k++) v.c[k] = tbl[((size_t)col * 4 + k) * rows + r];
        return v;
    }
};

__device__ __forceinline__ kb::Ext load_ext_aos(const uint32_t* p, uint32_t i) {
    return kb::Ext{{p[4 * i], p[4 * i + 1], p[4 * i + 2], p[4 * i + 3]}};
}

// value of column `col` at node t in {0, 2, 4} for row pair i
template <bool FIRST>
__device__ __forceinline__ typename KT<FIRST>::T leaf(const uint32_t* tbl, uint32_t col, uint32_t rows, uint32_t i, int t) {
    using K = KT<FIRST>;
    typename K::T r0 = K::load(tbl, col, rows, 2 * i);
    if (t == 0) return r0;
    typename K::T r1 = (2 * i + 1 < rows) ? K::load(tbl, col, rows, 2 * i + 1) : K::zero();
    typename K::T slope = K::sub(r1, r0);
    typename K::T s2 = K::add(slope, slope);
    if (t == 2) return K::add(s2, r0);
    return K::add(K::add(s2, s2), r0);
}

template <bool FIRST, int MAXR>
__device__ __forceinline__ kb::Ext run_program(const ZcArgs& a, uint32_t i, int t) {
    using K = KT<FIRST>;
    typename K::T reg[MAXR];
    kb::Ext acc = kb::ext_zero();
    uint32_t ci = 0;
    for (uint32_t k = 0; k < a.n_instr; k++) {
        const uint32_t op = a.prog[4 * k], dst = a.prog[4 * k + 1], x = a.prog[4 * k + 2], y = a.prog[4 * k + 3];
        switch (op) {
            case ZC_LOAD_MAIN: reg[dst] = leaf<FIRST>(a.main, x, a.rows, i, t); break;
            case ZC_LOAD_PREP: reg[dst] = leaf<FIRST>(a.prep, x, a.rows, i, t); break;
            case ZC_CONST: reg[dst] = K::from_f(x); break;               // host pre-converts to Montgomery
            case ZC_PUBLIC: reg[dst] = K::from_f(a.publics[x]); break;
            case ZC_ADD: reg[dst] = K::add(reg[x], reg[y]); break;
            case ZC_SUB: reg[dst] = K::sub(reg[x], reg[y]); break;
            case ZC_MUL: reg[dst] = K::mul(reg[x], reg[y]); break;
            case ZC_NEG: reg[dst] = K::sub(K::zero(), reg[x]); break;
            default: acc = kb::ext_add(acc, K::scale(load_ext_aos(a.alpha_pows, ci++), reg[x])); break;  // ASSERT_ZERO
        }
    }
    return acc;
}

__device__ __forceinline__ uint32_t zc_wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = kb::add(v, __shfl_xor(v, off));
    return v;
}

template <bool FIRST, int MAXR>
__global__ __launch_bounds__(256) void zc_sum_kernel(ZcArgs a) {
    using K = KT<FIRST>;
    __shared__ uint32_t scratch[4 * 12];
    const uint32_t terms = (a.rows + 1) / 2;
    kb::Ext y0 = kb::ext_zero(), y2 = kb::ext_zero(), y4 = kb::ext_zero();
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < terms; i += gridDim.x * 256u) {
        // GKR-opening batching term: sum_j gkr_pow[j] * value_j(t), main columns then preprocessed
        kb::Ext g0 = kb::ext_zero(), g2 = kb::ext_zero();
        for (uint32_t c = 0; c < a.main_w; c++) {
            const kb::Ext pw = load_ext_aos(a.gkr_pows, c);
            g0 = kb::ext_add(g0, K::scale(pw, leaf<FIRST>(a.main, c, a.rows, i, 0)));
            g2 = kb::ext_add(g2, K::scale(pw, leaf<FIRST>(a.main, c, a.rows, i, 2)));
        }
        for (uint32_t c = 0; c < a.prep_w; c++) {
            const kb::Ext pw = load_ext_aos(a.gkr_pows, a.main_w + c);
            g0 = kb::ext_add(g0, K::scale(pw, leaf<FIRST>(a.prep, c, a.rows, i, 0)));
            g2 = kb::ext_add(g2, K::scale(pw, leaf<FIRST>(a.prep, c, a.rows, i, 2)));
        }
        const kb::Ext g4 = kb::ext_sub(kb::ext_add(g2, g2), g0);
        kb::Ext a0 = g0;
        if (!FIRST) a0 = kb::ext_add(a0, run_program<FIRST, MAXR>(a, i, 0));   // round 0: constraints vanish at t = 0
        const kb::Ext a2 = kb::ext_add(run_program<FIRST, MAXR>(a, i, 2), g2);
        const kb::Ext a4 = kb::ext_add(run_program<FIRST, MAXR>(a, i, 4), g4);
        kb::Ext e;
#pragma unroll
        for (int k = 0; k < 4; k++) e.c[k] = a.eq[(size_t)k * a.eq_len + i];
        y0 = kb::ext_add(y0, kb::ext_mul(a0, e));
        y2 = kb::ext_add(y2, kb::ext_mul(a2, e));
        y4 = kb::ext_add(y4, kb::ext_mul(a4, e));
    }
    uint32_t v[12];
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = y0.c[k]; v[4 + k] = y2.c[k]; v[8 + k] = y4.c[k]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = zc_wave_sum(v[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 12; k++) scratch[wave * 12 + k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        const uint32_t k = threadIdx.x;
        a.partial[blockIdx.x * 12 + k] = kb::add(kb::add(scratch[k], scratch[12 + k]), kb::add(scratch[24 + k], scratch[36 + k]));
    }
}

// out[0..12) = summed partials; out[12..16) = eq[th] (zero if th is outside the table)
__global__ void zc_sum_partials_kernel(const uint32_t* __restrict__ partial, uint32_t n_blocks,
                                       const uint32_t* __restrict__ eq, uint32_t eq_len, uint32_t th,
                                       uint32_t* __restrict__ out) {
    const uint32_t k = threadIdx.x;
    if (k >= 16) return;
    if (k >= 12) { out[k] = th < eq_len ? eq[(size_t)(k - 12) * eq_len + th] : 0u; return; }
    uint32_t acc = 0;
    for (uint32_t b = 0; b < n_blocks; b++) acc = kb::add(acc, partial[b * 12 + k]);
    out[k] = acc;
}

// out[i][c] = x + alpha (y - x), x = row 2i, y = row 2i + 1 (zero beyond the real rows); out is an ext table
template <bool FIRST>
__global__ __launch_bounds__(256) void zc_fix_kernel(const uint32_t* __restrict__ in, uint32_t rows, uint32_t width,
                                                     kb::Ext alpha, uint32_t* __restrict__ out) {
    using K = KT<FIRST>;
    const uint32_t out_rows = (rows + 1) / 2;
    const size_t t = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= (size_t)out_rows * width) return;
    const uint32_t c = (uint32_t)(t / out_rows), i = (uint32_t)(t % out_rows);
    typename K::T x = K::load(in, c, rows, 2 * i);
    typename K::T y = (2 * i + 1 < rows) ? K::load(in, c, rows, 2 * i + 1) : K::zero();
    const kb::Ext r = kb::ext_add(K::scale(alpha, K::sub(y, x)), K::to_ext(x));
#pragma unroll
    for (int k = 0; k < 4; k++) out[((size_t)c * 4 + k) * out_rows + i] = r.c[k];
}

// ------------------------------------------------------------------------------------------ host side
using Ext = kb::Ext;
static Ext operator+(const Ext& a, const Ext& b) { return kb::ext_add(a, b); }
static Ext operator-(const Ext& a, const Ext& b) { return kb::ext_sub(a, b); }
static Ext operator*(const Ext& a, const Ext& b) { return kb::ext_mul(a, b); }
static Ext ext_c(uint32_t canonical) { return kb::ext_from_base(kb::to_monty(canonical)); }

using UniPoly = std::vector<Ext>;
static Ext uni_eval(const UniPoly& p, const Ext& x) {
    Ext acc = kb::ext_zero();
    for (size_t i = p.size(); i-- > 0;) acc = acc * x + p[i];
    return acc;
}
static UniPoly uni_add(const UniPoly& a, const UniPoly& b) {
    UniPoly r(std::max(a.size(), b.size()), kb::ext_zero());
    for (size_t i = 0; i < r.size(); i++) r[i] = (i < a.size() ? a[i] : kb::ext_zero()) + (i < b.size() ? b[i] : kb::ext_zero());
    return r;
}
static UniPoly uni_scale(UniPoly a, const Ext& k) { for (auto& c : a) c = c * k; return a; }
// Lagrange interpolation, same operation order as slop_algebra::interpolate_univariate_polynomial
static UniPoly interpolate(const std::vector<Ext>& xs, const std::vector<Ext>& ys) {
    UniPoly result{kb::ext_zero()};
    for (size_t i = 0; i < xs.size(); i++) {
        Ext den = kb::ext_one();
        UniPoly num{ys[i]};
        for (size_t j = 0; j < xs.size(); j++) {
            if (j == i) continue;
            den = den * (xs[i] - xs[j]);
            UniPoly shifted{kb::ext_zero()};
            shifted.insert(shifted.end(), num.begin(), num.end());
            num = uni_add(shifted, uni_scale(num, kb::ext_zero() - xs[j]));
        }
        result = uni_add(result, uni_scale(num, kb::ext_inv(den)));
    }
    return result;
}

struct VGeq {
    uint32_t threshold;
    Ext geq_c, eq_c;
    VGeq fix(const Ext& alpha) const {
        VGeq r;
        r.threshold = threshold >> 1;
        r.geq_c = geq_c;
        r.eq_c = (threshold & 1) == 0 ? (kb::ext_one() - alpha) * eq_c : alpha * (eq_c + geq_c) - geq_c;
        return r;
    }
    Ext at(size_t idx) const {
        if (idx < threshold) return kb::ext_zero();
        if (idx == threshold) return eq_c + geq_c;
        return geq_c;
    }
};

struct DevBuf {
    void* p = nullptr;
    hipStream_t s = nullptr;
    size_t n = 0;
    int alloc(size_t bytes, hipStream_t stream) {
        s = stream;
        n = bytes;
        return arena_alloc(&p, bytes, stream);
    }
    void release() { arena_free(p, n, s); p = nullptr; }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    uint32_t* u32() const { return (uint32_t*)p; }
};

struct ChipState {
    const sp1hip_zc_chip_t* in;
    std::vector<uint32_t> prog;     // allocated [n][4]
    uint32_t n_regs = 1;
    std::vector<Ext> alpha_pows, gkr_pows;
    DevBuf d_prog, d_alpha, d_gkr, d_partial, d_sums;
    std::unique_ptr<DevBuf> main_buf, prep_buf;   // ext tables of later rounds
    const uint32_t* d_main = nullptr;
    const uint32_t* d_prep = nullptr;
    uint64_t rows = 0;
    uint32_t num_vars = 0;
    Ext eq_adj, pad_adj;
    VGeq vgeq;
    UniPoly uni;
};

// linear-scan register allocation of the SSA program (host)
static int allocate_registers(const uint32_t* ssa, uint32_t n, std::vector<uint32_t>* out, uint32_t* n_regs) {
    std::vector<int> last_use(n, -1);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        SP1HIP_REQUIRE(op <= ZC_ASSERT_ZERO, "bad opcode in constraint program");
        if (op == ZC_ADD || op == ZC_SUB || op == ZC_MUL) {
            SP1HIP_REQUIRE(a < k && b < k, "constraint program is not in SSA order");
            last_use[a] = (int)k;
            last_use[b] = (int)k;
        } else if (op == ZC_NEG || op == ZC_ASSERT_ZERO) {
            SP1HIP_REQUIRE(a < k, "constraint program is not in SSA order");
            last_use[a] = (int)k;
        }
    }
    std::vector<uint32_t> free_regs, reg_of(n, 0xffffffffu);
    uint32_t regs = 0;
    out->assign((size_t)n * 4, 0);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = ssa[3 * k], a = ssa[3 * k + 1], b = ssa[3 * k + 2];
        const bool bin = op == ZC_ADD || op == ZC_SUB || op == ZC_MUL, un = op == ZC_NEG || op == ZC_ASSERT_ZERO;
        uint32_t ra = a, rb = b;
        if (bin || un) ra = reg_of[a];
        if (bin) rb = reg_of[b];
        if (bin || un) {
            if (last_use[a] == (int)k && reg_of[a] != 0xffffffffu) { free_regs.push_back(reg_of[a]); reg_of[a] = 0xffffffffu; }
            if (bin && b != a && last_use[b] == (int)k && reg_of[b] != 0xffffffffu) { free_regs.push_back(reg_of[b]); reg_of[b] = 0xffffffffu; }
        }
        uint32_t dst = 0;
        if (op != ZC_ASSERT_ZERO) {
            if (!free_regs.empty()) { dst = free_regs.back(); free_regs.pop_back(); }
            else dst = regs++;
            if (last_use[k] >= 0) reg_of[k] = dst;
            else free_regs.push_back(dst);      // dead value
        }
        uint32_t* o = out->data() + (size_t)k * 4;
        o[0] = op; o[1] = dst; o[2] = ra; o[3] = rb;
        if (op == ZC_CONST) o[2] = kb::to_monty(a % kb::P);
    }
    *n_regs = regs ? regs : 1;
    return SP1HIP_SUCCESS;
}

// host evaluation of the program on an all-zero row (padded_row_adjustment, shard.rs:L524-L536)
static Ext eval_zero_row(const ChipState& c, const uint32_t* publics) {
    const uint32_t n = (uint32_t)(c.prog.size() / 4);
    std::vector<uint32_t> reg(c.n_regs, 0);
    Ext acc = kb::ext_zero();
    uint32_t ci = 0;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t op = c.prog[4 * k], dst = c.prog[4 * k + 1], x = c.prog[4 * k + 2], y = c.prog[4 * k + 3];
        switch (op) {
            case ZC_LOAD_MAIN: case ZC_LOAD_PREP: reg[dst] = 0; break;
            case ZC_CONST: reg[dst] = x; break;
            case ZC_PUBLIC: reg[dst] = publics[x]; break;
            case ZC_ADD: reg[dst] = kb::add(reg[x], reg[y]); break;
            case ZC_SUB: reg[dst] = kb::sub(reg[x], reg[y]); break;
            case ZC_MUL: reg[dst] = kb::mul(reg[x], reg[y]); break;
            case ZC_NEG: reg[dst] = kb::neg(reg[x]); break;
            default: acc = acc + kb::ext_mul_base(c.alpha_pows[ci++], reg[x]); break;
        }
    }
    return acc;
}

template <bool FIRST>
static int launch_sum(const ZcArgs& a, uint32_t n_regs, uint32_t blocks, hipStream_t s) {
    if (n_regs <= 16) hipLaunchKernelGGL((zc_sum_kernel<FIRST, 16>), dim3(blocks), dim3(256), 0, s, a);
    else if (n_regs <= 64) hipLaunchKernelGGL((zc_sum_kernel<FIRST, 64>), dim3(blocks), dim3(256), 0, s, a);
    else if (n_regs <= 256) hipLaunchKernelGGL((zc_sum_kernel<FIRST, 256>), dim3(blocks), dim3(256), 0, s, a);
    else if (n_regs <= 1024) hipLaunchKernelGGL((zc_sum_kernel<FIRST, 1024>), dim3(blocks), dim3(256), 0, s, a);
    else { set_error("constraint program needs %u live registers (max 1024)", n_regs); return SP1HIP_ERROR_INVALID_ARGUMENT; }
    SP1HIP_LAUNCH_CHECK();
    return SP1HIP_SUCCESS;
}

struct ByteOut {
    std::vector<uint8_t> b;
    void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
    void ext(const Ext& e) {
        for (int k = 0; k < 4; k++) { uint32_t c = kb::from_monty(e.c[k]); for (int i = 0; i < 4; i++) b.push_back((uint8_t)(c >> (8 * i))); }
    }
};

}  // namespace sp1hip

using namespace sp1hip;

struct sp1hip_challenger_s;
namespace sp1hip {
// transcript hooks implemented in prover.hip
void challenger_observe(sp1hip_challenger_t* ch, uint32_t x);
kb::Ext challenger_sample_ext(sp1hip_challenger_t* ch);
void challenger_restore(sp1hip_challenger_t* dst, const sp1hip_challenger_t* src);
}

static int zerocheck_prove_impl(const sp1hip_zc_chip_t* chips, int n_chips, int max_log_row_count,
                               const sp1hip_ext_t* h_zeta, const sp1hip_ext_t* h_openings, sp1hip_ext_t alpha_c,
                               sp1hip_ext_t gkr_c, const uint32_t* h_publics, int n_publics,
                               sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len,
                               sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(chips && n_chips > 0 && h_zeta && h_openings && challenger && proof_len, "null argument");
    SP1HIP_REQUIRE(max_log_row_count >= 1 && max_log_row_count <= 30, "max_log_row_count out of range");
    SP1HIP_REQUIRE(h_publics || n_publics == 0, "null publics");
    const int L = max_log_row_count;
    size_t total_w = 0;
    for (int i = 0; i < n_chips; i++) {
        SP1HIP_REQUIRE(chips[i].program && chips[i].n_instr > 0, "empty constraint program");
        SP1HIP_REQUIRE(chips[i].real_rows <= ((uint64_t)1 << L), "chip taller than 2^max_log_row_count");
        SP1HIP_REQUIRE(chips[i].real_rows == 0 || (chips[i].d_main || chips[i].main_width == 0), "null main trace");
        SP1HIP_REQUIRE(chips[i].real_rows == 0 || (chips[i].d_prep || chips[i].prep_width == 0), "null preprocessed trace");
        total_w += chips[i].main_width + chips[i].prep_width;
    }
    const size_t need = 8 + (size_t)L * (8 + 80) + 16 + 8 + (size_t)L * 16 + 16 + 8 + (size_t)n_chips * 8 + total_w * 16;
    if (!h_proof || *proof_len < need) {
        *proof_len = need;
        set_error("sp1hip_zerocheck_prove: proof buffer too small, need %zu bytes", need);
        return SP1HIP_ERROR_BUFFER_TOO_SMALL;
    }
    hipStream_t s = S(stream);
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    const Ext alpha{{alpha_c.c[0], alpha_c.c[1], alpha_c.c[2], alpha_c.c[3]}};
    const Ext gkr{{gkr_c.c[0], gkr_c.c[1], gkr_c.c[2], gkr_c.c[3]}};
    std::vector<uint32_t> publics(h_publics, h_publics + n_publics);
    DevBuf d_publics;
    SP1HIP_TRY(d_publics.alloc((size_t)n_publics * 4, s));
    if (n_publics) SP1HIP_HIP(hipMemcpyAsync(d_publics.p, publics.data(), (size_t)n_publics * 4, hipMemcpyHostToDevice, s));

    int max_constraints = 0;
    for (int i = 0; i < n_chips; i++) max_constraints = std::max<int>(max_constraints, chips[i].num_constraints);
    std::vector<Ext> pows(max_constraints);
    { Ext cur = kb::ext_one(); for (auto& x : pows) { x = cur; cur = cur * alpha; } }

    std::vector<std::unique_ptr<ChipState>> st;
    std::vector<Ext> claims;
    size_t oo = 0;
    for (int i = 0; i < n_chips; i++) {
        std::unique_ptr<ChipState> c(new ChipState());
        c->in = &chips[i];
        SP1HIP_TRY(allocate_registers(chips[i].program, chips[i].n_instr, &c->prog, &c->n_regs));
        uint32_t asserts = 0;
        for (uint32_t k = 0; k < chips[i].n_instr; k++) {
            const uint32_t op = chips[i].program[3 * k], a = chips[i].program[3 * k + 1];
            if (op == ZC_ASSERT_ZERO) asserts++;
            if (op == ZC_LOAD_MAIN) SP1HIP_REQUIRE(a < chips[i].main_width, "main column out of range");
            if (op == ZC_LOAD_PREP) SP1HIP_REQUIRE(a < chips[i].prep_width, "preprocessed column out of range");
            if (op == ZC_PUBLIC) SP1HIP_REQUIRE((int)a < n_publics, "public value index out of range");
        }
        SP1HIP_REQUIRE(asserts == chips[i].num_constraints, "num_constraints does not match the program");
        // [alpha^(n-1), ..., alpha, 1] so that the folder matches the verifier's Horner order
        c->alpha_pows.assign(pows.begin(), pows.begin() + chips[i].num_constraints);
        std::reverse(c->alpha_pows.begin(), c->alpha_pows.end());
        { Ext cur = gkr; for (uint32_t k = 0; k < chips[i].main_width + chips[i].prep_width; k++) { c->gkr_pows.push_back(cur); cur = cur * gkr; } }
        c->pad_adj = eval_zero_row(*c, publics.data());
        Ext claim = kb::ext_zero();
        for (uint32_t k = 0; k < chips[i].main_width + chips[i].prep_width; k++, oo++) {
            const Ext o{{h_openings[oo].c[0], h_openings[oo].c[1], h_openings[oo].c[2], h_openings[oo].c[3]}};
            claim = claim + o * c->gkr_pows[k];
        }
        claims.push_back(claim);
        c->rows = chips[i].real_rows;
        c->num_vars = (uint32_t)L;
        c->eq_adj = kb::ext_one();
        c->vgeq = VGeq{(uint32_t)chips[i].real_rows, kb::ext_one(), kb::ext_zero()};
        c->d_main = chips[i].d_main;
        c->d_prep = chips[i].d_prep;
        SP1HIP_TRY(c->d_prog.alloc(c->prog.size() * 4, s));
        SP1HIP_TRY(c->d_alpha.alloc(c->alpha_pows.size() * 16, s));
        SP1HIP_TRY(c->d_gkr.alloc(c->gkr_pows.size() * 16, s));
        SP1HIP_TRY(c->d_partial.alloc(1024 * 12 * 4, s));
        SP1HIP_TRY(c->d_sums.alloc(16 * 4, s));
        SP1HIP_HIP(hipMemcpyAsync(c->d_prog.p, c->prog.data(), c->prog.size() * 4, hipMemcpyHostToDevice, s));
        if (!c->alpha_pows.empty())
            SP1HIP_HIP(hipMemcpyAsync(c->d_alpha.p, c->alpha_pows.data(), c->alpha_pows.size() * 16, hipMemcpyHostToDevice, s));
        if (!c->gkr_pows.empty())
            SP1HIP_HIP(hipMemcpyAsync(c->d_gkr.p, c->gkr_pows.data(), c->gkr_pows.size() * 16, hipMemcpyHostToDevice, s));
        st.push_back(std::move(c));
    }
    SP1HIP_HIP(hipStreamSynchronize(s));   // host staging vectors above may now be reused

    std::vector<Ext> zeta(L);
    memcpy(zeta.data(), h_zeta, (size_t)L * 16);
    const Ext lambda = challenger_sample_ext(challenger);
    DevBuf d_eq;
    SP1HIP_TRY(d_eq.alloc(((size_t)1 << (L - 1)) * 16, s));
    std::vector<UniPoly> msgs;
    std::vector<Ext> point;   // [alpha_last, ..., alpha_first]
    std::vector<Ext> round_claims = claims;
    std::vector<std::array<uint32_t, 16>> sums(n_chips);
    for (int r = 0; r < L; r++) {
        const int nv = L - r;                       // variables left
        const Ext last = zeta[nv - 1];
        // eq(zeta[0 .. nv-1), .) is shared by every chip with real rows
        SP1HIP_TRY(sp1hip_partial_lagrange(reinterpret_cast<const sp1hip_ext_t*>(zeta.data()), nv - 1, d_eq.u32(), stream));
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            if (c.rows == 0) continue;
            const uint32_t terms = (uint32_t)((c.rows + 1) / 2);
            uint32_t blocks = (terms + 255) / 256;
            if (blocks > 1024) blocks = 1024;
            ZcArgs a{};
            a.prog = c.d_prog.u32(); a.n_instr = (uint32_t)(c.prog.size() / 4);
            a.main = c.d_main; a.prep = c.d_prep; a.main_w = c.in->main_width; a.prep_w = c.in->prep_width;
            a.rows = (uint32_t)c.rows; a.eq = d_eq.u32(); a.eq_len = 1u << (nv - 1);
            a.alpha_pows = c.d_alpha.u32(); a.gkr_pows = c.d_gkr.u32(); a.publics = d_publics.u32(); a.partial = c.d_partial.u32();
            if (r == 0) SP1HIP_TRY(launch_sum<true>(a, c.n_regs, blocks, s));
            else SP1HIP_TRY(launch_sum<false>(a, c.n_regs, blocks, s));
            hipLaunchKernelGGL(zc_sum_partials_kernel, dim3(1), dim3(64), 0, s, c.d_partial.u32(), blocks, d_eq.u32(),
                               1u << (nv - 1), terms - 1, c.d_sums.u32());
            SP1HIP_LAUNCH_CHECK();
            SP1HIP_HIP(hipMemcpyAsync(sums[i].data(), c.d_sums.p, 64, hipMemcpyDeviceToHost, s));
        }
        SP1HIP_HIP(hipStreamSynchronize(s));
        // ---- univariate messages (sum_as_poly.rs:L187-L287)
        std::vector<UniPoly> uni(n_chips);
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            if (c.rows == 0) { uni[i] = UniPoly(5, kb::ext_zero()); continue; }
            const size_t th = (size_t)((c.rows + 1) / 2) - 1;
            const Ext eq_th{{sums[i][12], sums[i][13], sums[i][14], sums[i][15]}};
            const Ext msb = c.eq_adj * eq_th;
            const Ext y0s{{sums[i][0], sums[i][1], sums[i][2], sums[i][3]}}, y2s{{sums[i][4], sums[i][5], sums[i][6], sums[i][7]}},
                y4s{{sums[i][8], sums[i][9], sums[i][10], sums[i][11]}};
            const Ext two = ext_c(2), four = ext_c(4), three = ext_c(3), seven = ext_c(7);
            const Ext v0 = c.vgeq.fix(kb::ext_zero()).at(th), v2 = c.vgeq.fix(two).at(th), v4 = c.vgeq.fix(four).at(th);
            const Ext f0 = kb::ext_one() - last;
            const Ext y0 = y0s * (f0 * c.eq_adj) - c.pad_adj * v0 * msb * f0;
            const Ext f2 = last * three - kb::ext_one();
            const Ext y2 = y2s * (f2 * c.eq_adj) - c.pad_adj * v2 * msb * f2;
            const Ext f4 = last * seven - three;
            const Ext y4 = y4s * (f4 * c.eq_adj) - c.pad_adj * v4 * msb * f4;
            const Ext b = (kb::ext_one() - last) * kb::ext_inv(kb::ext_one() - (last + last));
            uni[i] = interpolate({kb::ext_zero(), kb::ext_one(), two, four, b},
                                 {y0, round_claims[i] - y0, y2, y4, kb::ext_zero()});
        }
        UniPoly rlc{kb::ext_zero()};
        for (auto& u : uni) rlc = uni_add(uni_scale(rlc, lambda), u);
        for (auto& cf : rlc)
            for (int k = 0; k < 4; k++) challenger_observe(challenger, cf.c[k]);
        msgs.push_back(rlc);
        const Ext a_r = challenger_sample_ext(challenger);
        point.insert(point.begin(), a_r);
        for (int i = 0; i < n_chips; i++) {
            round_claims[i] = uni_eval(uni[i], a_r);
            st[i]->uni = uni[i];
        }
        // ---- fix the last variable of every table (fix_last_variable.rs)
        for (int i = 0; i < n_chips; i++) {
            ChipState& c = *st[i];
            c.vgeq = c.vgeq.fix(a_r);
            if (c.rows == 0) continue;
            const uint64_t out_rows = (c.rows + 1) / 2;
            auto fix_table = [&](const uint32_t* in, uint32_t width, std::unique_ptr<DevBuf>& holder, const uint32_t** cur) -> int {
                if (width == 0) return SP1HIP_SUCCESS;
                std::unique_ptr<DevBuf> nb(new DevBuf());
                SP1HIP_TRY(nb->alloc((size_t)out_rows * width * 16, s));
                const size_t total = (size_t)out_rows * width;
                if (r == 0) hipLaunchKernelGGL(zc_fix_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, (uint32_t)c.rows, width, a_r, nb->u32());
                else hipLaunchKernelGGL(zc_fix_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, (uint32_t)c.rows, width, a_r, nb->u32());
                SP1HIP_LAUNCH_CHECK();
                holder = std::move(nb);    // the previous round's table is released stream-ordered
                *cur = holder->u32();
                return SP1HIP_SUCCESS;
            };
            SP1HIP_TRY(fix_table(c.d_main, c.in->main_width, c.main_buf, &c.d_main));
            SP1HIP_TRY(fix_table(c.d_prep, c.in->prep_width, c.prep_buf, &c.d_prep));
            c.eq_adj = c.eq_adj * (a_r * last + (kb::ext_one() - a_r) * (kb::ext_one() - last));
            c.rows = out_rows;
        }
    }
    // ---- proof: PartialSumcheckProof + per-chip component evaluations (prep then main)
    ByteOut w;
    w.u64((uint64_t)L);
    for (auto& m : msgs) { w.u64(m.size()); for (auto& cf : m) w.ext(cf); }
    Ext claimed = kb::ext_zero(), final_eval = kb::ext_zero();
    for (auto& cl : claims) claimed = claimed * lambda + cl;
    for (int i = 0; i < n_chips; i++) final_eval = final_eval * lambda + uni_eval(st[i]->uni, point.front());
    w.ext(claimed);
    w.u64(point.size());
    for (auto& x : point) w.ext(x);
    w.ext(final_eval);
    w.u64((uint64_t)n_chips);
    std::vector<std::vector<Ext>> chip_evals(n_chips);
    for (int i = 0; i < n_chips; i++) {
        ChipState& c = *st[i];
        const uint32_t wp = c.in->prep_width, wm = c.in->main_width;
        std::vector<uint32_t> hp((size_t)wp * 4, 0), hm((size_t)wm * 4, 0);
        if (c.rows) {   // one row left: ext table [1 x w] = w*4 words, column-major == (col, coord)
            if (wp) SP1HIP_HIP(hipMemcpyAsync(hp.data(), c.d_prep, hp.size() * 4, hipMemcpyDeviceToHost, s));
            if (wm) SP1HIP_HIP(hipMemcpyAsync(hm.data(), c.d_main, hm.size() * 4, hipMemcpyDeviceToHost, s));
            SP1HIP_HIP(hipStreamSynchronize(s));
        }
        for (uint32_t k = 0; k < wp; k++) chip_evals[i].push_back(Ext{{hp[4 * k], hp[4 * k + 1], hp[4 * k + 2], hp[4 * k + 3]}});
        for (uint32_t k = 0; k < wm; k++) chip_evals[i].push_back(Ext{{hm[4 * k], hm[4 * k + 1], hm[4 * k + 2], hm[4 * k + 3]}});
        w.u64(chip_evals[i].size());
        for (auto& e : chip_evals[i]) w.ext(e);
    }
    // observe the openings (shard.rs:L609-L640)
    challenger_observe(challenger, kb::to_monty((uint32_t)n_chips));
    for (int i = 0; i < n_chips; i++) {
        const uint32_t wp = st[i]->in->prep_width, wm = st[i]->in->main_width;
        challenger_observe(challenger, kb::to_monty(wp));
        for (uint32_t k = 0; k < wp; k++) for (int q = 0; q < 4; q++) challenger_observe(challenger, chip_evals[i][k].c[q]);
        challenger_observe(challenger, kb::to_monty(wm));
        for (uint32_t k = 0; k < wm; k++) for (int q = 0; q < 4; q++) challenger_observe(challenger, chip_evals[i][wp + k].c[q]);
    }
    if (w.b.size() != need) { set_error("internal error: zerocheck proof size %zu != %zu", w.b.size(), need); return SP1HIP_ERROR_RUNTIME; }
    memcpy(h_proof, w.b.data(), need);
    *proof_len = need;
    return SP1HIP_SUCCESS;
}

extern "C" int sp1hip_zerocheck_prove(const sp1hip_zc_chip_t* chips, int n_chips, int max_log_row_count,
                                      const sp1hip_ext_t* h_zeta, const sp1hip_ext_t* h_openings, sp1hip_ext_t alpha,
                                      sp1hip_ext_t gkr_batch, const uint32_t* h_publics, int n_publics,
                                      sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len,
                                      sp1hip_stream_t stream) {
    // the caller's transcript only advances if the proof is produced
    sp1hip_challenger_t* backup = nullptr;
    if (challenger) SP1HIP_TRY(sp1hip_challenger_clone(challenger, &backup));
    const int st = zerocheck_prove_impl(chips, n_chips, max_log_row_count, h_zeta, h_openings, alpha, gkr_batch, h_publics,
                                        n_publics, challenger, h_proof, proof_len, stream);
    if (st != SP1HIP_SUCCESS && challenger) challenger_restore(challenger, backup);
    sp1hip_challenger_free(backup);
    return st;
}
