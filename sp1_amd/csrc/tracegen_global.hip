// sp1_amd/csrc/tracegen_global.hip — trace generation ON THE DEVICE for the RISC-V machine's Global chip.
//
// The Global chip is a third of a core shard's cells: one row per global interaction event with a Poseidon2 permutation of the
// message, the lift of its digest onto the septic curve, and the running sum of the lifted points. The reference generates it on
// the GPU (/root/reference/sp1-gpu/crates/sys/lib/tracegen/riscv/global.cu:L1-L252: a decompress kernel, a scan of curve points,
// a finalize kernel; CPU definition crates/core/machine/src/global/mod.rs:L131-L260, operations/global_interaction.rs:L33-L103,
// operations/global_accumulation.rs:L28-L54, hypercube/src/septic_{extension,curve,digest}.rs). This is the MI355X counterpart:
//
//   global_lift_kernel     one lane per event: for offset = 0, 1, ...: m = message (kind folded into word 0, offset into word 7),
//                          h = Poseidon2(m), x = h[0..7] in F_p^7 = F_p[z] / (z^7 - 3 z - 5); if x^3 + 45 x + 41 z^3 is a square
//                          whose root y has y_6 (or -y_6) in [1, 63 2^24], the point is (x, +-y) — a receive takes the root in
//                          that range, a send its negative — and the row gets message, flags, limbs, x, y, offset, the byte
//                          decomposition of the range check and the 179 columns of the permutation (`populate_perm`).
//                          Square roots take the norm route (septic.py: n^((r+1)/2) / sqrt_p(N(n)), Frobenius maps from a table
//                          COMPUTED on the host at first use, Tonelli-Shanks in F_p).
//   global_chunk_sum / global_scan_totals / global_finalize
//                          cumulative_sum[i] = start + P_0 + ... + P_i by a three-phase scan over the group law (16 points per
//                          lane, one workgroup over the chunk totals), initial_digest[i] = cumulative_sum[i - 1] (start for i = 0);
//                          padding rows: the dummy point, start and start + dummy, the permutation of the zero state
//                          (global/mod.rs:L213-L236).
//
// Events: `GlobalInteractionEvent { message: [u32; 8], is_receive: bool, kind: u8 }` as 9 u32 words per event (the Rust struct's
// `#[repr(C)]` image: word 8 = is_receive | kind << 8). Output: the column-major [241][height] table of Montgomery words the
// prover consumes (column order: sp1_amd/machines/riscv.py::global_chip). Bit-identical to the host trace of
// sp1_amd/machines/{riscv_trace.py, septic.py} (tests/test_gpu_tracegen_global.py).
#include <memory>
#include <mutex>
#include <unordered_map>

#include "device_ctx.hpp"
#include "poseidon2.hpp"

namespace sp1hip {
namespace {

struct S7 { uint32_t c[7]; };                       // an element of F_p^7, Montgomery words

KB_HD S7 s7_zero() { S7 r; for (int i = 0; i < 7; i++) r.c[i] = 0; return r; }
KB_HD S7 s7_add(const S7& a, const S7& b) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = kb::add(a.c[i], b.c[i]); return r; }
KB_HD S7 s7_sub(const S7& a, const S7& b) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = kb::sub(a.c[i], b.c[i]); return r; }
KB_HD S7 s7_neg(const S7& a) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = kb::neg(a.c[i]); return r; }
KB_HD S7 s7_scale(const S7& a, uint32_t k) { S7 r; for (int i = 0; i < 7; i++) r.c[i] = kb::mul(a.c[i], k); return r; }
// a b mod z^7 - 3 z - 5 (septic_extension.rs:L307-L325)
KB_HD S7 s7_mul(const S7& a, const S7& b) {
    uint32_t t[13];
    for (int s = 0; s < 13; s++) t[s] = 0;
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) t[i + j] = kb::add(t[i + j], kb::mul(a.c[i], b.c[j]));
    const uint32_t c5 = kb::to_monty(5), c3 = kb::to_monty(3);
    S7 r;
    for (int i = 0; i < 7; i++) r.c[i] = t[i];
    for (int s = 7; s < 13; s++) {
        r.c[s - 7] = kb::add(r.c[s - 7], kb::mul(t[s], c5));
        r.c[s - 6] = kb::add(r.c[s - 6], kb::mul(t[s], c3));
    }
    return r;
}
KB_HD uint32_t fp_pow(uint32_t x, uint32_t e) {
    uint32_t r = kb::R1;
    for (; e; e >>= 1) { if (e & 1) r = kb::mul(r, x); x = kb::mul(x, x); }
    return r;
}
KB_HD uint32_t fp_inv(uint32_t x) { return fp_pow(x, kb::P - 2); }

// frob[k - 1][i] = (z^i)^(p^k), k = 1..6 (computed on the host: z^p by square-and-multiply)
struct FrobTables { S7 f[6][7]; };
KB_HD S7 s7_frob(const FrobTables& ft, const S7& a, int k) {
    S7 r = s7_zero();
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) r.c[j] = kb::add(r.c[j], kb::mul(a.c[i], ft.f[k - 1][i].c[j]));
    return r;
}
// (a^(r - 1), N(a)) with r = 1 + p + ... + p^6
KB_HD void s7_norm_parts(const FrobTables& ft, const S7& a, S7* t_out, uint32_t* n_out) {
    S7 t = s7_mul(s7_frob(ft, a, 1), s7_frob(ft, a, 2));            // a^(p + p^2)
    t = s7_mul(s7_mul(t, s7_frob(ft, t, 2)), s7_frob(ft, t, 4));     // ^(1 + p^2 + p^4)
    *t_out = t;
    *n_out = s7_mul(t, a).c[0];
}
KB_HD S7 s7_inv(const FrobTables& ft, const S7& a) {
    S7 t; uint32_t n;
    s7_norm_parts(ft, a, &t, &n);
    return s7_scale(t, fp_inv(n));
}
// Tonelli-Shanks in F_p, p - 1 = 2^24 127 (a root for quadratic residues)
KB_HD uint32_t fp_sqrt(uint32_t a) {
    const int S = 24;
    const uint32_t Q = 127;
    uint32_t c = fp_pow(kb::to_monty(3), Q), t = fp_pow(a, Q), r = fp_pow(a, (Q + 1) / 2);
    for (int i = 1; i < S; i++) {
        uint32_t b = t;
        for (int k = 0; k < S - 1 - i; k++) b = kb::mul(b, b);
        const bool m = b != kb::R1;
        if (m) r = kb::mul(r, c);
        c = kb::mul(c, c);
        if (m) t = kb::mul(t, c);
    }
    return r;
}
// a square root of n if it has one (ok): n^((r + 1) / 2) / sqrt_p(N(n)), (r + 1) / 2 = 1 + p ((p + 1) / 2) (1 + p^2 + p^4)
KB_HD bool s7_sqrt(const FrobTables& ft, const S7& n, S7* root) {
    S7 t; uint32_t nn;
    s7_norm_parts(ft, n, &t, &nn);
    if (fp_pow(nn, (kb::P - 1) / 2) != kb::R1) return false;
    S7 w = n, sq = n;
    for (int i = 1; i < 30; i++) {                                    // w = n^((p + 1) / 2) = n^(1 + 2^23 + ... + 2^29)
        sq = s7_mul(sq, sq);
        if (i >= 23) w = s7_mul(w, sq);
    }
    const S7 cand = s7_mul(s7_mul(s7_mul(s7_frob(ft, w, 1), s7_frob(ft, w, 3)), s7_frob(ft, w, 5)), n);
    *root = s7_scale(cand, fp_inv(fp_sqrt(nn)));
    return true;
}
KB_HD S7 s7_curve_rhs(const S7& x) {                                  // x^3 + 45 x + 41 z^3 (septic_curve.rs:L101-L113)
    S7 r = s7_add(s7_mul(s7_mul(x, x), x), s7_scale(x, kb::to_monty(45)));
    r.c[3] = kb::add(r.c[3], kb::to_monty(41));
    return r;
}
struct Pt { S7 x, y; };
KB_HD Pt ec_add(const FrobTables& ft, const Pt& p1, const Pt& p2) {  // add_incomplete (septic_curve.rs:L58-L63)
    const S7 slope = s7_mul(s7_sub(p2.y, p1.y), s7_inv(ft, s7_sub(p2.x, p1.x)));
    Pt r;
    r.x = s7_sub(s7_sub(s7_mul(slope, slope), p1.x), p2.x);
    r.y = s7_sub(s7_mul(slope, s7_sub(p1.x, r.x)), p1.y);
    return r;
}

// column offsets of the Global chip (riscv.py::global_chip)
constexpr int G_MESSAGE = 0, G_KIND = 8, G_M0_16 = 9, G_M0_8 = 10, G_X = 11, G_Y = 18, G_PERM = 25, G_OFFSET = 204, G_Y6 = 205,
              G_IS_REAL = 209, G_IS_RECV = 210, G_IS_SEND = 211, G_INDEX = 212, G_INIT_X = 213, G_INIT_Y = 220, G_CUM_X = 227,
              G_CUM_Y = 234, G_WIDTH = 241;
constexpr uint32_t Y6_LIMIT = 63u << 24;
// CURVE_CUMULATIVE_SUM_START / CURVE_WITNESS_DUMMY_POINT (hypercube/src/septic_curve.rs:L170-L196), canonical
constexpr uint32_t START_X[7] = {21053971, 90322736, 156256384, 23627556, 34171264, 126183783, 25645942};
constexpr uint32_t START_Y[7] = {2020310104, 1513506566, 1843922297, 2003644209, 805967281, 1882435203, 1623804682};
constexpr uint32_t DUMMY_X[7] = {57770625, 136856976, 72496438, 2651249, 55731748, 158816036, 118044313};
constexpr uint32_t DUMMY_Y[7] = {1250555984, 1592495468, 656721246, 420301347, 2125819749, 819876460, 17687681};

struct GlobalCtx {
    FrobTables* d_ft = nullptr;
    Pt start, dummy, start_plus_dummy;          // Montgomery
};

// `populate_perm` into the 179 permutation columns of row r (operations/poseidon2/trace.rs:L29-L152; the same walk as
// tracegen.hip's Poseidon2Wide kernel); returns the output state in s
__device__ void perm_rows(uint32_t (&s)[16], const p2::RoundConstants* rc, uint32_t* trace, uint64_t height, uint64_t r, bool store) {
    auto put = [&](int col, uint32_t v) { if (store) trace[(size_t)(G_PERM + col) * height + r] = v; };
    int col = 0;
#pragma unroll 1
    for (int round = 0; round < 8; round++) {
        for (int i = 0; i < 16; i++) put(col + i, s[i]);
        col += 16;
        if (round == 0) p2::external_linear(s);
        for (int i = 0; i < 16; i++) s[i] = p2::sbox(s[i], rc->ext[round][i]);
        p2::external_linear(s);
        if (round == 3) {
            for (int i = 0; i < 16; i++) put(128 + i, s[i]);
#pragma unroll 1
            for (int k = 0; k < 20; k++) {
                s[0] = p2::sbox(s[0], rc->internal[k]);
                p2::internal_linear_lazy(s);
                if (k < 19) put(144 + k, s[0]);
            }
            for (int i = 1; i < 16; i++) s[i] = kb::umin(s[i], s[i] - kb::P);
        }
    }
    for (int i = 0; i < 16; i++) put(163 + i, s[i]);
}

__global__ __launch_bounds__(256) void global_lift_kernel(uint32_t* __restrict__ trace, uint64_t height, const uint32_t* __restrict__ events,
                                                          uint64_t n_events, const p2::RoundConstants* __restrict__ rc,
                                                          const FrobTables* __restrict__ ftp, Pt dummy, Pt* __restrict__ points,
                                                          uint32_t* __restrict__ failed) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= height) return;
    auto put = [&](int col, uint32_t v) { trace[(size_t)col * height + r] = v; };
    if (r >= n_events) {                                              // padding row (global/mod.rs:L213-L236)
        for (int c = 0; c < G_WIDTH; c++) put(c, 0u);
        uint32_t s[16];
        for (int i = 0; i < 16; i++) s[i] = 0;
        perm_rows(s, rc, trace, height, r, true);
        for (int i = 0; i < 7; i++) { put(G_X + i, dummy.x.c[i]); put(G_Y + i, dummy.y.c[i]); }
        return;
    }
    const FrobTables& ft = *ftp;
    const uint32_t* ev = events + r * 9;
    const bool is_recv = (ev[8] & 0xffu) != 0;
    const uint32_t kind = (ev[8] >> 8) & 0xffu;
    uint32_t m[16];
    for (int i = 0; i < 8; i++) m[i] = kb::to_monty(ev[i] % kb::P);
    for (int i = 8; i < 16; i++) m[i] = 0;
    m[0] = kb::add(m[0], kb::to_monty((kind << 24) % kb::P));
    for (uint32_t offset = 0; offset < 256; offset++) {
        uint32_t s[16];
        for (int i = 0; i < 16; i++) s[i] = m[i];
        s[7] = kb::add(s[7], kb::to_monty(offset << 16));
        uint32_t s_in[16];
        for (int i = 0; i < 16; i++) s_in[i] = s[i];
        perm_rows(s, rc, trace, height, r, false);
        S7 x;
        for (int i = 0; i < 7; i++) x.c[i] = s[i];
        S7 root;
        if (!s7_sqrt(ft, s7_curve_rhs(x), &root)) continue;
        const uint32_t y6 = kb::from_monty(root.c[6]);
        const uint32_t neg6 = y6 ? kb::P - y6 : 0u;
        const bool pos_ok = y6 >= 1 && y6 <= Y6_LIMIT, neg_ok = neg6 >= 1 && neg6 <= Y6_LIMIT;
        if (!pos_ok && !neg_ok) continue;                             // the exception: neither root's last coordinate is in range
        S7 y = pos_ok ? root : s7_neg(root);                          // the "receive" root
        if (!is_recv) y = s7_neg(y);                                  // a send carries the negated point
        for (int c = 0; c < 8; c++) put(G_MESSAGE + c, kb::to_monty(ev[c] % kb::P));
        put(G_KIND, kb::to_monty(kind));
        put(G_M0_16, kb::to_monty(ev[0] & 0xffffu));
        put(G_M0_8, kb::to_monty((ev[0] >> 16) & 0xffu));
        for (int i = 0; i < 7; i++) { put(G_X + i, x.c[i]); put(G_Y + i, y.c[i]); }
        perm_rows(s_in, rc, trace, height, r, true);
        put(G_OFFSET, kb::to_monty(offset));
        const uint32_t yc6 = kb::from_monty(y.c[6]);
        const uint32_t rcv = is_recv ? yc6 - 1 : kb::P - yc6 - 1;
        for (int k = 0; k < 4; k++) put(G_Y6 + k, kb::to_monty((rcv >> (8 * k)) & 0xffu));
        put(G_IS_REAL, kb::R1);
        put(G_IS_RECV, is_recv ? kb::R1 : 0u);
        put(G_IS_SEND, is_recv ? 0u : kb::R1);
        put(G_INDEX, kb::to_monty((uint32_t)r));
        points[r] = Pt{x, y};
        return;
    }
    atomicAdd(failed, 1u);                                            // no curve point within 256 offsets (probability ~2^-256)
}

constexpr uint32_t G_CHUNK = 16;                                      // points one lane sums sequentially

// totals[c] = Q_{16 c} + ... (Q_0 = start + P_0, Q_i = P_i)
__global__ __launch_bounds__(256) void global_chunk_sum_kernel(const Pt* __restrict__ points, uint64_t n, Pt start, const FrobTables* __restrict__ ftp,
                                                               Pt* __restrict__ totals) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t lo = c * G_CHUNK;
    if (lo >= n) return;
    const FrobTables& ft = *ftp;
    Pt acc = points[lo];
    if (lo == 0) acc = ec_add(ft, start, acc);
    for (uint64_t i = lo + 1; i < lo + G_CHUNK && i < n; i++) acc = ec_add(ft, acc, points[i]);
    totals[c] = acc;
}

// inclusive scan of the chunk totals in place, ONE workgroup of 1024 lanes: serial over a lane's slice, Hillis-Steele over the
// slice totals in LDS, slice prefixes applied
__global__ __launch_bounds__(1024) void global_scan_totals_kernel(Pt* __restrict__ totals, uint64_t n_chunks, const FrobTables* __restrict__ ftp) {
    __shared__ Pt sh[1024];
    const FrobTables& ft = *ftp;
    const uint32_t t = threadIdx.x;
    const uint64_t per = (n_chunks + 1023) / 1024;
    const uint64_t lo = (uint64_t)t * per, hi = lo + per < n_chunks ? lo + per : n_chunks;
    const bool has = lo < n_chunks;
    Pt acc;
    if (has) {
        acc = totals[lo];
        for (uint64_t i = lo + 1; i < hi; i++) { acc = ec_add(ft, acc, totals[i]); totals[i] = acc; }
        sh[t] = acc;
    }
    __syncthreads();
    const uint32_t live = (uint32_t)((n_chunks + per - 1) / per);      // lanes with a slice
    for (uint32_t d = 1; d < live; d <<= 1) {
        Pt v;
        const bool take = has && t >= d;
        if (take) v = ec_add(ft, sh[t - d], sh[t]);
        __syncthreads();
        if (take) sh[t] = v;
        __syncthreads();
    }
    if (has && t > 0) {
        const Pt pre = sh[t - 1];                                     // everything before this lane's slice
        for (uint64_t i = lo; i < hi; i++) totals[i] = ec_add(ft, pre, totals[i]);
    }
}

__global__ __launch_bounds__(256) void global_finalize_kernel(uint32_t* __restrict__ trace, uint64_t height, const Pt* __restrict__ points, uint64_t n,
                                                              const Pt* __restrict__ totals, Pt start, Pt start_plus_dummy,
                                                              const FrobTables* __restrict__ ftp) {
    const uint64_t c = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t lo = c * G_CHUNK;
    if (lo >= height) return;
    const FrobTables& ft = *ftp;
    auto put_pt = [&](uint64_t r, int cx, int cy, const Pt& p) {
        for (int i = 0; i < 7; i++) { trace[(size_t)(cx + i) * height + r] = p.x.c[i]; trace[(size_t)(cy + i) * height + r] = p.y.c[i]; }
    };
    Pt prev = (c == 0 || lo >= n) ? start : totals[c - 1];            // cumulative sum before this chunk (padding-only chunks never use it)
    for (uint64_t r = lo; r < lo + G_CHUNK && r < height; r++) {
        if (r < n) {
            const Pt cur = ec_add(ft, prev, points[r]);
            put_pt(r, G_INIT_X, G_INIT_Y, prev);
            put_pt(r, G_CUM_X, G_CUM_Y, cur);
            prev = cur;
        } else {
            put_pt(r, G_INIT_X, G_INIT_Y, start);
            put_pt(r, G_CUM_X, G_CUM_Y, start_plus_dummy);
        }
    }
}

int get_global_ctx(const GlobalCtx** out) {
    static std::mutex g;
    static std::unordered_map<int, GlobalCtx*> per_device;
    int dev = 0;
    SP1HIP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g);
    GlobalCtx*& c = per_device[dev];
    if (!c) {
        std::unique_ptr<GlobalCtx> fresh(new GlobalCtx());
        // z^p, then row i of the k-th Frobenius power = (z^i)^(p^k): the linear map composed k times
        FrobTables ft;
        S7 z = s7_zero(), zp = s7_zero();
        z.c[1] = kb::R1;
        zp.c[0] = kb::R1;
        {
            S7 b = z;
            for (uint32_t e = kb::P; e; e >>= 1) { if (e & 1) zp = s7_mul(zp, b); b = s7_mul(b, b); }
        }
        S7 pw = s7_zero();
        pw.c[0] = kb::R1;
        for (int i = 0; i < 7; i++) { ft.f[0][i] = pw; pw = s7_mul(pw, zp); }
        for (int k = 1; k < 6; k++)
            for (int i = 0; i < 7; i++) {                              // frob^(k+1)(z^i) = frob(frob^k(z^i)): apply the k = 1 map to the previous row
                S7 r = s7_zero();
                for (int a = 0; a < 7; a++)
                    for (int j = 0; j < 7; j++) r.c[j] = kb::add(r.c[j], kb::mul(ft.f[k - 1][i].c[a], ft.f[0][a].c[j]));
                ft.f[k][i] = r;
            }
        for (int i = 0; i < 7; i++) {
            fresh->start.x.c[i] = kb::to_monty(START_X[i]); fresh->start.y.c[i] = kb::to_monty(START_Y[i]);
            fresh->dummy.x.c[i] = kb::to_monty(DUMMY_X[i]); fresh->dummy.y.c[i] = kb::to_monty(DUMMY_Y[i]);
        }
        fresh->start_plus_dummy = ec_add(ft, fresh->start, fresh->dummy);
        SP1HIP_HIP(hipMalloc((void**)&fresh->d_ft, sizeof ft));
        const hipError_t e = hipMemcpy(fresh->d_ft, &ft, sizeof ft, hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(fresh->d_ft); return map_hip_error(e, "uploading the Frobenius tables"); }
        c = fresh.release();
    }
    *out = c;
    return SP1HIP_SUCCESS;
}

}  // namespace
}  // namespace sp1hip

using namespace sp1hip;

extern "C" int sp1hip_tracegen_riscv_global(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                            sp1hip_stream_t stream) {
    SP1HIP_REQUIRE(d_trace && (d_events || n_events == 0), "null buffer");
    SP1HIP_REQUIRE(n_events <= height && height < ((uint64_t)1 << 31), "more events than rows, or a table taller than 2^31");
    if (height == 0) return SP1HIP_SUCCESS;
    hipStream_t s = S(stream);
    const DeviceCtx* ctx;
    SP1HIP_TRY(get_device_ctx(&ctx));
    const GlobalCtx* g;
    SP1HIP_TRY(get_global_ctx(&g));
    const uint64_t n_chunks = (n_events + G_CHUNK - 1) / G_CHUNK;
    Pt *d_points = nullptr, *d_totals = nullptr;
    uint32_t* d_failed = nullptr;
    const size_t pts_bytes = std::max<uint64_t>(n_events, 1) * sizeof(Pt), tot_bytes = std::max<uint64_t>(n_chunks, 1) * sizeof(Pt);
    SP1HIP_TRY(arena_alloc((void**)&d_points, pts_bytes, s));
    struct Free { void* p; size_t b; hipStream_t s; ~Free() { arena_free(p, b, s); } } f1{d_points, pts_bytes, s};
    SP1HIP_TRY(arena_alloc((void**)&d_totals, tot_bytes, s));
    Free f2{d_totals, tot_bytes, s};
    SP1HIP_TRY(arena_alloc((void**)&d_failed, 256, s));
    Free f3{d_failed, 256, s};
    SP1HIP_HIP(hipMemsetAsync(d_failed, 0, 4, s));
    hipLaunchKernelGGL(global_lift_kernel, dim3((unsigned)((height + 255) / 256)), dim3(256), 0, s, d_trace, height, d_events, n_events, ctx->d_rc,
                       g->d_ft, g->dummy, d_points, d_failed);
    SP1HIP_LAUNCH_CHECK();
    if (n_events) {
        hipLaunchKernelGGL(global_chunk_sum_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, s, d_points, n_events, g->start, g->d_ft, d_totals);
        SP1HIP_LAUNCH_CHECK();
        hipLaunchKernelGGL(global_scan_totals_kernel, dim3(1), dim3(1024), 0, s, d_totals, n_chunks, g->d_ft);
        SP1HIP_LAUNCH_CHECK();
    }
    const uint64_t fin_chunks = (height + G_CHUNK - 1) / G_CHUNK;
    hipLaunchKernelGGL(global_finalize_kernel, dim3((unsigned)((fin_chunks + 255) / 256)), dim3(256), 0, s, d_trace, height, d_points, n_events, d_totals,
                       g->start, g->start_plus_dummy, g->d_ft);
    SP1HIP_LAUNCH_CHECK();
    uint32_t failed = 0;
    SP1HIP_HIP(hipMemcpyAsync(&failed, d_failed, 4, hipMemcpyDeviceToHost, s));
    SP1HIP_HIP(hipStreamSynchronize(s));
    SP1HIP_REQUIRE(failed == 0, "an event has no curve point within 256 offsets");
    return SP1HIP_SUCCESS;
}
