// sp1_amd/csrc/zc_poly.hpp — polynomial identities over byte limbs as ONE fused piece of the zerocheck (hint kind 7).
//
// Reference: `FieldOpCols::eval_with_polynomials` + `eval_field_operation`
// (/root/reference/crates/core/machine/src/operations/field/field_op.rs:L367-L560, util_air.rs:L6-L27): with byte-limb polynomials
// a(x), b(x), result(x), carry(x), the modulus m(x) and a witness w(x), the 2 n - 1 coefficients of
//        V(x) = a(x) b(x) - result(x) - carry(x) m(x) - (w(x) - offset) (x - 2^8)
// are asserted zero one after the other: 63 constraints whose cones hold 1,024 byte products and 1,024 products by constants for
// a 32-limb field — through the interpreter a secp256k1 chip (ten such operations) is a 24,000-instruction program per row pair
// and node. The zerocheck batches constraint k with alpha^(n - 1 - k): consecutive coefficients carry consecutive powers, so
//        sum_k w_k V_k = w_0 V(rho),    rho = 1 / alpha,    w_0 = the power of the first coefficient,
// and V(rho) = A(rho) B(rho) + R(rho) with A, B, R AFFINE in the row (their coefficients over the columns depend on the proof's
// alpha only): a 32-limb multiplication costs ~250 column loads and one extension product instead of 2,000 products. Exact field
// arithmetic: the same element whichever way it is summed, so the proof bytes do not change.
//
// The hint (sp1_amd/air.py, hint_polynomial_identity) names the SSA values A_t[i], B_t[j] and the rest R[k]: "assert k is
// sum_t sum_{i + j = k} A_t[i] B_t[j] + R[k]". The planner extracts the affine form of every named value from the caller's SSA
// (a value that is not affine in the main columns drops the hint: the interpreter keeps those constraints) and checks the
// identity on a pseudo-random row like every other hint. Sums of several products (FieldInnerProductCols, a modulus read from
// memory) are terms t = 0, 1, ...; a term may have a third factor — `eval_variable`'s selectors (is_add, is_sub, is_mul) are
// one-coefficient polynomials, and the product of three polynomials collapses to the product of three affine forms the same way.
//
// Because the three forms are affine, their values at the nodes t = 0, 2, 4 of a row pair (and at the twelve nodes of a row quad in
// the bivariate rounds) follow from their values on the rows themselves: one workgroup loads every column of a row pair ONCE and
// leaves the sums of all three nodes — the three-nodes-per-pass form that lost occupancy for the Keccak pieces is free here.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <vector>

#include "kb31.hpp"

namespace sp1hip {

constexpr uint32_t ZC_HINT_POLY = 7, ZC_HINT_POLY_ARG = 8;
constexpr uint32_t ZC_POLY_MAX_TERMS = 4;
// device table of one identity (words): header [n_terms, n_rest, n_owned, 0, then per term n_0, n_1, n_2 (ZC_POLY_NONE: two factors)]
// (16 words), then segments of 8-word entries [column, 0, 0, 0, coefficient (4 words)]: per term one segment per factor (its constant
// first, column = ZC_POLY_ONE), then the rest (constant first), then the rest's owned columns (their GKR batching term rides along)
constexpr uint32_t ZC_POLY_HDR = 16, ZC_POLY_ENTRY = 8, ZC_POLY_ONE = 0xffffffffu, ZC_POLY_NONE = 0xffffffffu;

struct ZcLinForm {                                       // sum coefs[k] * main[cols[k]] + c0 (Montgomery words)
    std::vector<uint32_t> cols, coefs;
    uint32_t c0 = 0;
};
struct ZcPolyTerm { std::vector<ZcLinForm> f[3]; };      // two or three factors (f[2] empty: two)
struct ZcPoly {
    uint32_t first_constraint = 0, n_c = 0;
    std::vector<ZcPolyTerm> terms;
    std::vector<ZcLinForm> rest;
    std::vector<uint32_t> owned;                         // sorted: columns nothing else reads (this piece carries their GKR term)
};

// the affine forms of the SSA values `ids` over the main columns; false if one of them is not affine in main columns alone.
// ssa: [n][3] triples (HINT pseudo-instructions already turned into constants).
inline bool zc_poly_extract(const uint32_t* ssa, uint32_t n, const std::vector<uint32_t>& ids, std::vector<ZcLinForm>* out) {
    std::vector<uint8_t> need(n, 0);
    std::vector<uint32_t> stack;
    for (uint32_t id : ids) { if (id >= n) return false; stack.push_back(id); }
    while (!stack.empty()) {
        const uint32_t v = stack.back();
        stack.pop_back();
        if (need[v]) continue;
        need[v] = 1;
        const uint32_t op = ssa[3 * v];
        if (op == 4 || op == 5 || op == 6) { if (ssa[3 * v + 1] >= v || ssa[3 * v + 2] >= v) return false; stack.push_back(ssa[3 * v + 1]); stack.push_back(ssa[3 * v + 2]); }
        else if (op == 7) { if (ssa[3 * v + 1] >= v) return false; stack.push_back(ssa[3 * v + 1]); }
        else if (op != 0 && op != 2) return false;       // preprocessed columns, public values, asserts: not a main-affine value
    }
    typedef std::map<uint32_t, uint32_t> Terms;
    struct Form { Terms t; uint32_t c0 = 0; };
    std::map<uint32_t, Form> val;
    for (uint32_t v = 0; v < n; v++) {
        if (!need[v]) continue;
        const uint32_t op = ssa[3 * v], a = ssa[3 * v + 1], b = ssa[3 * v + 2];
        Form f;
        if (op == 0) f.t[a] = kb::to_monty(1u);
        else if (op == 2) f.c0 = kb::to_monty(a % kb::P);
        else if (op == 4 || op == 5) {
            f = val[a];
            const Form& g = val[b];
            for (auto& kv : g.t) {
                uint32_t& slot = f.t[kv.first];
                slot = op == 4 ? kb::add(slot, kv.second) : kb::sub(slot, kv.second);
            }
            f.c0 = op == 4 ? kb::add(f.c0, g.c0) : kb::sub(f.c0, g.c0);
        } else if (op == 7) {
            const Form& g = val[a];
            for (auto& kv : g.t) f.t[kv.first] = kb::neg(kv.second);
            f.c0 = kb::neg(g.c0);
        } else {                                         // MUL: one side must be a constant
            const Form *x = &val[a], *y = &val[b];
            if (!x->t.empty()) std::swap(x, y);
            if (!x->t.empty()) return false;
            const uint32_t k = x->c0;
            for (auto& kv : y->t) f.t[kv.first] = kb::mul(kv.second, k);
            f.c0 = kb::mul(y->c0, k);
        }
        for (auto it = f.t.begin(); it != f.t.end();) it = it->second == 0 ? f.t.erase(it) : std::next(it);
        val[v] = std::move(f);
    }
    out->clear();
    for (uint32_t id : ids) {
        const Form& f = val[id];
        ZcLinForm lf;
        lf.c0 = f.c0;
        for (auto& kv : f.t) { lf.cols.push_back(kv.first); lf.coefs.push_back(kv.second); }
        out->push_back(std::move(lf));
    }
    return true;
}

inline uint32_t zc_lin_eval(const ZcLinForm& f, const uint32_t* main_row) {
    uint32_t v = f.c0;
    for (size_t k = 0; k < f.cols.size(); k++) v = kb::add(v, kb::mul(f.coefs[k], main_row[f.cols[k]]));
    return v;
}

// host model: the constraints of the identity on one row of base-field words, coefficient by coefficient
template <class Sink>
inline void zc_poly_eval_row(const ZcPoly& p, const uint32_t* main_row, Sink&& sink) {
    std::vector<uint32_t> v(p.n_c, 0u);
    for (uint32_t k = 0; k < p.n_c; k++) v[k] = zc_lin_eval(p.rest[k], main_row);
    for (const ZcPolyTerm& t : p.terms) {
        std::vector<uint32_t> fv[3];
        for (int f = 0; f < 3; f++) for (const ZcLinForm& lf : t.f[f]) fv[f].push_back(zc_lin_eval(lf, main_row));
        if (fv[2].empty()) fv[2].push_back(kb::to_monty(1u));
        for (size_t i = 0; i < fv[0].size(); i++)
            for (size_t j = 0; j < fv[1].size(); j++)
                for (size_t l = 0; l < fv[2].size(); l++)
                    v[i + j + l] = kb::add(v[i + j + l], kb::mul(kb::mul(fv[0][i], fv[1][j]), fv[2][l]));
    }
    for (uint32_t k = 0; k < p.n_c; k++) sink(k, v[k]);
}

// One segment of the device table, prepared when the plan is made: its columns (slot 0 = the constant) and, per coefficient of
// every form in it, (index of the form = index of its weight, slot, coefficient) — the per-proof work is one multiply-add per entry.
struct ZcPolySeg {
    struct E { uint32_t i, slot, coef; };
    std::vector<uint32_t> cols;                          // slot s >= 1 is column cols[s - 1]
    std::vector<E> ent;
    bool with_const = true;
};
inline ZcPolySeg zc_poly_seg(const std::vector<ZcLinForm>& forms, const std::vector<uint32_t>& owned, bool skip_owned, bool only_owned) {
    ZcPolySeg sg;
    sg.with_const = !only_owned;
    std::map<uint32_t, uint32_t> slot_of;
    for (const ZcLinForm& f : forms)
        for (uint32_t c : f.cols) {
            const bool own = std::binary_search(owned.begin(), owned.end(), c);
            if ((own && skip_owned) || (!own && only_owned)) continue;
            slot_of.emplace(c, 0u);
        }
    for (auto& kv : slot_of) { sg.cols.push_back(kv.first); kv.second = (uint32_t)sg.cols.size(); }
    for (size_t i = 0; i < forms.size(); i++) {
        const ZcLinForm& f = forms[i];
        if (sg.with_const && f.c0) sg.ent.push_back({(uint32_t)i, 0u, f.c0});
        for (size_t k = 0; k < f.cols.size(); k++) {
            auto it = slot_of.find(f.cols[k]);
            if (it != slot_of.end()) sg.ent.push_back({(uint32_t)i, it->second, f.coefs[k]});
        }
    }
    return sg;
}
// (the segments of an identity, in table order: per term its three factors — an absent third one is an empty segment —, the rest,
// the rest's owned columns)
inline std::vector<ZcPolySeg> zc_poly_segments(const ZcPoly& p) {
    std::vector<ZcPolySeg> out;
    for (const ZcPolyTerm& t : p.terms) for (int f = 0; f < 3; f++) out.push_back(zc_poly_seg(t.f[f], p.owned, false, false));
    out.push_back(zc_poly_seg(p.rest, p.owned, true, false));
    out.push_back(zc_poly_seg(p.rest, p.owned, false, true));
    return out;
}

// the device table of one identity for this proof's alpha. w[k] = the batching power of constraint first_constraint + k (k < n_c),
// rho = 1 / alpha (so that w[i + j] = w[i] rho^j): A and the rest are weighted by w, B by the powers of rho. Appends to `blob`.
inline void zc_poly_table(const ZcPoly& p, const std::vector<ZcPolySeg>& segs, const kb::Ext* w, const kb::Ext& rho, std::vector<uint32_t>* blob,
                          std::vector<kb::Ext>* scratch) {
    std::vector<kb::Ext>& rp = scratch[0];
    std::vector<kb::Ext>& acc = scratch[1];
    rp.resize(p.n_c);
    { kb::Ext cur = kb::ext_one(); for (auto& x : rp) { x = cur; cur = kb::ext_mul(cur, rho); } }
    const size_t hdr = blob->size();
    blob->resize(hdr + ZC_POLY_HDR, 0u);
    (*blob)[hdr + 0] = (uint32_t)p.terms.size();
    for (size_t si = 0; si < segs.size(); si++) {
        const ZcPolySeg& sg = segs[si];
        const bool in_term = si < 3 * p.terms.size();
        if (in_term && si % 3 == 2 && p.terms[si / 3].f[2].empty()) { (*blob)[hdr + 4 + si] = ZC_POLY_NONE; continue; }
        const kb::Ext* weight = in_term && si % 3 != 0 ? rp.data() : w;               // (the first factor carries the batching powers)
        acc.assign(sg.cols.size() + 1, kb::ext_zero());
        for (const ZcPolySeg::E& e : sg.ent) acc[e.slot] = kb::ext_add(acc[e.slot], kb::ext_mul_base(weight[e.i], e.coef));
        auto push = [&](uint32_t col, const kb::Ext& c) { const uint32_t wds[8] = {col, 0u, 0u, 0u, c.c[0], c.c[1], c.c[2], c.c[3]}; blob->insert(blob->end(), wds, wds + 8); };
        if (sg.with_const) push(ZC_POLY_ONE, acc[0]);
        for (size_t k = 0; k < sg.cols.size(); k++) push(sg.cols[k], acc[k + 1]);
        const uint32_t n = (uint32_t)sg.cols.size();                                  // (counts exclude the constant entry)
        if (in_term) (*blob)[hdr + 4 + si] = n;
        else (*blob)[hdr + 1 + (si - 3 * p.terms.size())] = n;
    }
}

}  // namespace sp1hip
