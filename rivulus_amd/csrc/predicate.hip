// Predicate normalisation: rv_predicate -> AND list / CNF literals / composed BooleanArray.
// One unit of the backend library behind include/rivulus_gpu.h (gfx950 only; compiled with hipcc).  Shared helpers and the
// functions the units call across each other are declared in launch.hpp (namespace rvl).
#include "launch.hpp"

using namespace rvh;
using namespace rvl;

namespace rvl {

struct ExprNode {
    int kind;  // 0 term, 1 and, 2 or, 3 not
    int a, b;
    uint32_t term;
};
struct Literal {
    uint32_t term;
    bool neg;
    bool operator==(const Literal &o) const { return term == o.term && neg == o.neg; }
};
using Clause = std::vector<Literal>;

// postfix program -> tree (nodes in evaluation order, the root last)
std::vector<ExprNode> parse_expression(const uint8_t *expr, uint32_t n_expr, uint32_t n_terms) {
    require(n_expr >= 1 && n_expr <= 255, RV_ERR_INVALID_ARG, "predicate expression: 1..255 postfix entries");
    std::vector<ExprNode> nodes;
    std::vector<int> stack;
    for (uint32_t i = 0; i < n_expr; ++i) {
        const uint8_t op = expr[i];
        if (op < 0x80) {
            require(op < n_terms, RV_ERR_INVALID_ARG, fmt("predicate expression: entry %u pushes term %u of %u", i, op, n_terms));
            nodes.push_back(ExprNode{0, -1, -1, op});
        } else if (op == RV_EXPR_NOT) {
            require(!stack.empty(), RV_ERR_INVALID_ARG, fmt("predicate expression: NOT at entry %u has no operand", i));
            const int a = stack.back();
            stack.pop_back();
            nodes.push_back(ExprNode{3, a, -1, 0});
        } else {
            require(op == RV_EXPR_AND || op == RV_EXPR_OR, RV_ERR_INVALID_ARG, fmt("predicate expression: unknown entry 0x%02x", op));
            require(stack.size() >= 2, RV_ERR_INVALID_ARG, fmt("predicate expression: operator at entry %u has fewer than two operands", i));
            const int b = stack.back();
            stack.pop_back();
            const int a = stack.back();
            stack.pop_back();
            nodes.push_back(ExprNode{op == RV_EXPR_AND ? 1 : 2, a, b, 0});
        }
        stack.push_back(static_cast<int>(nodes.size()) - 1);
    }
    require(stack.size() == 1, RV_ERR_INVALID_ARG, "predicate expression must leave exactly one value");
    return nodes;
}

// conjunctive normal form of node `i` (negated when neg); false when it outgrows `cap` literals
bool cnf_of(const std::vector<ExprNode> &nodes, int i, bool neg, size_t cap, std::vector<Clause> &out) {
    const ExprNode &n = nodes[i];
    if (n.kind == 0) {
        out.push_back(Clause{Literal{n.term, neg}});
        return true;
    }
    if (n.kind == 3) return cnf_of(nodes, n.a, !neg, cap, out);
    std::vector<Clause> A, B;
    if (!cnf_of(nodes, n.a, neg, cap, A) || !cnf_of(nodes, n.b, neg, cap, B)) return false;
    const bool conj = (n.kind == 1) != neg;  // De Morgan: NOT(a AND b) = NOT a OR NOT b
    if (conj) {
        out = std::move(A);
        for (auto &c : B)
            if (std::find(out.begin(), out.end(), c) == out.end()) out.push_back(std::move(c));
    } else {  // OR distributes over the clauses of both sides
        size_t lits = 0;
        for (const Clause &ca : A)
            for (const Clause &cb : B) {
                Clause c = ca;
                bool tautology = false;
                for (const Literal &l : cb) {
                    if (std::find(c.begin(), c.end(), Literal{l.term, !l.neg}) != c.end()) tautology = true;
                    if (std::find(c.begin(), c.end(), l) == c.end()) c.push_back(l);
                }
                if (tautology || std::find(out.begin(), out.end(), c) != out.end()) continue;  // t OR NOT t
                lits += c.size();
                if (lits > 4 * cap) return false;
                out.push_back(std::move(c));
            }
    }
    size_t lits = 0;
    for (auto &c : out) lits += c.size();
    return lits <= 4 * cap;  // generous while composing; the caller applies the real cap to the final form
}
size_t literal_count(const std::vector<Clause> &f) {
    size_t n = 0;
    for (auto &c : f) n += c.size();
    return n;
}

// copy of a column's validity bits re-based to bit 0 (attached to a String truth bitmap when nulls propagate strictly)
DevBufRef rebased_validity(rv_ctx *ctx, const rv_dcolumn *col) {
    const uint64_t n = col->length;
    DevBufRef v = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n) + 8, 16));
    if (n) {
        hipLaunchKernelGGL(rvk::copy_bits_kernel, dim3(grid_for_words(ctx, (n + 63) / 64, 256)), dim3(256), 0, ctx->stream,
                           static_cast<const uint8_t *>(col->validity->ptr), static_cast<uint64_t>(col->validity->bytes), col->offset, n,
                           static_cast<uint64_t *>(v->ptr));
        RV_HIP(hipGetLastError());
    }
    return v;
}

// The reference's own composition, on the device: every term a BooleanArray, the expression the BooleanArray
// operators, the result the predicate of RecordBatch::filter.  Used when one pass cannot hold the predicate.
std::unique_ptr<rv_dcolumn> compose_predicate(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_term *terms,
                                              rv_null_policy policy, const std::vector<ExprNode> &nodes) {
    std::vector<std::unique_ptr<rv_dcolumn>> val(nodes.size());
    auto term_array = [&](const rv_term &t) -> std::unique_ptr<rv_dcolumn> {
        require(t.column < ncols, RV_ERR_INVALID_ARG, fmt("term references column %u of %u", t.column, ncols));
        const rv_dcolumn *col = cols[t.column];
        rv_dcolumn *o = nullptr;
        if (policy == RV_NULL_DROPS) {  // nullable: null where the cell is null
            const rv_status st = rv_compare_term(ctx, col, &t, &o);
            if (st != RV_OK) throw Error(st, last_error());
        } else if (col->dtype == RV_STRING) {  // eager mask (plan.rs:112-130): a definite bool per row
            o = string_term_mask(ctx, col, t, policy);
        } else {
            rv_term one = t;
            one.column = 0;
            rv_dcolumn *none = nullptr;
            run_fused_pass(ctx, &col, 1, &one, 1, policy, nullptr, 0, &none, &o);
        }
        return std::unique_ptr<rv_dcolumn>(o);
    };
    for (size_t i = 0; i < nodes.size(); ++i) {
        const ExprNode &n = nodes[i];
        rv_dcolumn *o = nullptr;
        if (n.kind == 0) {
            val[i] = term_array(terms[n.term]);
            continue;
        }
        if (n.kind == 3) bool_op(ctx, 2, val[n.a].get(), nullptr, &o);
        else bool_op(ctx, n.kind == 1 ? 0 : 1, val[n.a].get(), val[n.b].get(), &o);
        val[i].reset(o);
        // operands may be shared by several parents in principle; a postfix program uses each value once
        val[n.a].reset();
        if (n.kind != 3) val[n.b].reset();
    }
    return std::move(val.back());
}

// What a term rewritten to `mask is true` stood for (FNV-1a over its column's buffer, operator and literal -- the bytes of a String
// literal): travels in the rewritten term's unused literal, and fused_begin keys the selectivity memory on it instead of on the
// temporary bitmap's address, which the pool hands to the next query's mask as well (s == "a" and s == "b" were one predicate).
static uint64_t term_identity(const rv_dcolumn *const *cols, uint32_t ncols, const rv_term &t, uint64_t h = 0xcbf29ce484222325ull) {
    auto mix = [&](uint64_t v) {
        for (int b = 0; b < 8; ++b) h = (h ^ ((v >> (8 * b)) & 0xFF)) * 0x100000001b3ull;
    };
    const rv_dcolumn *c = t.column < ncols ? cols[t.column] : nullptr;
    mix(c && c->values ? (c->values->id ? c->values->id : reinterpret_cast<uint64_t>(c->values->ptr)) : 0);  // see predicate_signature
    mix(c ? static_cast<uint64_t>(c->dtype) : 0);
    mix(static_cast<uint64_t>(t.op));
    mix(static_cast<uint64_t>(t.lit_type));
    if (t.lit_type == RV_STRING) {
        for (uint64_t i = 0; i < t.lit.s.len; ++i) h = (h ^ static_cast<unsigned char>(t.lit.s.ptr[i])) * 0x100000001b3ull;
        mix(t.lit.s.len);
    } else if (t.lit_type != RV_NULL) {
        mix(static_cast<uint64_t>(t.lit.i));
    }
    return h | 1;  // never 0: 0 means "an ordinary Boolean column"
}

void normalize_predicate(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_predicate *pred, Normalized &out) {
    // `x is true` has no literal: whatever the caller left in that field must not reach the selectivity memory's key, where a non-zero
    // literal of an IS_TRUE term means "a rewritten term's identity" (term_identity below)
    std::vector<rv_term> user_terms(pred->terms, pred->terms + pred->n_terms);
    for (rv_term &t : user_terms)
        if (t.op == RV_IS_TRUE) t.lit.i = 0;
    const rv_term *terms = user_terms.data();
    const uint32_t nterms = pred->n_terms;
    const rv_null_policy policy = pred->nulls;
    require(nterms >= 1 && nterms <= static_cast<uint32_t>(rvk::kMaxTerms), RV_ERR_UNSUPPORTED,
            fmt("predicate needs 1..%d terms, got %u", rvk::kMaxTerms, nterms));
    for (uint32_t t = 0; t < nterms; ++t) {
        require(terms[t].column < ncols, RV_ERR_INVALID_ARG, fmt("term %u references column %u of %u", t, terms[t].column, ncols));
        const rv_dtype dt = cols[terms[t].column]->dtype;
        require(is_value_type(dt) || dt == RV_BOOLEAN || dt == RV_STRING, RV_ERR_UNSUPPORTED,
                "predicate columns must be Int64, Float64, Boolean or String on the device path");
    }
    out.cols.assign(cols, cols + ncols);

    // ---- the expression: plain AND, conjunctive normal form, or too large for one pass -------------------------
    std::vector<ExprNode> nodes;
    std::vector<Clause> form;  // empty when the predicate is the AND of `and_terms`
    std::vector<uint32_t> and_terms;
    bool negate_result = false, compose = false;
    if (pred->expr == nullptr) {
        for (uint32_t t = 0; t < nterms; ++t) and_terms.push_back(t);
    } else {
        nodes = parse_expression(pred->expr, pred->n_expr, nterms);
        std::vector<Clause> pos, negf;
        const size_t cap = static_cast<size_t>(rvk::kMaxTerms);
        const bool okp = cnf_of(nodes, static_cast<int>(nodes.size()) - 1, false, cap, pos) && literal_count(pos) <= cap;
        const bool okn = cnf_of(nodes, static_cast<int>(nodes.size()) - 1, true, cap, negf) && literal_count(negf) <= cap;
        bool pure_and = okp && !pos.empty();
        for (auto &c : pos) pure_and = pure_and && c.size() == 1 && !c[0].neg;
        // Simplification may have dropped every literal of a column (`a AND (NOT a OR b OR NOT b)` is `a`), but under
        // RV_NULL_DROPS the nulls of every column the expression READS still drop the row (BooleanArray::and / or / not
        // propagate them, boolean.rs:120-165): the plain AND only stands in when its terms cover those columns.
        if (pure_and && policy == RV_NULL_DROPS)
            for (const ExprNode &n : nodes) {
                if (n.kind != 0 || !cols[terms[n.term].column]->validity) continue;
                bool covered = false;
                for (auto &c : pos) covered = covered || terms[c[0].term].column == terms[n.term].column;
                pure_and = pure_and && covered;
            }
        if (pure_and) {
            for (auto &c : pos) and_terms.push_back(c[0].term);
        } else if (okp && (!okn || literal_count(pos) <= literal_count(negf))) {
            form = std::move(pos);
        } else if (okn) {
            form = std::move(negf);
            negate_result = true;
        } else {
            compose = true;
        }
        if (!pure_and && !compose && form.empty()) {
            // every clause was a tautology: the expression is constantly true (false when negated) wherever it is
            // not null -- one always-true literal keeps the kernels' term list non-empty
            const uint32_t any = nodes.front().term;  // a postfix program starts with a term
            form.push_back(Clause{Literal{any, false}, Literal{any, true}});
        }
    }
    const bool is_expr = !form.empty();
    // the columns the ORIGINAL expression reads (simplification may have dropped literals, never null propagation)
    std::vector<uint32_t> read_cols;
    if (is_expr || compose)
        for (const ExprNode &n : nodes)
            if (n.kind == 0 && std::find(read_cols.begin(), read_cols.end(), terms[n.term].column) == read_cols.end())
                read_cols.push_back(terms[n.term].column);

    // ---- column budget of one pass ---------------------------------------------------------------------------
    std::vector<uint32_t> used_terms = and_terms;
    if (is_expr)
        for (auto &c : form)
            for (auto &l : c)
                if (std::find(used_terms.begin(), used_terms.end(), l.term) == used_terms.end()) used_terms.push_back(l.term);
    std::vector<uint32_t> value_cols, bool_cols;
    size_t string_terms = 0;
    auto note = [](std::vector<uint32_t> &v, uint32_t c) {
        if (std::find(v.begin(), v.end(), c) == v.end()) v.push_back(c);
    };
    for (uint32_t t : used_terms) {
        const uint32_t c = terms[t].column;
        if (is_value_type(cols[c]->dtype)) note(value_cols, c);
        else if (cols[c]->dtype == RV_BOOLEAN) note(bool_cols, c);
        else ++string_terms;  // every String term gets its own truth bitmap
    }
    if (is_expr && pred->nulls == RV_NULL_DROPS)
        for (uint32_t c : read_cols)
            if (cols[c]->dtype == RV_BOOLEAN && cols[c]->validity) note(bool_cols, c);
    const bool too_many_values = value_cols.size() > static_cast<size_t>(rvk::kMaxValueCols);
    size_t string_nulls_only = 0;  // nullable String columns read for their nulls only: one Boolean slot each
    if (is_expr && pred->nulls == RV_NULL_DROPS)
        for (uint32_t c : read_cols) {
            if (cols[c]->dtype != RV_STRING || !cols[c]->validity) continue;
            bool used = false;
            for (uint32_t t : used_terms) used = used || terms[t].column == c;
            if (!used) ++string_nulls_only;
        }
    const bool too_many_bools = bool_cols.size() + string_terms + string_nulls_only > static_cast<size_t>(rvk::kMaxBoolCols);
    if (!compose && (too_many_values || (is_expr && too_many_bools))) {
        compose = true;
        if (nodes.empty()) {  // the AND of the terms as a tree
            for (uint32_t t = 0; t < nterms; ++t) {
                nodes.push_back(ExprNode{0, -1, -1, t});
                if (t) {
                    const int b = static_cast<int>(nodes.size()) - 1, a = t == 1 ? 0 : b - 1;
                    nodes.push_back(ExprNode{1, a, b, 0});
                }
            }
        }
    }
    if (compose) {
        out.masks.emplace_back(compose_predicate(ctx, cols, ncols, terms, policy, nodes));
        rv_term r{};
        r.column = static_cast<uint32_t>(out.cols.size());
        r.op = RV_IS_TRUE;
        uint64_t id = 0xcbf29ce484222325ull;
        for (uint32_t t = 0; t < nterms; ++t) id = term_identity(cols, ncols, terms[t], id);
        for (const ExprNode &nd : nodes) id = (id ^ static_cast<uint64_t>(nd.kind + 7 * nd.term)) * 0x100000001b3ull;
        r.lit.i = static_cast<int64_t>(id | 1);
        out.cols.push_back(out.masks.back().get());
        out.terms.assign(1, r);
        return;
    }

    // ---- String terms -> truth bitmaps ---------------------------------------------------------------------------
    const bool strict = is_expr && policy == RV_NULL_DROPS;
    std::vector<rv_term> rewritten(terms, terms + nterms);
    for (uint32_t t : used_terms) {
        const rv_dcolumn *c = cols[terms[t].column];
        if (c->dtype != RV_STRING) continue;
        out.masks.emplace_back(string_term_mask(ctx, c, terms[t], policy));
        if (strict && c->validity) {  // its nulls have to drop the row even under a NOT: a nullable BooleanArray
            out.masks.back()->validity = rebased_validity(ctx, c);
            out.masks.back()->null_count = -1;
        }
        rv_term r{};
        r.column = static_cast<uint32_t>(out.cols.size());
        r.op = RV_IS_TRUE;
        r.lit.i = static_cast<int64_t>(term_identity(cols, ncols, terms[t]));
        out.cols.push_back(out.masks.back().get());
        rewritten[t] = r;
    }
    if (is_expr) {
        for (auto &c : form)
            for (size_t i = 0; i < c.size(); ++i) {
                out.terms.push_back(rewritten[c[i].term]);
                out.ex.negate.push_back(c[i].neg);
                out.ex.group_end.push_back(i + 1 == c.size());
            }
        out.ex.negate_result = negate_result;
        out.ex.strict = strict;
        if (strict)
            for (uint32_t c : read_cols) out.ex.strict_cols.push_back(cols[c]->dtype == RV_STRING ? UINT32_MAX : c);
        // a String column's nulls travel with its truth bitmap (validity attached above); a String column whose
        // literals were all simplified away still drops its null rows: a Boolean stand-in that is only its validity
        if (strict) {
            std::vector<uint32_t> covered;
            for (uint32_t t : used_terms)
                if (cols[terms[t].column]->dtype == RV_STRING && cols[terms[t].column]->validity) {
                    out.ex.strict_cols.push_back(rewritten[t].column);
                    covered.push_back(terms[t].column);
                }
            for (uint32_t c : read_cols) {
                if (cols[c]->dtype != RV_STRING || !cols[c]->validity || std::find(covered.begin(), covered.end(), c) != covered.end()) continue;
                auto m = std::make_unique<rv_dcolumn>();
                m->dtype = RV_BOOLEAN;
                m->length = cols[c]->length;
                m->validity = rebased_validity(ctx, cols[c]);
                m->values = m->validity;
                out.masks.emplace_back(std::move(m));
                out.ex.strict_cols.push_back(static_cast<uint32_t>(out.cols.size()));
                out.cols.push_back(out.masks.back().get());
            }
        }
        out.ex.strict_cols.erase(std::remove(out.ex.strict_cols.begin(), out.ex.strict_cols.end(), UINT32_MAX), out.ex.strict_cols.end());
        out.has_ex = true;
        return;
    }
    for (uint32_t t : and_terms) out.terms.push_back(rewritten[t]);
    if (!too_many_bools) return;

    // ---- AND only, more Boolean / String predicate columns than one pass reads: fold them into one truth bitmap ----
    const uint64_t n = cols[0]->length;
    rvk::BoolFold f{};
    std::vector<rv_term> kept;
    uint64_t folded_id = 0xcbf29ce484222325ull;
    for (const rv_term &t : out.terms) {
        const rv_dcolumn *c = out.cols[t.column];
        if (c->dtype != RV_BOOLEAN) {
            kept.push_back(t);
            continue;
        }
        // what the folded bitmap stands for: a rewritten term carries its identity, a Boolean column is its buffer
        folded_id = (t.op == RV_IS_TRUE && t.lit.i != 0) ? (folded_id ^ static_cast<uint64_t>(t.lit.i)) * 0x100000001b3ull
                                                         : term_identity(out.cols.data(), static_cast<uint32_t>(out.cols.size()), t, folded_id);
        require(f.nterms < rvk::kMaxTerms, RV_ERR_UNSUPPORTED, "too many predicate terms");
        f.cols[f.nterms] = dev_view(c);
        f.terms[f.nterms] = lower_term(t, RV_BOOLEAN, policy, static_cast<uint32_t>(f.nterms));
        ++f.nterms;
    }
    auto m = std::make_unique<rv_dcolumn>();
    m->dtype = RV_BOOLEAN;
    m->length = n;
    m->null_count = 0;
    m->values = pool_alloc(ctx, std::max<size_t>(bitmap_words_bytes(n) + 8, 16));
    RV_HIP(hipMemsetAsync(m->values->ptr, 0, std::max<size_t>(bitmap_words_bytes(n) + 8, 16), ctx->stream));
    f.n = n;
    f.out_words = static_cast<uint64_t *>(m->values->ptr);
    if (n) {
        hipLaunchKernelGGL(rvk::bool_fold_kernel, dim3(static_cast<uint32_t>(((n + 63) / 64 + 255) / 256)), dim3(256), 0, ctx->stream, f);
        RV_HIP(hipGetLastError());
    }
    rv_term r{};
    r.column = static_cast<uint32_t>(out.cols.size());
    r.op = RV_IS_TRUE;
    r.lit.i = static_cast<int64_t>(folded_id | 1);
    out.masks.emplace_back(std::move(m));
    out.cols.push_back(out.masks.back().get());
    kept.push_back(r);
    out.terms = std::move(kept);
}

}  // namespace rvl
