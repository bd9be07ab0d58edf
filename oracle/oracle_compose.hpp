// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Semantics the reference does NOT implement but the new backend adds (SURVEY.md
// section 8c, "paths with no reference implementation"): compare terms and AND of
// compares over typed RecordBatch columns, and SUM/COUNT over survivors.  They are
// COMPOSED from restated reference primitives only:
//   cell compare        AnyValue PartialEq/PartialOrd     series.rs:87-117
//   null propagation    BooleanArray::and (strict)        boolean.rs:120-135
//   row selection       RecordBatch::filter (Some(true))  record_batch.rs:235-240
//   eager null rule     PhysicalPlan::Filter mask loop    plan.rs:112-130
// The one rule added for the streaming composition: a compare over a NULL CELL is
// null.  Aggregates have no reference at all: parity unpinned.
#pragma once

#include "oracle_eager.hpp"

namespace rvo {

enum class NullPolicy { Drops, IsLeast };
enum class TermOp { Eq, Ne, Lt, Gt, Le, Ge, IsTrue };

struct Term {
    size_t column;
    TermOp op;
    AnyValue literal;
};

inline BinaryOperator to_binary(TermOp op) {
    switch (op) {
        case TermOp::Eq: return BinaryOperator::Eq;
        case TermOp::Ne: return BinaryOperator::NotEq;
        case TermOp::Lt: return BinaryOperator::Lt;
        case TermOp::Gt: return BinaryOperator::Gt;
        case TermOp::Le: return BinaryOperator::LtEq;
        case TermOp::Ge: return BinaryOperator::GtEq;
        default: throw Panic("IsTrue is not a binary operator");
    }
}

// One cell as the row enum the eager path would hold (Null when the validity bit is 0).
inline AnyValue cell_value(const ArrayRef &a, size_t i) {
    switch (a->data_type()) {
        case DataType::Int64: {
            auto v = std::static_pointer_cast<const Int64Array>(a)->value(i);
            return v ? AnyValue(*v) : AnyValue::null();
        }
        case DataType::Float64: {
            auto v = std::static_pointer_cast<const Float64Array>(a)->value(i);
            return v ? AnyValue(*v) : AnyValue::null();
        }
        case DataType::Boolean: {
            auto v = std::static_pointer_cast<const BooleanArray>(a)->value(i);
            return v ? AnyValue(*v) : AnyValue::null();
        }
        case DataType::String: {
            auto v = std::static_pointer_cast<const StringArray>(a)->value(i);
            return v ? AnyValue(*v) : AnyValue::null();
        }
        default: return AnyValue::null();
    }
}

// Streaming composition: compare term -> nullable BooleanArray (null where the cell is null).
inline std::shared_ptr<BooleanArray> compare_array(const ArrayRef &a, TermOp op, const AnyValue &lit) {
    if (op == TermOp::IsTrue) {
        auto b = std::dynamic_pointer_cast<const BooleanArray>(a);
        if (!b) throw Err("Predicate must be a BooleanArray");
        BooleanArrayBuilder out;
        for (size_t i = 0; i < b->len(); ++i) {
            auto v = b->value(i);
            if (v) out.append_value(*v);
            else out.append_null();
        }
        return out.finish();
    }
    BooleanArrayBuilder out;
    for (size_t i = 0; i < a->len(); ++i) {
        AnyValue cell = cell_value(a, i);
        if (cell.is_null()) out.append_null();
        else out.append_value(any_compare(to_binary(op), cell, lit));
    }
    return out.finish();
}

// AND of terms -> the BooleanArray handed to RecordBatch::filter.
//   Drops:   compare_array per term, folded with BooleanArray::and.
//   IsLeast: the eager mask (plan.rs:112-130) per term, ANDed (== chained .filter() calls).
//   expr (postfix over term indices, 0x80 AND / 0x81 OR / 0x82 NOT; rv_predicate::expr): the same per-term
//            arrays combined with BooleanArray::and / or / not (Drops: strict null propagation, boolean.rs:120-165;
//            IsLeast: the eager masks are null-free, so the operators are plain Boolean algebra on them).
inline std::shared_ptr<BooleanArray> term_mask(const std::vector<ArrayRef> &cols, const Term &t, NullPolicy policy) {
    if (policy == NullPolicy::Drops) return compare_array(cols.at(t.column), t.op, t.literal);
    const auto &c = cols.at(t.column);
    std::vector<bool> mask(c->len());
    for (size_t i = 0; i < mask.size(); ++i) {  // the eager mask loop, plan.rs:112-130
        AnyValue cell = cell_value(c, i);
        mask[i] = (t.op == TermOp::IsTrue) ? any_eq(cell, AnyValue(true)) : any_compare(to_binary(t.op), cell, t.literal);
    }
    return BooleanArray::from_bools(mask);
}
inline std::shared_ptr<BooleanArray> evaluate_expression(const std::vector<ArrayRef> &cols, const std::vector<Term> &terms,
                                                         NullPolicy policy, const std::vector<uint8_t> &expr) {
    std::vector<std::shared_ptr<BooleanArray>> stack;
    for (uint8_t op : expr) {
        if (op < 0x80) {
            stack.push_back(term_mask(cols, terms.at(op), policy));
        } else if (op == 0x82) {
            if (stack.empty()) throw Err("predicate expression: NOT without operand");
            stack.back() = stack.back()->logical_not();
        } else {
            if (stack.size() < 2) throw Err("predicate expression: operator without operands");
            auto b = stack.back();
            stack.pop_back();
            auto a = stack.back();
            stack.back() = op == 0x80 ? a->logical_and(*b) : a->logical_or(*b);
        }
    }
    if (stack.size() != 1) throw Err("predicate expression must leave exactly one value");
    return stack.back();
}

inline std::shared_ptr<BooleanArray> evaluate_predicate(const std::vector<ArrayRef> &cols,
                                                        const std::vector<Term> &terms, NullPolicy policy,
                                                        const std::vector<uint8_t> &expr = {}) {
    if (terms.empty()) throw Err("predicate needs at least one term");
    if (!expr.empty()) return evaluate_expression(cols, terms, policy, expr);
    size_t n = cols.empty() ? 0 : cols[0]->len();
    if (policy == NullPolicy::Drops) {
        std::shared_ptr<BooleanArray> acc;
        for (auto &t : terms) {
            auto b = compare_array(cols.at(t.column), t.op, t.literal);
            acc = acc ? acc->logical_and(*b) : b;
        }
        return acc;
    }
    std::vector<bool> mask(n, true);
    for (auto &t : terms) {
        const auto &c = cols.at(t.column);
        for (size_t i = 0; i < n; ++i) {
            AnyValue cell = cell_value(c, i);
            bool keep = (t.op == TermOp::IsTrue) ? any_eq(cell, AnyValue(true))
                                                 : any_compare(to_binary(t.op), cell, t.literal);
            mask[i] = mask[i] && keep;
        }
    }
    return BooleanArray::from_bools(mask);
}

inline SchemaRef positional_schema(const std::vector<ArrayRef> &cols) {
    std::vector<Field> f;
    for (size_t i = 0; i < cols.size(); ++i) f.push_back(Field{"c" + std::to_string(i), cols[i]->data_type(), true});
    return std::make_shared<Schema>(f);
}

// One-shot: SelectStream(FilterStream(batch)) on a single batch.
inline RecordBatch filter_project(const std::vector<ArrayRef> &cols, const std::vector<Term> &terms,
                                  NullPolicy policy, const std::vector<size_t> &proj, const std::vector<uint8_t> &expr = {}) {
    RecordBatch batch = RecordBatch::try_new(positional_schema(cols), cols);
    ArrayRef pred = evaluate_predicate(cols, terms, policy, expr);
    return batch.filter(pred).select_columns(proj);
}

// A FilterStream whose predicate is computed per batch by evaluate_predicate: what the
// streaming planner's Filter arm becomes once extract_boolean_predicate_column
// (streaming_planner.rs:137-168) is replaced by compare/AND lowering.
class PredicateFilterStream : public DataStream {
  public:
    PredicateFilterStream(DataStreamRef input, std::vector<Term> terms, NullPolicy policy, std::vector<uint8_t> expr = {})
        : input_(std::move(input)), terms_(std::move(terms)), policy_(policy), expr_(std::move(expr)) {}
    SchemaRef schema() const override { return input_->schema(); }
    std::optional<RecordBatch> next_batch() override {
        auto batch = input_->next_batch();
        if (!batch) return std::nullopt;
        ArrayRef pred = evaluate_predicate(batch->columns(), terms_, policy_, expr_);
        try {
            return batch->filter(pred);
        } catch (const Err &e) {
            throw StreamError::execution(e.what());
        }
    }

  private:
    DataStreamRef input_;
    std::vector<Term> terms_;
    NullPolicy policy_;
    std::vector<uint8_t> expr_;
};

// The faithful streaming pipeline: chunk into batch_rows-row zero-copy slices ->
// MemoryStream -> PredicateFilterStream -> SelectStream -> concat (streaming.rs:343-352).
inline RecordBatch stream_filter_project(const std::vector<ArrayRef> &cols, size_t batch_rows,
                                         const std::vector<Term> &terms, NullPolicy policy,
                                         const std::vector<size_t> &proj, const std::vector<uint8_t> &expr = {}) {
    RecordBatch whole = RecordBatch::try_new(positional_schema(cols), cols);
    std::vector<RecordBatch> batches;
    for (size_t off = 0; off < whole.num_rows(); off += batch_rows)
        batches.push_back(whole.slice(off, std::min(batch_rows, whole.num_rows() - off)));
    std::vector<std::string> names;
    for (size_t p : proj) names.push_back(whole.schema()->field(p).name);
    DataStreamRef s = std::make_unique<MemoryStream>(whole.schema(), std::move(batches));
    s = std::make_unique<PredicateFilterStream>(std::move(s), terms, policy, expr);
    auto sel = std::make_unique<SelectStream>(std::move(s), names);
    return sel->concatenate();
}

// SUM / COUNT over survivors -- NO reference implementation; parity unpinned.
// COUNT = surviving rows; SUM(Int64) wraps in two's complement over surviving non-null
// cells; SUM(Float64) adds surviving non-null cells in row order.
struct AggResult {
    int64_t sum_i = 0;
    double sum_f = 0.0;
    uint64_t count = 0;
};
inline AggResult filter_agg(const std::vector<ArrayRef> &cols, const std::vector<Term> &terms, NullPolicy policy,
                            size_t agg_col, const std::vector<uint8_t> &expr = {}) {
    auto pred = evaluate_predicate(cols, terms, policy, expr);
    AggResult r;
    const auto &c = cols.at(agg_col);
    for (size_t i = 0; i < pred->len(); ++i) {
        auto p = pred->value(i);
        if (!(p && *p)) continue;
        ++r.count;
        AnyValue cell = cell_value(c, i);
        if (cell.v.index() == 1)
            r.sum_i = static_cast<int64_t>(static_cast<uint64_t>(r.sum_i) + static_cast<uint64_t>(std::get<1>(cell.v)));
        else if (cell.v.index() == 2)
            r.sum_f += std::get<2>(cell.v);
    }
    return r;
}

// Synthetic generator shared (bit for bit) with the device kernel -- SURVEY.md section 8d.
inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace rvo
