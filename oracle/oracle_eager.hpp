// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the row-enum ("eager") side of CleConor/rivulus:
//   AnyValue / Series / DataFrame        src/datatypes/series.rs, dataframe.rs
//   Expr / BinaryOperator                src/expressions/expr.rs
//   LogicalPlan / optimizer / LazyFrame  src/logical_plan/{plan,optimizer,builder}.rs
//   PhysicalPlan (Source/Select/Filter/Limit) + planner
//                                        src/physical_plan/{plan,planner}.rs
//   StreamingPhysicalPlan + planner      src/physical_plan/{streaming,streaming_planner}.rs
// HashJoin and CsvFileSource are outside the hot path
// and are not restated.
#pragma once

#include <cstdlib>
#include <set>
#include <string_view>
#include <variant>

#include "oracle_batch.hpp"

namespace rvo {

// ---------------------------------------------------------------------------
// AnyValue -- series.rs:6-13, PartialEq :87-98, PartialOrd :100-117
// ---------------------------------------------------------------------------
// The String payload laid out as Rust lays out `String`: pointer / capacity / length, 24 bytes, a heap allocation per
// clone of a non-empty string (std::string is 32 bytes with libstdc++ and keeps short strings inline) -- so that the
// cell below is 32 bytes: the UPPER bound SURVEY.md section 8d(i) gives for the reference's cell (24-32, compiler-dependent).
// Edition 2024 needs rustc >= 1.85, where String's capacity has a niche and the enum most likely takes 24 bytes: the timed
// restatement then moves up to a third more bytes per cell than the reference would -- it never moves fewer.
class RustString {
  public:
    RustString() = default;
    RustString(const std::string &s) { assign(s.data(), s.size()); }
    RustString(const char *s) { assign(s, std::strlen(s)); }
    RustString(const RustString &o) { assign(o.ptr_, o.len_); }
    RustString(RustString &&o) noexcept : ptr_(o.ptr_), cap_(o.cap_), len_(o.len_) { o.ptr_ = nullptr, o.cap_ = o.len_ = 0; }
    RustString &operator=(RustString o) noexcept {
        std::swap(ptr_, o.ptr_), std::swap(cap_, o.cap_), std::swap(len_, o.len_);
        return *this;
    }
    ~RustString() { std::free(ptr_); }
    std::string_view view() const { return std::string_view(ptr_ ? ptr_ : "", len_); }
    operator std::string() const { return std::string(view()); }
    bool operator==(const RustString &o) const { return view() == o.view(); }
    int compare(const RustString &o) const { return view().compare(o.view()); }  // byte-wise, like Rust's str

  private:
    void assign(const char *s, size_t n) {
        if (n) {
            ptr_ = static_cast<char *>(std::malloc(n));
            if (!ptr_) throw std::bad_alloc();
            std::memcpy(ptr_, s, n);
        }
        cap_ = len_ = n;
    }
    char *ptr_ = nullptr;
    size_t cap_ = 0, len_ = 0;
};
static_assert(sizeof(RustString) == 24, "String is pointer + capacity + length");

struct AnyValue {
    std::variant<std::monostate, int64_t, double, RustString, bool> v;
    AnyValue() = default;
    AnyValue(int64_t x) : v(x) {}
    AnyValue(int x) : v(static_cast<int64_t>(x)) {}
    AnyValue(double x) : v(x) {}
    AnyValue(const char *s) : v(RustString(s)) {}
    AnyValue(const std::string &s) : v(RustString(s)) {}
    AnyValue(bool b) : v(b) {}
    static AnyValue null() { return AnyValue(); }

    bool is_null() const { return v.index() == 0; }
    DataType data_type() const {  // series.rs:20-28
        switch (v.index()) {
            case 0: return DataType::Null;
            case 1: return DataType::Int64;
            case 2: return DataType::Float64;
            case 3: return DataType::String;
            default: return DataType::Boolean;
        }
    }
    std::string display() const {  // series.rs:61-71
        switch (v.index()) {
            case 0: return "null";
            case 1: return std::to_string(std::get<1>(v));
            case 2: {
                char buf[64];
                snprintf(buf, sizeof buf, "%g", std::get<2>(v));
                return buf;
            }
            case 3: return std::get<3>(v);
            default: return std::get<4>(v) ? "true" : "false";
        }
    }
};
static_assert(sizeof(AnyValue) == 32, "the reference's cell: a 24-byte String payload + tag (SURVEY.md section 8d)");

// series.rs:87-98
inline bool any_eq(const AnyValue &a, const AnyValue &b) {
    if (a.v.index() != b.v.index()) return false;
    switch (a.v.index()) {
        case 0: return true;
        case 1: return std::get<1>(a.v) == std::get<1>(b.v);
        case 2: return std::get<2>(a.v) == std::get<2>(b.v);  // IEEE: NaN != NaN, -0.0 == 0.0
        case 3: return std::get<3>(a.v) == std::get<3>(b.v);
        default: return std::get<4>(a.v) == std::get<4>(b.v);
    }
}

// series.rs:100-117.  -1 Less, 0 Equal, +1 Greater, nullopt == None.
inline std::optional<int> any_partial_cmp(const AnyValue &a, const AnyValue &b) {
    if (a.is_null() && b.is_null()) return 0;
    if (a.is_null()) return -1;
    if (b.is_null()) return 1;
    if (a.v.index() != b.v.index()) return std::nullopt;  // cross-type
    auto cmp3 = [](auto x, auto y) -> std::optional<int> {
        if (x < y) return -1;
        if (x > y) return 1;
        if (x == y) return 0;
        return std::nullopt;  // NaN
    };
    switch (a.v.index()) {
        case 1: return cmp3(std::get<1>(a.v), std::get<1>(b.v));
        case 2: return cmp3(std::get<2>(a.v), std::get<2>(b.v));
        case 3: {
            int c = std::get<3>(a.v).compare(std::get<3>(b.v));
            return c < 0 ? -1 : (c > 0 ? 1 : 0);
        }
        default: return cmp3(static_cast<int>(std::get<4>(a.v)), static_cast<int>(std::get<4>(b.v)));
    }
}

// expr.rs:15-29
enum class BinaryOperator { Plus, Minus, Multiply, Divide, Eq, NotEq, Lt, Gt, LtEq, GtEq, And, Or };

inline const char *op_name(BinaryOperator op) {
    static const char *n[] = {"Plus", "Minus", "Multiply", "Divide", "Eq", "NotEq",
                              "Lt",   "Gt",    "LtEq",     "GtEq",   "And", "Or"};
    return n[static_cast<int>(op)];
}
inline bool is_compare(BinaryOperator op) {
    return op == BinaryOperator::Eq || op == BinaryOperator::NotEq || op == BinaryOperator::Lt ||
           op == BinaryOperator::Gt || op == BinaryOperator::LtEq || op == BinaryOperator::GtEq;
}

// plan.rs:113-120: the per-row decision of PhysicalPlan::Filter.  Rust derives <,<=,>,>=
// from partial_cmp (None => false) and != as !eq.
inline bool any_compare(BinaryOperator op, const AnyValue &row, const AnyValue &lit) {
    switch (op) {
        case BinaryOperator::Eq: return any_eq(row, lit);
        case BinaryOperator::NotEq: return !any_eq(row, lit);
        default: break;
    }
    auto c = any_partial_cmp(row, lit);
    if (!c) return false;
    switch (op) {
        case BinaryOperator::Lt: return *c < 0;
        case BinaryOperator::Gt: return *c > 0;
        case BinaryOperator::LtEq: return *c <= 0;
        case BinaryOperator::GtEq: return *c >= 0;
        default: throw Panic("any_compare: not a comparison operator");
    }
}

// ---------------------------------------------------------------------------
// Series -- series.rs:119-229 ; SeriesError :176-183
// ---------------------------------------------------------------------------
struct SeriesError : std::runtime_error {
    enum Kind { MixedTypes, EmptyData, OutOfBounds } kind;
    SeriesError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};

class Series {
  public:
    Series(const std::string &name, std::vector<AnyValue> data) : name_(name), data_(std::move(data)) {  // :185-221
        if (data_.empty()) throw SeriesError(SeriesError::EmptyData, "Empty series not allowed");
        std::optional<DataType> first;
        for (auto &v : data_)
            if (!v.is_null()) {
                first = v.data_type();
                break;
            }
        DataType dtype = first.value_or(DataType::Null);
        for (auto &v : data_) {
            if (v.is_null()) continue;
            DataType cur = v.data_type();
            if (!compatible(dtype, cur))
                throw SeriesError(SeriesError::MixedTypes, std::string("Mixed types in series: expected ") +
                                                               dtype_name(dtype) + ", found " + dtype_name(cur));
            if (dtype == DataType::Int64 && cur == DataType::Float64) dtype = DataType::Float64;  // :210-212
        }
        dtype_ = dtype;
    }
    static Series empty(const std::string &name, DataType dtype) {  // :223-229
        Series s;
        s.name_ = name;
        s.dtype_ = dtype;
        return s;
    }
    const std::string &name() const { return name_; }
    size_t len() const { return data_.size(); }
    bool is_empty() const { return data_.empty(); }
    DataType dtype() const { return dtype_; }
    const std::vector<AnyValue> &data() const { return data_; }
    const AnyValue &operator[](size_t i) const {  // :276-287
        if (i >= data_.size())
            throw Panic("Index " + std::to_string(i) + " out of bounds for series of length " +
                        std::to_string(data_.size()));
        return data_[i];
    }

  private:
    Series() : dtype_(DataType::Null) {}
    static bool compatible(DataType e, DataType f) {  // :259-268
        if (e == f) return true;
        return (e == DataType::Int64 && f == DataType::Float64) || (e == DataType::Float64 && f == DataType::Int64);
    }
    std::string name_;
    std::vector<AnyValue> data_;
    DataType dtype_;
};

// ---------------------------------------------------------------------------
// DataFrame -- dataframe.rs:7-110
// ---------------------------------------------------------------------------
struct DataFrameError : std::runtime_error {
    enum Kind { LengthMismatch, DuplicateColumn, ColumnNotFound } kind;
    DataFrameError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};

class DataFrame {
  public:
    DataFrame() = default;
    explicit DataFrame(std::vector<Series> columns) {  // dataframe.rs:29-56
        if (columns.empty()) return;
        std::set<std::string> seen;
        for (auto &c : columns)
            if (!seen.insert(c.name()).second)
                throw DataFrameError(DataFrameError::DuplicateColumn, "Duplicate column name: '" + c.name() + "'");
        size_t expected = columns[0].len();
        for (auto &c : columns)
            if (c.len() != expected)
                throw DataFrameError(DataFrameError::LengthMismatch,
                                     "Column lengths mismatch: expected " + std::to_string(expected) + ", found " +
                                         std::to_string(c.len()) + " for column '" + c.name() + "'");
        columns_ = std::move(columns);
    }
    size_t height() const { return columns_.empty() ? 0 : columns_[0].len(); }
    size_t width() const { return columns_.size(); }
    bool is_empty() const { return columns_.empty(); }  // NB: "no columns", dataframe.rs:80-82
    const Series *column(const std::string &name) const {
        for (auto &s : columns_)
            if (s.name() == name) return &s;
        return nullptr;
    }
    std::vector<std::string> column_names() const {
        std::vector<std::string> n;
        for (auto &s : columns_) n.push_back(s.name());
        return n;
    }
    const std::vector<Series> &columns() const { return columns_; }
    DataFrame select(const std::vector<std::string> &names) const {  // dataframe.rs:96-109 (dups allowed)
        DataFrame out;
        for (auto &n : names) {
            auto s = column(n);
            if (!s) throw DataFrameError(DataFrameError::ColumnNotFound, "Column not found: '" + n + "'");
            out.columns_.push_back(*s);
        }
        return out;
    }

  private:
    std::vector<Series> columns_;
};

// ---------------------------------------------------------------------------
// Expr -- expr.rs:3-139
// ---------------------------------------------------------------------------
struct Expr;
using ExprPtr = std::shared_ptr<const Expr>;
struct Expr {
    enum Kind { Column, Literal, Binary, Alias } kind;
    std::string name;  // Column name / Alias name
    AnyValue literal;
    ExprPtr left, right;  // Binary: left,right ; Alias: left = inner
    BinaryOperator op = BinaryOperator::Eq;

    static Expr col(const std::string &n) {
        Expr e;
        e.kind = Column;
        e.name = n;
        return e;
    }
    static Expr lit(AnyValue v) {
        Expr e;
        e.kind = Literal;
        e.literal = std::move(v);
        return e;
    }
    Expr alias(const std::string &n) const {
        Expr e;
        e.kind = Alias;
        e.name = n;
        e.left = std::make_shared<Expr>(*this);
        return e;
    }
    Expr binary(BinaryOperator o, const Expr &other) const {
        Expr e;
        e.kind = Binary;
        e.op = o;
        e.left = std::make_shared<Expr>(*this);
        e.right = std::make_shared<Expr>(other);
        return e;
    }
    Expr eq(const Expr &o) const { return binary(BinaryOperator::Eq, o); }
    Expr neq(const Expr &o) const { return binary(BinaryOperator::NotEq, o); }
    Expr lt(const Expr &o) const { return binary(BinaryOperator::Lt, o); }
    Expr gt(const Expr &o) const { return binary(BinaryOperator::Gt, o); }
    Expr lte(const Expr &o) const { return binary(BinaryOperator::LtEq, o); }
    Expr gte(const Expr &o) const { return binary(BinaryOperator::GtEq, o); }
    Expr and_(const Expr &o) const { return binary(BinaryOperator::And, o); }
    Expr or_(const Expr &o) const { return binary(BinaryOperator::Or, o); }
    Expr add(const Expr &o) const { return binary(BinaryOperator::Plus, o); }
};

// ---------------------------------------------------------------------------
// LogicalPlan -- logical_plan/plan.rs:9-286 (Join / CsvFileSource not restated)
// ---------------------------------------------------------------------------
struct LogicalPlanError : std::runtime_error {
    enum Kind { ColumnNotFound } kind;
    std::string name;
    LogicalPlanError(const std::string &n)
        : std::runtime_error("Column not found: '" + n + "'"), kind(ColumnNotFound), name(n) {}
};

using NamedSchema = std::vector<std::pair<std::string, DataType>>;

struct LogicalPlan;
using LogicalPlanPtr = std::shared_ptr<const LogicalPlan>;
struct LogicalPlan {
    enum Kind { DataFrameSource, Select, Filter, Limit } kind;
    DataFrame df;               // DataFrameSource
    NamedSchema source_schema;  // DataFrameSource
    LogicalPlanPtr input;
    std::vector<Expr> expressions;  // Select
    Expr predicate;                 // Filter
    size_t n = 0;                   // Limit

    NamedSchema schema() const {  // plan.rs:63-113
        switch (kind) {
            case DataFrameSource: return source_schema;
            case Select: {
                auto in = input->schema();
                NamedSchema out;
                for (auto &e : expressions) out.push_back(resolve_expr_schema(e, in));
                return out;
            }
            default: return input->schema();
        }
    }

    void validate() const {  // plan.rs:115-202
        switch (kind) {
            case DataFrameSource: {
                auto names = df.column_names();
                for (auto &p : source_schema) {
                    bool found = false;
                    for (auto &n : names) found |= (n == p.first);
                    if (!found) throw LogicalPlanError(p.first);
                }
                return;
            }
            case Select: {
                input->validate();
                auto in = input->schema();
                for (auto &e : expressions) validate_expr_columns(e, in);
                return;
            }
            case Filter: {
                input->validate();
                validate_expr_columns(predicate, input->schema());  // against the INPUT schema :139-146
                return;
            }
            case Limit: input->validate(); return;
        }
    }

    static std::pair<std::string, DataType> resolve_expr_schema(const Expr &e, const NamedSchema &in) {  // :204-233
        switch (e.kind) {
            case Expr::Column: {
                DataType t = DataType::Null;
                for (auto &p : in)
                    if (p.first == e.name) {
                        t = p.second;
                        break;
                    }
                return {e.name, t};
            }
            case Expr::Alias: return {e.name, resolve_expr_schema(*e.left, in).second};
            case Expr::Binary: {
                auto l = resolve_expr_schema(*e.left, in);
                auto r = resolve_expr_schema(*e.right, in);
                return {l.first, infer_binary(l.second, e.op, r.second)};
            }
            default: return {"literal", e.literal.data_type()};
        }
    }
    static DataType infer_binary(DataType l, BinaryOperator op, DataType r) {  // :235-262
        if (is_compare(op) || op == BinaryOperator::And || op == BinaryOperator::Or) return DataType::Boolean;
        if (l == DataType::Float64 || r == DataType::Float64) return DataType::Float64;
        if (l == DataType::Int64 && r == DataType::Int64) return DataType::Int64;
        if (l == DataType::Null) return r;
        if (r == DataType::Null) return l;
        return DataType::Null;
    }
    static void validate_expr_columns(const Expr &e, const NamedSchema &schema) {  // :264-285
        switch (e.kind) {
            case Expr::Column: {
                for (auto &p : schema)
                    if (p.first == e.name) return;
                throw LogicalPlanError(e.name);
            }
            case Expr::Binary:
                validate_expr_columns(*e.left, schema);
                validate_expr_columns(*e.right, schema);
                return;
            case Expr::Alias: validate_expr_columns(*e.left, schema); return;
            default: return;
        }
    }
};

// optimizer.rs:6-100 -- the single rewrite Select(Filter(x)) -> Filter(Select(x))
struct QueryOptimizer {
    static LogicalPlanPtr optimize(LogicalPlanPtr p) { return push_predicates_down(std::move(p)); }

    static LogicalPlanPtr push_predicates_down(LogicalPlanPtr plan) {
        switch (plan->kind) {
            case LogicalPlan::Select: {
                const auto &in = plan->input;
                if (in->kind == LogicalPlan::Filter) {
                    if (predicate_uses_only_selected(in->predicate, plan->expressions)) {
                        auto sel = std::make_shared<LogicalPlan>();
                        sel->kind = LogicalPlan::Select;
                        sel->input = in->input;  // NOT recursed into (optimizer.rs:24-30)
                        sel->expressions = plan->expressions;
                        auto fil = std::make_shared<LogicalPlan>();
                        fil->kind = LogicalPlan::Filter;
                        fil->input = sel;
                        fil->predicate = in->predicate;
                        return fil;
                    }
                    return plan;  // rebuilt unchanged (optimizer.rs:31-38)
                }
                auto out = std::make_shared<LogicalPlan>(*plan);
                out->input = push_predicates_down(in);
                return out;
            }
            case LogicalPlan::Filter: {
                auto out = std::make_shared<LogicalPlan>(*plan);
                out->input = push_predicates_down(plan->input);
                return out;
            }
            default: return plan;  // Limit and sources: `other => other` (:62)
        }
    }
    static void extract_column_names(const Expr &e, std::vector<std::string> &out) {  // :75-86
        switch (e.kind) {
            case Expr::Column: out.push_back(e.name); return;
            case Expr::Binary:
                extract_column_names(*e.left, out);
                extract_column_names(*e.right, out);
                return;
            case Expr::Alias: extract_column_names(*e.left, out); return;
            default: return;
        }
    }
    static bool predicate_uses_only_selected(const Expr &pred, const std::vector<Expr> &exprs) {  // :66-73,:88-99
        std::vector<std::string> pc, sc;
        extract_column_names(pred, pc);
        for (auto &e : exprs) {
            if (e.kind == Expr::Column) sc.push_back(e.name);
            else if (e.kind == Expr::Alias && e.left->kind == Expr::Column) sc.push_back(e.left->name);  // alias-blind
        }
        for (auto &c : pc) {
            bool found = false;
            for (auto &s : sc) found |= (s == c);
            if (!found) return false;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------
// PhysicalPlan (eager) -- physical_plan/plan.rs:8-210 ; ExecutionError :36-62
// ---------------------------------------------------------------------------
struct ExecutionError : std::runtime_error {
    enum Kind { ColumnNotFound, InvalidOperation, DataFrameErr, SeriesErr, General } kind;
    ExecutionError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};

struct PhysicalPlan;
using PhysicalPlanPtr = std::shared_ptr<const PhysicalPlan>;
struct PhysicalPlan {
    enum Kind { DataFrameSource, Select, Filter, Limit } kind;
    DataFrame df;
    PhysicalPlanPtr input;
    std::vector<std::string> columns, final_names;  // Select
    std::string column;                             // Filter
    AnyValue value;
    BinaryOperator op = BinaryOperator::Eq;
    size_t n = 0;

    DataFrame execute() const {
        try {
            return execute_inner();
        } catch (const DataFrameError &e) {
            throw ExecutionError(ExecutionError::DataFrameErr, std::string("DataFrame error: ") + e.what());
        } catch (const SeriesError &e) {
            throw ExecutionError(ExecutionError::SeriesErr, std::string("Series error: ") + e.what());
        }
    }

  private:
    DataFrame execute_inner() const {
        switch (kind) {
            case DataFrameSource: return df;  // plan.rs:67
            case Select: {                     // plan.rs:68-96
                DataFrame in = input->execute();
                for (auto &c : columns)
                    if (!in.column(c))
                        throw ExecutionError(ExecutionError::ColumnNotFound, "Column not found: '" + c + "'");
                DataFrame sel = in.select(columns);
                std::vector<Series> renamed;
                for (size_t i = 0; i < sel.columns().size() && i < final_names.size(); ++i)
                    renamed.emplace_back(final_names[i], sel.columns()[i].data());  // Series::new: EmptyData on 0 rows
                return DataFrame(std::move(renamed));
            }
            case Filter: {  // plan.rs:97-150
                DataFrame in = input->execute();
                const Series *fs = in.column(column);
                if (!fs) throw ExecutionError(ExecutionError::ColumnNotFound, "Column not found: '" + column + "'");
                std::vector<bool> mask;
                mask.reserve(in.height());
                for (auto &row : fs->data()) {  // HOT LOOP 1 :112-130
                    if (!is_compare(op))
                        throw ExecutionError(ExecutionError::InvalidOperation,
                                             std::string("Invalid operation: ") + op_name(op) +
                                                 " not supported for types " + dtype_name(fs->dtype()) + " and " +
                                                 dtype_name(value.data_type()));
                    mask.push_back(any_compare(op, row, value));
                }
                std::vector<Series> out;
                for (auto &s : in.columns()) {  // HOT LOOP 2 :132-147
                    std::vector<AnyValue> kept;
                    for (size_t i = 0; i < s.len(); ++i)
                        if (mask[i]) kept.push_back(s.data()[i]);
                    if (kept.empty()) out.push_back(Series::empty(s.name(), s.dtype()));
                    else out.emplace_back(s.name(), std::move(kept));
                }
                return DataFrame(std::move(out));
            }
            case Limit: {  // plan.rs:151-173
                DataFrame in = input->execute();
                if (n == 0 || in.is_empty()) {
                    std::vector<Series> e;
                    for (auto &s : in.columns()) e.push_back(Series::empty(s.name(), s.dtype()));
                    return DataFrame(std::move(e));
                }
                size_t limit = std::min(n, in.height());
                std::vector<Series> out;
                for (auto &s : in.columns()) {
                    std::vector<AnyValue> d(s.data().begin(), s.data().begin() + limit);
                    out.emplace_back(s.name(), std::move(d));
                }
                return DataFrame(std::move(out));
            }
        }
        throw Panic("unreachable");
    }
};

// planner.rs:8-39 ConversionError
struct ConversionError : std::runtime_error {
    enum Kind {
        UnsupportedExpression,
        UnsupportedFilter,
        InvalidFilterStructure,
        FilterLeftNotColumn,
        FilterRightNotLiteral,
        UnsupportedFilterOperator,
        InvalidSelectExpression
    } kind;
    ConversionError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};

// planner.rs:113-132
inline std::pair<std::string, std::string> convert_select_expr(const Expr &e) {
    switch (e.kind) {
        case Expr::Column: return {e.name, e.name};
        case Expr::Alias:
            if (e.left->kind == Expr::Column) return {e.left->name, e.name};
            throw ConversionError(ConversionError::UnsupportedExpression, "Unsupported expression");
        case Expr::Binary: throw ConversionError(ConversionError::UnsupportedExpression, "Unsupported expression");
        default:
            throw ConversionError(ConversionError::InvalidSelectExpression,
                                  "Select expression must be a column or alias");
    }
}

struct FilterTriple {
    std::string column;
    AnyValue value;
    BinaryOperator op;
};
// planner.rs:134-189
inline FilterTriple convert_filter_predicate(const Expr &p) {
    if (p.kind != Expr::Binary) {
        const char *t = p.kind == Expr::Column ? "Column" : (p.kind == Expr::Literal ? "Literal" : "Alias");
        throw ConversionError(ConversionError::InvalidFilterStructure,
                              std::string("Filter must be a binary comparison, found: ") + t);
    }
    if (p.op == BinaryOperator::And || p.op == BinaryOperator::Or)
        throw ConversionError(ConversionError::UnsupportedFilter,
                              "Unsupported filter: only simple column comparisons supported");
    if (!is_compare(p.op))
        throw ConversionError(ConversionError::UnsupportedFilterOperator,
                              std::string("Unsupported binary operator in filter: ") + op_name(p.op));
    if (p.left->kind != Expr::Column)
        throw ConversionError(ConversionError::FilterLeftNotColumn, "Filter left side must be a column reference");
    if (p.right->kind != Expr::Literal)
        throw ConversionError(ConversionError::FilterRightNotLiteral, "Filter right side must be a literal value");
    return {p.left->name, p.right->literal, p.op};
}

// planner.rs:41-111
inline PhysicalPlanPtr logical_to_physical(const LogicalPlanPtr &l) {
    auto out = std::make_shared<PhysicalPlan>();
    switch (l->kind) {
        case LogicalPlan::DataFrameSource:
            out->kind = PhysicalPlan::DataFrameSource;
            out->df = l->df;
            return out;
        case LogicalPlan::Select: {
            out->kind = PhysicalPlan::Select;
            out->input = logical_to_physical(l->input);
            for (auto &e : l->expressions) {
                auto pr = convert_select_expr(e);
                out->columns.push_back(pr.first);
                out->final_names.push_back(pr.second);
            }
            return out;
        }
        case LogicalPlan::Filter: {
            out->kind = PhysicalPlan::Filter;
            out->input = logical_to_physical(l->input);
            auto t = convert_filter_predicate(l->predicate);
            out->column = t.column;
            out->value = t.value;
            out->op = t.op;
            return out;
        }
        case LogicalPlan::Limit:
            out->kind = PhysicalPlan::Limit;
            out->input = logical_to_physical(l->input);
            out->n = l->n;
            return out;
    }
    throw Panic("unreachable");
}

// ---------------------------------------------------------------------------
// Streaming physical plan -- streaming.rs:29-133, chunker :135-233, collect :235-238, :343-352
// ---------------------------------------------------------------------------
struct StreamingPlannerError : std::runtime_error {
    enum Kind { ExpressionError, StreamingExecution } kind;
    StreamingPlannerError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};
struct StreamingExecutionError : std::runtime_error {
    enum Kind { Stream, Conversion, InvalidOperation } kind;
    StreamingExecutionError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};

// streaming.rs:135-233: nulls become 0 / 0.0 / false WITHOUT validity (String keeps nulls)
inline std::vector<RecordBatch> dataframe_to_batches(const DataFrame &df, size_t batch_size) {
    std::vector<RecordBatch> batches;
    if (df.is_empty()) return batches;
    size_t num_rows = df.height();
    size_t num_batches = (num_rows + batch_size - 1) / batch_size;
    std::vector<Field> fields;
    for (auto &s : df.columns()) fields.push_back(Field{s.name(), s.dtype(), true});
    auto schema = std::make_shared<Schema>(fields);
    for (size_t b = 0; b < num_batches; ++b) {
        size_t start = b * batch_size, end = std::min((b + 1) * batch_size, num_rows);
        std::vector<ArrayRef> arrays;
        for (auto &s : df.columns()) {
            switch (s.dtype()) {
                case DataType::Int64: {
                    std::vector<int64_t> v;
                    for (size_t i = start; i < end; ++i) {
                        const AnyValue &a = s[i];
                        if (a.v.index() == 1) v.push_back(std::get<1>(a.v));
                        else if (a.is_null()) v.push_back(0);
                        else throw Panic("Type mismatch in Int64 series");
                    }
                    arrays.push_back(Int64Array::from_values(std::move(v)));
                    break;
                }
                case DataType::Float64: {
                    std::vector<double> v;
                    for (size_t i = start; i < end; ++i) {
                        const AnyValue &a = s[i];
                        if (a.v.index() == 2) v.push_back(std::get<2>(a.v));
                        else if (a.is_null()) v.push_back(0.0);
                        else throw Panic("Type mismatch in Float64 series");  // reference defect 4
                    }
                    arrays.push_back(Float64Array::from_values(std::move(v)));
                    break;
                }
                case DataType::String: {
                    std::vector<std::optional<std::string>> v;
                    for (size_t i = start; i < end; ++i) {
                        const AnyValue &a = s[i];
                        if (a.v.index() == 3) v.push_back(std::get<3>(a.v));
                        else if (a.is_null()) v.push_back(std::nullopt);
                        else throw Panic("Type mismatch in String series");
                    }
                    arrays.push_back(std::make_shared<StringArray>(v));
                    break;
                }
                case DataType::Boolean: {
                    std::vector<bool> v;
                    for (size_t i = start; i < end; ++i) {
                        const AnyValue &a = s[i];
                        if (a.v.index() == 4) v.push_back(std::get<4>(a.v));
                        else if (a.is_null()) v.push_back(false);
                        else throw Panic("Type mismatch in Boolean series");
                    }
                    arrays.push_back(BooleanArray::from_bools(v));
                    break;
                }
                case DataType::Null: arrays.push_back(std::make_shared<NullArray>(end - start)); break;
            }
        }
        try {
            batches.push_back(RecordBatch::try_new(schema, std::move(arrays)));
        } catch (const Err &e) {
            throw StreamingExecutionError(StreamingExecutionError::Conversion,
                                          std::string("Conversion error: ") + e.what());
        }
    }
    return batches;
}

struct StreamingPhysicalPlan;
using StreamingPlanPtr = std::shared_ptr<const StreamingPhysicalPlan>;
struct StreamingPhysicalPlan {
    enum Kind { MemorySource, DataFrameSource, Filter, Select, Limit } kind;
    std::vector<RecordBatch> batches;  // MemorySource
    DataFrame df;
    size_t batch_size = 1024;
    StreamingPlanPtr input;
    std::string predicate_column;
    std::vector<std::string> columns;
    size_t n = 0;

    DataStreamRef execute() const {  // streaming.rs:71-133
        try {
            switch (kind) {
                case MemorySource: {
                    if (batches.empty())
                        throw StreamingExecutionError(StreamingExecutionError::InvalidOperation,
                                                      "Invalid operation: Cannot create stream from empty batch list");
                    return std::make_unique<MemoryStream>(batches[0].schema(), batches);
                }
                case DataFrameSource: {
                    auto b = dataframe_to_batches(df, batch_size);
                    SchemaRef s = b.empty() ? std::make_shared<Schema>() : b[0].schema();
                    return std::make_unique<MemoryStream>(s, std::move(b));
                }
                case Filter: return std::make_unique<FilterStream>(input->execute(), predicate_column);
                case Select: return std::make_unique<SelectStream>(input->execute(), columns);
                case Limit: return std::make_unique<LimitStream>(input->execute(), n);
            }
        } catch (const StreamError &e) {
            throw StreamingExecutionError(StreamingExecutionError::Stream, std::string("Stream error: ") + e.what());
        }
        throw Panic("unreachable");
    }

    RecordBatch collect() const {  // streaming.rs:235-238 + :343-352
        auto stream = execute();
        try {
            auto schema = stream->schema();
            auto all = stream->collect();
            if (all.empty()) return RecordBatch::empty(schema);
            try {
                return RecordBatch::concat(all);
            } catch (const Err &e) {
                throw StreamingExecutionError(StreamingExecutionError::Conversion,
                                              std::string("Conversion error: ") + e.what());
            }
        } catch (const StreamError &e) {
            throw StreamingExecutionError(StreamingExecutionError::Stream, std::string("Stream error: ") + e.what());
        }
    }
};

// streaming_planner.rs:102-135 (alias name dropped :110-113)
inline std::vector<std::string> extract_column_names_from_expressions(const std::vector<Expr> &exprs) {
    std::vector<std::string> out;
    for (auto &e : exprs) {
        if (e.kind == Expr::Column) out.push_back(e.name);
        else if (e.kind == Expr::Alias) {
            if (e.left->kind == Expr::Column) out.push_back(e.left->name);
            else
                throw StreamingPlannerError(StreamingPlannerError::ExpressionError,
                                            "Expression conversion error: Complex expressions with aliases not yet "
                                            "supported");
        } else
            throw StreamingPlannerError(StreamingPlannerError::ExpressionError,
                                        "Expression conversion error: Complex expressions not yet supported in "
                                        "streaming mode");
    }
    return out;
}

// streaming_planner.rs:137-168: the reference rejects every BinaryExpr
inline std::string extract_boolean_predicate_column(const Expr &p) {
    if (p.kind == Expr::Column) return p.name;
    if (p.kind == Expr::Binary) {
        if (p.left->kind == Expr::Column)
            throw StreamingPlannerError(
                StreamingPlannerError::ExpressionError,
                "Expression conversion error: Binary expressions not yet supported in streaming mode. Found "
                "expression on column '" +
                    p.left->name +
                    "'. Currently only simple boolean column references are supported (e.g., "
                    ".filter(col('is_active')))");
        throw StreamingPlannerError(StreamingPlannerError::ExpressionError,
                                    "Expression conversion error: Complex binary expressions not supported in "
                                    "streaming mode");
    }
    throw StreamingPlannerError(StreamingPlannerError::ExpressionError,
                                "Expression conversion error: Unsupported filter expression type");
}

// streaming_planner.rs:29-100
inline StreamingPlanPtr logical_to_streaming(const LogicalPlanPtr &l) {
    auto out = std::make_shared<StreamingPhysicalPlan>();
    switch (l->kind) {
        case LogicalPlan::DataFrameSource:
            out->kind = StreamingPhysicalPlan::DataFrameSource;
            out->df = l->df;
            out->batch_size = 1024;  // :32
            return out;
        case LogicalPlan::Select:
            out->kind = StreamingPhysicalPlan::Select;
            out->input = logical_to_streaming(l->input);
            out->columns = extract_column_names_from_expressions(l->expressions);
            return out;
        case LogicalPlan::Filter:
            out->kind = StreamingPhysicalPlan::Filter;
            out->input = logical_to_streaming(l->input);
            out->predicate_column = extract_boolean_predicate_column(l->predicate);
            return out;
        case LogicalPlan::Limit:
            out->kind = StreamingPhysicalPlan::Limit;
            out->input = logical_to_streaming(l->input);
            out->n = l->n;
            return out;
    }
    throw Panic("unreachable");
}

// ---------------------------------------------------------------------------
// LazyFrame -- logical_plan/builder.rs:12-114 ; QueryError :16-24
// ---------------------------------------------------------------------------
struct QueryError : std::runtime_error {
    enum Kind { LogicalPlan, Execution, StreamingPlanner } kind;
    std::string column;  // for LogicalPlanError::ColumnNotFound
    QueryError(Kind k, const std::string &m, std::string col = "") : std::runtime_error(m), kind(k), column(std::move(col)) {}
};

class LazyFrame {
  public:
    static LazyFrame from_dataframe(const DataFrame &df) {  // builder.rs:27-40
        auto p = std::make_shared<LogicalPlan>();
        p->kind = LogicalPlan::DataFrameSource;
        p->df = df;
        for (auto &s : df.columns()) p->source_schema.emplace_back(s.name(), s.dtype());
        return LazyFrame(p);
    }
    LazyFrame select(std::vector<Expr> exprs) const {
        auto p = std::make_shared<LogicalPlan>();
        p->kind = LogicalPlan::Select;
        p->input = plan_;
        p->expressions = std::move(exprs);
        return LazyFrame(p);
    }
    LazyFrame filter(Expr pred) const {
        auto p = std::make_shared<LogicalPlan>();
        p->kind = LogicalPlan::Filter;
        p->input = plan_;
        p->predicate = std::move(pred);
        return LazyFrame(p);
    }
    LazyFrame limit(size_t n) const {
        auto p = std::make_shared<LogicalPlan>();
        p->kind = LogicalPlan::Limit;
        p->input = plan_;
        p->n = n;
        return LazyFrame(p);
    }
    const LogicalPlanPtr &logical_plan() const { return plan_; }

    DataFrame collect() const {  // builder.rs:96-104
        auto opt = QueryOptimizer::optimize(plan_);
        try {
            opt->validate();
        } catch (const LogicalPlanError &e) {
            throw QueryError(QueryError::LogicalPlan, std::string("Logical plan error: ") + e.what(), e.name);
        }
        try {
            return logical_to_physical(opt)->execute();
        } catch (const ConversionError &e) {
            throw QueryError(QueryError::Execution, std::string("Execution error: ") + e.what());
        } catch (const ExecutionError &e) {
            throw QueryError(QueryError::Execution, std::string("Execution error: ") + e.what());
        }
    }
    RecordBatch collect_streaming() const {  // builder.rs:106-113
        auto opt = QueryOptimizer::optimize(plan_);
        try {
            opt->validate();
        } catch (const LogicalPlanError &e) {
            throw QueryError(QueryError::LogicalPlan, std::string("Logical plan error: ") + e.what(), e.name);
        }
        StreamingPlanPtr sp;
        try {
            sp = logical_to_streaming(opt);
        } catch (const StreamingPlannerError &e) {
            throw QueryError(QueryError::StreamingPlanner, std::string("Streaming planner error: ") + e.what());
        }
        try {
            return sp->collect();
        } catch (const StreamingExecutionError &e) {
            throw QueryError(QueryError::Execution, std::string("Execution error: ") + e.what());
        }
    }

  private:
    explicit LazyFrame(LogicalPlanPtr p) : plan_(std::move(p)) {}
    LogicalPlanPtr plan_;
};

}  // namespace rvo
