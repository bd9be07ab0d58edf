// Tests of the C++ host layer (rivulus_amd/host/rivulus_host.hpp).  They re-express the
// reference's own unit tests for this path (file:line in each case) against device-resident
// arrays, and compare values with the CPU oracle where the reference only checks shapes.
//   host_tests --cpu   planner / lowering logic only (no device)
//   host_tests         everything (needs an MI355X)
// Output: "ok <name>" / "FAIL <name>: why"; exit status 0 iff all pass.
#include <unistd.h>
#include <fstream>
#include <cmath>
#include <cstdio>
#include <functional>

#include "../../oracle/oracle_compose.hpp"  // checker only
#include "../../oracle/oracle_csv.hpp"      // checker only
#include "../../rivulus_amd/host/rivulus_host.hpp"

using namespace rivulus;
using namespace rivulus::execution;
using namespace rivulus::expressions;
using namespace rivulus::physical_plan;

namespace {
struct Case {
    const char *name;
    bool needs_gpu;
    std::function<void()> fn;
};
std::vector<Case> &cases() {
    static std::vector<Case> c;
    return c;
}
struct Reg {
    Reg(const char *n, bool g, std::function<void()> f) { cases().push_back({n, g, std::move(f)}); }
};
struct Fail : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define GPU_TEST(name) \
    static void name(); \
    static Reg reg_##name(#name, true, name); \
    static void name()
#define CPU_TEST(name) \
    static void name(); \
    static Reg reg_##name(#name, false, name); \
    static void name()
#define CHECK(cond) \
    do { \
        if (!(cond)) throw Fail(std::string(__FILE__ ":") + std::to_string(__LINE__) + " CHECK(" #cond ")"); \
    } while (0)
template <class E, class F>
bool throws(F f) {
    try {
        f();
    } catch (const E &) {
        return true;
    }
    return false;
}
template <class F>
std::string error_text(F f) {
    try {
        f();
    } catch (const std::exception &e) {
        return e.what();
    }
    return "<no error>";
}

ContextRef g_ctx;
const ContextRef &ctx() {
    if (!g_ctx) g_ctx = std::make_shared<Context>(0);
    return g_ctx;
}
using OB = std::optional<bool>;
const OB N = std::nullopt;

// the reference fixtures with the String column replaced by a Float64 one (strings are
// outside the device path): id / score / active
SchemaRef test_schema() {
    return std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}, {"score", DataType::Float64, true}, {"active", DataType::Boolean, false}});
}
RecordBatch test_batch() {  // record_batch.rs:594-604 shape: 3 rows, one null in column 1
    return RecordBatch::try_new(test_schema(), {Int64Array::from_values(ctx(), {1, 2, 3}),
                                                Float64Array::create(ctx(), {85.5, 0.0, 78.5}, std::vector<bool>{true, false, true}),
                                                BooleanArray::from_bools(ctx(), {true, false, true})});
}
RecordBatch stream_batch(int64_t id) {  // stream.rs:236-250: 2 rows, active = [true, false]
    return RecordBatch::try_new(test_schema(), {Int64Array::from_values(ctx(), {id, id + 1}),
                                                Float64Array::from_values(ctx(), {id * 1.5, id * 2.5}),
                                                BooleanArray::from_bools(ctx(), {true, false})});
}
std::optional<int64_t> i64_at(const ArrayRef &a, size_t i) { return std::dynamic_pointer_cast<const Int64Array>(a)->value(i); }
std::optional<double> f64_at(const ArrayRef &a, size_t i) { return std::dynamic_pointer_cast<const Float64Array>(a)->value(i); }
std::optional<bool> bool_at(const ArrayRef &a, size_t i) { return std::dynamic_pointer_cast<const BooleanArray>(a)->value(i); }

// oracle twin of a device batch (for value-level comparison)
rvo::ArrayRef to_oracle(const ArrayRef &a) {
    const size_t n = a->len();
    std::optional<std::vector<bool>> valid;
    if (a->has_null_bitmap()) valid = std::vector<bool>(n);
    switch (a->data_type()) {
        case DataType::Int64: {
            auto p = std::dynamic_pointer_cast<const Int64Array>(a);
            std::vector<int64_t> v(n);
            for (size_t i = 0; i < n; ++i) {
                v[i] = p->raw_value(i);
                if (valid) (*valid)[i] = p->value(i).has_value();
            }
            return std::make_shared<rvo::Int64Array>(v, valid);
        }
        case DataType::Float64: {
            auto p = std::dynamic_pointer_cast<const Float64Array>(a);
            std::vector<double> v(n);
            for (size_t i = 0; i < n; ++i) {
                v[i] = p->raw_value(i);
                if (valid) (*valid)[i] = p->value(i).has_value();
            }
            return std::make_shared<rvo::Float64Array>(v, valid);
        }
        case DataType::String: {
            auto p = std::dynamic_pointer_cast<const StringArray>(a);
            std::vector<std::optional<std::string>> v(n);
            for (size_t i = 0; i < n; ++i) v[i] = p->value(i);
            return std::make_shared<rvo::StringArray>(v);
        }
        default: {
            auto p = std::dynamic_pointer_cast<const BooleanArray>(a);
            std::vector<std::optional<bool>> v(n);
            for (size_t i = 0; i < n; ++i) v[i] = p->value(i);
            return rvo::BooleanArray::make(v);
        }
    }
}
bool same(const ArrayRef &dev, const rvo::ArrayRef &ora) {
    if (dev->len() != ora->len() || dev->null_count() != ora->null_count()) return false;
    if (dev->has_null_bitmap() != (ora->null_count() > 0)) return false;  // bitmap dropped when no null (primitive.rs:179-185)
    for (size_t i = 0; i < dev->len(); ++i) {
        switch (dev->data_type()) {
            case DataType::String: {
                auto o = std::static_pointer_cast<const rvo::StringArray>(ora);
                if (std::dynamic_pointer_cast<const StringArray>(dev)->value(i) != o->value(i)) return false;
                break;
            }
            case DataType::Int64: {
                auto o = std::static_pointer_cast<const rvo::Int64Array>(ora);
                if (i64_at(dev, i) != o->value(i) || std::dynamic_pointer_cast<const Int64Array>(dev)->raw_value(i) != o->values()[i]) return false;
                break;
            }
            case DataType::Float64: {
                auto o = std::static_pointer_cast<const rvo::Float64Array>(ora);
                const double a = std::dynamic_pointer_cast<const Float64Array>(dev)->raw_value(i), b = o->values()[i];
                if (f64_at(dev, i).has_value() != o->value(i).has_value() || std::memcmp(&a, &b, 8) != 0) return false;
                break;
            }
            default:
                if (bool_at(dev, i) != std::static_pointer_cast<const rvo::BooleanArray>(ora)->value(i)) return false;
        }
    }
    return true;
}
}  // namespace

// ============================ planners (no device) ============================
CPU_TEST(planner_convert_filter_predicate) {  // planner.rs:134-189
    auto t = convert_filter_predicate(Expr::col("age").gte(Expr::lit(30)));
    CHECK(t.column == "age" && t.op == RV_GE && std::get<int64_t>(t.literal) == 30);
    auto kind = [](const Expr &e) {
        try {
            convert_filter_predicate(e);
        } catch (const ConversionError &c) {
            return static_cast<int>(c.kind);
        }
        return -1;
    };
    auto cmp = Expr::col("a").gt(Expr::lit(1));
    CHECK(kind(cmp.and_(cmp)) == ConversionError::UnsupportedFilter);
    CHECK(kind(Expr::col("a").add(Expr::lit(1))) == ConversionError::UnsupportedFilterOperator);
    CHECK(kind(Expr::lit(1).gt(Expr::lit(1))) == ConversionError::FilterLeftNotColumn);
    CHECK(kind(Expr::col("a").gt(Expr::col("b"))) == ConversionError::FilterRightNotLiteral);
    CHECK(kind(Expr::col("a")) == ConversionError::InvalidFilterStructure);
    auto s = convert_select_expr(Expr::col("age").alias("years"));
    CHECK(s.first == "age" && s.second == "years");
}
CPU_TEST(planner_reference_streaming_grammar) {  // streaming_planner.rs:102-168, :331-381
    CHECK(extract_boolean_predicate_column(Expr::col("active")) == "active");
    CHECK(error_text([] { extract_boolean_predicate_column(Expr::col("age").gt(Expr::lit(30))); }).find("Binary expressions not yet supported") != std::string::npos);
    CHECK(throws<StreamingPlannerError>([] { extract_column_names_from_expressions({Expr::col("age").add(Expr::lit(10))}); }));
    auto names = extract_column_names_from_expressions({Expr::col("name"), Expr::col("city").alias("location")});
    CHECK(names.size() == 2 && names[1] == "city");  // alias dropped (:110-113)
}
CPU_TEST(planner_lower_predicate) {  // the lowering that replaces the rejection above
    auto lp = lower_predicate(Expr::col("f").gt(Expr::lit(Literal(0.5))).and_(Expr::col("x").lt(Expr::lit(200))).and_(Expr::col("active")));
    const auto &t = lp.terms;
    CHECK(t.size() == 3 && t[0].column == "f" && t[0].op == RV_GT && t[1].op == RV_LT && t[2].op == RV_IS_TRUE);
    CHECK(lp.expr.empty());  // an AND tree is the plain term list
    // BinaryOperator::Or (expr.rs:28): (a > 1 OR b > 1) AND c  ->  a b OR c AND in postfix
    auto lo = lower_predicate(Expr::col("a").gt(Expr::lit(1)).or_(Expr::col("b").gt(Expr::lit(1))).and_(Expr::col("c")));
    CHECK(lo.terms.size() == 3 && lo.terms[2].op == RV_IS_TRUE);
    CHECK((lo.expr == std::vector<uint8_t>{0, 1, RV_EXPR_OR, 2, RV_EXPR_AND}));
    CHECK(throws<StreamingPlannerError>([] { lower_predicate(Expr::col("a").add(Expr::lit(1))); }));
}
CPU_TEST(schema_basics) {  // schema.rs:1-76
    auto s = test_schema();
    CHECK(s->num_fields() == 3 && s->index_of("score") == 1u && !s->index_of("zzz") && s->field_by_name("active")->data_type() == DataType::Boolean);
    CHECK(*s == *test_schema() && Schema::empty().is_empty());
}

// ============================ arrays ============================
GPU_TEST(primitive_array_reference_vectors) {  // primitive.rs:228-306
    auto a = Int64Array::create(ctx(), {1, 2, 3, 4, 5}, std::vector<bool>{true, false, true, false, true});
    CHECK(a->len() == 5 && a->null_count() == 2 && a->has_null_bitmap());
    CHECK(a->value(0) == 1 && !a->value(1) && a->value(2) == 3 && !a->value(3) && a->value(4) == 5);
    auto b = Int64Array::create(ctx(), {1, 2, 3, 4, 5, 6}, std::vector<bool>{true, false, true, false, true, false});
    auto s = std::dynamic_pointer_cast<const Int64Array>(b->slice(2, 3));
    CHECK(s->len() == 3 && s->value(0) == 3 && !s->value(1) && s->value(2) == 5 && s->null_count() == 1);
    auto e = Int64Array::from_values(ctx(), {});
    CHECK(e->len() == 0 && e->null_count() == 0 && !e->has_null_bitmap());
    CHECK(throws<Panic>([&] { b->slice(4, 3); }));
    CHECK(throws<Panic>([&] { b->value(6); }));
}
GPU_TEST(boolean_array_reference_vectors) {  // boolean.rs:625-690
    auto a = BooleanArray::create(ctx(), {true, false, true, N, false});
    auto b = BooleanArray::create(ctx(), {true, true, false, true, N});
    auto r = a->logical_and(*b);
    CHECK(r->value(0) == OB(true) && r->value(1) == OB(false) && r->value(2) == OB(false) && !r->value(3) && !r->value(4));
    auto o = a->logical_or(*BooleanArray::create(ctx(), {false, true, false, true, N}));
    CHECK(o->value(0) == OB(true) && o->value(1) == OB(true) && o->value(2) == OB(true) && !o->value(3) && !o->value(4));
    auto n = BooleanArray::create(ctx(), {true, false, N, true})->logical_not();
    CHECK(n->value(0) == OB(false) && n->value(1) == OB(true) && !n->value(2) && n->value(3) == OB(false));
    auto c = BooleanArray::create(ctx(), {true, false, true, N, false, true});
    CHECK(c->count_true() == 3 && c->count_false() == 2);
    CHECK(error_text([&] { BooleanArray::create(ctx(), {true, false})->logical_and(*BooleanArray::create(ctx(), {true})); }) ==
          "Array lengths must match for logical operations");
    CHECK(!BooleanArray::from_bools(ctx(), {true, false})->has_null_bitmap());
}

// ============================ RecordBatch ============================
GPU_TEST(record_batch_try_new) {  // record_batch.rs:606-644
    auto b = test_batch();
    CHECK(b.num_rows() == 3 && b.num_columns() == 3 && *b.schema() == *test_schema());
    auto bad = std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}, {"score", DataType::Boolean, true}});
    CHECK(throws<Error>([&] { RecordBatch::try_new(bad, {Int64Array::from_values(ctx(), {1, 2, 3}), Float64Array::from_values(ctx(), {1.0, 2.0, 3.0})}); }));
    CHECK(throws<Error>([&] {
        RecordBatch::try_new(test_schema(), {Int64Array::from_values(ctx(), {1, 2, 3}), Float64Array::from_values(ctx(), {1.0}), BooleanArray::from_bools(ctx(), {true, false, true})});
    }));
    CHECK(b.column_by_name("id") && !b.column_by_name("nonexistent"));
    CHECK(throws<Panic>([&] { b.column(5); }));
}
GPU_TEST(record_batch_slice) {  // record_batch.rs:699-749
    auto b = test_batch();
    auto s = b.slice(1, 2);
    CHECK(s.num_rows() == 2 && i64_at(s.column(0), 0) == 2 && i64_at(s.column(0), 1) == 3);
    CHECK(b.slice(1, 0).is_empty() && b.slice(0, 3).num_rows() == 3 && b.slice(2, 1).num_rows() == 1);
    CHECK(throws<Panic>([&] { b.slice(2, 5); }));
}
GPU_TEST(record_batch_take) {  // record_batch.rs:751-791
    auto b = test_batch();
    auto t = b.take({2, 0, 1});
    CHECK(t.num_rows() == 3 && i64_at(t.column(0), 0) == 3 && i64_at(t.column(0), 1) == 1 && i64_at(t.column(0), 2) == 2);
    CHECK(!f64_at(t.column(1), 2) && std::dynamic_pointer_cast<const Float64Array>(t.column(1))->raw_value(2) == 0.0);  // placeholder
    CHECK(b.take({}).is_empty());
    CHECK(error_text([&] { b.take({0, 5, 1}); }) == "Index 5 out of bounds for 3 rows");
    auto no_null = b.take({0, 2});
    CHECK(!no_null.column(1)->has_null_bitmap());  // no null survived -> bitmap dropped
}
GPU_TEST(record_batch_select_columns) {  // record_batch.rs:793-819
    auto b = test_batch();
    auto s = b.select_columns({0, 2});
    CHECK(s.num_columns() == 2 && s.schema()->field(0).name() == "id" && s.schema()->field(1).name() == "active");
    auto n = b.select_columns_by_name({"score", "id"});
    CHECK(n.schema()->field(0).name() == "score" && n.schema()->field(1).name() == "id");
    CHECK(error_text([&] { b.select_columns_by_name({"zzz"}); }) == "Column 'zzz' not found");
}
GPU_TEST(record_batch_filter) {  // record_batch.rs:821-879
    auto b = test_batch();
    auto f = b.filter(BooleanArray::from_bools(ctx(), {true, false, true}));
    CHECK(f.num_rows() == 2 && f.num_columns() == 3 && i64_at(f.column(0), 0) == 1 && i64_at(f.column(0), 1) == 3);
    CHECK(b.filter(BooleanArray::all_true(ctx(), 3)).num_rows() == 3);
    CHECK(b.filter(BooleanArray::all_false(ctx(), 3)).is_empty());
    CHECK(b.filter(BooleanArray::create(ctx(), {true, N, false})).num_rows() == 1);  // nulls treated as false (:237)
    CHECK(error_text([&] { b.filter(BooleanArray::from_bools(ctx(), {true, false, true, true})); }) == "Predicate length 4 doesn't match batch length 3");
    CHECK(error_text([&] { b.filter(Int64Array::from_values(ctx(), {1, 0, 1})); }) == "Predicate must be a BooleanArray");
}
GPU_TEST(record_batch_concat) {  // record_batch.rs:881-949
    auto c = RecordBatch::concat({stream_batch(1), stream_batch(3)});
    CHECK(c.num_rows() == 4 && c.num_columns() == 3);
    for (int i = 0; i < 4; ++i) CHECK(i64_at(c.column(0), i) == i + 1);
    auto e = RecordBatch::concat({RecordBatch::empty(ctx(), test_schema()), RecordBatch::empty(ctx(), test_schema())});
    CHECK(e.num_rows() == 0 && e.is_empty());
    auto s1 = std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}});
    auto s2 = std::make_shared<Schema>(std::vector<Field>{{"x", DataType::Float64, false}});
    CHECK(throws<Error>([&] { RecordBatch::concat({RecordBatch::empty(ctx(), s1), RecordBatch::empty(ctx(), s2)}); }));
    CHECK(error_text([] { RecordBatch::concat({}); }) == "Cannot concatenate empty batch list");
}
GPU_TEST(record_batch_string_columns_on_device) {  // record_batch.rs:594-604 fixture; take/filter/concat of strings :163-170, :277-342
    auto schema = std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}, {"name", DataType::String, true}});
    auto names = StringArray::create(ctx(), {"Alice", std::nullopt, "Charlie"});
    CHECK(names->len() == 3 && names->null_count() == 1 && names->has_null_bitmap() && names->total_bytes() == 12);
    CHECK(*names->value(0) == "Alice" && !names->value(1) && *names->value(2) == "Charlie");
    auto b = RecordBatch::try_new(schema, {Int64Array::from_values(ctx(), {1, 2, 3}), names});
    CHECK(b.slice(1, 2).num_rows() == 2 && b.select_columns_by_name({"id"}).num_columns() == 1);
    auto f = b.filter(BooleanArray::from_bools(ctx(), {true, false, true}));  // record_batch.rs:821-840 with a String column
    CHECK(f.num_rows() == 2);
    auto fn = std::dynamic_pointer_cast<const StringArray>(f.column(1));
    CHECK(fn && *fn->value(0) == "Alice" && *fn->value(1) == "Charlie" && !fn->has_null_bitmap());  // no null survived
    auto t = b.take({2, 1, 1, 0});  // record_batch.rs:751-770
    auto tn = std::dynamic_pointer_cast<const StringArray>(t.column(1));
    CHECK(tn && *tn->value(0) == "Charlie" && !tn->value(1) && !tn->value(2) && *tn->value(3) == "Alice" && tn->null_count() == 2);
    auto c = RecordBatch::concat({b.slice(1, 2), b, f});  // record_batch.rs:881-922
    auto cn = std::dynamic_pointer_cast<const StringArray>(c.column(1));
    CHECK(c.num_rows() == 7 && cn && !cn->value(0) && *cn->value(1) == "Charlie" && *cn->value(2) == "Alice" && !cn->value(3) &&
          *cn->value(6) == "Charlie" && cn->total_bytes() == 7 + 12 + 12);
    auto e = RecordBatch::empty(ctx(), schema);
    CHECK(e.num_rows() == 0 && e.column(1)->data_type() == DataType::String);
}
GPU_TEST(record_batch_null_columns) {  // NullArray rides through take / filter / concat (record_batch.rs:176, :339)
    auto schema = std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}, {"nothing", DataType::Null, true}});
    auto b = RecordBatch::try_new(schema, {Int64Array::from_values(ctx(), {1, 2, 3, 4}), NullArray::create(ctx(), 4)});
    CHECK(b.column(1)->null_count() == 4 && b.column(1)->data_type() == DataType::Null);
    auto f = b.filter(BooleanArray::from_bools(ctx(), {true, false, true, true}));
    CHECK(f.num_rows() == 3 && f.column(1)->len() == 3 && f.column(1)->null_count() == 3);
    auto t = b.take({3, 3});
    CHECK(t.column(1)->len() == 2 && t.column(1)->data_type() == DataType::Null);
    auto c = RecordBatch::concat({f, t, b.slice(1, 2)});
    CHECK(c.num_rows() == 7 && c.column(1)->len() == 7 && c.column(1)->null_count() == 7);
    CHECK(RecordBatch::empty(ctx(), schema).column(1)->len() == 0);
}
GPU_TEST(dataframe_source_is_the_references_chunker) {  // streaming.rs:135-233, :85-94
    using namespace physical_plan;
    DeviceFrame df;
    df.names = {"id", "score", "name", "active"};
    df.columns = {Int64Array::create(ctx(), {1, 2, 3, 4, 5}, std::vector<bool>{true, false, true, true, true}),
                  Float64Array::create(ctx(), {1.5, 2.5, 3.5, 4.5, 5.5}, std::vector<bool>{true, true, false, true, true}),
                  StringArray::create(ctx(), {"a", std::nullopt, "c", "d", "e"}), BooleanArray::create(ctx(), {true, std::nullopt, false, true, true})};
    auto batches = dataframe_to_batches(df, 2);
    CHECK(batches.size() == 3 && batches[0].num_rows() == 2 && batches[2].num_rows() == 1);
    CHECK(batches[0].schema()->field(0).is_nullable() && batches[0].schema()->field(2).data_type() == DataType::String);
    auto id0 = std::dynamic_pointer_cast<const Int64Array>(batches[0].column(0));
    CHECK(!id0->has_null_bitmap() && *id0->value(0) == 1 && *id0->value(1) == 0);  // null -> 0, bitmap gone
    auto sc1 = std::dynamic_pointer_cast<const Float64Array>(batches[1].column(1));
    CHECK(!sc1->has_null_bitmap() && *sc1->value(0) == 0.0 && *sc1->value(1) == 4.5);
    auto nm0 = std::dynamic_pointer_cast<const StringArray>(batches[0].column(2));
    CHECK(*nm0->value(0) == "a" && !nm0->value(1));  // String keeps its nulls
    auto ac0 = std::dynamic_pointer_cast<const BooleanArray>(batches[0].column(3));
    CHECK(!ac0->has_null_bitmap() && *ac0->value(0) == true && *ac0->value(1) == false);
    // as a plan source: filter on the Boolean column, then collect (the null `active` of row 2 became false)
    auto out = StreamingPhysicalPlan::filter(StreamingPhysicalPlan::dataframe_source(df, 2), "active")->collect(ctx());
    CHECK(out.num_rows() == 3 && i64_at(out.column(0), 0) == 1 && i64_at(out.column(0), 1) == 4 && i64_at(out.column(0), 2) == 5);
    CHECK(StreamingPhysicalPlan::dataframe_source(DeviceFrame{}, 4)->collect_batches().empty());
}
GPU_TEST(config1_shape_on_the_device) {  // filter(age > 25).select([name]) with a String column riding along (plan.rs:504-525 data)
    using namespace physical_plan;
    DeviceFrame df;
    df.names = {"name", "age"};
    df.columns = {StringArray::from_strings(ctx(), {"Alice", "Bob", "Charlie"}), Int64Array::from_values(ctx(), {25, 30, 35})};
    auto plan = PhysicalPlan::select(PhysicalPlan::filter(PhysicalPlan::source(df), CompareTerm{"age", RV_GT, Literal(int64_t(25))}), {"name"}, {"name"});
    DeviceFrame out = plan->execute();
    CHECK(out.names.size() == 1 && out.names[0] == "name");
    auto n = std::dynamic_pointer_cast<const StringArray>(out.columns[0]);
    CHECK(n && n->len() == 2 && *n->value(0) == "Bob" && *n->value(1) == "Charlie");
}
// BASELINE configs[0] as SURVEY.md section 8d states it (1 000 rows, name = "n{i}", age = 18 + splitmix64(7 + i) % 50, three
// spellings), with every Expr taken through the reference's planner hooks -- convert_filter_predicate / convert_select_expr
// (planner.rs:134-189, :113-132), the Filter / Select arms of logical_to_physical (planner.rs:69-80) -- into the eager DEVICE
// plan, and compared with the oracle's LazyFrame::collect() on the same frame.
GPU_TEST(config1_thousand_rows_through_the_planner_hooks) {
    using namespace physical_plan;
    std::vector<std::string> names;
    std::vector<int64_t> ages;
    std::vector<rvo::AnyValue> onames, oages;
    for (uint64_t i = 0; i < 1000; ++i) {
        names.push_back("n" + std::to_string(i));
        ages.push_back(18 + static_cast<int64_t>(rvo::splitmix64(7 + i) % 50));
        onames.push_back(rvo::AnyValue(names.back()));
        oages.push_back(rvo::AnyValue(ages.back()));
    }
    DeviceFrame df;
    df.names = {"name", "age"};
    df.columns = {StringArray::from_strings(ctx(), names), Int64Array::from_values(ctx(), ages)};
    const Expr predicate = Expr::col("age").gt(Expr::lit(25));
    auto filter_of = [&](PhysicalPlanPtr in) { return PhysicalPlan::filter(std::move(in), convert_filter_predicate(predicate)); };
    auto select_of = [&](PhysicalPlanPtr in, const std::vector<Expr> &exprs) {
        std::vector<std::string> cols, finals;
        for (auto &e : exprs) {
            auto [c, f] = convert_select_expr(e);
            cols.push_back(c);
            finals.push_back(f);
        }
        return PhysicalPlan::select(std::move(in), cols, finals);
    };
    const rvo::DataFrame odf({rvo::Series("name", onames), rvo::Series("age", oages)});
    const auto opred = rvo::Expr::col("age").gt(rvo::Expr::lit(rvo::AnyValue(25)));

    // 1. select([name]).filter(age > 25): `age` is gone when the filter runs
    try {
        filter_of(select_of(PhysicalPlan::source(df), {Expr::col("name")}))->execute();
        CHECK(false);
    } catch (const ExecutionError &e) {
        CHECK(e.kind == ExecutionError::ColumnNotFound && std::string(e.what()) == "Column not found: 'age'");
    }
    CHECK(throws<rvo::QueryError>([&] { rvo::LazyFrame::from_dataframe(odf).select({rvo::Expr::col("name")}).filter(opred).collect(); }));
    // 2. select([name, age]).filter(age > 25)
    DeviceFrame a = filter_of(select_of(PhysicalPlan::source(df), {Expr::col("name"), Expr::col("age")}))->execute();
    auto oa = rvo::LazyFrame::from_dataframe(odf).select({rvo::Expr::col("name"), rvo::Expr::col("age")}).filter(opred).collect();
    // 3. filter(age > 25).select([name])
    DeviceFrame b = select_of(filter_of(PhysicalPlan::source(df)), {Expr::col("name")})->execute();
    auto ob = rvo::LazyFrame::from_dataframe(odf).filter(opred).select({rvo::Expr::col("name")}).collect();
    CHECK(a.width() == 2 && a.height() == oa.height() && oa.height() > 800 && b.width() == 1 && b.height() == ob.height() && b.names == ob.column_names());
    auto an = std::dynamic_pointer_cast<const StringArray>(a.columns[0]), bn = std::dynamic_pointer_cast<const StringArray>(b.columns[0]);
    auto aa = std::dynamic_pointer_cast<const Int64Array>(a.columns[1]);
    for (size_t i = 0; i < a.height(); ++i) {
        CHECK(rvo::any_eq((*oa.column("name"))[i], rvo::AnyValue(*an->value(i))) && rvo::any_eq((*oa.column("age"))[i], rvo::AnyValue(*aa->value(i))));
        CHECK(rvo::any_eq((*ob.column("name"))[i], rvo::AnyValue(*bn->value(i))));
    }
    // an aliased select renames (plan.rs:83-94) and an unsupported predicate is rejected by the hook, as in the reference
    DeviceFrame r = select_of(PhysicalPlan::source(df), {Expr::col("age").alias("years")})->execute();
    CHECK(r.names == std::vector<std::string>{"years"} && r.height() == 1000);
    CHECK(throws<ConversionError>([&] { convert_filter_predicate(predicate.and_(predicate)); }));
}
// ---- CsvFileStream (file_stream.rs:370-458) ------------------------------------------------------------
static std::string write_temp_csv(const std::string &name, const std::string &text) {
    const char *dir = std::getenv("TMPDIR");
    const std::string path = std::string(dir ? dir : "/tmp") + "/rivulus_host_" + name + "_" + std::to_string(::getpid()) + ".csv";
    std::ofstream f(path);
    f << text;
    return path;
}
static SchemaRef csv_schema() {  // create_test_schema, file_stream.rs:389-396
    return std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}, {"name", DataType::String, true},
                                                       {"score", DataType::Float64, true}, {"active", DataType::Boolean, false}});
}
static const char *kTestCsv = "id,name,score,active\n1,Alice,85.5,true\n2,Bob,92.0,false\n3,Charlie,78.5,true\n4,,90.0,false\n5,Eve,null,true\n";

GPU_TEST(csv_file_stream_basic_and_nulls) {  // file_stream.rs:398-415, :431-446
    const std::string path = write_temp_csv("basic", kTestCsv);
    CsvFileStream stream(ctx(), path, csv_schema(), 10, std::nullopt, CsvNulls::AsIntended);
    CHECK(*stream.schema() == *csv_schema());
    auto batch = stream.next_batch();
    CHECK(batch && batch->num_rows() == 5 && batch->num_columns() == 4);
    auto name = std::dynamic_pointer_cast<const StringArray>(batch->column(1));
    auto score = std::dynamic_pointer_cast<const Float64Array>(batch->column(2));
    auto id = std::dynamic_pointer_cast<const Int64Array>(batch->column(0));
    auto active = std::dynamic_pointer_cast<const BooleanArray>(batch->column(3));
    CHECK(*name->value(0) == "Alice" && !name->value(3) && *name->value(4) == "Eve" && name->null_count() == 1);  // row 3: null name
    CHECK(*score->value(0) == 85.5 && *score->value(3) == 90.0 && !score->value(4) && score->null_count() == 1);    // row 4: null score
    CHECK(score->raw_value(4) == 0.0);                                                                             // placeholder (:246-249)
    CHECK(!id->has_null_bitmap() && *id->value(4) == 5 && *active->value(1) == false && *active->value(4) == true);
    CHECK(!stream.next_batch());
    // the reference's own arrays -- the DEFAULT of the drop-in: `nulls` handed over as validity (file_stream.rs:236-243) -> inverted bitmap
    CsvFileStream ref(ctx(), path, csv_schema(), 10);
    auto rb = ref.next_batch();
    auto rscore = std::dynamic_pointer_cast<const Float64Array>(rb->column(2));
    CHECK(rscore->null_count() == 4 && !rscore->value(0) && *rscore->value(4) == 0.0);
    std::remove(path.c_str());
}
GPU_TEST(csv_adaptive_batch_size_and_empty_file) {  // file_stream.rs:417-429, :448-457
    CHECK(calculate_adaptive_batch_size(*csv_schema()) == 100000);
    const std::string path = write_temp_csv("empty", "id,name\n");
    auto schema = std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}, {"name", DataType::String, true}});
    CsvFileStream stream(ctx(), path, schema, 10);
    CHECK(!stream.next_batch());
    std::remove(path.c_str());
    CHECK(error_text([&] { CsvFileStream(ctx(), "/nonexistent/dir/x.csv", schema, 10); }).rfind("Failed to open file: ", 0) == 0);
}
GPU_TEST(csv_batches_blank_lines_delimiter_and_parse_errors) {
    std::string text = "id;name;score;active\n";
    for (int i = 0; i < 25; ++i) text += std::to_string(i) + " ; n" + std::to_string(i) + ";" + std::to_string(i * 0.5) + "; T \r\n" + (i % 5 == 0 ? "\n   \n" : "");
    const std::string path = write_temp_csv("batches", text);
    CsvFileStream stream(ctx(), path, csv_schema(), 10, ';');
    size_t rows = 0, batches = 0;
    while (auto b = stream.next_batch()) {
        CHECK(b->num_rows() == (batches < 2 ? 10u : 5u));
        auto id = std::dynamic_pointer_cast<const Int64Array>(b->column(0));
        CHECK(*id->value(0) == static_cast<int64_t>(rows));
        rows += b->num_rows();
        ++batches;
    }
    CHECK(rows == 25 && batches == 3);
    std::remove(path.c_str());
    const std::string bad = write_temp_csv("bad", "id,name,score,active\n1,A,1.5,true\nx2,B,2.5,false\n");
    CsvFileStream s2(ctx(), bad, csv_schema(), 10);
    CHECK(error_text([&] { s2.next_batch(); }) == "Stream execution error: Parse error: Line 3, field 0: Cannot parse 'x2' as Int64");
    std::remove(bad.c_str());
    const std::string shortl = write_temp_csv("short", "id,name,score,active\n1,A,1.5\n");
    CsvFileStream s3(ctx(), shortl, csv_schema(), 10);
    CHECK(error_text([&] { s3.next_batch(); }) == "Stream execution error: Parse error: Line 2: Expected 4 fields, found 3");
    std::remove(shortl.c_str());
}
GPU_TEST(csv_source_through_the_gpu_filter_project_plan) {  // CsvFileSource -> Filter -> Select (streaming.rs:95-105)
    using namespace physical_plan;
    const std::string path = write_temp_csv("plan", kTestCsv);
    const execution::LoweredPredicate pred{CompareTerm{"score", RV_GT, Literal(80.0)}, CompareTerm{"active", RV_EQ, Literal(true)}};
    auto plan = StreamingPhysicalPlan::gpu_filter_project(
        StreamingPhysicalPlan::csv_file_source(ctx(), path, csv_schema(), 2, std::nullopt, execution::CsvNulls::AsIntended), pred, {"name", "id"});
    RecordBatch out = plan->collect(ctx());
    auto name = std::dynamic_pointer_cast<const StringArray>(out.column(0));
    auto id = std::dynamic_pointer_cast<const Int64Array>(out.column(1));
    CHECK(out.num_rows() == 1 && *name->value(0) == "Alice" && *id->value(0) == 1);  // Eve's null score drops her (RV_NULL_DROPS)
    // the default reads the file the way the reference does: the batch that holds a null score comes out with its score
    // validity inverted (file_stream.rs:236-243), batches without a null are untouched -- Alice (batch 0) still qualifies
    auto as_ref = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::csv_file_source(ctx(), path, csv_schema(), 2), pred, {"name", "id"});
    {
        std::vector<rvo::Field> of{{"id", rvo::DataType::Int64, true}, {"name", rvo::DataType::String, true}, {"score", rvo::DataType::Float64, true},
                                   {"active", rvo::DataType::Boolean, true}};
        rvo::CsvFileStream ora(path, std::make_shared<rvo::Schema>(of), 2);
        size_t want_rows = 0;
        while (auto ob = ora.next_batch()) {
            auto sc = std::static_pointer_cast<const rvo::Float64Array>(ob->column(2));
            auto ac = std::static_pointer_cast<const rvo::BooleanArray>(ob->column(3));
            for (size_t i = 0; i < ob->num_rows(); ++i) {
                auto v = sc->value(i);
                auto a = ac->value(i);
                if (v && *v > 80.0 && a && *a) ++want_rows;
            }
        }
        CHECK(as_ref->collect(ctx()).num_rows() == want_rows);
    }
    std::remove(path.c_str());
    auto missing = StreamingPhysicalPlan::csv_file_source(ctx(), "/nonexistent/x.csv", csv_schema());
    CHECK(error_text([&] { missing->execute(); }).rfind("Invalid operation: Failed to open file: ", 0) == 0);
}
GPU_TEST(eager_filter_string_equality) {  // plan.rs:527-547: name == "Bob" -> 1 row, every column kept
    using namespace physical_plan;
    DeviceFrame df;
    df.names = {"name", "age", "score"};
    df.columns = {StringArray::from_strings(ctx(), {"Alice", "Bob", "Charlie"}), Int64Array::from_values(ctx(), {25, 30, 35}),
                  Float64Array::from_values(ctx(), {85.5, 92.0, 78.5})};
    DeviceFrame out = PhysicalPlan::filter(PhysicalPlan::source(df), CompareTerm{"name", RV_EQ, Literal(std::string("Bob"))})->execute();
    CHECK(out.columns.size() == 3 && out.columns[0]->len() == 1);
    CHECK(*std::dynamic_pointer_cast<const StringArray>(out.columns[0])->value(0) == "Bob");
    CHECK(*std::dynamic_pointer_cast<const Int64Array>(out.columns[1])->value(0) == 30);
    // a String literal against an Int64 column is a cross-type compare: only != holds (series.rs:100-117)
    DeviceFrame none = PhysicalPlan::filter(PhysicalPlan::source(df), CompareTerm{"age", RV_EQ, Literal(std::string("30"))})->execute();
    DeviceFrame all = PhysicalPlan::filter(PhysicalPlan::source(df), CompareTerm{"age", RV_NE, Literal(std::string("30"))})->execute();
    CHECK(none.columns[0]->len() == 0 && all.columns[0]->len() == 3);
}
GPU_TEST(record_batch_large_filter_matches_oracle) {  // record_batch.rs:1075-1103 shape, value-checked
    const size_t n = 10000;
    std::vector<int64_t> ids(n);
    std::vector<double> sc(n);
    std::vector<bool> valid(n), act(n), pv(n), pvalid(n);
    for (size_t i = 0; i < n; ++i) {
        ids[i] = static_cast<int64_t>(i);
        sc[i] = std::sin(static_cast<double>(i));
        valid[i] = rvo::splitmix64(i) % 10 != 0;
        act[i] = rvo::splitmix64(i + 7) % 2;
        pv[i] = rvo::splitmix64(i + 99) % 3 == 0;
        pvalid[i] = rvo::splitmix64(i + 5) % 20 != 0;
    }
    std::vector<std::optional<bool>> pred(n);
    for (size_t i = 0; i < n; ++i) pred[i] = pvalid[i] ? OB(pv[i]) : N;
    auto b = RecordBatch::try_new(test_schema(), {Int64Array::from_values(ctx(), ids), Float64Array::create(ctx(), sc, valid), BooleanArray::from_bools(ctx(), act)});
    auto got = b.filter(BooleanArray::create(ctx(), pred));
    std::vector<rvo::ArrayRef> ocols{to_oracle(b.column(0)), to_oracle(b.column(1)), to_oracle(b.column(2))};
    auto exp = rvo::RecordBatch::try_new(rvo::positional_schema(ocols), ocols).filter(rvo::BooleanArray::make(pred));
    CHECK(got.num_rows() == exp.num_rows());
    for (size_t c = 0; c < 3; ++c) CHECK(same(got.column(c), exp.column(c)));
}

// ============================ streams ============================
GPU_TEST(memory_stream) {  // stream.rs:252-300
    MemoryStream s(test_schema(), {stream_batch(1), stream_batch(3)});
    CHECK(s.next_batch()->num_rows() == 2 && s.next_batch()->num_rows() == 2 && !s.next_batch());
    auto other = std::make_shared<Schema>(std::vector<Field>{{"x", DataType::Int64, false}});
    CHECK(throws<StreamError>([&] { MemoryStream bad(other, {stream_batch(1)}); }));
}
GPU_TEST(filter_stream_boolean_predicate) {  // stream.rs:375-432
    FilterStream f(MemoryStream::from_single_batch(stream_batch(1)), "active");
    auto r = f.next_batch();
    CHECK(r && r->num_rows() == 1 && i64_at(r->column(0), 0) == 1 && f.schema()->num_fields() == 3);
    auto none = RecordBatch::try_new(test_schema(), {Int64Array::from_values(ctx(), {1, 2}), Float64Array::from_values(ctx(), {1.0, 2.0}), BooleanArray::from_bools(ctx(), {false, false})});
    FilterStream g(MemoryStream::from_single_batch(none), "active");
    auto e = g.next_batch();
    CHECK(e && e->num_rows() == 0);  // the empty batch is still emitted (:156-158)
    FilterStream m(std::make_unique<MemoryStream>(test_schema(), std::vector<RecordBatch>{stream_batch(1), stream_batch(3)}), "active");
    CHECK(m.next_batch()->num_rows() == 1 && m.next_batch()->num_rows() == 1 && !m.next_batch());
    FilterStream miss(MemoryStream::from_single_batch(stream_batch(1)), "missing");
    CHECK(error_text([&] { miss.next_batch(); }) == "Stream execution error: Column 'missing' not found in schema");
    FilterStream notbool(MemoryStream::from_single_batch(stream_batch(1)), "id");
    CHECK(error_text([&] { notbool.next_batch(); }) == "Stream execution error: Predicate column 'id' is not of boolean type");
}
GPU_TEST(select_stream) {  // stream.rs:436-532
    SelectStream s(MemoryStream::from_single_batch(stream_batch(1)), {"active", "id"});
    auto r = s.next_batch();
    CHECK(r->num_columns() == 2 && r->num_rows() == 2 && r->schema()->field(0).name() == "active" && r->schema()->field(1).name() == "id");
    try {
        SelectStream bad(MemoryStream::from_single_batch(stream_batch(1)), {"nonexistent"});
        CHECK(false);
    } catch (const StreamError &e) {
        CHECK(e.kind == StreamError::Execution && e.message.find("nonexistent") != std::string::npos);
    }
    auto f = std::make_unique<FilterStream>(MemoryStream::from_single_batch(stream_batch(1)), "active");
    SelectStream chained(std::move(f), {"score"});
    auto c = chained.next_batch();
    CHECK(c && c->num_columns() == 1 && c->num_rows() == 1 && c->schema()->field(0).name() == "score");
}
GPU_TEST(limit_stream) {  // streaming.rs:464-498
    LimitStream a(std::make_unique<MemoryStream>(test_schema(), std::vector<RecordBatch>{stream_batch(1), stream_batch(3)}), 2);
    CHECK(a.next_batch()->num_rows() == 2 && !a.next_batch());
    LimitStream b(std::make_unique<MemoryStream>(test_schema(), std::vector<RecordBatch>{stream_batch(1), stream_batch(3)}), 3);
    CHECK(b.next_batch()->num_rows() == 2 && b.next_batch()->num_rows() == 1 && !b.next_batch());
}
GPU_TEST(streaming_plan_reference_cases) {  // streaming.rs:391-462
    using P = StreamingPhysicalPlan;
    auto src = [] { return P::memory_source({stream_batch(1), stream_batch(3)}); };
    CHECK(src()->collect(ctx()).num_rows() == 4);
    auto f = P::filter(src(), "active")->collect(ctx());
    CHECK(f.num_rows() == 2 && f.num_columns() == 3 && i64_at(f.column(0), 0) == 1 && i64_at(f.column(0), 1) == 3);
    auto s = P::select(src(), {"id", "score"})->collect(ctx());
    CHECK(s.num_rows() == 4 && s.num_columns() == 2 && s.schema()->field(0).name() == "id");
    CHECK(P::limit(src(), 3)->collect(ctx()).num_rows() == 3);
    auto c = P::limit(P::select(P::filter(src(), "active"), {"score"}), 1)->collect(ctx());
    CHECK(c.num_rows() == 1 && c.num_columns() == 1 && c.schema()->field(0).name() == "score");
    CHECK(throws<StreamingExecutionError>([&] { P::memory_source({})->collect(ctx()); }));
    CHECK(P::memory_source({stream_batch(1), stream_batch(3)})->collect_batches().size() == 2);  // :500-516
}

// CsvFileStream against the oracle's restatement of file_stream.rs on random files: CsvNulls::AsReference must give the
// reference's arrays bit for bit (its inverted validity included), CsvNulls::AsIntended the same cells with null == null.
GPU_TEST(csv_file_stream_matches_the_oracle_restatement) {
    const char *ints[] = {"0", "7", "-12", "+5", "9223372036854775807", "-9223372036854775808", "", "null", " 42 "};
    const char *floats[] = {"1.5", "-0.0", "1e3", ".5", "2.", "inf", "-Infinity", "NaN", "", "null", "  3.25", "1E-2"};
    const char *names[] = {"Alice", "Bob", "", "null", " padded ", "x y", "\xc3\x9cn\xc3\xaf"};
    const char *bools[] = {"true", "FALSE", "t", "f", "1", "0", "", "null", "True"};
    auto schema = csv_schema();
    auto oschema = std::make_shared<rvo::Schema>(std::vector<rvo::Field>{{"id", rvo::DataType::Int64, false}, {"name", rvo::DataType::String, true},
                                                                         {"score", rvo::DataType::Float64, true}, {"active", rvo::DataType::Boolean, false}});
    for (uint64_t seed = 0; seed < 12; ++seed) {
        uint64_t z = seed * 1000;
        auto pick = [&](size_t n) { return static_cast<size_t>(rvo::splitmix64(++z) % n); };
        std::string text = "id,name,score,active\n";
        const size_t rows = 1 + pick(400);
        const bool crlf = seed % 3 == 1, nulls_free = seed % 4 == 3;
        for (size_t r = 0; r < rows; ++r) {
            if (pick(17) == 0) text += crlf ? "   \r\n" : "\n";  // blank lines are skipped, not counted
            std::string line = std::string(ints[pick(nulls_free ? 6 : 9)]) + "," + names[pick(7)] + "," + floats[pick(nulls_free ? 8 : 12)] + "," + bools[pick(nulls_free ? 6 : 9)];
            text += line + (crlf ? "\r\n" : "\n");
        }
        const std::string path = write_temp_csv("fuzz" + std::to_string(seed), text);
        const size_t batch_size = 1 + pick(64);
        CsvFileStream asref(ctx(), path, schema, batch_size), fixed(ctx(), path, schema, batch_size, std::nullopt, CsvNulls::AsIntended);  // default == the reference
        rvo::CsvFileStream ora(path, oschema, batch_size);
        for (;;) {
            auto a = asref.next_batch();
            auto f = fixed.next_batch();
            auto o = ora.next_batch();
            CHECK(a.has_value() == o.has_value() && f.has_value() == o.has_value());
            if (!o) break;
            CHECK(a->num_rows() == o->num_rows() && f->num_rows() == o->num_rows());
            for (size_t c = 0; c < 4; ++c) {
                const auto &oc = o->column(c);
                const bool obitmap = c == 0   ? std::static_pointer_cast<const rvo::Int64Array>(oc)->null_bitmap() != nullptr
                                     : c == 2 ? std::static_pointer_cast<const rvo::Float64Array>(oc)->null_bitmap() != nullptr
                                              : oc->null_count() > 0;
                CHECK(a->column(c)->has_null_bitmap() == obitmap && a->column(c)->null_count() == oc->null_count());
                for (size_t i = 0; i < o->num_rows(); ++i) {
                    if (c == 0) {
                        auto oi = std::static_pointer_cast<const rvo::Int64Array>(oc);
                        CHECK(i64_at(a->column(c), i) == oi->value(i));
                        // AsIntended: the same raw cells, null exactly where the reference's `nulls` flag was true
                        auto fi = std::dynamic_pointer_cast<const Int64Array>(f->column(c));
                        CHECK(fi->raw_value(i) == oi->values()[i]);
                        if (obitmap) CHECK(fi->value(i).has_value() == !oi->value(i).has_value());
                    } else if (c == 2) {
                        auto of = std::static_pointer_cast<const rvo::Float64Array>(oc);
                        const double x = std::dynamic_pointer_cast<const Float64Array>(a->column(c))->raw_value(i), y = of->values()[i];
                        CHECK(std::memcmp(&x, &y, 8) == 0 && f64_at(a->column(c), i).has_value() == of->value(i).has_value());
                        if (obitmap) CHECK(f64_at(f->column(c), i).has_value() == !of->value(i).has_value());
                    }
                }
                if (c == 1 || c == 3) CHECK(same(a->column(c), oc) && same(f->column(c), oc));  // String / Boolean: right in the reference
            }
        }
        unlink(path.c_str());
    }
    // a malformed cell: the same error text from both, raised by the batch that holds the line
    const std::string bad = write_temp_csv("badcell", "id,name,score,active\n1,a,1.0,true\n2,b,zz,false\n");
    CsvFileStream h(ctx(), bad, schema, 10);
    rvo::CsvFileStream o(bad, oschema, 10);
    CHECK(error_text([&] { h.next_batch(); }) == error_text([&] { o.next_batch(); }));
    unlink(bad.c_str());
}

// ============================ the new operator: compare / AND lowering ============================
GPU_TEST(gpu_filter_project_stream_matches_composed_oracle) {  // BASELINE config 3 shape through seam S1
    const size_t n = 50000, batch_rows = 1024;  // 1024 == the reference's streaming batch size (streaming_planner.rs:32)
    std::vector<double> f(n);
    std::vector<int64_t> x(n);
    std::vector<bool> vf(n), vx(n);
    for (size_t i = 0; i < n; ++i) {
        f[i] = static_cast<double>(rvo::splitmix64(43 + i) >> 11) * 0x1.0p-53;
        x[i] = static_cast<int64_t>(rvo::splitmix64(42 + i) % 1000);
        vf[i] = rvo::splitmix64(44 + i) % 100 >= 5;
        vx[i] = rvo::splitmix64(45 + i) % 100 >= 5;
    }
    auto schema = std::make_shared<Schema>(std::vector<Field>{{"f", DataType::Float64, true}, {"x", DataType::Int64, true}});
    auto whole = RecordBatch::try_new(schema, {Float64Array::create(ctx(), f, vf), Int64Array::create(ctx(), x, vx)});
    std::vector<RecordBatch> batches;
    for (size_t off = 0; off < n; off += batch_rows) batches.push_back(whole.slice(off, std::min(batch_rows, n - off)));  // zero-copy slices

    // collect_streaming() of  filter((f > 0.5) AND (x < 200)).select([f, x])
    auto terms = lower_predicate(Expr::col("f").gt(Expr::lit(Literal(0.5))).and_(Expr::col("x").lt(Expr::lit(200))));
    auto plan = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::memory_source(batches), terms, {"f", "x"});
    auto got = plan->collect(ctx());

    std::vector<rvo::ArrayRef> ocols{std::make_shared<rvo::Float64Array>(f, vf), std::make_shared<rvo::Int64Array>(x, vx)};
    auto exp = rvo::stream_filter_project(ocols, batch_rows, {{0, rvo::TermOp::Gt, rvo::AnyValue(0.5)}, {1, rvo::TermOp::Lt, rvo::AnyValue(200)}},
                                          rvo::NullPolicy::Drops, {0, 1});
    CHECK(got.num_rows() == exp.num_rows() && got.num_rows() > 4000);
    CHECK(same(got.column(0), exp.column(0)) && same(got.column(1), exp.column(1)));
}
GPU_TEST(gpu_filter_project_over_a_dataframe_source_is_fused) {  // Select(Filter(DataFrameSource)) -> rv_filter_project_chunked
    using namespace physical_plan;
    const size_t n = 50007, batch_rows = 1024;
    std::vector<double> f(n);
    std::vector<int64_t> x(n);
    std::vector<bool> vf(n), vx(n);
    std::vector<std::optional<std::string>> names(n);
    for (size_t i = 0; i < n; ++i) {
        f[i] = static_cast<double>(rvo::splitmix64(43 + i) >> 11) * 0x1.0p-53;
        x[i] = static_cast<int64_t>(rvo::splitmix64(42 + i) % 1000);
        vf[i] = rvo::splitmix64(44 + i) % 100 >= 5;
        vx[i] = rvo::splitmix64(45 + i) % 100 >= 5;
        if (rvo::splitmix64(46 + i) % 10 != 0) names[i] = "n" + std::to_string(i % 37);
    }
    DeviceFrame df;
    df.names = {"f", "x", "name"};
    df.columns = {Float64Array::create(ctx(), f, vf), Int64Array::create(ctx(), x, vx), StringArray::create(ctx(), names)};
    auto pred = lower_predicate(Expr::col("f").gt(Expr::lit(Literal(0.5))).and_(Expr::col("x").lt(Expr::lit(200))));
    const std::vector<std::string> sel{"x", "name", "f"};
    auto fused = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::dataframe_source(df, batch_rows), pred, sel);
    CHECK(dynamic_cast<GpuChunkedFilterProjectStream *>(fused->execute().get()) != nullptr);
    auto fb = fused->collect_batches();
    // the unfused pipeline over the chunker's batches, batch by batch
    auto ub = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::memory_source(dataframe_to_batches(df, batch_rows)), pred, sel)->collect_batches();
    CHECK(fb.size() == ub.size() && fb.size() == (n + batch_rows - 1) / batch_rows);
    size_t total = 0;
    for (size_t b = 0; b < fb.size(); ++b) {
        CHECK(fb[b].num_rows() == ub[b].num_rows() && fb[b].schema()->field(1).name() == "name");
        for (size_t c = 0; c < 3; ++c) CHECK(same(fb[b].column(c), to_oracle(ub[b].column(c))));
        total += fb[b].num_rows();
    }
    // and the oracle on the null-filled columns (dataframe_to_batches: null cells of Int64 / Float64 become 0 / 0.0)
    std::vector<double> ff(f);
    std::vector<int64_t> xf(x);
    for (size_t i = 0; i < n; ++i) {
        if (!vf[i]) ff[i] = 0.0;
        if (!vx[i]) xf[i] = 0;
    }
    std::vector<rvo::ArrayRef> ocols{std::make_shared<rvo::Float64Array>(ff, std::nullopt), std::make_shared<rvo::Int64Array>(xf, std::nullopt),
                                     std::make_shared<rvo::StringArray>(names)};
    auto exp = rvo::stream_filter_project(ocols, batch_rows, {{0, rvo::TermOp::Gt, rvo::AnyValue(0.5)}, {1, rvo::TermOp::Lt, rvo::AnyValue(200)}},
                                          rvo::NullPolicy::Drops, {1, 2, 0});
    auto got = fused->collect(ctx());
    CHECK(got.num_rows() == exp.num_rows() && got.num_rows() == total && total > 4000);
    for (size_t c = 0; c < 3; ++c) CHECK(same(got.column(c), exp.column(c)));
    // batch sizes that do not divide the table, a window smaller than the table, Limit behind the fused operator
    auto odd = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::dataframe_source(df, 1000), pred, sel)->collect_batches();
    CHECK(odd.size() == 51 && odd.back().num_rows() <= 7);
    auto lim = StreamingPhysicalPlan::limit(fused, 10)->collect(ctx());
    CHECK(lim.num_rows() == 10 && same(lim.column(0), to_oracle(got.column(0)->slice(0, 10))));
    // errors as from the unfused operator
    auto error_of = [&](StreamingPlanPtr source) -> std::string {
        try {
            StreamingPhysicalPlan::gpu_filter_project(std::move(source), pred, {"zzz"})->collect(ctx());
        } catch (const StreamingExecutionError &e) {
            return e.what();
        }
        return "no error";
    };
    const std::string fused_error = error_of(StreamingPhysicalPlan::dataframe_source(df, batch_rows));
    CHECK(fused_error == "Stream error: Stream execution error: Column 'zzz' not found in schema");
    CHECK(fused_error == error_of(StreamingPhysicalPlan::memory_source(dataframe_to_batches(df, batch_rows))));
}
GPU_TEST(limit_is_pushed_down_into_the_device_streams) {  // streaming.rs:246-288 downstream of the fused operator
    using namespace physical_plan;
    // a resident table of 2e8 rows (rv_generate: x = splitmix64(42 + i) % 1000), 1024-row batches
    const uint64_t n = 200000000;
    rv_synth_spec spec{};
    spec.dtype = RV_INT64;
    spec.seed = 42;
    spec.length = n;
    spec.modulus = 1000;
    rv_dcolumn *h = nullptr;
    check(rv_generate(ctx()->raw(), &spec, &h));
    DeviceFrame df;
    df.names = {"x"};
    df.columns = {Array::adopt(ctx(), h)};
    auto scanned = [&] {
        int64_t v = 0;
        check(rv_ctx_get_option(ctx()->raw(), "fused_rows_scanned", &v));
        return static_cast<uint64_t>(v);
    };
    auto first_survivors = [&](int64_t lit, bool eq, size_t k) {  // the first k surviving values, from the generator on the host
        std::vector<int64_t> out;
        for (uint64_t i = 0; i < n && out.size() < k; ++i) {
            const int64_t v = static_cast<int64_t>(rvo::splitmix64(42 + i) % 1000);
            if (eq ? v == lit : v > lit) out.push_back(v);
        }
        return out;
    };
    // limit(10) behind filter(x > 899): the first window is sized for 10 rows, not 2^28
    auto pred = lower_predicate(Expr::col("x").gt(Expr::lit(899)));
    auto plan = StreamingPhysicalPlan::limit(StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::dataframe_source(df, 1024), pred, {"x"}), 10);
    uint64_t before = scanned();
    auto got = plan->collect(ctx());
    const uint64_t touched = scanned() - before;
    CHECK(got.num_rows() == 10 && same(got.column(0), std::make_shared<rvo::Int64Array>(first_survivors(899, false, 10), std::nullopt)));
    CHECK(touched > 0 && touched < 10000000);  // < 1e7 of 2e8 rows (the unhinted operator filters 2^28 per refill)
    // ... also through a Select between the Limit and the operator, and through the handle-array stream
    auto plan2 = StreamingPhysicalPlan::limit(
        StreamingPhysicalPlan::select(StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::dataframe_source(df, 1024), pred, {"x"}), {"x"}), 10);
    before = scanned();
    CHECK(plan2->collect(ctx()).num_rows() == 10 && scanned() - before < 10000000);
    // a miss doubles the window: x == 5 keeps 0.1 %, the estimate before the first window comes from the 10 % query above
    auto rare = lower_predicate(Expr::col("x").eq(Expr::lit(5)));
    auto plan3 = StreamingPhysicalPlan::limit(StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::dataframe_source(df, 1024), rare, {"x"}), 3000);
    before = scanned();
    auto got3 = plan3->collect(ctx());
    CHECK(got3.num_rows() == 3000 && same(got3.column(0), std::make_shared<rvo::Int64Array>(first_survivors(5, true, 3000), std::nullopt)));
    CHECK(scanned() - before < 30000000);  // ~3e6 rows hold 3000 matches; a few doublings, far from the whole table
    // without a limit the whole table is filtered (and the hint is only a hint: pulling past it keeps working)
    auto s = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::dataframe_source(df, 1 << 20), pred, {"x"})->execute();
    s->limit_hint(5);
    size_t rows = 0;
    while (auto b = s->next_batch()) rows += b->num_rows();
    CHECK(rows > 19900000 && rows < 20100000);
}
GPU_TEST(a_failed_refill_leaves_the_chunked_stream_where_it_was) {
    using namespace physical_plan;
    std::vector<int64_t> x(5000);
    for (size_t i = 0; i < x.size(); ++i) x[i] = static_cast<int64_t>(i % 10);
    DeviceFrame df;
    df.names = {"x"};
    df.columns = {Int64Array::from_values(ctx(), x)};
    auto s = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::dataframe_source(df, 1024), lower_predicate(Expr::col("x").gt(Expr::lit(7))), {"x"})
                 ->execute();
    check(rv_ctx_set_option(ctx()->raw(), "inject_failure", 1));
    CHECK(throws<StreamError>([&] { s->next_batch(); }));  // the very first window fails ...
    size_t rows = 0, batches = 0;
    while (auto b = s->next_batch()) rows += b->num_rows(), ++batches;  // ... and the stream starts over, nothing half-updated
    CHECK(rows == 1000 && batches == 5);
}
GPU_TEST(gpu_filter_project_stream_or_of_compares) {  // BinaryOperator::Or through seam S1, strict nulls (boolean.rs:137-152)
    const size_t n = 30000, batch_rows = 1024;
    std::vector<double> f(n);
    std::vector<int64_t> x(n);
    std::vector<bool> vf(n), vx(n);
    std::vector<OB> act(n);
    for (size_t i = 0; i < n; ++i) {
        f[i] = static_cast<double>(rvo::splitmix64(43 + i) >> 11) * 0x1.0p-53;
        x[i] = static_cast<int64_t>(rvo::splitmix64(42 + i) % 1000);
        vf[i] = rvo::splitmix64(44 + i) % 100 >= 5;
        vx[i] = rvo::splitmix64(45 + i) % 100 >= 5;
        act[i] = rvo::splitmix64(46 + i) % 10 == 0 ? N : OB(rvo::splitmix64(47 + i) % 3 == 0);
    }
    auto schema = std::make_shared<Schema>(std::vector<Field>{{"f", DataType::Float64, true}, {"x", DataType::Int64, true}, {"active", DataType::Boolean, true}});
    auto whole = RecordBatch::try_new(schema, {Float64Array::create(ctx(), f, vf), Int64Array::create(ctx(), x, vx), BooleanArray::create(ctx(), act)});
    std::vector<RecordBatch> batches;
    for (size_t off = 0; off < n; off += batch_rows) batches.push_back(whole.slice(off, std::min(batch_rows, n - off)));
    // filter((f > 0.9 OR x < 50) AND active).select([x, f])
    auto pred = lower_predicate(Expr::col("f").gt(Expr::lit(Literal(0.9))).or_(Expr::col("x").lt(Expr::lit(50))).and_(Expr::col("active")));
    auto got = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::memory_source(batches), pred, {"x", "f"})->collect(ctx());
    std::vector<rvo::ArrayRef> ocols{std::make_shared<rvo::Float64Array>(f, vf), std::make_shared<rvo::Int64Array>(x, vx),
                                     rvo::BooleanArray::make(act)};
    auto exp = rvo::stream_filter_project(ocols, batch_rows,
                                          {{0, rvo::TermOp::Gt, rvo::AnyValue(0.9)}, {1, rvo::TermOp::Lt, rvo::AnyValue(50)}, {2, rvo::TermOp::IsTrue, rvo::AnyValue(true)}},
                                          rvo::NullPolicy::Drops, {1, 0}, pred.expr);
    CHECK(got.num_rows() == exp.num_rows() && got.num_rows() > 500);
    CHECK(same(got.column(0), exp.column(0)) && same(got.column(1), exp.column(1)));
}
GPU_TEST(gpu_filter_project_stream_empty_and_errors) {
    auto plan = StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::memory_source({stream_batch(1), stream_batch(3)}),
                                                          {CompareTerm{"id", RV_GT, Literal(int64_t(100))}}, {"id", "active"});
    auto r = plan->collect(ctx());
    CHECK(r.num_rows() == 0 && r.num_columns() == 2);  // every batch still emitted, concat of empties
    CHECK(throws<StreamingExecutionError>([&] {
        StreamingPhysicalPlan::gpu_filter_project(StreamingPhysicalPlan::memory_source({stream_batch(1)}), {CompareTerm{"zzz", RV_GT, Literal(int64_t(1))}}, {"id"})->collect(ctx());
    }));
}

// ============================ eager PhysicalPlan over typed device columns ============================
DeviceFrame people() {  // plan.rs:295-327 without the String column: age / score
    return DeviceFrame{{"age", "score"}, {Int64Array::from_values(ctx(), {25, 30, 35}), Float64Array::from_values(ctx(), {85.5, 92.0, 78.5})}};
}
GPU_TEST(eager_filter_gt) {  // plan.rs:504-525
    auto r = PhysicalPlan::filter(PhysicalPlan::source(people()), CompareTerm{"age", RV_GT, Literal(int64_t(25))})->execute();
    CHECK(r.height() == 2 && r.width() == 2 && i64_at(*r.column("age"), 0) == 30 && i64_at(*r.column("age"), 1) == 35);
}
GPU_TEST(eager_filter_lt_float_and_no_match) {  // plan.rs:549-589
    auto r = PhysicalPlan::filter(PhysicalPlan::source(people()), CompareTerm{"score", RV_LT, Literal(90.0)})->execute();
    CHECK(r.height() == 2 && i64_at(*r.column("age"), 0) == 25 && i64_at(*r.column("age"), 1) == 35);
    auto none = PhysicalPlan::filter(PhysicalPlan::source(people()), CompareTerm{"age", RV_GT, Literal(int64_t(100))})->execute();
    CHECK(none.height() == 0 && none.width() == 2 && none.names == std::vector<std::string>({"age", "score"}));
}
GPU_TEST(eager_errors_and_chain) {  // plan.rs:591-612, :706-769
    try {
        PhysicalPlan::filter(PhysicalPlan::source(people()), CompareTerm{"nonexistent", RV_EQ, Literal(int64_t(0))})->execute();
        CHECK(false);
    } catch (const ExecutionError &e) {
        CHECK(e.kind == ExecutionError::ColumnNotFound && std::string(e.what()) == "Column not found: 'nonexistent'");
    }
    auto r = PhysicalPlan::select(PhysicalPlan::filter(PhysicalPlan::source(people()), CompareTerm{"age", RV_GE, Literal(int64_t(30))}), {"score"}, {"points"})->execute();
    CHECK(r.height() == 2 && r.width() == 1 && r.names[0] == "points" && f64_at(r.columns[0], 0) == 92.0 && f64_at(r.columns[0], 1) == 78.5);
    CHECK(throws<ExecutionError>([&] { PhysicalPlan::select(PhysicalPlan::source(people()), {"zzz"}, {"zzz"})->execute(); }));
}
GPU_TEST(eager_nulls_sort_lowest) {  // plan.rs:112-130 + series.rs:105-107: <, <=, != keep null rows
    DeviceFrame df{{"v"}, {Int64Array::create(ctx(), {5, 0, 50}, std::vector<bool>{true, false, true})}};
    auto lt = PhysicalPlan::filter(PhysicalPlan::source(df), CompareTerm{"v", RV_LT, Literal(int64_t(10))})->execute();
    CHECK(lt.height() == 2 && i64_at(lt.columns[0], 0) == 5 && !i64_at(lt.columns[0], 1));  // 5 and the null
    CHECK(PhysicalPlan::filter(PhysicalPlan::source(df), CompareTerm{"v", RV_GT, Literal(int64_t(10))})->execute().height() == 1);
    CHECK(PhysicalPlan::filter(PhysicalPlan::source(df), CompareTerm{"v", RV_NE, Literal(int64_t(5))})->execute().height() == 2);
    CHECK(PhysicalPlan::filter(PhysicalPlan::source(df), CompareTerm{"v", RV_EQ, Literal()})->execute().height() == 1);  // == Null keeps the null row
}

int main(int argc, char **argv) {
    const bool cpu_only = argc > 1 && std::string(argv[1]) == "--cpu";
    int failed = 0, ran = 0;
    for (auto &c : cases()) {
        if (cpu_only && c.needs_gpu) continue;
        ++ran;
        try {
            c.fn();
            std::printf("ok %s\n", c.name);
        } catch (const std::exception &e) {
            std::printf("FAIL %s: %s\n", c.name, e.what());
            ++failed;
        }
    }
    g_ctx.reset();
    std::printf("%d cases, %d failed\n", ran, failed);
    return failed ? 1 : 0;
}
