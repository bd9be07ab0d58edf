// ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// Pins the CPU restatement against the known-answer vectors held by the reference's
// own inline unit tests.  Each case names the reference test it
// re-expresses (file:line).  Output: one "ok <name>" / "FAIL <name>: why" line per case;
// exit status 0 iff all pass.  Run by tests/test_oracle_kat.py.
#include <cmath>
#include <cstdio>
#include <functional>
#include <limits>

#include <unistd.h>

#include "oracle_compose.hpp"
#include "oracle_csv.hpp"

using namespace rvo;

namespace {
struct Case {
    const char *name;
    std::function<void()> fn;
};
std::vector<Case> &cases() {
    static std::vector<Case> c;
    return c;
}
struct Reg {
    Reg(const char *n, std::function<void()> f) { cases().push_back({n, std::move(f)}); }
};
struct Fail : std::runtime_error {
    using std::runtime_error::runtime_error;
};
#define KAT(name) \
    static void name(); \
    static Reg reg_##name(#name, name); \
    static void name()
#define CHECK(cond) \
    do { \
        if (!(cond)) throw Fail(std::string(__FILE__ ":") + std::to_string(__LINE__) + " CHECK(" #cond ")"); \
    } while (0)
template <class F>
bool panics(F f) {
    try {
        f();
    } catch (const Panic &) {
        return true;
    } catch (const std::out_of_range &) {
        return true;
    }
    return false;
}
template <class E, class F>
bool throws(F f) {
    try {
        f();
    } catch (const E &) {
        return true;
    }
    return false;
}

std::vector<bool> pattern(size_t len) {  // bitmap.rs:207-210
    std::vector<bool> v(len);
    for (size_t i = 0; i < len; ++i) v[i] = ((i % 3 == 0) != (i % 5 == 0));
    return v;
}
size_t ones_in(const std::vector<bool> &v, size_t a, size_t b) {
    size_t c = 0;
    for (size_t i = a; i < b; ++i) c += v[i];
    return c;
}
using OB = std::optional<bool>;
const OB N = std::nullopt;

// --- fixtures ------------------------------------------------------------------
SchemaRef rb_schema() {  // record_batch.rs:586-592, stream.rs:228-234
    return std::make_shared<Schema>(std::vector<Field>{
        {"id", DataType::Int64, false}, {"name", DataType::String, true}, {"active", DataType::Boolean, false}});
}
std::vector<ArrayRef> rb_columns() {  // record_batch.rs:594-604
    return {Int64Array::from_values({1, 2, 3}),
            std::make_shared<StringArray>(std::vector<std::optional<std::string>>{"Alice", std::nullopt, "Charlie"}),
            BooleanArray::from_bools({true, false, true})};
}
RecordBatch rb_batch() { return RecordBatch::try_new(rb_schema(), rb_columns()); }
RecordBatch stream_batch(int64_t id) {  // stream.rs:236-250, streaming.rs:375-389
    return RecordBatch::try_new(
        rb_schema(),
        {Int64Array::from_values({id, id + 1}),
         std::make_shared<StringArray>(std::vector<std::optional<std::string>>{
             "name_" + std::to_string(id), "name_" + std::to_string(id + 1)}),
         BooleanArray::from_bools({true, false})});
}
DataFrame people() {  // plan.rs:295-327 (also planner.rs:200-232, builder.rs:128-160)
    return DataFrame({Series("name", {AnyValue("Alice"), AnyValue("Bob"), AnyValue("Charlie")}),
                      Series("age", {AnyValue(25), AnyValue(30), AnyValue(35)}),
                      Series("score", {AnyValue(85.5), AnyValue(92.0), AnyValue(78.5)})});
}
DataFrame people_active() {  // streaming_planner.rs:176-209
    return DataFrame({Series("name", {AnyValue("Alice"), AnyValue("Bob"), AnyValue("Charlie")}),
                      Series("age", {AnyValue(25), AnyValue(30), AnyValue(35)}),
                      Series("active", {AnyValue(true), AnyValue(false), AnyValue(true)})});
}
PhysicalPlanPtr src(const DataFrame &df) {
    auto p = std::make_shared<PhysicalPlan>();
    p->kind = PhysicalPlan::DataFrameSource;
    p->df = df;
    return p;
}
PhysicalPlanPtr filt(PhysicalPlanPtr in, const std::string &col, AnyValue v, BinaryOperator op) {
    auto p = std::make_shared<PhysicalPlan>();
    p->kind = PhysicalPlan::Filter;
    p->input = std::move(in);
    p->column = col;
    p->value = std::move(v);
    p->op = op;
    return p;
}
PhysicalPlanPtr sel(PhysicalPlanPtr in, std::vector<std::string> cols) {
    auto p = std::make_shared<PhysicalPlan>();
    p->kind = PhysicalPlan::Select;
    p->input = std::move(in);
    p->columns = cols;
    p->final_names = cols;
    return p;
}
PhysicalPlanPtr lim(PhysicalPlanPtr in, size_t n) {
    auto p = std::make_shared<PhysicalPlan>();
    p->kind = PhysicalPlan::Limit;
    p->input = std::move(in);
    p->n = n;
    return p;
}
StreamingPlanPtr mem_source(std::vector<RecordBatch> b) {
    auto p = std::make_shared<StreamingPhysicalPlan>();
    p->kind = StreamingPhysicalPlan::MemorySource;
    p->batches = std::move(b);
    return p;
}
StreamingPlanPtr s_filter(StreamingPlanPtr in, const std::string &c) {
    auto p = std::make_shared<StreamingPhysicalPlan>();
    p->kind = StreamingPhysicalPlan::Filter;
    p->input = std::move(in);
    p->predicate_column = c;
    return p;
}
StreamingPlanPtr s_select(StreamingPlanPtr in, std::vector<std::string> c) {
    auto p = std::make_shared<StreamingPhysicalPlan>();
    p->kind = StreamingPhysicalPlan::Select;
    p->input = std::move(in);
    p->columns = std::move(c);
    return p;
}
StreamingPlanPtr s_limit(StreamingPlanPtr in, size_t n) {
    auto p = std::make_shared<StreamingPhysicalPlan>();
    p->kind = StreamingPhysicalPlan::Limit;
    p->input = std::move(in);
    p->n = n;
    return p;
}
std::optional<int64_t> i64_at(const ArrayRef &a, size_t i) {
    return std::dynamic_pointer_cast<const Int64Array>(a)->value(i);
}
}  // namespace

// ============================ BitMap (bitmap.rs tests) ============================
KAT(bitmap_new_is_zeroed) {  // bitmap.rs:212-228
    BitMap bm = BitMap::zeros(100);
    CHECK(bm.bit_count() == 100 && bm.offset() == 0 && bm.buffer()->size() == 13);
    for (size_t i = 0; i < 100; ++i) CHECK(!bm.get_bit(i));
    CHECK(bm.count_ones() == 0);
}
KAT(bitmap_roundtrip_unaligned_37) {  // bitmap.rs:230-241
    auto v = pattern(37);
    BitMap bm = BitMap::from_bools(v);
    CHECK(bm.bit_count() == 37 && bm.offset() == 0);
    for (size_t i = 0; i < 37; ++i) CHECK(bm.get_bit(i) == v[i]);
    CHECK(bm.count_ones() == ones_in(v, 0, 37));
}
KAT(bitmap_all_true_65) {  // bitmap.rs:243-252
    BitMap bm = BitMap::from_bools(std::vector<bool>(65, true));
    for (size_t i = 0; i < 65; ++i) CHECK(bm.get_bit(i));
    CHECK(bm.count_ones() == 65);
}
KAT(bitmap_slice_view_7_25) {  // bitmap.rs:254-281
    auto v = pattern(64);
    BitMap bm = BitMap::from_bools(v);
    BitMap s = bm.slice(7, 25);
    CHECK(s.buffer().get() == bm.buffer().get());
    CHECK(s.offset() == 7 && s.bit_count() == 25);
    for (size_t i = 0; i < 25; ++i) CHECK(s.get_bit(i) == v[7 + i]);
    CHECK(s.count_ones() == ones_in(v, 7, 32));
}
KAT(bitmap_chained_slice) {  // bitmap.rs:283-309
    auto v = pattern(91);
    BitMap bm = BitMap::from_bools(v);
    BitMap s1 = bm.slice(10, 50).slice(7, 20), s2 = bm.slice(17, 20);
    CHECK(s1.bit_count() == s2.bit_count() && s1.offset() == s2.offset());
    for (size_t i = 0; i < 20; ++i) CHECK(s1.get_bit(i) == s2.get_bit(i));
    CHECK(s1.count_ones() == s2.count_ones());
}
KAT(bitmap_zero_length_slice_and_panics) {  // bitmap.rs:311-334
    BitMap bm = BitMap::from_bools(pattern(13));
    BitMap s = bm.slice(5, 0);
    CHECK(s.bit_count() == 0 && s.count_ones() == 0);
    CHECK(panics([] { BitMap::zeros(16).slice(9, 8); }));
    CHECK(panics([] { BitMap::zeros(10).get_bit(10); }));
}
KAT(bitmap_byte_boundaries) {  // bitmap.rs:336-361
    std::vector<bool> v(17);
    for (size_t i = 0; i < 17; ++i) v[i] = (i == 7 || i == 8 || i == 16);
    BitMap bm = BitMap::from_bools(v);
    for (size_t i = 0; i < 17; ++i) CHECK(bm.get_bit(i) == v[i]);
    CHECK(bm.count_ones() == 3);
    BitMap s = bm.slice(6, 6);
    for (size_t i = 0; i < 6; ++i) CHECK(s.get_bit(i) == v[6 + i]);
    CHECK(s.count_ones() == ones_in(v, 6, 12));
    // layout: bit 7 -> byte0 0x80 ; bit 8 -> byte1 0x01 ; bit 16 -> byte2 0x01 (LSB first)
    CHECK((*bm.buffer())[0] == 0x80 && (*bm.buffer())[1] == 0x01 && (*bm.buffer())[2] == 0x01);
}
KAT(bitmap_all_true_tail_masked) {  // bitmap.rs:21-38
    BitMap bm = BitMap::all_true(11);
    CHECK((*bm.buffer())[0] == 0xFF && (*bm.buffer())[1] == 0x07);
}
KAT(bitmap_builder_flush_and_has_nulls) {  // bitmap.rs:142-188
    BitmapBuilder b;
    for (int i = 0; i < 10; ++i) b.append(true);
    CHECK(!b.has_nulls());
    b.append(false);
    CHECK(b.has_nulls());
    BitMap bm = b.finish();
    CHECK(bm.bit_count() == 11 && bm.buffer()->size() == 2 && (*bm.buffer())[1] == 0x03);
    BitmapBuilder e;
    CHECK(!e.has_nulls());
}

// ============================ PrimitiveArray ============================
KAT(primitive_some_nulls) {  // primitive.rs:228-244
    Int64Array a({1, 2, 3, 4, 5}, std::vector<bool>{true, false, true, false, true});
    CHECK(a.len() == 5 && a.null_count() == 2 && a.null_bitmap());
    CHECK(a.value(0) == 1 && !a.value(1) && a.value(2) == 3 && !a.value(3) && a.value(4) == 5);
}
KAT(primitive_all_nulls_and_empty) {  // primitive.rs:246-267
    Int64Array a({0, 0, 0}, std::vector<bool>{false, false, false});
    CHECK(a.null_count() == 3);
    for (size_t i = 0; i < 3; ++i) CHECK(!a.value(i));
    Int64Array e({}, std::nullopt);
    CHECK(e.len() == 0 && e.null_count() == 0 && !e.null_bitmap());
}
KAT(primitive_slice_2_3) {  // primitive.rs:283-306
    Int64Array a({1, 2, 3, 4, 5, 6}, std::vector<bool>{true, false, true, false, true, false});
    auto s = std::dynamic_pointer_cast<const Int64Array>(a.slice(2, 3));
    CHECK(s->len() == 3 && s->offset() == 2);
    CHECK(s->value(0) == 3 && !s->value(1) && s->value(2) == 5);
    CHECK(s->null_count() == 1);
    CHECK(panics([&] { a.slice(4, 3); }));
    CHECK(panics([&] { a.value(6); }));
}
KAT(primitive_builder) {  // primitive.rs:546-566
    PrimitiveArrayBuilder<int64_t> b;
    b.append_value(10);
    b.append_null(0);
    b.append_value(20);
    b.append_value(30);
    b.append_null(0);
    auto a = b.finish();
    CHECK(a->len() == 5 && a->null_count() == 2);
    CHECK(a->value(0) == 10 && !a->value(1) && a->value(2) == 20 && a->value(3) == 30 && !a->value(4));
    CHECK(a->values()[1] == 0 && a->values()[4] == 0);
}
KAT(primitive_builder_f64_every_third_null) {  // primitive.rs:568-586
    PrimitiveArrayBuilder<double> b;
    for (int i = 0; i < 50; ++i) {
        if (i % 3 == 0) b.append_null(0.0);
        else b.append_value(i * 1.5);
    }
    auto a = b.finish();
    CHECK(a->len() == 50 && a->data_type() == DataType::Float64 && a->null_count() == 17);
}
KAT(primitive_builder_no_nulls_drops_bitmap) {  // primitive.rs:588-604
    PrimitiveArrayBuilder<int64_t> b;
    for (int i = 1; i <= 5; ++i) b.append_value(i * 10);
    auto a = b.finish();
    CHECK(a->len() == 5 && a->null_count() == 0 && !a->null_bitmap());
    for (size_t i = 0; i < 5; ++i) CHECK(a->value(i) == static_cast<int64_t>((i + 1) * 10));
}

// ============================ BooleanArray ============================
KAT(boolean_and) {  // boolean.rs:625-638
    auto a = BooleanArray::make({true, false, true, N, false});
    auto b = BooleanArray::make({true, true, false, true, N});
    auto r = a->logical_and(*b);
    CHECK(r->value(0) == OB(true) && r->value(1) == OB(false) && r->value(2) == OB(false));
    CHECK(!r->value(3) && !r->value(4));  // null&&true = null ; false&&null = null (strict)
}
KAT(boolean_or) {  // boolean.rs:640-654
    auto a = BooleanArray::make({true, false, true, N, false});
    auto b = BooleanArray::make({false, true, false, true, N});
    auto r = a->logical_or(*b);
    CHECK(r->value(0) == OB(true) && r->value(1) == OB(true) && r->value(2) == OB(true));
    CHECK(!r->value(3) && !r->value(4));
}
KAT(boolean_not) {  // boolean.rs:656-666
    auto r = BooleanArray::make({true, false, N, true})->logical_not();
    CHECK(r->value(0) == OB(false) && r->value(1) == OB(true) && !r->value(2) && r->value(3) == OB(false));
}
KAT(boolean_count_true_false) {  // boolean.rs:668-682
    auto a = BooleanArray::make({true, false, true, N, false, true});
    CHECK(a->count_true() == 3 && a->count_false() == 2);
}
KAT(boolean_and_mismatched) {  // boolean.rs:684-690
    auto a = BooleanArray::make({true, false});
    auto b = BooleanArray::make({true});
    CHECK(throws<Err>([&] { a->logical_and(*b); }));
}
KAT(boolean_false_under_null_and_bitmap_drop) {  // boolean.rs:29-32, :36-40
    auto a = BooleanArray::make({true, N, true});
    CHECK(a->null_bitmap() && !a->values_bitmap()->get_bit(1));
    auto b = BooleanArray::from_bools({true, false});
    CHECK(!b->null_bitmap());
    CHECK(BooleanArray::make(std::vector<OB>(17, OB(true)))->total_bytes() == 3);  // boolean.rs:617-621
}

// ============================ RecordBatch ============================
KAT(rb_try_new_valid_and_errors) {  // record_batch.rs:606-644
    auto b = rb_batch();
    CHECK(b.num_rows() == 3 && b.num_columns() == 3 && *b.schema() == *rb_schema());
    auto bad_schema = std::make_shared<Schema>(
        std::vector<Field>{{"id", DataType::Int64, false}, {"name", DataType::Float64, true}});
    CHECK(throws<Err>([&] {
        RecordBatch::try_new(bad_schema,
                             {Int64Array::from_values({1, 2, 3}),
                              std::make_shared<StringArray>(std::vector<std::optional<std::string>>{"Alice"})});
    }));
    CHECK(throws<Err>([&] {
        RecordBatch::try_new(rb_schema(),
                             {Int64Array::from_values({1, 2, 3}),
                              std::make_shared<StringArray>(std::vector<std::optional<std::string>>{"Alice"}),
                              BooleanArray::from_bools({true, false, true})});
    }));
}
KAT(rb_empty) {  // record_batch.rs:646-655
    auto b = RecordBatch::empty(std::make_shared<Schema>());
    CHECK(b.num_rows() == 0 && b.num_columns() == 0 && b.is_empty());
}
KAT(rb_column_access) {  // record_batch.rs:657-697
    auto b = rb_batch();
    CHECK(b.column(0)->data_type() == DataType::Int64 && b.column(1)->data_type() == DataType::String &&
          b.column(2)->data_type() == DataType::Boolean);
    CHECK(b.column_by_name("id") && b.column_by_name("active") && !b.column_by_name("nonexistent"));
    CHECK(panics([&] { b.column(5); }));
}
KAT(rb_slice) {  // record_batch.rs:699-749
    auto b = rb_batch();
    auto s = b.slice(1, 2);
    CHECK(s.num_rows() == 2 && s.num_columns() == 3);
    CHECK(i64_at(s.column(0), 0) == 2 && i64_at(s.column(0), 1) == 3);
    CHECK(b.slice(1, 0).is_empty() && b.slice(0, 3).num_rows() == 3 && b.slice(2, 1).num_rows() == 1);
    CHECK(panics([&] { b.slice(2, 5); }));
}
KAT(rb_take) {  // record_batch.rs:751-791
    auto b = rb_batch();
    auto t = b.take({2, 0, 1});
    CHECK(t.num_rows() == 3 && t.num_columns() == 3);
    CHECK(i64_at(t.column(0), 0) == 3 && i64_at(t.column(0), 1) == 1 && i64_at(t.column(0), 2) == 2);
    CHECK(b.take({}).is_empty());
    try {
        b.take({0, 5, 1});
        CHECK(false);
    } catch (const Err &e) {
        CHECK(std::string(e.what()) == "Index 5 out of bounds for 3 rows");
    }
}
KAT(rb_select_columns) {  // record_batch.rs:793-819
    auto b = rb_batch();
    auto s = b.select_columns({0, 2});
    CHECK(s.num_rows() == 3 && s.num_columns() == 2 && s.schema()->field(0).name == "id" &&
          s.schema()->field(1).name == "active");
    auto n = b.select_columns_by_name({"name", "id"});
    CHECK(n.schema()->field(0).name == "name" && n.schema()->field(1).name == "id");
    try {
        b.select_columns_by_name({"zzz"});
        CHECK(false);
    } catch (const Err &e) {
        CHECK(std::string(e.what()) == "Column 'zzz' not found");
    }
}
KAT(rb_filter) {  // record_batch.rs:821-866
    auto b = rb_batch();
    auto f = b.filter(BooleanArray::from_bools({true, false, true}));
    CHECK(f.num_rows() == 2 && f.num_columns() == 3);
    CHECK(i64_at(f.column(0), 0) == 1 && i64_at(f.column(0), 1) == 3);
    CHECK(b.filter(BooleanArray::all_true(3)).num_rows() == 3);
    CHECK(b.filter(BooleanArray::all_false(3)).is_empty());
}
KAT(rb_filter_with_nulls) {  // record_batch.rs:868-879
    auto f = rb_batch().filter(BooleanArray::make({true, N, false}));
    CHECK(f.num_rows() == 1);
    CHECK(i64_at(f.column(0), 0) == 1);
}
KAT(rb_filter_length_and_type_errors) {  // record_batch.rs:222-233
    try {
        rb_batch().filter(BooleanArray::from_bools({true, false, true, true}));
        CHECK(false);
    } catch (const Err &e) {
        CHECK(std::string(e.what()) == "Predicate length 4 doesn't match batch length 3");
    }
    try {
        rb_batch().filter(Int64Array::from_values({1, 0, 1}));
        CHECK(false);
    } catch (const Err &e) {
        CHECK(std::string(e.what()) == "Predicate must be a BooleanArray");
    }
}
KAT(rb_concat) {  // record_batch.rs:881-949
    auto b1 = RecordBatch::try_new(
        rb_schema(), {Int64Array::from_values({1, 2}),
                      std::make_shared<StringArray>(std::vector<std::optional<std::string>>{"A", std::nullopt}),
                      BooleanArray::from_bools({true, false})});
    auto b2 = RecordBatch::try_new(rb_schema(),
                                   {Int64Array::from_values({3, 4}),
                                    std::make_shared<StringArray>(std::vector<std::optional<std::string>>{"B", "C"}),
                                    BooleanArray::from_bools({true, true})});
    auto c = RecordBatch::concat({b1, b2});
    CHECK(c.num_rows() == 4 && c.num_columns() == 3);
    for (int i = 0; i < 4; ++i) CHECK(i64_at(c.column(0), i) == i + 1);
    auto e = RecordBatch::concat({RecordBatch::empty(rb_schema()), RecordBatch::empty(rb_schema())});
    CHECK(e.num_rows() == 0 && e.is_empty());
    auto s1 = std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}});
    auto s2 = std::make_shared<Schema>(std::vector<Field>{{"name", DataType::String, false}});
    CHECK(throws<Err>([&] { RecordBatch::concat({RecordBatch::empty(s1), RecordBatch::empty(s2)}); }));
    try {
        RecordBatch::concat({});
        CHECK(false);
    } catch (const Err &e2) {
        CHECK(std::string(e2.what()) == "Cannot concatenate empty batch list");
    }
}
KAT(rb_take_null_placeholder_and_bitmap_drop) {  // record_batch.rs:142-146 + primitive.rs:179-185
    auto a = std::make_shared<Int64Array>(std::vector<int64_t>{7, 8, 9, 10},
                                          std::vector<bool>{true, false, true, false});
    auto schema = std::make_shared<Schema>(std::vector<Field>{{"v", DataType::Int64, true}});
    auto b = RecordBatch::try_new(schema, {a});
    auto t = std::dynamic_pointer_cast<const Int64Array>(b.take({1, 0}).column(0));
    CHECK(t->null_bitmap() && !t->value(0) && t->values()[0] == 0 && t->value(1) == 7);
    auto u = std::dynamic_pointer_cast<const Int64Array>(b.take({0, 2}).column(0));
    CHECK(!u->null_bitmap() && u->value(0) == 7 && u->value(1) == 9);  // no null survived -> bitmap dropped
}
KAT(rb_chain_slice_filter_select) {  // record_batch.rs:1116-1130 (chain of zero-copy + filter)
    auto b = rb_batch();
    auto r = b.slice(0, 3).filter(BooleanArray::from_bools({true, false, true})).select_columns({0});
    CHECK(r.num_rows() == 2 && r.num_columns() == 1);
}

// ============================ Streams ============================
KAT(stream_memory) {  // stream.rs:252-300
    MemoryStream s(rb_schema(), {stream_batch(1), stream_batch(3)});
    CHECK(*s.schema() == *rb_schema());
    CHECK(s.next_batch()->num_rows() == 2);
    CHECK(s.next_batch()->num_rows() == 2);
    CHECK(!s.next_batch());
    auto other = std::make_shared<Schema>(std::vector<Field>{{"x", DataType::Int64, false}});
    CHECK(throws<StreamError>([&] { MemoryStream bad(other, {stream_batch(1)}); }));
}
KAT(stream_filter_boolean_predicate) {  // stream.rs:375-387
    FilterStream f(MemoryStream::from_single_batch(stream_batch(1)), "active");
    auto r = f.next_batch();
    CHECK(r && r->num_rows() == 1);
    CHECK(f.schema()->num_fields() == 3);
}
KAT(stream_filter_no_matches_still_emits) {  // stream.rs:389-411
    auto b = RecordBatch::try_new(rb_schema(),
                                  {Int64Array::from_values({1, 2}),
                                   std::make_shared<StringArray>(std::vector<std::optional<std::string>>{"a", "b"}),
                                   BooleanArray::from_bools({false, false})});
    FilterStream f(MemoryStream::from_single_batch(b), "active");
    auto r = f.next_batch();
    CHECK(r && r->num_rows() == 0);
}
KAT(stream_filter_multiple_batches) {  // stream.rs:413-432
    FilterStream f(std::make_unique<MemoryStream>(rb_schema(), std::vector<RecordBatch>{stream_batch(1), stream_batch(3)}),
                   "active");
    CHECK(f.next_batch()->num_rows() == 1);
    CHECK(f.next_batch()->num_rows() == 1);
    CHECK(!f.next_batch());
}
KAT(stream_filter_errors) {  // stream.rs:139-154
    FilterStream f(MemoryStream::from_single_batch(stream_batch(1)), "missing");
    try {
        f.next_batch();
        CHECK(false);
    } catch (const StreamError &e) {
        CHECK(std::string(e.what()) == "Stream execution error: Column 'missing' not found in schema");
    }
    FilterStream g(MemoryStream::from_single_batch(stream_batch(1)), "id");
    try {
        g.next_batch();
        CHECK(false);
    } catch (const StreamError &e) {
        CHECK(std::string(e.what()) == "Stream execution error: Predicate column 'id' is not of boolean type");
    }
}
KAT(stream_select) {  // stream.rs:436-512
    SelectStream s(MemoryStream::from_single_batch(stream_batch(1)), {"id", "name"});
    CHECK(s.schema()->num_fields() == 2 && s.schema()->field(0).name == "id" && s.schema()->field(1).name == "name");
    SelectStream one(MemoryStream::from_single_batch(stream_batch(1)), {"name"});
    auto r = one.next_batch();
    CHECK(r->num_columns() == 1 && r->num_rows() == 2 && r->schema()->field(0).name == "name");
    SelectStream re(MemoryStream::from_single_batch(stream_batch(1)), {"active", "id"});
    auto rr = re.next_batch();
    CHECK(rr->schema()->field(0).name == "active" && rr->schema()->field(1).name == "id");
    try {
        SelectStream bad(MemoryStream::from_single_batch(stream_batch(1)), {"nonexistent"});
        CHECK(false);
    } catch (const StreamError &e) {
        CHECK(e.kind == StreamError::Execution && std::string(e.what()).find("nonexistent") != std::string::npos);
    }
}
KAT(stream_filter_then_select) {  // stream.rs:516-532
    auto f = std::make_unique<FilterStream>(MemoryStream::from_single_batch(stream_batch(1)), "active");
    SelectStream s(std::move(f), {"name"});
    auto r = s.next_batch();
    CHECK(r && r->num_columns() == 1 && r->num_rows() == 1 && r->schema()->field(0).name == "name");
}
KAT(stream_concatenate_empty) {  // stream.rs:41-53
    auto m = MemoryStream::empty(rb_schema());
    auto r = m->concatenate();
    CHECK(r.num_rows() == 0 && r.num_columns() == 3);
}

// ============================ Streaming physical plan ============================
KAT(streaming_plan_memory_filter_select_limit) {  // streaming.rs:391-462
    CHECK(mem_source({stream_batch(1), stream_batch(3)})->collect().num_rows() == 4);
    auto f = s_filter(mem_source({stream_batch(1), stream_batch(3)}), "active")->collect();
    CHECK(f.num_rows() == 2 && f.num_columns() == 3);
    CHECK(i64_at(f.column(0), 0) == 1 && i64_at(f.column(0), 1) == 3);
    auto s = s_select(mem_source({stream_batch(1), stream_batch(3)}), {"id", "name"})->collect();
    CHECK(s.num_rows() == 4 && s.num_columns() == 2 && s.schema()->field(0).name == "id");
    auto l = s_limit(mem_source({stream_batch(1), stream_batch(3)}), 3)->collect();
    CHECK(l.num_rows() == 3 && l.num_columns() == 3);
    auto c = s_limit(s_select(s_filter(mem_source({stream_batch(1), stream_batch(3)}), "active"), {"name"}), 1)->collect();
    CHECK(c.num_rows() == 1 && c.num_columns() == 1 && c.schema()->field(0).name == "name");
    CHECK(throws<StreamingExecutionError>([] { mem_source({})->collect(); }));
}
KAT(limit_stream_batches) {  // streaming.rs:464-498
    LimitStream a(std::make_unique<MemoryStream>(rb_schema(), std::vector<RecordBatch>{stream_batch(1), stream_batch(3)}), 2);
    CHECK(a.next_batch()->num_rows() == 2);
    CHECK(!a.next_batch());
    LimitStream b(std::make_unique<MemoryStream>(rb_schema(), std::vector<RecordBatch>{stream_batch(1), stream_batch(3)}), 3);
    CHECK(b.next_batch()->num_rows() == 2);
    CHECK(b.next_batch()->num_rows() == 1);
    CHECK(!b.next_batch());
}
KAT(streaming_planner_conversions) {  // streaming_planner.rs:211-355
    auto lf = LazyFrame::from_dataframe(people_active());
    auto r0 = logical_to_streaming(lf.logical_plan())->collect();
    CHECK(r0.num_rows() == 3 && r0.num_columns() == 3);
    auto r1 = logical_to_streaming(lf.select({Expr::col("name"), Expr::col("age")}).logical_plan())->collect();
    CHECK(r1.num_columns() == 2 && r1.schema()->field(0).name == "name" && r1.schema()->field(1).name == "age");
    auto r2 = logical_to_streaming(lf.filter(Expr::col("active")).logical_plan())->collect();
    CHECK(r2.num_rows() == 2);
    auto r3 = logical_to_streaming(lf.limit(2).logical_plan())->collect();
    CHECK(r3.num_rows() == 2 && r3.num_columns() == 3);
    auto r4 = logical_to_streaming(lf.filter(Expr::col("active")).select({Expr::col("name")}).limit(1).logical_plan())->collect();
    CHECK(r4.num_rows() == 1 && r4.num_columns() == 1 && r4.schema()->field(0).name == "name");
}
KAT(streaming_planner_rejects_binary) {  // streaming_planner.rs:331-381
    auto lf = LazyFrame::from_dataframe(people_active());
    CHECK(throws<StreamingPlannerError>([&] {
        logical_to_streaming(lf.select({Expr::col("age").add(Expr::lit(AnyValue(10)))}).logical_plan());
    }));
    try {
        logical_to_streaming(lf.filter(Expr::col("age").gt(Expr::lit(AnyValue(30)))).logical_plan());
        CHECK(false);
    } catch (const StreamingPlannerError &e) {
        CHECK(std::string(e.what()).find("Binary expressions not yet supported") != std::string::npos);
    }
}
KAT(streaming_alias_dropped) {  // streaming_planner.rs:110-113 (reference defect 2, bug-compatible)
    auto names = extract_column_names_from_expressions({Expr::col("city").alias("location")});
    CHECK(names.size() == 1 && names[0] == "city");
}
KAT(chunker_drops_nulls) {  // streaming.rs:177,188,212 (reference defect 3)
    DataFrame df({Series("v", {AnyValue(5), AnyValue::null(), AnyValue(7)})});
    auto b = dataframe_to_batches(df, 1024);
    auto a = std::dynamic_pointer_cast<const Int64Array>(b[0].column(0));
    CHECK(!a->null_bitmap() && a->value(1) == 0);
    CHECK(dataframe_to_batches(df, 2).size() == 2);
    CHECK(dataframe_to_batches(DataFrame(), 1024).empty());
}

// ============================ AnyValue / eager plan ============================
KAT(anyvalue_partial_ord) {  // series.rs:349-366
    CHECK(any_partial_cmp(AnyValue(1), AnyValue(2)) == -1);
    CHECK(any_partial_cmp(AnyValue(1.0), AnyValue(2.0)) == -1);
    CHECK(any_partial_cmp(AnyValue("a"), AnyValue("b")) == -1);
    CHECK(any_partial_cmp(AnyValue(false), AnyValue(true)) == -1);
    CHECK(any_partial_cmp(AnyValue::null(), AnyValue(0)) == -1);
    CHECK(any_partial_cmp(AnyValue::null(), AnyValue(false)) == -1);
    CHECK(!any_partial_cmp(AnyValue(1), AnyValue("1")));
}
KAT(anyvalue_truth_table_nulls_nan_crosstype) {  // series.rs:87-117 read as source (no reference test)
    using B = BinaryOperator;
    AnyValue nul = AnyValue::null(), i5 = AnyValue(5), f5 = AnyValue(5.0);
    double nan = std::numeric_limits<double>::quiet_NaN();
    // null cell vs non-null literal: Less  => < <= != keep, > >= == drop
    CHECK(any_compare(B::Lt, nul, i5) && any_compare(B::LtEq, nul, i5) && any_compare(B::NotEq, nul, i5));
    CHECK(!any_compare(B::Gt, nul, i5) && !any_compare(B::GtEq, nul, i5) && !any_compare(B::Eq, nul, i5));
    // null literal
    CHECK(any_compare(B::Eq, nul, nul) && any_compare(B::LtEq, nul, nul) && any_compare(B::GtEq, nul, nul));
    CHECK(!any_compare(B::NotEq, nul, nul) && !any_compare(B::Lt, nul, nul) && !any_compare(B::Gt, nul, nul));
    CHECK(any_compare(B::Gt, i5, nul) && any_compare(B::GtEq, i5, nul) && any_compare(B::NotEq, i5, nul));
    CHECK(!any_compare(B::Lt, i5, nul) && !any_compare(B::LtEq, i5, nul) && !any_compare(B::Eq, i5, nul));
    // cross-type: every op false except !=
    for (B op : {B::Eq, B::Lt, B::Gt, B::LtEq, B::GtEq}) CHECK(!any_compare(op, i5, f5));
    CHECK(any_compare(B::NotEq, i5, f5));
    // NaN: every op false except != ; -0.0 == 0.0
    for (B op : {B::Eq, B::Lt, B::Gt, B::LtEq, B::GtEq}) {
        CHECK(!any_compare(op, AnyValue(nan), f5));
        CHECK(!any_compare(op, f5, AnyValue(nan)));
    }
    CHECK(any_compare(B::NotEq, AnyValue(nan), AnyValue(nan)));
    CHECK(any_compare(B::Eq, AnyValue(-0.0), AnyValue(0.0)) && any_compare(B::LtEq, AnyValue(-0.0), AnyValue(0.0)) &&
          !any_compare(B::Lt, AnyValue(-0.0), AnyValue(0.0)));
}
KAT(series_inference) {  // series.rs:185-229
    CHECK(throws<SeriesError>([] { Series("x", {}); }));
    CHECK(Series("x", {AnyValue::null(), AnyValue(1)}).dtype() == DataType::Int64);
    CHECK(Series("x", {AnyValue(1), AnyValue(2.5)}).dtype() == DataType::Float64);  // promotion :210-212
    CHECK(Series("x", {AnyValue::null()}).dtype() == DataType::Null);
    CHECK(throws<SeriesError>([] { Series("x", {AnyValue(1), AnyValue("s")}); }));
}
KAT(eager_select) {  // plan.rs:423-502
    auto r = sel(src(people()), {"name"})->execute();
    CHECK(r.width() == 1 && r.height() == 3 && r.column_names() == std::vector<std::string>{"name"});
    CHECK(any_eq(r.columns()[0][0], AnyValue("Alice")) && any_eq(r.columns()[0][2], AnyValue("Charlie")));
    auto m = sel(src(people()), {"name", "age"})->execute();
    CHECK(m.width() == 2 && m.height() == 3);
    auto o = sel(src(people()), {"score", "name", "age"})->execute();
    CHECK((o.column_names() == std::vector<std::string>{"score", "name", "age"}));
    try {
        sel(src(people()), {"nonexistent"})->execute();
        CHECK(false);
    } catch (const ExecutionError &e) {
        CHECK(e.kind == ExecutionError::ColumnNotFound && std::string(e.what()) == "Column not found: 'nonexistent'");
    }
}
KAT(eager_filter_gt) {  // plan.rs:504-525
    auto r = filt(src(people()), "age", AnyValue(25), BinaryOperator::Gt)->execute();
    CHECK(r.height() == 2 && r.width() == 3);
    CHECK(any_eq((*r.column("age"))[0], AnyValue(30)) && any_eq((*r.column("age"))[1], AnyValue(35)));
}
KAT(eager_filter_eq_string) {  // plan.rs:527-547
    auto r = filt(src(people()), "name", AnyValue("Bob"), BinaryOperator::Eq)->execute();
    CHECK(r.height() == 1 && r.width() == 3 && any_eq((*r.column("name"))[0], AnyValue("Bob")));
}
KAT(eager_filter_lt_float) {  // plan.rs:549-569
    auto r = filt(src(people()), "score", AnyValue(90.0), BinaryOperator::Lt)->execute();
    CHECK(r.height() == 2);
    CHECK(any_eq((*r.column("name"))[0], AnyValue("Alice")) && any_eq((*r.column("name"))[1], AnyValue("Charlie")));
}
KAT(eager_filter_no_matches) {  // plan.rs:571-589
    auto r = filt(src(people()), "age", AnyValue(100), BinaryOperator::Gt)->execute();
    CHECK(r.height() == 0 && r.width() == 3);
    CHECK((r.column_names() == std::vector<std::string>{"name", "age", "score"}));
}
KAT(eager_filter_missing_column) {  // plan.rs:591-612
    try {
        filt(src(people()), "nonexistent", AnyValue(0), BinaryOperator::Eq)->execute();
        CHECK(false);
    } catch (const ExecutionError &e) {
        CHECK(e.kind == ExecutionError::ColumnNotFound);
    }
}
KAT(eager_chained) {  // plan.rs:671-704
    auto r = lim(filt(sel(src(people()), {"name", "age", "score"}), "age", AnyValue(25), BinaryOperator::Gt), 1)->execute();
    CHECK(r.height() == 1 && r.width() == 3 && any_eq((*r.column("name"))[0], AnyValue("Bob")));
}
KAT(eager_filter_then_select) {  // plan.rs:706-735
    auto r = sel(filt(src(people()), "age", AnyValue(30), BinaryOperator::GtEq), {"name", "score"})->execute();
    CHECK(r.height() == 2 && r.width() == 2 && (r.column_names() == std::vector<std::string>{"name", "score"}));
    CHECK(any_eq((*r.column("name"))[0], AnyValue("Bob")) && any_eq((*r.column("name"))[1], AnyValue("Charlie")));
}
KAT(eager_error_propagates) {  // plan.rs:739-769
    try {
        sel(filt(src(people()), "nonexistent", AnyValue(0), BinaryOperator::Eq), {"name"})->execute();
        CHECK(false);
    } catch (const ExecutionError &e) {
        CHECK(e.kind == ExecutionError::ColumnNotFound);
    }
}
KAT(eager_select_on_empty_errors) {  // plan.rs:88-93 + series.rs:186-188
    try {
        sel(filt(src(people()), "age", AnyValue(100), BinaryOperator::Gt), {"name"})->execute();
        CHECK(false);
    } catch (const ExecutionError &e) {
        CHECK(e.kind == ExecutionError::SeriesErr && std::string(e.what()) == "Series error: Empty series not allowed");
    }
}
KAT(eager_nulls_sort_lowest) {  // plan.rs:112-130 with series.rs:105-107 (no reference test)
    DataFrame df({Series("v", {AnyValue(5), AnyValue::null(), AnyValue(50)})});
    CHECK(filt(src(df), "v", AnyValue(10), BinaryOperator::Lt)->execute().height() == 2);     // 5 and null
    CHECK(filt(src(df), "v", AnyValue(10), BinaryOperator::Gt)->execute().height() == 1);     // 50
    CHECK(filt(src(df), "v", AnyValue(5), BinaryOperator::NotEq)->execute().height() == 2);   // null, 50
}

// ============================ LazyFrame (config 1 plumbing) ============================
KAT(lazy_collect_select) {  // builder.rs:435-448
    auto r = LazyFrame::from_dataframe(people()).select({Expr::col("name"), Expr::col("age")}).collect();
    CHECK(r.width() == 2 && r.height() == 3 && (r.column_names() == std::vector<std::string>{"name", "age"}));
}
KAT(lazy_collect_filter) {  // builder.rs:450-461
    auto r = LazyFrame::from_dataframe(people()).filter(Expr::col("age").gt(Expr::lit(AnyValue(25)))).collect();
    CHECK(r.height() == 2 && r.width() == 3);
}
KAT(lazy_collect_limit) {  // builder.rs:463-473
    auto r = LazyFrame::from_dataframe(people()).limit(2).collect();
    CHECK(r.height() == 2 && r.width() == 3);
}
KAT(lazy_validate_errors) {  // builder.rs:496-532
    for (int which = 0; which < 2; ++which) {
        try {
            auto lf = LazyFrame::from_dataframe(people());
            if (which == 0) lf.select({Expr::col("nonexistent")}).collect();
            else lf.filter(Expr::col("nonexistent").gt(Expr::lit(AnyValue(0)))).collect();
            CHECK(false);
        } catch (const QueryError &e) {
            CHECK(e.kind == QueryError::LogicalPlan && e.column == "nonexistent");
        }
    }
}
KAT(lazy_schema_inference) {  // builder.rs:536-566
    auto lf = LazyFrame::from_dataframe(people()).select({Expr::col("name"), Expr::col("age").alias("user_age")});
    auto s = lf.logical_plan()->schema();
    CHECK(s.size() == 2 && s[0].first == "name" && s[1].first == "user_age" && s[0].second == DataType::String &&
          s[1].second == DataType::Int64);
    auto a = LazyFrame::from_dataframe(people()).select({Expr::col("age").add(Expr::col("score")).alias("age_plus_score")});
    auto s2 = a.logical_plan()->schema();
    CHECK(s2.size() == 1 && s2[0].first == "age_plus_score" && s2[0].second == DataType::Float64);
}
KAT(lazy_streaming) {  // builder.rs:569-614
    auto r = LazyFrame::from_dataframe(people()).select({Expr::col("name"), Expr::col("age")}).collect_streaming();
    CHECK(r.num_columns() == 2 && r.num_rows() == 3 && r.schema()->field(0).name == "name" &&
          r.schema()->field(1).name == "age");
    CHECK(throws<QueryError>([] { LazyFrame::from_dataframe(people()).filter(Expr::col("active")).collect_streaming(); }));
    auto e = LazyFrame::from_dataframe(people()).select({Expr::col("name")}).collect();
    auto s = LazyFrame::from_dataframe(people()).select({Expr::col("name")}).collect_streaming();
    CHECK(e.width() == s.num_columns() && e.height() == s.num_rows());
}
KAT(lazy_select_then_filter_same_column) {  // builder.rs:618-633
    auto lf = LazyFrame::from_dataframe(people()).select({Expr::col("name")}).filter(Expr::col("name").eq(Expr::lit(AnyValue("Alice"))));
    auto r1 = lf.collect(), r2 = lf.collect();
    CHECK(r1.height() == 1 && r1.width() == 1 && r2.height() == r1.height());
}
KAT(config1_literal_order_errors) {  // BASELINE config 1 as written; README.md:60-63; SURVEY 3.1
    try {
        LazyFrame::from_dataframe(people()).select({Expr::col("name")}).filter(Expr::col("age").gt(Expr::lit(AnyValue(25)))).collect();
        CHECK(false);
    } catch (const QueryError &e) {
        CHECK(e.kind == QueryError::LogicalPlan && e.column == "age");
        CHECK(std::string(e.what()) == "Logical plan error: Column not found: 'age'");
    }
    auto a = LazyFrame::from_dataframe(people()).select({Expr::col("name"), Expr::col("age")}).filter(Expr::col("age").gt(Expr::lit(AnyValue(25)))).collect();
    CHECK(a.height() == 2 && a.width() == 2);  // main.rs:49-52
    auto b = LazyFrame::from_dataframe(people()).filter(Expr::col("age").gt(Expr::lit(AnyValue(25)))).select({Expr::col("name")}).collect();
    CHECK(b.height() == 2 && b.width() == 1);  // main.rs:59-62
}
// BASELINE configs[0] exactly as SURVEY.md section 8d states it: a 1 000-row frame name = "n{i}", age = 18 + splitmix64(7 + i) % 50,
// and the three spellings of the query (builder.rs:57-73, :96-104).  Expected rows are computed here with a plain loop.
DataFrame config1_frame(std::vector<std::string> *kept_names, std::vector<int64_t> *kept_ages) {
    std::vector<AnyValue> names, ages;
    for (uint64_t i = 0; i < 1000; ++i) {
        const int64_t age = 18 + static_cast<int64_t>(splitmix64(7 + i) % 50);
        names.push_back(AnyValue("n" + std::to_string(i)));
        ages.push_back(AnyValue(age));
        if (age > 25) {
            if (kept_names) kept_names->push_back("n" + std::to_string(i));
            if (kept_ages) kept_ages->push_back(age);
        }
    }
    return DataFrame({Series("name", names), Series("age", ages)});
}
KAT(config1_thousand_rows_three_spellings) {
    std::vector<std::string> want_names;
    std::vector<int64_t> want_ages;
    const DataFrame df = config1_frame(&want_names, &want_ages);
    CHECK(want_names.size() > 800 && want_names.size() < 900);  // ages 18..67 uniform: 42 of 50 values survive
    const auto pred = Expr::col("age").gt(Expr::lit(AnyValue(25)));
    // 1. select([name]).filter(age > 25): the filter sees a frame without `age`
    try {
        LazyFrame::from_dataframe(df).select({Expr::col("name")}).filter(pred).collect();
        CHECK(false);
    } catch (const QueryError &e) {
        CHECK(e.kind == QueryError::LogicalPlan && e.column == "age");
        CHECK(std::string(e.what()) == "Logical plan error: Column not found: 'age'");
    }
    // 2. select([name, age]).filter(age > 25)
    auto a = LazyFrame::from_dataframe(df).select({Expr::col("name"), Expr::col("age")}).filter(pred).collect();
    CHECK(a.width() == 2 && a.height() == want_names.size());
    // 3. filter(age > 25).select([name])
    auto b = LazyFrame::from_dataframe(df).filter(pred).select({Expr::col("name")}).collect();
    CHECK(b.width() == 1 && b.height() == want_names.size() && b.column_names() == std::vector<std::string>{"name"});
    for (size_t i = 0; i < want_names.size(); ++i) {
        CHECK(any_eq((*a.column("name"))[i], AnyValue(want_names[i])) && any_eq((*a.column("age"))[i], AnyValue(want_ages[i])));
        CHECK(any_eq((*b.column("name"))[i], AnyValue(want_names[i])));
    }
}
// ---- CsvFileStream (file_stream.rs:370-458: the reference's four inline tests) ------------------------------------------
static std::string write_temp(const std::string &text) {
    char name[] = "/tmp/rvo_csv_XXXXXX";
    const int fd = mkstemp(name);
    if (fd < 0 || write(fd, text.data(), text.size()) != static_cast<ssize_t>(text.size())) throw Err("cannot write temp file");
    close(fd);
    return name;
}
static SchemaRef csv_schema() {
    return std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}, {"name", DataType::String, true},
                                                       {"score", DataType::Float64, true}, {"active", DataType::Boolean, false}});
}
KAT(csv_file_stream_reference_tests) {
    const std::string path = write_temp("id,name,score,active\n1,Alice,85.5,true\n2,Bob,92.0,false\n3,Charlie,78.5,true\n4,,90.0,false\n5,Eve,null,true\n");
    CsvFileStream s(path, csv_schema(), 10);
    CHECK(*s.schema() == *csv_schema());
    auto b = s.next_batch();
    CHECK(b && b->num_rows() == 5 && b->num_columns() == 4);  // :401-417, :433-446
    CHECK(!s.next_batch());
    CHECK(csv_adaptive_batch_size(*csv_schema()) == 100000);  // :419-431: 8 + 32 + 8 + 1 = 49 bytes per row, clamped
    // the arrays behind the row counts (the reference's tests stop at the shapes)
    auto name = std::static_pointer_cast<const StringArray>(b->column(1));
    CHECK(*name->value(0) == "Alice" && !name->value(3) && name->null_count() == 1);
    auto id = std::static_pointer_cast<const Int64Array>(b->column(0));
    CHECK(id->null_bitmap() == nullptr && *id->value(4) == 5);
    // the defect: one null score -> nulls [F,F,F,F,T] handed over as VALIDITY: rows 0..3 read as null, row 4 (the null) as 0.0
    auto score = std::static_pointer_cast<const Float64Array>(b->column(2));
    CHECK(score->null_count() == 4 && !score->value(0) && score->value(4) && *score->value(4) == 0.0 && score->values()[0] == 85.5);
    auto active = std::static_pointer_cast<const BooleanArray>(b->column(3));
    CHECK(*active->value(0) == true && *active->value(1) == false);
    unlink(path.c_str());
    const std::string empty = write_temp("id,name\n");  // :448-458: header only
    CsvFileStream e(empty, std::make_shared<Schema>(std::vector<Field>{{"id", DataType::Int64, false}, {"name", DataType::String, true}}), 10);
    CHECK(!e.next_batch());
    unlink(empty.c_str());
    CHECK(throws<Err>([] { CsvFileStream("/nonexistent/x.csv", csv_schema()); }));
    int64_t i = 0;
    double f = 0;
    CHECK(rust_parse_i64("-9223372036854775808", i) && i == INT64_MIN && !rust_parse_i64("9223372036854775808", i) && !rust_parse_i64("+", i) &&
          !rust_parse_i64("1_0", i) && rust_parse_i64("+7", i) && i == 7 && !rust_parse_i64("1.0", i));
    CHECK(rust_parse_f64("1e3", f) && f == 1000.0 && rust_parse_f64(".5", f) && f == 0.5 && rust_parse_f64("1.", f) && rust_parse_f64("-Infinity", f) && f < 0 &&
          rust_parse_f64("NaN", f) && f != f && !rust_parse_f64("0x10", f) && !rust_parse_f64("e5", f) && !rust_parse_f64(".", f) && !rust_parse_f64("1e", f));
}
KAT(optimizer_pushdown_and_alias_blindness) {  // optimizer.rs:17-39, :66-100 (0 reference tests)
    auto lf = LazyFrame::from_dataframe(people());
    auto p = QueryOptimizer::optimize(lf.filter(Expr::col("age").gt(Expr::lit(AnyValue(25)))).select({Expr::col("name"), Expr::col("age")}).logical_plan());
    CHECK(p->kind == LogicalPlan::Filter && p->input->kind == LogicalPlan::Select);
    auto q = QueryOptimizer::optimize(lf.filter(Expr::col("age").gt(Expr::lit(AnyValue(25)))).select({Expr::col("name")}).logical_plan());
    CHECK(q->kind == LogicalPlan::Select && q->input->kind == LogicalPlan::Filter);
    // alias-blind rewrite then fails validation: 'age' not in [a]
    CHECK(throws<QueryError>([&] {
        lf.filter(Expr::col("age").gt(Expr::lit(AnyValue(30)))).select({Expr::col("age").alias("a")}).collect();
    }));
}
KAT(planner_filter_grammar) {  // planner.rs:134-189
    auto t = convert_filter_predicate(Expr::col("age").gte(Expr::lit(AnyValue(30))));
    CHECK(t.column == "age" && t.op == BinaryOperator::GtEq && any_eq(t.value, AnyValue(30)));
    auto kind_of = [](const Expr &e) {
        try {
            convert_filter_predicate(e);
        } catch (const ConversionError &c) {
            return static_cast<int>(c.kind);
        }
        return -1;
    };
    auto cmp = Expr::col("a").gt(Expr::lit(AnyValue(1)));
    CHECK(kind_of(cmp.and_(cmp)) == ConversionError::UnsupportedFilter);
    CHECK(kind_of(Expr::col("a").add(Expr::lit(AnyValue(1)))) == ConversionError::UnsupportedFilterOperator);
    CHECK(kind_of(Expr::lit(AnyValue(1)).gt(Expr::lit(AnyValue(1)))) == ConversionError::FilterLeftNotColumn);
    CHECK(kind_of(Expr::col("a").gt(Expr::col("b"))) == ConversionError::FilterRightNotLiteral);
    CHECK(kind_of(Expr::col("a")) == ConversionError::InvalidFilterStructure);
    auto s = convert_select_expr(Expr::col("age").alias("years"));
    CHECK(s.first == "age" && s.second == "years");
}

// ============================ composed semantics (new backend) ============================
KAT(compose_compare_and_filter_project) {
    auto f = std::make_shared<Float64Array>(std::vector<double>{0.9, 0.1, 0.7, 0.8},
                                            std::vector<bool>{true, true, false, true});
    auto x = std::make_shared<Int64Array>(std::vector<int64_t>{100, 100, 100, 500},
                                          std::vector<bool>{true, true, true, true});
    std::vector<Term> terms{{0, TermOp::Gt, AnyValue(0.5)}, {1, TermOp::Lt, AnyValue(200)}};
    auto drops = filter_project({f, x}, terms, NullPolicy::Drops, {0, 1});
    CHECK(drops.num_rows() == 1);  // row 0 only: row 2 has a null f, row 3 fails x<200
    auto least = filter_project({f, x}, {{0, TermOp::Lt, AnyValue(0.5)}}, NullPolicy::IsLeast, {0});
    CHECK(least.num_rows() == 2);  // row 1 (0.1) and row 2 (null sorts lowest)
    auto lf = std::dynamic_pointer_cast<const Float64Array>(least.column(0));
    CHECK(lf->null_bitmap() && !lf->value(1) && lf->values()[1] == 0.0);
    auto streamed = stream_filter_project({f, x}, 3, terms, NullPolicy::Drops, {0, 1});
    CHECK(streamed.num_rows() == 1);
}

int main(int argc, char **argv) {
    int failed = 0;
    for (auto &c : cases()) {
        if (argc > 1 && std::string(argv[1]) != c.name) continue;
        try {
            c.fn();
            std::printf("ok %s\n", c.name);
        } catch (const std::exception &e) {
            std::printf("FAIL %s: %s\n", c.name, e.what());
            ++failed;
        }
    }
    std::printf("%zu cases, %d failed\n", cases().size(), failed);
    return failed ? 1 : 0;
}
