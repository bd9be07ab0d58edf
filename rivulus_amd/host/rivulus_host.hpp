// C++ host layer above the C ABI (include/rivulus_gpu.h): the mirror of the reference's
// operator interfaces for the filter / project / scan path, with every array resident in
// HBM.  Same names, argument meaning and error texts as the reference (Rust), so code written
// against the reference reads the same here:
//
//   execution::{DataType, Field, Schema}                    src/execution/schema.rs:1-76
//   execution::{PrimitiveArray<T>, BooleanArray}             src/execution/array/{primitive,boolean}.rs
//   execution::RecordBatch                                   src/execution/record_batch.rs:8-422
//   execution::{DataStream, MemoryStream, FilterStream,
//               SelectStream, LimitStream}                   src/execution/stream.rs:25-213, streaming.rs:246-288
//   execution::GpuFilterProjectStream                        the operator the new backend adds at seam S1
//   expressions::{Expr, BinaryOperator}                      src/expressions/expr.rs:3-139
//   physical_plan::{convert_filter_predicate, convert_select_expr,
//                   extract_boolean_predicate_column,
//                   extract_column_names_from_expressions}   planner.rs:113-189, streaming_planner.rs:102-168
//   physical_plan::lower_predicate                           compare / AND / OR lowering (replaces the rejection at
//                                                            streaming_planner.rs:141-162)
//   physical_plan::StreamingPhysicalPlan                     streaming.rs:29-133, :235-238, :343-352
//   physical_plan::PhysicalPlan (Source/Filter/Select)       plan.rs:8-150 over typed device columns
//
// String columns live on the device as well (bytes + int32 offsets + validity): they ride through
// filter / take / concat and `name == "Bob"` compare terms run on the device (byte-wise str ordering).
// There is no CPU fallback anywhere in this layer.
#pragma once

#include <cstring>
#include <exception>
#include <cstdlib>
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <fstream>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "../../include/rivulus_gpu.h"

namespace rivulus {

// Err(String) of the reference; `status` is the C-ABI code it came with.
struct Error : std::runtime_error {
    rv_status status;
    Error(rv_status s, const std::string &m) : std::runtime_error(m), status(s) {}
};
struct Panic : std::runtime_error {  // Rust assert!/panic!
    using std::runtime_error::runtime_error;
};

inline void check(rv_status s) {
    if (s != RV_OK) throw Error(s, rv_last_error());
}

// One device + one stream (rv_ctx).  Shared by the arrays created from it.
class Context {
  public:
    explicit Context(int device = 0) { check(rv_ctx_create(device, &ctx_)); }
    ~Context() { rv_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    rv_ctx *raw() const { return ctx_; }

  private:
    rv_ctx *ctx_ = nullptr;
};
using ContextRef = std::shared_ptr<Context>;

namespace execution {

// ---------------------------------------------------------------------------------------
// schema.rs:1-76
// ---------------------------------------------------------------------------------------
enum class DataType { Null, Boolean, Int64, Float64, String };
inline const char *to_string(DataType t) {
    static const char *n[] = {"Null", "Boolean", "Int64", "Float64", "String"};
    return n[static_cast<int>(t)];
}
inline rv_dtype to_rv(DataType t) { return static_cast<rv_dtype>(static_cast<int>(t)); }
inline DataType from_rv(rv_dtype t) { return static_cast<DataType>(static_cast<int>(t)); }

class Field {
  public:
    Field(std::string name, DataType data_type, bool nullable) : name_(std::move(name)), data_type_(data_type), nullable_(nullable) {}
    const std::string &name() const { return name_; }
    DataType data_type() const { return data_type_; }
    bool is_nullable() const { return nullable_; }
    bool operator==(const Field &o) const { return name_ == o.name_ && data_type_ == o.data_type_ && nullable_ == o.nullable_; }

  private:
    std::string name_;
    DataType data_type_;
    bool nullable_;
};

class Schema {
  public:
    Schema() = default;
    explicit Schema(std::vector<Field> fields) : fields_(std::move(fields)) {}
    static Schema empty() { return Schema(); }
    const std::vector<Field> &fields() const { return fields_; }
    const Field &field(size_t i) const { return fields_.at(i); }
    const Field *field_by_name(const std::string &n) const {
        for (auto &f : fields_)
            if (f.name() == n) return &f;
        return nullptr;
    }
    std::optional<size_t> index_of(const std::string &n) const {
        for (size_t i = 0; i < fields_.size(); ++i)
            if (fields_[i].name() == n) return i;
        return std::nullopt;
    }
    size_t num_fields() const { return fields_.size(); }
    bool is_empty() const { return fields_.empty(); }
    bool operator==(const Schema &o) const { return fields_ == o.fields_; }
    bool operator!=(const Schema &o) const { return !(*this == o); }

  private:
    std::vector<Field> fields_;
};
using SchemaRef = std::shared_ptr<const Schema>;

// ---------------------------------------------------------------------------------------
// trait Array (array/mod.rs:10-16): a device column handle
// ---------------------------------------------------------------------------------------
class Array;
using ArrayRef = std::shared_ptr<const Array>;

class Array {
  public:
    virtual ~Array() {
        if (handle_) rv_free(ctx_ ? ctx_->raw() : nullptr, handle_);
    }
    virtual size_t len() const { return info().length; }
    virtual DataType data_type() const { return from_rv(info().dtype); }
    virtual size_t null_count() const {  // primitive.rs:90-105
        uint64_t n = 0;
        check(rv_null_count(ctx_->raw(), handle_, &n));
        return n;
    }
    virtual ArrayRef slice(size_t offset, size_t length) const {  // primitive.rs:107-117 (zero copy)
        if (offset + length > len()) throw Panic("Slice out of bounds");
        rv_dcolumn *out = nullptr;
        check(rv_slice(ctx_->raw(), handle_, offset, length, &out));
        return adopt(ctx_, out);
    }
    bool has_null_bitmap() const { return info().has_validity != 0; }  // `null_bitmap.is_some()`
    rv_dcolumn *handle() const { return handle_; }
    const ContextRef &context() const { return ctx_; }
    bool on_device() const { return handle_ != nullptr; }

    // adopts a handle returned by the C ABI; the concrete type follows its dtype
    static ArrayRef adopt(const ContextRef &ctx, rv_dcolumn *h);

  protected:
    Array() = default;
    Array(ContextRef ctx, rv_dcolumn *h) : ctx_(std::move(ctx)), handle_(h) {}
    rv_column_info info() const {
        rv_column_info i{};
        check(rv_column_info_get(ctx_->raw(), handle_, &i));
        return i;
    }
    // host copy of the logical range, fetched once (element access is for tests and display)
    struct HostCopy {
        std::vector<uint64_t> values;   // 8-byte cells, or bit bytes packed in the low part for Boolean
        std::vector<uint8_t> bits;      // Boolean value bits
        std::vector<uint8_t> validity;  // empty: no null bitmap
    };
    const HostCopy &host() const {
        if (!host_) {
            auto h = std::make_shared<HostCopy>();
            const rv_column_info i = info();
            const size_t n = i.length, nb = (n + 7) / 8;
            int hv = 0;
            if (i.has_validity) h->validity.resize(std::max<size_t>(nb, 1));
            if (i.dtype == RV_BOOLEAN) {
                h->bits.resize(std::max<size_t>(nb, 1));
                check(rv_download(ctx_->raw(), handle_, h->bits.data(), i.has_validity ? h->validity.data() : nullptr, &hv));
            } else {
                h->values.resize(std::max<size_t>(n, 1));
                check(rv_download(ctx_->raw(), handle_, h->values.data(), i.has_validity ? h->validity.data() : nullptr, &hv));
            }
            host_ = h;
        }
        return *host_;
    }
    bool host_valid(size_t i) const {
        const HostCopy &h = host();
        return h.validity.empty() || ((h.validity[i / 8] >> (i % 8)) & 1);
    }
    ContextRef ctx_;
    rv_dcolumn *handle_ = nullptr;
    mutable std::shared_ptr<HostCopy> host_;
};

inline std::vector<uint8_t> pack_bits(const std::vector<bool> &v) {  // BitMap::from_bool_slice, bitmap.rs:44-59
    std::vector<uint8_t> out((v.size() + 7) / 8, 0);
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i]) out[i / 8] |= static_cast<uint8_t>(1u << (i % 8));
    return out;
}

// PrimitiveArray<i64|f64> -- primitive.rs:20-122
template <class T>
class PrimitiveArray : public Array {
    static_assert(sizeof(T) == 8, "Int64 / Float64");

  public:
    static constexpr DataType kType = std::is_same<T, double>::value ? DataType::Float64 : DataType::Int64;
    PrimitiveArray(ContextRef ctx, rv_dcolumn *h) : Array(std::move(ctx), h) {}

    // PrimitiveArray::new(values, validity) -- primitive.rs:31-42
    static std::shared_ptr<const PrimitiveArray> create(const ContextRef &ctx, const std::vector<T> &values,
                                                        const std::optional<std::vector<bool>> &validity = std::nullopt) {
        std::vector<uint8_t> vbits;
        if (validity) {
            if (validity->size() != values.size()) throw Error(RV_ERR_LENGTH_MISMATCH, "validity length != values length");
            vbits = pack_bits(*validity);
            if (vbits.empty()) vbits.push_back(0);
        }
        rv_column c{};
        c.dtype = to_rv(kType);
        c.values = values.data();
        c.validity = validity ? vbits.data() : nullptr;
        c.length = values.size();
        rv_dcolumn *h = nullptr;
        check(rv_upload(ctx->raw(), &c, &h));
        return std::make_shared<const PrimitiveArray>(ctx, h);
    }
    static std::shared_ptr<const PrimitiveArray> from_values(const ContextRef &ctx, const std::vector<T> &values) {  // :44-46
        return create(ctx, values);
    }
    // primitive.rs:48-60
    std::optional<T> value(size_t index) const {
        if (index >= len()) throw Panic("Index " + std::to_string(index) + " out of bounds");
        if (!host_valid(index)) return std::nullopt;
        T out;
        std::memcpy(&out, &host().values[index], 8);
        return out;
    }
    // the raw slot, placeholder included (take_array writes 0 / 0.0 under nulls, record_batch.rs:142-146)
    T raw_value(size_t index) const {
        T out;
        std::memcpy(&out, &host().values.at(index), 8);
        return out;
    }
};
using Int64Array = PrimitiveArray<int64_t>;
using Float64Array = PrimitiveArray<double>;

// BooleanArray -- boolean.rs:9-180
class BooleanArray : public Array {
  public:
    BooleanArray(ContextRef ctx, rv_dcolumn *h) : Array(std::move(ctx), h) {}
    // BooleanArray::new(Vec<Option<bool>>) -- boolean.rs:19-50 (false under null, bitmap dropped when no null)
    static std::shared_ptr<const BooleanArray> create(const ContextRef &ctx, const std::vector<std::optional<bool>> &v) {
        std::vector<bool> vals(v.size()), valid(v.size());
        bool any_null = false;
        for (size_t i = 0; i < v.size(); ++i) {
            vals[i] = v[i].value_or(false);
            valid[i] = v[i].has_value();
            any_null |= !valid[i];
        }
        return upload(ctx, vals, any_null ? std::optional<std::vector<bool>>(valid) : std::nullopt);
    }
    static std::shared_ptr<const BooleanArray> from_bools(const ContextRef &ctx, const std::vector<bool> &v) {  // :52-55
        return upload(ctx, v, std::nullopt);
    }
    static std::shared_ptr<const BooleanArray> all_true(const ContextRef &ctx, size_t n) { return from_bools(ctx, std::vector<bool>(n, true)); }
    static std::shared_ptr<const BooleanArray> all_false(const ContextRef &ctx, size_t n) { return from_bools(ctx, std::vector<bool>(n, false)); }

    std::optional<bool> value(size_t index) const {  // boolean.rs:92-104
        if (index >= len()) throw Panic("Index " + std::to_string(index) + " out of bounds");
        if (!host_valid(index)) return std::nullopt;
        return ((host().bits[index / 8] >> (index % 8)) & 1) != 0;
    }
    // boolean.rs:120-165: strict null propagation; errors keep the reference text
    std::shared_ptr<const BooleanArray> logical_and(const BooleanArray &o) const { return binary(&rv_boolean_and, o); }
    std::shared_ptr<const BooleanArray> logical_or(const BooleanArray &o) const { return binary(&rv_boolean_or, o); }
    std::shared_ptr<const BooleanArray> logical_not() const {
        rv_dcolumn *out = nullptr;
        check(rv_boolean_not(ctx_->raw(), handle_, &out));
        return std::make_shared<const BooleanArray>(ctx_, out);
    }
    size_t count_true() const {  // boolean.rs:167-172
        uint64_t t = 0, f = 0;
        check(rv_boolean_count(ctx_->raw(), handle_, &t, &f));
        return t;
    }
    size_t count_false() const {
        uint64_t t = 0, f = 0;
        check(rv_boolean_count(ctx_->raw(), handle_, &t, &f));
        return f;
    }

  private:
    static std::shared_ptr<const BooleanArray> upload(const ContextRef &ctx, const std::vector<bool> &vals,
                                                      const std::optional<std::vector<bool>> &valid) {
        std::vector<uint8_t> vb = pack_bits(vals), mb;
        if (vb.empty()) vb.push_back(0);
        if (valid) {
            mb = pack_bits(*valid);
            if (mb.empty()) mb.push_back(0);
        }
        rv_column c{};
        c.dtype = RV_BOOLEAN;
        c.values = vb.data();
        c.validity = valid ? mb.data() : nullptr;
        c.length = vals.size();
        rv_dcolumn *h = nullptr;
        check(rv_upload(ctx->raw(), &c, &h));
        return std::make_shared<const BooleanArray>(ctx, h);
    }
    using BinFn = rv_status (*)(rv_ctx *, const rv_dcolumn *, const rv_dcolumn *, rv_dcolumn **);
    std::shared_ptr<const BooleanArray> binary(BinFn fn, const BooleanArray &o) const {
        rv_dcolumn *out = nullptr;
        check(fn(ctx_->raw(), handle_, o.handle_, &out));
        return std::make_shared<const BooleanArray>(ctx_, out);
    }
};

// StringArray -- string.rs:8-190: UTF-8 bytes + int32 offsets + optional validity, on the device
class StringArray : public Array {
  public:
    StringArray(ContextRef ctx, rv_dcolumn *h) : Array(std::move(ctx), h) {}
    // StringArray::new(Vec<Option<String>>) -- string.rs:19-57: a null spans no bytes, the bitmap is
    // dropped when there is no null
    static std::shared_ptr<const StringArray> create(const ContextRef &ctx, const std::vector<std::optional<std::string>> &v) {
        std::vector<int32_t> offsets{0};
        std::string data;
        std::vector<bool> valid(v.size());
        bool any_null = false;
        for (size_t i = 0; i < v.size(); ++i) {
            valid[i] = v[i].has_value();
            any_null |= !valid[i];
            if (v[i]) data += *v[i];
            offsets.push_back(static_cast<int32_t>(data.size()));
        }
        std::vector<uint8_t> vb;
        if (any_null) {
            vb = pack_bits(valid);
            if (vb.empty()) vb.push_back(0);
        }
        rv_column c{};
        c.dtype = RV_STRING;
        c.values = data.data();
        c.data_bytes = data.size();
        c.offsets = offsets.data();
        c.validity = any_null ? vb.data() : nullptr;
        c.length = v.size();
        rv_dcolumn *h = nullptr;
        check(rv_upload(ctx->raw(), &c, &h));
        return std::make_shared<const StringArray>(ctx, h);
    }
    static std::shared_ptr<const StringArray> from_strings(const ContextRef &ctx, const std::vector<std::string> &v) {  // string.rs:59-62
        return create(ctx, std::vector<std::optional<std::string>>(v.begin(), v.end()));
    }
    std::optional<std::string> value(size_t i) const {  // string.rs:81-99
        if (i >= len()) throw Panic("Index " + std::to_string(i) + " out of bounds");
        const Strings &h = strings();
        if (!h.validity.empty() && !((h.validity[i / 8] >> (i % 8)) & 1)) return std::nullopt;
        return std::string(h.data.begin() + h.offsets[i], h.data.begin() + h.offsets[i + 1]);
    }
    size_t total_bytes() const { return info().data_bytes; }  // bytes of the logical elements

  private:
    struct Strings {
        std::vector<int32_t> offsets;
        std::vector<uint8_t> data, validity;
    };
    const Strings &strings() const {  // host copy, fetched once (element access is for tests and display)
        if (!strings_) {
            auto h = std::make_shared<Strings>();
            const rv_column_info i = info();
            h->offsets.resize(i.length + 1);
            h->data.resize(std::max<size_t>(i.data_bytes, 1));
            if (i.has_validity) h->validity.resize(std::max<size_t>((i.length + 7) / 8, 1));
            int hv = 0;
            check(rv_download_string(ctx_->raw(), handle_, h->offsets.data(), h->data.data(), i.has_validity ? h->validity.data() : nullptr, &hv));
            strings_ = h;
        }
        return *strings_;
    }
    mutable std::shared_ptr<Strings> strings_;
};

// NullArray -- null.rs:5-66: a length, every element null
class NullArray : public Array {
  public:
    NullArray(ContextRef ctx, rv_dcolumn *h) : Array(std::move(ctx), h) {}
    static std::shared_ptr<const NullArray> create(const ContextRef &ctx, size_t n) {
        rv_column c{};
        c.dtype = RV_NULL;
        c.length = n;
        rv_dcolumn *h = nullptr;
        check(rv_upload(ctx->raw(), &c, &h));
        return std::make_shared<const NullArray>(ctx, h);
    }
};

inline ArrayRef Array::adopt(const ContextRef &ctx, rv_dcolumn *h) {
    rv_column_info i{};
    check(rv_column_info_get(ctx->raw(), h, &i));
    switch (i.dtype) {
        case RV_INT64: return std::make_shared<const Int64Array>(ctx, h);
        case RV_FLOAT64: return std::make_shared<const Float64Array>(ctx, h);
        case RV_BOOLEAN: return std::make_shared<const BooleanArray>(ctx, h);
        case RV_STRING: return std::make_shared<const StringArray>(ctx, h);
        case RV_NULL: return std::make_shared<const NullArray>(ctx, h);
        default: rv_free(ctx->raw(), h); throw Error(RV_ERR_UNSUPPORTED, "unsupported device array type");
    }
}

// ---------------------------------------------------------------------------------------
// RecordBatch -- record_batch.rs:8-422
// ---------------------------------------------------------------------------------------
class RecordBatch {
  public:
    RecordBatch() : schema_(std::make_shared<Schema>()), num_rows_(0) {}
    // record_batch.rs:16-58
    static RecordBatch try_new(SchemaRef schema, std::vector<ArrayRef> columns) {
        if (schema->num_fields() != columns.size())
            throw Error(RV_ERR_INVALID_ARG, "Schema has " + std::to_string(schema->num_fields()) + " fields but " +
                                                std::to_string(columns.size()) + " columns provided");
        const size_t rows = columns.empty() ? 0 : columns[0]->len();
        for (size_t i = 0; i < columns.size(); ++i)
            if (columns[i]->len() != rows)
                throw Error(RV_ERR_LENGTH_MISMATCH, "Column " + std::to_string(i) + " has length " + std::to_string(columns[i]->len()) +
                                                        " but expected " + std::to_string(rows));
        for (size_t i = 0; i < columns.size(); ++i)
            if (schema->field(i).data_type() != columns[i]->data_type())
                throw Error(RV_ERR_TYPE_MISMATCH, "Column " + std::to_string(i) + " has type " + to_string(columns[i]->data_type()) +
                                                      " but schema expects " + to_string(schema->field(i).data_type()));
        return RecordBatch(std::move(schema), std::move(columns), rows);
    }
    static RecordBatch new_unchecked(SchemaRef schema, std::vector<ArrayRef> columns, size_t num_rows) {
        return RecordBatch(std::move(schema), std::move(columns), num_rows);
    }
    const SchemaRef &schema() const { return schema_; }
    size_t num_rows() const { return num_rows_; }
    size_t num_columns() const { return columns_.size(); }
    const ArrayRef &column(size_t i) const {
        if (i >= columns_.size()) throw Panic("index out of bounds");
        return columns_[i];
    }
    const ArrayRef *column_by_name(const std::string &n) const {
        auto i = schema_->index_of(n);
        return i ? &columns_[*i] : nullptr;
    }
    const std::vector<ArrayRef> &columns() const { return columns_; }
    bool is_empty() const { return num_rows_ == 0; }

    RecordBatch slice(size_t offset, size_t length) const {  // :92-106
        if (offset + length > num_rows_) throw Panic("Slice out of bounds");
        std::vector<ArrayRef> cols;
        for (auto &c : columns_) cols.push_back(c->slice(offset, length));
        return RecordBatch(schema_, std::move(cols), length);
    }
    RecordBatch select_columns(const std::vector<size_t> &indices) const {  // :180-206
        for (size_t i : indices)
            if (i >= num_columns())
                throw Error(RV_ERR_OUT_OF_BOUNDS, "Column index " + std::to_string(i) + " out of bounds for " + std::to_string(num_columns()) + " columns");
        std::vector<Field> f;
        std::vector<ArrayRef> c;
        for (size_t i : indices) {
            f.push_back(schema_->field(i));
            c.push_back(columns_[i]);
        }
        return RecordBatch(std::make_shared<Schema>(f), std::move(c), num_rows_);
    }
    RecordBatch select_columns_by_name(const std::vector<std::string> &names) const {  // :208-219
        std::vector<size_t> idx;
        for (auto &n : names) {
            auto i = schema_->index_of(n);
            if (!i) throw Error(RV_ERR_INVALID_ARG, "Column '" + n + "' not found");
            idx.push_back(*i);
        }
        return select_columns(idx);
    }
    // record_batch.rs:221-243 -> rv_filter (one fused device pass per group of columns)
    RecordBatch filter(const ArrayRef &predicate) const {
        if (predicate->len() != num_rows_)
            throw Error(RV_ERR_LENGTH_MISMATCH, "Predicate length " + std::to_string(predicate->len()) + " doesn't match batch length " + std::to_string(num_rows_));
        if (predicate->data_type() != DataType::Boolean) throw Error(RV_ERR_TYPE_MISMATCH, "Predicate must be a BooleanArray");
        auto handles = device_handles("filter");
        std::vector<rv_dcolumn *> out(columns_.size(), nullptr);
        uint64_t rows = 0;
        check(rv_filter(predicate->context()->raw(), handles.data(), static_cast<uint32_t>(handles.size()), predicate->handle(), out.data(), &rows));
        return adopt_all(predicate->context(), out, rows);
    }
    // record_batch.rs:108-129 -> rv_take
    RecordBatch take(const std::vector<size_t> &indices) const {
        for (size_t i : indices)
            if (i >= num_rows_) throw Error(RV_ERR_OUT_OF_BOUNDS, "Index " + std::to_string(i) + " out of bounds for " + std::to_string(num_rows_) + " rows");
        if (columns_.empty()) return RecordBatch(schema_, {}, indices.size());
        auto handles = device_handles("take");
        std::vector<uint64_t> idx(indices.begin(), indices.end());
        std::vector<rv_dcolumn *> out(columns_.size(), nullptr);
        check(rv_take(ctx()->raw(), handles.data(), static_cast<uint32_t>(handles.size()), idx.data(), idx.size(), out.data()));
        return adopt_all(ctx(), out, indices.size());
    }
    // record_batch.rs:245-275 -> rv_concat per column position
    static RecordBatch concat(const std::vector<RecordBatch> &batches) {
        if (batches.empty()) throw Error(RV_ERR_INVALID_ARG, "Cannot concatenate empty batch list");
        for (size_t i = 1; i < batches.size(); ++i)
            if (*batches[i].schema_ != *batches[0].schema_) throw Error(RV_ERR_TYPE_MISMATCH, "All batches must have the same schema");
        size_t total = 0;
        for (auto &b : batches) total += b.num_rows_;
        std::vector<ArrayRef> cols;
        for (size_t c = 0; c < batches[0].num_columns(); ++c) {
            std::vector<const rv_dcolumn *> parts;
            for (auto &b : batches) {
                if (!b.columns_[c]->on_device()) throw Error(RV_ERR_UNSUPPORTED, "concat: String columns are outside the device path");
                parts.push_back(b.columns_[c]->handle());
            }
            rv_dcolumn *out = nullptr;
            check(rv_concat(batches[0].ctx()->raw(), parts.data(), static_cast<uint32_t>(parts.size()), &out));
            cols.push_back(Array::adopt(batches[0].ctx(), out));
        }
        return RecordBatch(batches[0].schema_, std::move(cols), total);
    }
    // record_batch.rs:402-421
    static RecordBatch empty(const ContextRef &ctx, SchemaRef schema) {
        std::vector<ArrayRef> cols;
        for (auto &f : schema->fields()) {
            switch (f.data_type()) {
                case DataType::Int64: cols.push_back(Int64Array::from_values(ctx, {})); break;
                case DataType::Float64: cols.push_back(Float64Array::from_values(ctx, {})); break;
                case DataType::Boolean: cols.push_back(BooleanArray::from_bools(ctx, {})); break;
                case DataType::String: cols.push_back(StringArray::create(ctx, {})); break;
                default: cols.push_back(NullArray::create(ctx, 0)); break;
            }
        }
        return RecordBatch(std::move(schema), std::move(cols), 0);
    }
    ContextRef ctx() const {
        for (auto &c : columns_)
            if (c->on_device()) return c->context();
        throw Error(RV_ERR_INVALID_ARG, "batch has no device column");
    }

  private:
    RecordBatch(SchemaRef s, std::vector<ArrayRef> c, size_t n) : schema_(std::move(s)), columns_(std::move(c)), num_rows_(n) {}
    std::vector<const rv_dcolumn *> device_handles(const char *what) const {
        std::vector<const rv_dcolumn *> h;
        for (auto &c : columns_) {
            if (!c->on_device()) throw Error(RV_ERR_UNSUPPORTED, std::string(what) + ": String columns are outside the device path");
            h.push_back(c->handle());
        }
        return h;
    }
    RecordBatch adopt_all(const ContextRef &ctx, const std::vector<rv_dcolumn *> &out, size_t rows) const {
        std::vector<ArrayRef> cols;
        for (auto *h : out) cols.push_back(Array::adopt(ctx, h));
        return RecordBatch(schema_, std::move(cols), rows);
    }
    SchemaRef schema_;
    std::vector<ArrayRef> columns_;
    size_t num_rows_;
};

// ---------------------------------------------------------------------------------------
// streams -- stream.rs:7-213
// ---------------------------------------------------------------------------------------
struct StreamError : std::runtime_error {
    enum Kind { Execution, SchemaMismatch, Exhausted, Io } kind;
    std::string message;  // StreamError::Execution { message }
    StreamError(Kind k, const std::string &display, std::string msg = "") : std::runtime_error(display), kind(k), message(std::move(msg)) {}
    static StreamError execution(const std::string &m) { return StreamError(Execution, "Stream execution error: " + m, m); }
    static StreamError schema_mismatch() { return StreamError(SchemaMismatch, "Schema mismatch"); }
};

class DataStream {  // trait DataStream, stream.rs:25-54
  public:
    virtual ~DataStream() = default;
    virtual SchemaRef schema() const = 0;
    virtual std::optional<RecordBatch> next_batch() = 0;
    // Not in the reference's trait: a consumer that will take at most `rows` rows (LimitStream, streaming.rs:246-288) says
    // so once, before it pulls.  Row-preserving operators pass it on; the device operators size the window they filter
    // ahead by it instead of by their default of 2^28 rows.  A hint, never a contract: pulling past it still works.
    virtual void limit_hint(size_t /*rows*/) {}
    std::vector<RecordBatch> collect() {
        std::vector<RecordBatch> out;
        while (auto b = next_batch()) out.push_back(std::move(*b));
        return out;
    }
};
using DataStreamRef = std::unique_ptr<DataStream>;

class MemoryStream : public DataStream {  // stream.rs:58-114
  public:
    MemoryStream(SchemaRef schema, std::vector<RecordBatch> batches) : schema_(std::move(schema)), batches_(std::move(batches)) {
        for (auto &b : batches_)
            if (*b.schema() != *schema_) throw StreamError::schema_mismatch();
    }
    static std::unique_ptr<MemoryStream> from_single_batch(RecordBatch b) {
        auto s = b.schema();
        std::vector<RecordBatch> v;
        v.push_back(std::move(b));
        return std::make_unique<MemoryStream>(s, std::move(v));
    }
    static std::unique_ptr<MemoryStream> empty(SchemaRef s) { return std::make_unique<MemoryStream>(std::move(s), std::vector<RecordBatch>{}); }
    SchemaRef schema() const override { return schema_; }
    std::optional<RecordBatch> next_batch() override {
        if (index_ < batches_.size()) return batches_[index_++];
        return std::nullopt;
    }

  private:
    SchemaRef schema_;
    std::vector<RecordBatch> batches_;
    size_t index_ = 0;
};

// FilterStream -- stream.rs:116-163: predicate = a Boolean column of the batch
class FilterStream : public DataStream {
  public:
    FilterStream(DataStreamRef input, std::string predicate_column) : input_(std::move(input)), predicate_column_(std::move(predicate_column)) {}
    SchemaRef schema() const override { return input_->schema(); }
    std::optional<RecordBatch> next_batch() override {
        auto batch = input_->next_batch();
        if (!batch) return std::nullopt;
        auto idx = batch->schema()->index_of(predicate_column_);
        if (!idx) throw StreamError::execution("Column '" + predicate_column_ + "' not found in schema");
        const auto &pred = batch->column(*idx);
        if (pred->data_type() != DataType::Boolean) throw StreamError::execution("Predicate column '" + predicate_column_ + "' is not of boolean type");
        try {
            return batch->filter(pred);  // empty batches are still emitted (:156-158)
        } catch (const Error &e) {
            throw StreamError::execution(e.what());
        }
    }

  private:
    DataStreamRef input_;
    std::string predicate_column_;
};

class SelectStream : public DataStream {  // stream.rs:165-213
  public:
    SelectStream(DataStreamRef input, std::vector<std::string> column_names) : input_(std::move(input)), column_names_(std::move(column_names)) {
        auto in = input_->schema();
        std::vector<Field> f;
        for (auto &n : column_names_) {
            auto fld = in->field_by_name(n);
            if (!fld) throw StreamError::execution("Column '" + n + "' not found in schema");
            f.push_back(*fld);
        }
        output_schema_ = std::make_shared<Schema>(f);
    }
    SchemaRef schema() const override { return output_schema_; }
    void limit_hint(size_t rows) override { input_->limit_hint(rows); }  // a projection keeps every row
    std::optional<RecordBatch> next_batch() override {
        auto batch = input_->next_batch();
        if (!batch) return std::nullopt;
        try {
            return batch->select_columns_by_name(column_names_);
        } catch (const Error &e) {
            throw StreamError::execution(e.what());
        }
    }

  private:
    DataStreamRef input_;
    std::vector<std::string> column_names_;
    SchemaRef output_schema_;
};

class LimitStream : public DataStream {  // streaming.rs:246-288
  public:
    LimitStream(DataStreamRef input, size_t limit) : input_(std::move(input)), limit_(limit) { input_->limit_hint(limit_); }
    void limit_hint(size_t rows) override { input_->limit_hint(std::min(rows, limit_)); }
    SchemaRef schema() const override { return input_->schema(); }
    std::optional<RecordBatch> next_batch() override {
        if (rows_returned_ >= limit_) return std::nullopt;  // the device stops launching chunks here
        auto batch = input_->next_batch();
        if (!batch) return std::nullopt;
        const size_t remaining = limit_ - rows_returned_;
        if (batch->num_rows() <= remaining) {
            rows_returned_ += batch->num_rows();
            return batch;
        }
        rows_returned_ += remaining;
        return batch->slice(0, remaining);
    }

  private:
    DataStreamRef input_;
    size_t limit_, rows_returned_ = 0;
};

// One lowered compare term: batch column name <op> literal (planner.rs:134-189 grammar).
using Literal = std::variant<std::monostate, int64_t, double, bool, std::string>;  // AnyValue of a literal
// CsvFileStream -- file_stream.rs:10-336: schema-driven CSV scan into device RecordBatches.
// Split on the delimiter, trim, "" / "null" = null, first line is the header, blank lines skipped,
// batch_size rows per batch (default: calculate_adaptive_batch_size, :345-368).
//
// Null bitmaps of Int64 / Float64 columns: the reference hands its `nulls` flags (true = null) to
// PrimitiveArray::new, whose second argument is a VALIDITY vector (true = valid, primitive.rs:31-33), so
// there a column with a null comes out with every cell's validity inverted (file_stream.rs:213-249).
// CsvNulls::AsReference -- THE DEFAULT: this layer is a drop-in, and a drop-in gives the reference's arrays bit for
// bit, its defect included -- reproduces that; CsvNulls::AsIntended marks the null cells as null and must be asked
// for ("CSV nulls").  String and Boolean columns are right in the reference.
enum class CsvNulls { AsIntended, AsReference };

inline size_t calculate_adaptive_batch_size(const Schema &schema) {  // file_stream.rs:345-368
    size_t row_bytes = 0;
    for (auto &f : schema.fields()) {
        switch (f.data_type()) {
            case DataType::Int64:
            case DataType::Float64: row_bytes += 8; break;
            case DataType::Boolean: row_bytes += 1; break;
            case DataType::String: row_bytes += 32; break;
            default: break;
        }
    }
    if (row_bytes == 0) return 10000;
    return std::min<size_t>(100000, std::max<size_t>(1000, (8u * 1024 * 1024) / row_bytes));
}

class CsvFileStream : public DataStream {
  public:
    // Err(String) of CsvFileStream::new (file_stream.rs:21-41) -> Error(RV_ERR_INVALID_ARG, same text)
    CsvFileStream(ContextRef ctx, const std::string &path, SchemaRef schema, std::optional<size_t> batch_size = std::nullopt,
                  std::optional<char> delimiter = std::nullopt, CsvNulls nulls = CsvNulls::AsReference)
        : ctx_(std::move(ctx)), file_(path), schema_(std::move(schema)), batch_size_(batch_size ? *batch_size : calculate_adaptive_batch_size(*schema_)),
          delimiter_(delimiter.value_or(',')), nulls_(nulls) {
        if (!file_) throw Error(RV_ERR_INVALID_ARG, "Failed to open file: " + std::string(std::strerror(errno)));
        for (auto &f : schema_->fields())
            if (f.data_type() == DataType::Null) throw Error(RV_ERR_UNSUPPORTED, "Null columns are outside the device path");
    }
    SchemaRef schema() const override { return schema_; }
    size_t batch_size() const { return batch_size_; }

    std::optional<RecordBatch> next_batch() override {  // read_batch, file_stream.rs:124-197
        if (finished_) return std::nullopt;
        const size_t ncols = schema_->num_fields();
        std::vector<std::vector<int64_t>> ints(ncols);
        std::vector<std::vector<double>> floats(ncols);
        std::vector<std::vector<std::optional<std::string>>> strings(ncols);
        std::vector<std::vector<std::optional<bool>>> bools(ncols);
        std::vector<std::vector<bool>> is_null(ncols);
        std::string line;
        if (current_line_ == 0) {  // header
            if (!std::getline(file_, line)) {
                finished_ = true;
                return std::nullopt;
            }
            ++current_line_;
        }
        size_t rows = 0;
        while (rows < batch_size_) {
            if (!std::getline(file_, line)) {
                finished_ = true;
                break;
            }
            ++current_line_;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (trim(line).empty()) continue;
            parse_line(line, ints, floats, strings, bools, is_null);
            ++rows;
        }
        if (rows == 0) return std::nullopt;
        std::vector<ArrayRef> cols;  // build_record_batch, file_stream.rs:199-327
        for (size_t c = 0; c < ncols; ++c) {
            const bool any_null = std::find(is_null[c].begin(), is_null[c].end(), true) != is_null[c].end();
            auto validity = [&]() -> std::optional<std::vector<bool>> {
                if (!any_null) return std::nullopt;
                std::vector<bool> v(is_null[c].size());
                for (size_t i = 0; i < v.size(); ++i) v[i] = nulls_ == CsvNulls::AsReference ? is_null[c][i] : !is_null[c][i];
                return v;
            };
            switch (schema_->field(c).data_type()) {
                case DataType::Int64: cols.push_back(Int64Array::create(ctx_, ints[c], validity())); break;
                case DataType::Float64: cols.push_back(Float64Array::create(ctx_, floats[c], validity())); break;
                case DataType::String: cols.push_back(StringArray::create(ctx_, strings[c])); break;
                default: cols.push_back(BooleanArray::create(ctx_, bools[c])); break;
            }
        }
        try {
            return RecordBatch::try_new(schema_, std::move(cols));
        } catch (const Error &e) {
            throw StreamError::execution(std::string("Failed to create RecordBatch: ") + e.what());
        }
    }

  private:
    static std::string trim(const std::string &s) {
        size_t b = 0, e = s.size();
        while (b < e && std::isspace(static_cast<unsigned char>(s[b]))) ++b;
        while (e > b && std::isspace(static_cast<unsigned char>(s[e - 1]))) --e;
        return s.substr(b, e - b);
    }
    [[noreturn]] void fail(const std::string &what) const { throw StreamError::execution("Parse error: " + what); }
    void parse_line(const std::string &line, std::vector<std::vector<int64_t>> &ints, std::vector<std::vector<double>> &floats,
                    std::vector<std::vector<std::optional<std::string>>> &strings, std::vector<std::vector<std::optional<bool>>> &bools,
                    std::vector<std::vector<bool>> &is_null) const {  // file_stream.rs:43-122
        std::vector<std::string> fields;
        size_t start = 0;
        for (;;) {
            const size_t p = line.find(delimiter_, start);
            fields.push_back(trim(line.substr(start, p == std::string::npos ? std::string::npos : p - start)));
            if (p == std::string::npos) break;
            start = p + 1;
        }
        if (fields.size() != schema_->num_fields())
            fail("Line " + std::to_string(current_line_) + ": Expected " + std::to_string(schema_->num_fields()) + " fields, found " + std::to_string(fields.size()));
        // the row is appended only once every field has parsed (the reference returns before pushing anything)
        struct Cell {
            bool null = false;
            int64_t i = 0;
            double f = 0;
            bool b = false;
        };
        std::vector<Cell> cells(fields.size());
        for (size_t c = 0; c < fields.size(); ++c) {
            const std::string &f = fields[c];
            Cell &cell = cells[c];
            cell.null = f.empty() || f == "null";
            if (cell.null) continue;
            auto bad = [&](const char *type) { fail("Line " + std::to_string(current_line_) + ", field " + std::to_string(c) + ": Cannot parse '" + f + "' as " + type); };
            switch (schema_->field(c).data_type()) {
                case DataType::Int64: {  // str::parse::<i64>: optional sign, decimal digits only
                    size_t k = (f[0] == '+' || f[0] == '-') ? 1 : 0;
                    bool ok = k < f.size();
                    for (size_t q = k; q < f.size(); ++q) ok = ok && std::isdigit(static_cast<unsigned char>(f[q]));
                    errno = 0;
                    char *end = nullptr;
                    const long long v = ok ? std::strtoll(f.c_str(), &end, 10) : 0;
                    if (!ok || errno == ERANGE || *end) bad("Int64");
                    cell.i = v;
                    break;
                }
                case DataType::Float64: {  // str::parse::<f64>: decimal / exponent forms, inf, infinity, nan (any case); no hex
                    bool ok = true;
                    for (char ch : f) ok = ok && (std::isalnum(static_cast<unsigned char>(ch)) || ch == '+' || ch == '-' || ch == '.');
                    if (f.find_first_of("xXpP") != std::string::npos) ok = false;
                    char *end = nullptr;
                    const double v = ok ? std::strtod(f.c_str(), &end) : 0.0;
                    if (!ok || end == f.c_str() || *end) bad("Float64");
                    cell.f = v;
                    break;
                }
                case DataType::Boolean: {
                    std::string l = f;
                    for (auto &ch : l) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
                    if (l == "true" || l == "t" || l == "1") cell.b = true;
                    else if (l == "false" || l == "f" || l == "0") cell.b = false;
                    else bad("Boolean");
                    break;
                }
                default: break;
            }
        }
        for (size_t c = 0; c < fields.size(); ++c) {
            const Cell &cell = cells[c];
            is_null[c].push_back(cell.null);
            switch (schema_->field(c).data_type()) {
                case DataType::Int64: ints[c].push_back(cell.null ? 0 : cell.i); break;       // placeholder 0 (:218-221)
                case DataType::Float64: floats[c].push_back(cell.null ? 0.0 : cell.f); break;  // placeholder 0.0 (:246-249)
                case DataType::String: strings[c].push_back(cell.null ? std::nullopt : std::optional<std::string>(fields[c])); break;
                default: bools[c].push_back(cell.null ? std::nullopt : std::optional<bool>(cell.b)); break;
            }
        }
    }

    ContextRef ctx_;
    std::ifstream file_;
    SchemaRef schema_;
    size_t batch_size_;
    char delimiter_;
    CsvNulls nulls_;
    size_t current_line_ = 0;
    bool finished_ = false;
};

struct CompareTerm {
    std::string column;
    rv_cmp op;
    Literal literal;  // monostate == Literal(AnyValue::Null); unused for RV_IS_TRUE
};

// A lowered predicate: compare terms + (for OR) the postfix program over them (rv_predicate::expr; empty: AND of
// the terms).  Converts from a plain term list, which is what AND-only call sites pass.
struct LoweredPredicate {
    std::vector<CompareTerm> terms;
    std::vector<uint8_t> expr;
    LoweredPredicate() = default;
    LoweredPredicate(std::vector<CompareTerm> t) : terms(std::move(t)) {}          // NOLINT: implicit on purpose
    LoweredPredicate(std::initializer_list<CompareTerm> t) : terms(t) {}            // NOLINT
};

inline rv_term to_rv_term(const CompareTerm &t, uint32_t column_index) {
    rv_term r{};
    r.column = column_index;
    r.op = t.op;
    switch (t.literal.index()) {
        case 0: r.lit_type = RV_NULL; break;
        case 1: r.lit_type = RV_INT64; r.lit.i = std::get<1>(t.literal); break;
        case 2: r.lit_type = RV_FLOAT64; r.lit.f = std::get<2>(t.literal); break;
        case 3: r.lit_type = RV_BOOLEAN; r.lit.i = std::get<3>(t.literal); break;
        default: {  // String literal: borrowed from the term for the duration of the call
            const std::string &str = std::get<4>(t.literal);
            r.lit_type = RV_STRING;
            r.lit.s.ptr = str.data();
            r.lit.s.len = str.size();
            break;
        }
    }
    return r;
}

// Per-batch survivor counts in memory the device can write (rv_host_alloc): rv_filter_project_chunked then lets the fused
// pass drop the counts there itself -- 8 bytes per batch cross PCIe once and the host copies nothing.
class PinnedCounts {
  public:
    PinnedCounts() = default;
    PinnedCounts(const PinnedCounts &) = delete;
    PinnedCounts &operator=(const PinnedCounts &) = delete;
    ~PinnedCounts() { release(); }
    uint64_t *reserve(const ContextRef &ctx, size_t n) {
        if (n > cap_ || ctx != ctx_) {
            release();
            void *p = nullptr;
            check(rv_host_alloc(ctx->raw(), std::max<size_t>(n, 1024) * 8, &p));
            ptr_ = static_cast<uint64_t *>(p);
            cap_ = std::max<size_t>(n, 1024);
            ctx_ = ctx;
        }
        return ptr_;
    }
    uint64_t operator[](size_t i) const { return ptr_[i]; }
    void swap(PinnedCounts &o) {
        std::swap(ptr_, o.ptr_);
        std::swap(cap_, o.cap_);
        std::swap(ctx_, o.ctx_);
    }

  private:
    void release() {
        if (ptr_) rv_host_free(ctx_->raw(), ptr_);
        ptr_ = nullptr;
        cap_ = 0;
    }
    ContextRef ctx_;
    uint64_t *ptr_ = nullptr;
    size_t cap_ = 0;
};

// How many input rows a device stream filters ahead when its consumer has announced a limit (DataStream::limit_hint):
// the survivors still owed divided by the selectivity seen so far (before the first window: the context's last fused
// launch, else 1 in 256), with a margin, and at least twice the previous window after a window that fell short.
class LimitWindow {
  public:
    void hint(size_t rows) { hint_ = hint_ ? std::min(*hint_, rows) : rows; }
    // rows to scan next, at most `cap`
    size_t next(const ContextRef &ctx, size_t cap) {
        if (!hint_ || survivors_ >= *hint_) return cap;  // no limit known / already met: the default window
        double sel = scanned_ ? std::max(static_cast<double>(survivors_) / static_cast<double>(scanned_), 1e-7) : 1.0 / 256;
        if (!scanned_) {
            int64_t ppm = -1;
            if (rv_ctx_get_option(ctx->raw(), "last_selectivity_ppm", &ppm) == RV_OK && ppm > 0) sel = std::max(1e-6 * static_cast<double>(ppm), 1e-7);
        }
        double want = static_cast<double>(*hint_ - survivors_) / sel * 1.5 + 1024.0;
        want = std::max(want, 2.0 * static_cast<double>(last_));
        const size_t rows = want >= static_cast<double>(cap) ? cap : static_cast<size_t>(want);
        return std::max<size_t>(1, rows);
    }
    void record(size_t scanned, size_t survivors) {
        last_ = scanned;
        scanned_ += scanned;
        survivors_ += survivors;
    }
    size_t scanned() const { return scanned_; }
    bool hinted() const { return hint_.has_value(); }

  private:
    std::optional<size_t> hint_;
    size_t scanned_ = 0, survivors_ = 0, last_ = 0;
};

// GpuFilterProjectStream -- SelectStream(FilterStream(input)) fused, the operator behind seam S1 (INTEGRATION.md
// section 3).  The reference pulls 1024-row batches (streaming_planner.rs:32); a device launch per such batch would
// cost ~30 us for ~10 ns of work, so the operator pulls a WINDOW of input batches ahead (at most window_batches
// batches / window_rows rows), hands all of them to ONE rv_filter_project_batches call (one pass over HBM, one
// read-back of the per-batch row counts) and then emits the output batches one by one as zero-copy slices of the
// joint result -- batch for batch what FilterStream + SelectStream would have produced.
class GpuFilterProjectStream : public DataStream {
  public:
    GpuFilterProjectStream(DataStreamRef input, LoweredPredicate predicate, std::vector<std::string> projection,
                           rv_null_policy nulls = RV_NULL_DROPS, size_t window_batches = 4096, size_t window_rows = size_t(1) << 28)
        : input_(std::move(input)), terms_(std::move(predicate.terms)), expr_(std::move(predicate.expr)), projection_(std::move(projection)), nulls_(nulls),
          window_batches_(std::max<size_t>(1, window_batches)), window_rows_(window_rows) {
        auto in = input_->schema();
        std::vector<Field> f;
        for (auto &n : projection_) {
            auto fld = in->field_by_name(n);
            if (!fld) throw StreamError::execution("Column '" + n + "' not found in schema");
            f.push_back(*fld);
        }
        for (auto &t : terms_)
            if (!in->field_by_name(t.column)) throw StreamError::execution("Column '" + t.column + "' not found in schema");
        output_schema_ = std::make_shared<Schema>(f);
    }
    SchemaRef schema() const override { return output_schema_; }
    void limit_hint(size_t rows) override { limit_.hint(rows); }
    size_t rows_scanned() const { return limit_.scanned(); }

    std::optional<RecordBatch> next_batch() override {
        if (next_ready_ == ready_.size()) refill();
        if (next_ready_ < ready_.size()) return std::move(ready_[next_ready_++]);
        if (pending_error_) {  // raised by the call that would have returned the failing batch, as without the look-ahead
            auto e = pending_error_;
            pending_error_ = nullptr;
            std::rethrow_exception(e);
        }
        return std::nullopt;
    }

  private:
    void refill() {
        ready_.clear();
        next_ready_ = 0;
        if (exhausted_ || pending_error_) return;
        std::vector<RecordBatch> window;
        size_t rows = 0;
        size_t want_rows = window_rows_;  // under a limit: only as far ahead as the rows still owed need
        try {
            while (window.size() < window_batches_ && rows < want_rows) {
                auto b = input_->next_batch();
                if (!b) {
                    exhausted_ = true;
                    break;
                }
                if (window.empty() && b->num_columns() > 0) want_rows = limit_.next(b->ctx(), window_rows_);
                rows += b->num_rows();
                window.push_back(std::move(*b));
            }
        } catch (const StreamError &) {
            pending_error_ = std::current_exception();  // the batches pulled so far are still delivered first
            exhausted_ = true;
        }
        if (window.empty()) return;
        try {
            // device columns referenced by the predicate or the projection, each once (the same slots in every batch)
            const RecordBatch &first = window[0];
            std::vector<size_t> batch_index;
            auto slot_of = [&](const std::string &name) -> uint32_t {
                const size_t bi = *first.schema()->index_of(name);
                for (size_t k = 0; k < batch_index.size(); ++k)
                    if (batch_index[k] == bi) return static_cast<uint32_t>(k);
                batch_index.push_back(bi);
                return static_cast<uint32_t>(batch_index.size() - 1);
            };
            std::vector<rv_term> rt;
            for (auto &t : terms_) rt.push_back(to_rv_term(t, slot_of(t.column)));
            std::vector<uint32_t> proj;
            for (auto &n : projection_) proj.push_back(slot_of(n));
            const size_t ncols = batch_index.size(), nb = window.size(), np = proj.size();
            std::vector<const rv_dcolumn *> cols(nb * ncols);
            for (size_t b = 0; b < nb; ++b)
                for (size_t c = 0; c < ncols; ++c) cols[b * ncols + c] = window[b].column(batch_index[c])->handle();
            rv_predicate pred{rt.data(), static_cast<uint32_t>(rt.size()), nulls_, expr_.empty() ? nullptr : expr_.data(), static_cast<uint32_t>(expr_.size())};
            const ContextRef ctx = first.ctx();
            std::vector<rv_dcolumn *> out(np ? np : 1, nullptr);
            std::vector<uint64_t> out_rows(nb, 0);
            std::vector<int64_t> out_nulls(nb * (np ? np : 1), 0);
            uint64_t total = 0;
            check(rv_filter_project_batches(ctx->raw(), cols.data(), static_cast<uint32_t>(nb), static_cast<uint32_t>(ncols), &pred, proj.data(),
                                            static_cast<uint32_t>(np), out.data(), out_rows.data(), out_nulls.data(), &total));
            std::vector<ArrayRef> joined;
            for (size_t j = 0; j < np; ++j) joined.push_back(Array::adopt(ctx, out[j]));
            limit_.record(rows, total);
            uint64_t at = 0;
            for (size_t b = 0; b < nb; ++b) {  // every input batch yields its output batch, empty ones included (stream.rs:156-158)
                std::vector<ArrayRef> arrays;
                for (size_t j = 0; j < np; ++j) {
                    rv_dcolumn *piece = nullptr;
                    check(rv_slice_known(ctx->raw(), joined[j]->handle(), at, out_rows[b], out_nulls[b * np + j], &piece));
                    arrays.push_back(Array::adopt(ctx, piece));
                }
                ready_.push_back(RecordBatch::new_unchecked(output_schema_, std::move(arrays), out_rows[b]));
                at += out_rows[b];
            }
        } catch (const Error &e) {
            ready_.clear();
            pending_error_ = std::make_exception_ptr(StreamError::execution(e.what()));
        }
    }

    DataStreamRef input_;
    std::vector<CompareTerm> terms_;
    std::vector<uint8_t> expr_;
    std::vector<std::string> projection_;
    rv_null_policy nulls_;
    size_t window_batches_, window_rows_;
    SchemaRef output_schema_;
    std::vector<RecordBatch> ready_;
    size_t next_ready_ = 0;
    bool exhausted_ = false;
    std::exception_ptr pending_error_;
    LimitWindow limit_;
};

// GpuChunkedFilterProjectStream -- Select(Filter(DataFrameSource)) fused: the table stays whole in HBM and the library
// is told the batch size (rv_filter_project_chunked) instead of being handed one handle per 1024-row slice.  Emits
// exactly the batches SelectStream(FilterStream(MemoryStream(dataframe_to_batches(df, batch_size)))) would, one per input
// batch, empty ones included; `columns` are the frame's columns after dataframe_to_batches' null fill.
class GpuChunkedFilterProjectStream : public DataStream {
  public:
    GpuChunkedFilterProjectStream(std::vector<std::string> names, std::vector<ArrayRef> columns, size_t batch_size, LoweredPredicate predicate,
                                  std::vector<std::string> projection, rv_null_policy nulls = RV_NULL_DROPS, size_t window_rows = size_t(1) << 28)
        : names_(std::move(names)), columns_(std::move(columns)), batch_size_(batch_size), terms_(std::move(predicate.terms)),
          expr_(std::move(predicate.expr)), projection_(std::move(projection)), nulls_(nulls) {
        std::vector<Field> in;
        for (size_t c = 0; c < columns_.size(); ++c) in.push_back(Field{names_[c], columns_[c]->data_type(), true});
        Schema input_schema(in);
        std::vector<Field> f;
        for (auto &n : projection_) {
            auto fld = input_schema.field_by_name(n);
            if (!fld) throw StreamError::execution("Column '" + n + "' not found in schema");
            f.push_back(*fld);
        }
        for (auto &t : terms_)
            if (!input_schema.field_by_name(t.column)) throw StreamError::execution("Column '" + t.column + "' not found in schema");
        output_schema_ = std::make_shared<Schema>(f);
        rows_ = columns_.empty() ? 0 : columns_[0]->len();
        window_batches_ = std::max<size_t>(1, window_rows / std::max<size_t>(1, batch_size_));
        auto slot_of = [&](const std::string &name) -> uint32_t {
            const size_t ci = *input_schema.index_of(name);
            for (size_t k = 0; k < used_.size(); ++k)
                if (used_[k] == ci) return static_cast<uint32_t>(k);
            used_.push_back(ci);
            return static_cast<uint32_t>(used_.size() - 1);
        };
        for (auto &t : terms_) rt_.push_back(to_rv_term(t, slot_of(t.column)));
        for (auto &n : projection_) proj_.push_back(slot_of(n));
    }
    ~GpuChunkedFilterProjectStream() override {  // a window still in flight: waited for and dropped
        if (ahead_.pending) {
            std::vector<rv_dcolumn *> out(proj_.size() ? proj_.size() : 1, nullptr);
            rv_pending *pending = ahead_.pending;
            ahead_.pending = nullptr;
            if (rv_filter_project_window_finish(columns_[0]->context()->raw(), pending, out.data(), nullptr, nullptr) == RV_OK)
                for (size_t j = 0; j < proj_.size(); ++j) rv_free(columns_[0]->context()->raw(), out[j]);
        }
    }
    SchemaRef schema() const override { return output_schema_; }
    void limit_hint(size_t rows) override { limit_.hint(rows); }
    size_t rows_scanned() const { return limit_.scanned(); }

    std::optional<RecordBatch> next_batch() override {
        if (next_in_window_ == window_batches_out_) {
            if (next_row_ >= rows_ && !ahead_.pending) return std::nullopt;
            refill();
        }
        const size_t b = next_in_window_++, np = proj_.size();
        const ContextRef ctx = columns_[0]->context();
        std::vector<ArrayRef> arrays;
        for (size_t j = 0; j < np; ++j) {
            rv_dcolumn *piece = nullptr;
            check_stream(rv_slice_known(ctx->raw(), joined_[j]->handle(), at_, window_rows_out_[b], window_nulls_[b * np + j], &piece));
            arrays.push_back(Array::adopt(ctx, piece));
        }
        at_ += window_rows_out_[b];
        return RecordBatch::new_unchecked(output_schema_, std::move(arrays), window_rows_out_[b]);
    }

  private:
    static void check_stream(rv_status st) {
        try {
            check(st);
        } catch (const Error &e) {
            throw StreamError::execution(e.what());
        }
    }
    // A window whose pass is queued on the device (rv_filter_project_chunked_begin): its views, predicate and count block stay put
    // until rv_filter_project_window_finish has consumed the pending handle.
    struct Ahead {
        rv_pending *pending = nullptr;
        std::vector<ArrayRef> views;
        std::vector<const rv_dcolumn *> cols;
        rv_predicate pred{};
        size_t len = 0, nb = 0;
    };
    void begin_window() {  // rows [next_row_, next_row_ + len): a whole number of batches
        const ContextRef ctx = columns_[0]->context();
        // under a limit (LimitStream told us how many rows it will take) only as far ahead as the rows still owed need
        const size_t want = limit_.next(ctx, window_batches_ * batch_size_);
        const size_t want_batches = std::max<size_t>(1, (want + batch_size_ - 1) / batch_size_);
        Ahead a;
        a.len = std::min(rows_ - next_row_, std::min(window_batches_, want_batches) * batch_size_);
        a.nb = (a.len + batch_size_ - 1) / batch_size_;
        for (size_t c : used_) {
            a.views.push_back(columns_[c]->slice(next_row_, a.len));
            a.cols.push_back(a.views.back()->handle());
        }
        a.pred = rv_predicate{rt_.data(), static_cast<uint32_t>(rt_.size()), nulls_, expr_.empty() ? nullptr : expr_.data(), static_cast<uint32_t>(expr_.size())};
        ahead_ = std::move(a);
        uint64_t *rows_out = next_rows_out_.reserve(ctx, ahead_.nb);  // pinned: written by the device (PinnedCounts)
        check_stream(rv_filter_project_chunked_begin(ctx->raw(), ahead_.cols.data(), static_cast<uint32_t>(ahead_.cols.size()), batch_size_, &ahead_.pred, proj_.data(),
                                                     static_cast<uint32_t>(proj_.size()), rows_out, ahead_.nb, &ahead_.pending));
        next_row_ += ahead_.len;
    }
    void refill() {  // the next window of the table
        const ContextRef ctx = columns_[0]->context();
        const size_t np = proj_.size();
        const size_t row_before = next_row_;
        if (!ahead_.pending) {
            try {
                begin_window();
            } catch (...) {
                next_row_ = row_before;  // a failed refill leaves the stream where it was (next_batch() raises again)
                throw;
            }
        }
        std::vector<rv_dcolumn *> out(np ? np : 1, nullptr);
        // filled in locals and committed only when the call succeeded
        std::vector<int64_t> nulls_out(ahead_.nb * (np ? np : 1), 0);
        uint64_t total = 0;
        rv_pending *pending = ahead_.pending;
        ahead_.pending = nullptr;  // finish consumes it, also on error
        const rv_status st = rv_filter_project_window_finish(ctx->raw(), pending, out.data(), nulls_out.data(), &total);
        if (st != RV_OK) {
            next_row_ -= ahead_.len;
            check_stream(st);
        }
        joined_.clear();
        for (size_t j = 0; j < np; ++j) joined_.push_back(Array::adopt(ctx, out[j]));
        window_rows_out_.swap(next_rows_out_);
        window_batches_out_ = ahead_.nb;
        window_nulls_ = std::move(nulls_out);
        limit_.record(ahead_.len, total);
        next_in_window_ = 0;
        at_ = 0;
        // Two windows in flight (stream.rs:25-28: the consumer pulls batch by batch): the next window's pass is queued now and runs
        // while this window's batches are handed on -- the device does not wait for the host between windows.  Not under a limit:
        // how far to look ahead then follows from what this window kept.
        if (next_row_ < rows_ && !limit_.hinted()) {
            try {
                begin_window();
            } catch (...) {  // reported by the refill that needs the window
                ahead_.pending = nullptr;
            }
        }
    }

    std::vector<std::string> names_;
    std::vector<ArrayRef> columns_;
    size_t batch_size_;
    std::vector<CompareTerm> terms_;
    std::vector<uint8_t> expr_;
    std::vector<std::string> projection_;
    rv_null_policy nulls_;
    SchemaRef output_schema_;
    std::vector<size_t> used_;  // frame columns the predicate or the projection reads, each once
    std::vector<rv_term> rt_;
    std::vector<uint32_t> proj_;
    size_t rows_ = 0, next_row_ = 0, window_batches_ = 1;
    std::vector<ArrayRef> joined_;  // the current window's outputs, all its batches back to back
    PinnedCounts window_rows_out_, next_rows_out_;  // the current window's per-batch counts / the block the next refill fills
    size_t window_batches_out_ = 0;
    std::vector<int64_t> window_nulls_;
    size_t next_in_window_ = 0;
    uint64_t at_ = 0;
    LimitWindow limit_;
    Ahead ahead_;
};

}  // namespace execution

// ---------------------------------------------------------------------------------------
// expressions -- expr.rs:3-139
// ---------------------------------------------------------------------------------------
namespace expressions {
enum class BinaryOperator { Plus, Minus, Multiply, Divide, Eq, NotEq, Lt, Gt, LtEq, GtEq, And, Or };
struct Expr;
using ExprPtr = std::shared_ptr<const Expr>;
struct Expr {
    enum Kind { Column, Literal, BinaryExpr, Alias } kind = Column;
    std::string name;             // Column / Alias
    execution::Literal literal;   // Literal
    ExprPtr left, right;          // BinaryExpr (left, right) / Alias (left)
    BinaryOperator op = BinaryOperator::Eq;

    static Expr col(const std::string &n) {
        Expr e;
        e.kind = Column;
        e.name = n;
        return e;
    }
    static Expr lit(execution::Literal v) {
        Expr e;
        e.kind = Literal;
        e.literal = std::move(v);
        return e;
    }
    static Expr lit(int v) { return lit(execution::Literal(static_cast<int64_t>(v))); }
    static Expr lit(const char *v) { return lit(execution::Literal(std::string(v))); }
    Expr alias(const std::string &n) const {
        Expr e;
        e.kind = Alias;
        e.name = n;
        e.left = std::make_shared<Expr>(*this);
        return e;
    }
    Expr binary(BinaryOperator o, const Expr &r) const {
        Expr e;
        e.kind = BinaryExpr;
        e.op = o;
        e.left = std::make_shared<Expr>(*this);
        e.right = std::make_shared<Expr>(r);
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
}  // namespace expressions

// ---------------------------------------------------------------------------------------
// physical_plan -- planner.rs, streaming_planner.rs, streaming.rs, plan.rs
// ---------------------------------------------------------------------------------------
namespace physical_plan {
using execution::CompareTerm;
using expressions::BinaryOperator;
using expressions::Expr;

struct ConversionError : std::runtime_error {  // planner.rs:8-39
    enum Kind { UnsupportedExpression, UnsupportedFilter, InvalidFilterStructure, FilterLeftNotColumn, FilterRightNotLiteral,
                UnsupportedFilterOperator, InvalidSelectExpression } kind;
    ConversionError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};
struct StreamingPlannerError : std::runtime_error {  // streaming_planner.rs:11-27
    enum Kind { ExpressionError } kind = ExpressionError;
    using std::runtime_error::runtime_error;
};

inline bool is_compare(BinaryOperator op) { return op >= BinaryOperator::Eq && op <= BinaryOperator::GtEq; }
inline rv_cmp to_cmp(BinaryOperator op) {
    switch (op) {
        case BinaryOperator::Eq: return RV_EQ;
        case BinaryOperator::NotEq: return RV_NE;
        case BinaryOperator::Lt: return RV_LT;
        case BinaryOperator::Gt: return RV_GT;
        case BinaryOperator::LtEq: return RV_LE;
        default: return RV_GE;
    }
}

// planner.rs:113-132
inline std::pair<std::string, std::string> convert_select_expr(const Expr &e) {
    switch (e.kind) {
        case Expr::Column: return {e.name, e.name};
        case Expr::Alias:
            if (e.left->kind == Expr::Column) return {e.left->name, e.name};
            throw ConversionError(ConversionError::UnsupportedExpression, "Unsupported expression");
        case Expr::BinaryExpr: throw ConversionError(ConversionError::UnsupportedExpression, "Unsupported expression");
        default: throw ConversionError(ConversionError::InvalidSelectExpression, "Select expression must be a column or alias");
    }
}

// planner.rs:134-189: the eager grammar -- exactly one `Column <cmp> Literal`
inline CompareTerm convert_filter_predicate(const Expr &p) {
    if (p.kind != Expr::BinaryExpr) {
        const char *t = p.kind == Expr::Column ? "Column" : (p.kind == Expr::Literal ? "Literal" : "Alias");
        throw ConversionError(ConversionError::InvalidFilterStructure, std::string("Filter must be a binary comparison, found: ") + t);
    }
    if (p.op == BinaryOperator::And || p.op == BinaryOperator::Or)
        throw ConversionError(ConversionError::UnsupportedFilter, "Unsupported filter: only simple column comparisons supported");
    if (!is_compare(p.op)) throw ConversionError(ConversionError::UnsupportedFilterOperator, "Unsupported binary operator in filter");
    if (p.left->kind != Expr::Column) throw ConversionError(ConversionError::FilterLeftNotColumn, "Filter left side must be a column reference");
    if (p.right->kind != Expr::Literal) throw ConversionError(ConversionError::FilterRightNotLiteral, "Filter right side must be a literal value");
    return CompareTerm{p.left->name, to_cmp(p.op), p.right->literal};
}

// streaming_planner.rs:102-135 (alias name dropped, :110-113)
inline std::vector<std::string> extract_column_names_from_expressions(const std::vector<Expr> &exprs) {
    std::vector<std::string> out;
    for (auto &e : exprs) {
        if (e.kind == Expr::Column) out.push_back(e.name);
        else if (e.kind == Expr::Alias && e.left->kind == Expr::Column) out.push_back(e.left->name);
        else if (e.kind == Expr::Alias) throw StreamingPlannerError("Expression conversion error: Complex expressions with aliases not yet supported");
        else throw StreamingPlannerError("Expression conversion error: Complex expressions not yet supported in streaming mode");
    }
    return out;
}

// streaming_planner.rs:137-168: what the REFERENCE accepts (a bare Boolean column)
inline std::string extract_boolean_predicate_column(const Expr &p) {
    if (p.kind == Expr::Column) return p.name;
    if (p.kind == Expr::BinaryExpr) {
        if (p.left->kind == Expr::Column)
            throw StreamingPlannerError("Expression conversion error: Binary expressions not yet supported in streaming mode. Found expression on column '" +
                                        p.left->name + "'. Currently only simple boolean column references are supported (e.g., .filter(col('is_active')))");
        throw StreamingPlannerError("Expression conversion error: Complex binary expressions not supported in streaming mode");
    }
    throw StreamingPlannerError("Expression conversion error: Unsupported filter expression type");
}

// The lowering the new backend adds: `Column <cmp> Literal`, a bare Boolean column, and AND / OR trees of those
// (BinaryOperator::{And, Or}, expr.rs:27-28) become the term list + postfix program of one fused device pass.
// Arithmetic stays unsupported (ExpressionError, like the reference).
inline void lower_predicate(const Expr &p, execution::LoweredPredicate &out, bool &has_or) {
    auto push_term = [&](CompareTerm t) {
        if (out.terms.size() >= 16) throw StreamingPlannerError("Expression conversion error: more than 16 compare terms in one filter");
        out.expr.push_back(static_cast<uint8_t>(out.terms.size()));
        out.terms.push_back(std::move(t));
    };
    if (p.kind == Expr::Column) return push_term(CompareTerm{p.name, RV_IS_TRUE, {}});
    if (p.kind == Expr::BinaryExpr && (p.op == BinaryOperator::And || p.op == BinaryOperator::Or)) {
        lower_predicate(*p.left, out, has_or);
        lower_predicate(*p.right, out, has_or);
        out.expr.push_back(p.op == BinaryOperator::And ? RV_EXPR_AND : RV_EXPR_OR);
        has_or = has_or || p.op == BinaryOperator::Or;
        return;
    }
    if (p.kind == Expr::BinaryExpr && is_compare(p.op) && p.left->kind == Expr::Column && p.right->kind == Expr::Literal)
        return push_term(CompareTerm{p.left->name, to_cmp(p.op), p.right->literal});
    throw StreamingPlannerError("Expression conversion error: only AND / OR of `column <cmp> literal` terms and Boolean columns run on the device");
}
inline execution::LoweredPredicate lower_predicate(const Expr &p) {
    execution::LoweredPredicate out;
    bool has_or = false;
    lower_predicate(p, out, has_or);
    if (!has_or) out.expr.clear();  // the AND of the terms: the plain term list (pipelined begin / finish path)
    return out;
}

// StreamingPhysicalPlan -- streaming.rs:29-133, collect :235-238 + :343-352
struct StreamingExecutionError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct DeviceFrame {  // the typed-column stand-in for datatypes::DataFrame
    std::vector<std::string> names;
    std::vector<execution::ArrayRef> columns;
    size_t height() const { return columns.empty() ? 0 : columns[0]->len(); }
    size_t width() const { return columns.size(); }
    const execution::ArrayRef *column(const std::string &n) const {
        for (size_t i = 0; i < names.size(); ++i)
            if (names[i] == n) return &columns[i];
        return nullptr;
    }
};

// dataframe_to_batches -- streaming.rs:135-233: batch_size-row chunks (zero-copy slices); the null cells of
// Int64 / Float64 / Boolean columns become 0 / 0.0 / false and the column loses its bitmap (the reference
// builds those arrays with from_values / from_bools); String columns keep their nulls; every field nullable.
inline std::vector<execution::RecordBatch> dataframe_to_batches(const DeviceFrame &df, size_t batch_size) {
    using namespace execution;
    std::vector<RecordBatch> batches;
    if (df.columns.empty() || df.height() == 0) return batches;  // df.is_empty()
    if (batch_size == 0) throw Panic("attempt to divide by zero");
    std::vector<Field> fields;
    for (size_t c = 0; c < df.columns.size(); ++c) fields.push_back(Field{df.names[c], df.columns[c]->data_type(), true});
    auto schema = std::make_shared<Schema>(fields);
    const size_t rows = df.height();
    for (size_t start = 0; start < rows; start += batch_size) {
        const size_t len = std::min(batch_size, rows - start);
        std::vector<ArrayRef> arrays;
        for (auto &col : df.columns) {
            ArrayRef piece = col->slice(start, len);
            rv_dcolumn *filled = nullptr;
            check(rv_fill_nulls(piece->context()->raw(), piece->handle(), &filled));
            arrays.push_back(Array::adopt(piece->context(), filled));
        }
        batches.push_back(RecordBatch::try_new(schema, std::move(arrays)));
    }
    return batches;
}

class StreamingPhysicalPlan;
using StreamingPlanPtr = std::shared_ptr<const StreamingPhysicalPlan>;
class StreamingPhysicalPlan {
  public:
    enum Kind { MemorySource, DataFrameSource, CsvFileSource, Filter, GpuFilterProject, Select, Limit } kind = MemorySource;
    // DataFrameSource (streaming.rs:85-94)
    DeviceFrame df;
    size_t df_batch_size = 0;
    // CsvFileSource (streaming.rs:95-105)
    ContextRef csv_ctx;
    std::string csv_path;
    execution::SchemaRef csv_schema;
    std::optional<size_t> csv_batch_size;
    std::optional<char> csv_delimiter;
    execution::CsvNulls csv_nulls = execution::CsvNulls::AsReference;
    std::vector<execution::RecordBatch> batches;
    StreamingPlanPtr input;
    std::string predicate_column;
    execution::LoweredPredicate predicate;
    std::vector<std::string> columns;
    size_t n = 0;

    static StreamingPlanPtr memory_source(std::vector<execution::RecordBatch> b) {
        auto p = std::make_shared<StreamingPhysicalPlan>();
        p->kind = MemorySource;
        p->batches = std::move(b);
        return p;
    }
    static StreamingPlanPtr dataframe_source(DeviceFrame df, size_t batch_size) {
        auto p = std::make_shared<StreamingPhysicalPlan>();
        p->kind = DataFrameSource;
        p->df = std::move(df);
        p->df_batch_size = batch_size;
        return p;
    }
    static StreamingPlanPtr csv_file_source(ContextRef ctx, std::string path, execution::SchemaRef schema, std::optional<size_t> batch_size = std::nullopt,
                                            std::optional<char> delimiter = std::nullopt, execution::CsvNulls nulls = execution::CsvNulls::AsReference) {
        auto p = std::make_shared<StreamingPhysicalPlan>();
        p->kind = CsvFileSource;
        p->csv_ctx = std::move(ctx);
        p->csv_path = std::move(path);
        p->csv_schema = std::move(schema);
        p->csv_batch_size = batch_size;
        p->csv_delimiter = delimiter;
        p->csv_nulls = nulls;
        return p;
    }
    static StreamingPlanPtr filter(StreamingPlanPtr in, std::string predicate_column) {
        auto p = std::make_shared<StreamingPhysicalPlan>();
        p->kind = Filter;
        p->input = std::move(in);
        p->predicate_column = std::move(predicate_column);
        return p;
    }
    // Filter(expr) followed by Select(columns): one fused operator
    static StreamingPlanPtr gpu_filter_project(StreamingPlanPtr in, execution::LoweredPredicate predicate, std::vector<std::string> columns) {
        auto p = std::make_shared<StreamingPhysicalPlan>();
        p->kind = GpuFilterProject;
        p->input = std::move(in);
        p->predicate = std::move(predicate);
        p->columns = std::move(columns);
        return p;
    }
    static StreamingPlanPtr select(StreamingPlanPtr in, std::vector<std::string> columns) {
        auto p = std::make_shared<StreamingPhysicalPlan>();
        p->kind = Select;
        p->input = std::move(in);
        p->columns = std::move(columns);
        return p;
    }
    static StreamingPlanPtr limit(StreamingPlanPtr in, size_t n) {
        auto p = std::make_shared<StreamingPhysicalPlan>();
        p->kind = Limit;
        p->input = std::move(in);
        p->n = n;
        return p;
    }

    execution::DataStreamRef execute() const {  // streaming.rs:71-133
        using namespace execution;
        try {
            switch (kind) {
                case MemorySource:
                    if (batches.empty()) throw StreamingExecutionError("Invalid operation: Cannot create stream from empty batch list");
                    return std::make_unique<MemoryStream>(batches[0].schema(), batches);
                case DataFrameSource: {  // streaming.rs:85-94: an empty frame gives an empty stream with an empty schema
                    auto b = dataframe_to_batches(df, df_batch_size);
                    auto schema = b.empty() ? std::make_shared<const Schema>() : b[0].schema();
                    return std::make_unique<MemoryStream>(schema, std::move(b));
                }
                case CsvFileSource:
                    try {
                        return std::make_unique<CsvFileStream>(csv_ctx, csv_path, csv_schema, csv_batch_size, csv_delimiter, csv_nulls);
                    } catch (const Error &e) {
                        throw StreamingExecutionError(std::string("Invalid operation: ") + e.what());  // streaming.rs:102-103
                    }
                case Filter: return std::make_unique<FilterStream>(input->execute(), predicate_column);
                case GpuFilterProject:
                    // over a resident frame the chunker and the operator fuse: no per-batch handles (rv_filter_project_chunked)
                    if (input->kind == DataFrameSource && !input->df.columns.empty() && input->df.height() > 0 && input->df_batch_size > 0) {
                        std::vector<ArrayRef> filled;
                        for (auto &col : input->df.columns) {  // dataframe_to_batches' null fill, once per column instead of once per batch
                            rv_dcolumn *f = nullptr;
                            check(rv_fill_nulls(col->context()->raw(), col->handle(), &f));
                            filled.push_back(Array::adopt(col->context(), f));
                        }
                        return std::make_unique<GpuChunkedFilterProjectStream>(input->df.names, std::move(filled), input->df_batch_size, predicate, columns);
                    }
                    return std::make_unique<GpuFilterProjectStream>(input->execute(), predicate, columns);
                case Select: return std::make_unique<SelectStream>(input->execute(), columns);
                case Limit: return std::make_unique<LimitStream>(input->execute(), n);
            }
        } catch (const StreamError &e) {
            throw StreamingExecutionError(std::string("Stream error: ") + e.what());
        }
        throw Panic("unreachable");
    }
    std::vector<execution::RecordBatch> collect_batches() const {
        auto s = execute();
        try {
            return s->collect();
        } catch (const execution::StreamError &e) {
            throw StreamingExecutionError(std::string("Stream error: ") + e.what());
        }
    }
    // collect(): concat of every batch, or RecordBatch::empty(schema) when the stream yields nothing
    execution::RecordBatch collect(const ContextRef &ctx) const {
        auto s = execute();
        try {
            auto schema = s->schema();
            auto all = s->collect();
            if (all.empty()) return execution::RecordBatch::empty(ctx, schema);
            try {
                return execution::RecordBatch::concat(all);
            } catch (const Error &e) {
                throw StreamingExecutionError(std::string("Conversion error: ") + e.what());
            }
        } catch (const execution::StreamError &e) {
            throw StreamingExecutionError(std::string("Stream error: ") + e.what());
        }
    }
};

// Eager PhysicalPlan (plan.rs:8-150) over a frame of named, typed device columns.  Filter keeps
// ALL columns and uses the AnyValue ordering for nulls (RV_NULL_IS_LEAST); Select renames.
struct ExecutionError : std::runtime_error {  // plan.rs:36-62
    enum Kind { ColumnNotFound, InvalidOperation, General } kind;
    ExecutionError(Kind k, const std::string &m) : std::runtime_error(m), kind(k) {}
};
class PhysicalPlan;
using PhysicalPlanPtr = std::shared_ptr<const PhysicalPlan>;
class PhysicalPlan {
  public:
    enum Kind { DataFrameSource, Select, Filter } kind = DataFrameSource;
    DeviceFrame df;
    PhysicalPlanPtr input;
    std::vector<std::string> columns, final_names;
    CompareTerm term;  // Filter { column, value, op }

    static PhysicalPlanPtr source(DeviceFrame f) {
        auto p = std::make_shared<PhysicalPlan>();
        p->df = std::move(f);
        return p;
    }
    static PhysicalPlanPtr filter(PhysicalPlanPtr in, CompareTerm t) {
        auto p = std::make_shared<PhysicalPlan>();
        p->kind = Filter;
        p->input = std::move(in);
        p->term = std::move(t);
        return p;
    }
    static PhysicalPlanPtr select(PhysicalPlanPtr in, std::vector<std::string> cols, std::vector<std::string> finals) {
        auto p = std::make_shared<PhysicalPlan>();
        p->kind = Select;
        p->input = std::move(in);
        p->columns = std::move(cols);
        p->final_names = std::move(finals);
        return p;
    }
    DeviceFrame execute() const {
        switch (kind) {
            case DataFrameSource: return df;  // plan.rs:67
            case Select: {                     // plan.rs:68-96
                DeviceFrame in = input->execute();
                DeviceFrame out;
                for (auto &c : columns)
                    if (!in.column(c)) throw ExecutionError(ExecutionError::ColumnNotFound, "Column not found: '" + c + "'");
                for (size_t i = 0; i < columns.size() && i < final_names.size(); ++i) {
                    out.names.push_back(final_names[i]);
                    out.columns.push_back(*in.column(columns[i]));
                }
                return out;
            }
            case Filter: {  // plan.rs:97-150: mask over one column, every column kept
                DeviceFrame in = input->execute();
                const execution::ArrayRef *fc = in.column(term.column);
                if (!fc) throw ExecutionError(ExecutionError::ColumnNotFound, "Column not found: '" + term.column + "'");
                std::vector<const rv_dcolumn *> cols;
                std::vector<uint32_t> proj;
                uint32_t pred_col = 0;
                for (size_t i = 0; i < in.columns.size(); ++i) {
                    if (!in.columns[i]->on_device()) throw Error(RV_ERR_UNSUPPORTED, "String columns are outside the device path");
                    if (&in.columns[i] == fc) pred_col = static_cast<uint32_t>(i);
                    cols.push_back(in.columns[i]->handle());
                    proj.push_back(static_cast<uint32_t>(i));
                }
                rv_term t = execution::to_rv_term(term, pred_col);
                rv_predicate pred{&t, 1, RV_NULL_IS_LEAST, nullptr, 0};
                std::vector<rv_dcolumn *> out(cols.size(), nullptr);
                uint64_t rows = 0;
                const ContextRef ctx = (*fc)->context();
                check(rv_filter_project(ctx->raw(), cols.data(), static_cast<uint32_t>(cols.size()), &pred, proj.data(),
                                        static_cast<uint32_t>(proj.size()), out.data(), &rows, nullptr));
                DeviceFrame res;
                res.names = in.names;
                for (auto *h : out) res.columns.push_back(execution::Array::adopt(ctx, h));
                return res;
            }
        }
        throw Panic("unreachable");
    }
};

}  // namespace physical_plan
}  // namespace rivulus
