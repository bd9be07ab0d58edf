// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of RecordBatch (src/execution/record_batch.rs) and of the
// pull-based stream operators (src/execution/stream.rs, LimitStream from
// src/physical_plan/streaming.rs:246-288) of CleConor/rivulus.
#pragma once

#include "oracle_arrays.hpp"

namespace rvo {

// ---------------------------------------------------------------------------
// RecordBatch -- record_batch.rs:8-422
// ---------------------------------------------------------------------------
class RecordBatch {
  public:
    RecordBatch() : schema_(std::make_shared<Schema>()), num_rows_(0) {}
    RecordBatch(SchemaRef schema, std::vector<ArrayRef> columns, size_t num_rows)  // new_unchecked :60-66
        : schema_(std::move(schema)), columns_(std::move(columns)), num_rows_(num_rows) {}

    // record_batch.rs:16-58
    static RecordBatch try_new(SchemaRef schema, std::vector<ArrayRef> columns) {
        if (schema->num_fields() != columns.size())
            throw Err("Schema has " + std::to_string(schema->num_fields()) + " fields but " +
                      std::to_string(columns.size()) + " columns provided");
        size_t num_rows = columns.empty() ? 0 : columns[0]->len();
        for (size_t i = 0; i < columns.size(); ++i)
            if (columns[i]->len() != num_rows)
                throw Err("Column " + std::to_string(i) + " has length " + std::to_string(columns[i]->len()) +
                          " but expected " + std::to_string(num_rows));
        for (size_t i = 0; i < columns.size(); ++i)
            if (schema->field(i).data_type != columns[i]->data_type())
                throw Err("Column " + std::to_string(i) + " has type " + dtype_name(columns[i]->data_type()) +
                          " but schema expects " + dtype_name(schema->field(i).data_type));
        return RecordBatch(std::move(schema), std::move(columns), num_rows);
    }

    const SchemaRef &schema() const { return schema_; }
    size_t num_rows() const { return num_rows_; }
    size_t num_columns() const { return columns_.size(); }
    const ArrayRef &column(size_t i) const {
        rv_assert(i < columns_.size(), "index out of bounds");  // Vec index panic
        return columns_[i];
    }
    const ArrayRef *column_by_name(const std::string &name) const {  // :84-86
        auto idx = schema_->index_of(name);
        return idx ? &columns_[*idx] : nullptr;
    }
    const std::vector<ArrayRef> &columns() const { return columns_; }
    bool is_empty() const { return num_rows_ == 0; }

    // record_batch.rs:92-106
    RecordBatch slice(size_t offset, size_t length) const {
        rv_assert(offset + length <= num_rows_, "Slice out of bounds");
        std::vector<ArrayRef> cols;
        for (auto &c : columns_) cols.push_back(c->slice(offset, length));
        return RecordBatch(schema_, std::move(cols), length);
    }

    // record_batch.rs:108-129
    RecordBatch take(const std::vector<size_t> &indices) const {
        for (size_t idx : indices)
            if (idx >= num_rows_)
                throw Err("Index " + std::to_string(idx) + " out of bounds for " + std::to_string(num_rows_) +
                          " rows");
        std::vector<ArrayRef> cols;
        for (auto &c : columns_) cols.push_back(take_array(c, indices));
        return RecordBatch(schema_, std::move(cols), indices.size());
    }

    // record_batch.rs:131-178: gather through builders; null slots get placeholder 0 / 0.0,
    // Boolean and String are rebuilt from Option vectors.
    static ArrayRef take_array(const ArrayRef &array, const std::vector<size_t> &indices) {
        switch (array->data_type()) {
            case DataType::Int64: {
                auto src = std::dynamic_pointer_cast<const Int64Array>(array);
                PrimitiveArrayBuilder<int64_t> b;
                b.reserve(indices.size());
                for (size_t i : indices) {
                    auto v = src->value(i);
                    if (v) b.append_value(*v);
                    else b.append_null(0);
                }
                return b.finish();
            }
            case DataType::Float64: {
                auto src = std::dynamic_pointer_cast<const Float64Array>(array);
                PrimitiveArrayBuilder<double> b;
                b.reserve(indices.size());
                for (size_t i : indices) {
                    auto v = src->value(i);
                    if (v) b.append_value(*v);
                    else b.append_null(0.0);
                }
                return b.finish();
            }
            case DataType::String: {
                auto src = std::dynamic_pointer_cast<const StringArray>(array);
                std::vector<std::optional<std::string>> vals;
                for (size_t i : indices) vals.push_back(src->value(i));
                return std::make_shared<StringArray>(vals);
            }
            case DataType::Boolean: {
                auto src = std::dynamic_pointer_cast<const BooleanArray>(array);
                std::vector<std::optional<bool>> vals;
                for (size_t i : indices) vals.push_back(src->value(i));
                return BooleanArray::make(vals);
            }
            case DataType::Null: return std::make_shared<NullArray>(indices.size());
        }
        throw Panic("unreachable");
    }

    // record_batch.rs:180-206
    RecordBatch select_columns(const std::vector<size_t> &indices) const {
        for (size_t idx : indices)
            if (idx >= num_columns())
                throw Err("Column index " + std::to_string(idx) + " out of bounds for " +
                          std::to_string(num_columns()) + " columns");
        std::vector<Field> fields;
        std::vector<ArrayRef> cols;
        for (size_t idx : indices) {
            fields.push_back(schema_->field(idx));
            cols.push_back(columns_[idx]);
        }
        return RecordBatch(std::make_shared<Schema>(fields), std::move(cols), num_rows_);
    }

    // record_batch.rs:208-219
    RecordBatch select_columns_by_name(const std::vector<std::string> &names) const {
        std::vector<size_t> indices;
        for (auto &n : names) {
            auto idx = schema_->index_of(n);
            if (!idx) throw Err("Column '" + n + "' not found");
            indices.push_back(*idx);
        }
        return select_columns(indices);
    }

    // record_batch.rs:221-243: null predicate => not selected (:237)
    RecordBatch filter(const ArrayRef &predicate) const {
        if (predicate->len() != num_rows_)
            throw Err("Predicate length " + std::to_string(predicate->len()) + " doesn't match batch length " +
                      std::to_string(num_rows_));
        auto bools = std::dynamic_pointer_cast<const BooleanArray>(predicate);
        if (!bools) throw Err("Predicate must be a BooleanArray");
        std::vector<size_t> selected;
        for (size_t i = 0; i < bools->len(); ++i) {
            auto v = bools->value(i);
            if (v && *v) selected.push_back(i);
        }
        return take(selected);
    }

    // record_batch.rs:245-275
    static RecordBatch concat(const std::vector<RecordBatch> &batches) {
        if (batches.empty()) throw Err("Cannot concatenate empty batch list");
        const auto &first_schema = batches[0].schema_;
        for (size_t i = 1; i < batches.size(); ++i)
            if (*batches[i].schema_ != *first_schema) throw Err("All batches must have the same schema");
        size_t total = 0;
        for (auto &b : batches) total += b.num_rows_;
        std::vector<ArrayRef> cols;
        for (size_t c = 0; c < first_schema->num_fields(); ++c) {
            std::vector<ArrayRef> parts;
            for (auto &b : batches) parts.push_back(b.columns_[c]);
            cols.push_back(concat_arrays(parts));
        }
        return RecordBatch(first_schema, std::move(cols), total);
    }

    // record_batch.rs:277-342: every element re-appended through a builder
    static ArrayRef concat_arrays(const std::vector<ArrayRef> &arrays) {
        if (arrays.empty()) throw Err("Cannot concatenate empty array list");
        switch (arrays[0]->data_type()) {
            case DataType::Int64: {
                PrimitiveArrayBuilder<int64_t> b;
                for (auto &a : arrays) {
                    auto p = std::dynamic_pointer_cast<const Int64Array>(a);
                    for (size_t i = 0; i < p->len(); ++i) {
                        auto v = p->value(i);
                        if (v) b.append_value(*v);
                        else b.append_null(0);
                    }
                }
                return b.finish();
            }
            case DataType::Float64: {
                PrimitiveArrayBuilder<double> b;
                for (auto &a : arrays) {
                    auto p = std::dynamic_pointer_cast<const Float64Array>(a);
                    for (size_t i = 0; i < p->len(); ++i) {
                        auto v = p->value(i);
                        if (v) b.append_value(*v);
                        else b.append_null(0.0);
                    }
                }
                return b.finish();
            }
            case DataType::String: {
                std::vector<std::optional<std::string>> all;
                for (auto &a : arrays) {
                    auto p = std::dynamic_pointer_cast<const StringArray>(a);
                    for (size_t i = 0; i < p->len(); ++i) all.push_back(p->value(i));
                }
                return std::make_shared<StringArray>(all);
            }
            case DataType::Boolean: {
                std::vector<std::optional<bool>> all;
                for (auto &a : arrays) {
                    auto p = std::dynamic_pointer_cast<const BooleanArray>(a);
                    for (size_t i = 0; i < p->len(); ++i) all.push_back(p->value(i));
                }
                return BooleanArray::make(all);
            }
            case DataType::Null: {
                size_t total = 0;
                for (auto &a : arrays) total += a->len();
                return std::make_shared<NullArray>(total);
            }
        }
        throw Panic("unreachable");
    }

    // record_batch.rs:402-421
    static RecordBatch empty(SchemaRef schema) {
        std::vector<ArrayRef> cols;
        for (auto &f : schema->fields) {
            switch (f.data_type) {
                case DataType::Int64: cols.push_back(Int64Array::from_values({})); break;
                case DataType::Float64: cols.push_back(Float64Array::from_values({})); break;
                case DataType::String:
                    cols.push_back(std::make_shared<StringArray>(std::vector<std::optional<std::string>>{}));
                    break;
                case DataType::Boolean: cols.push_back(BooleanArray::from_bools({})); break;
                case DataType::Null: cols.push_back(std::make_shared<NullArray>(0)); break;
            }
        }
        return RecordBatch(std::move(schema), std::move(cols), 0);
    }

  private:
    SchemaRef schema_;
    std::vector<ArrayRef> columns_;
    size_t num_rows_;
};

// ---------------------------------------------------------------------------
// StreamError -- stream.rs:7-23
// ---------------------------------------------------------------------------
struct StreamError : std::runtime_error {
    enum Kind { Execution, SchemaMismatch, Exhausted, Io } kind;
    StreamError(Kind k, const std::string &display) : std::runtime_error(display), kind(k) {}
    static StreamError execution(const std::string &message) {
        return StreamError(Execution, "Stream execution error: " + message);
    }
    static StreamError schema_mismatch() { return StreamError(SchemaMismatch, "Schema mismatch"); }
};

// trait DataStream -- stream.rs:25-54
class DataStream {
  public:
    virtual ~DataStream() = default;
    virtual SchemaRef schema() const = 0;
    virtual std::optional<RecordBatch> next_batch() = 0;

    std::vector<RecordBatch> collect() {  // stream.rs:30-39
        std::vector<RecordBatch> out;
        while (auto b = next_batch()) out.push_back(std::move(*b));
        return out;
    }
    RecordBatch concatenate() {  // stream.rs:41-53
        auto s = schema();
        auto batches = collect();
        if (batches.empty()) return RecordBatch::empty(s);
        try {
            return RecordBatch::concat(batches);
        } catch (const Err &e) {
            throw StreamError::execution(e.what());
        }
    }
};
using DataStreamRef = std::unique_ptr<DataStream>;

// MemoryStream -- stream.rs:58-114
class MemoryStream : public DataStream {
  public:
    MemoryStream(SchemaRef schema, std::vector<RecordBatch> batches)  // :66-80
        : schema_(std::move(schema)), batches_(std::move(batches)) {
        for (auto &b : batches_)
            if (*b.schema() != *schema_) throw StreamError::schema_mismatch();
    }
    static std::unique_ptr<MemoryStream> from_single_batch(RecordBatch b) {
        auto s = b.schema();
        std::vector<RecordBatch> v;
        v.push_back(std::move(b));
        return std::make_unique<MemoryStream>(s, std::move(v));
    }
    static std::unique_ptr<MemoryStream> empty(SchemaRef s) {
        return std::make_unique<MemoryStream>(std::move(s), std::vector<RecordBatch>{});
    }
    SchemaRef schema() const override { return schema_; }
    std::optional<RecordBatch> next_batch() override {  // :105-113
        if (index_ < batches_.size()) return batches_[index_++];
        return std::nullopt;
    }

  private:
    SchemaRef schema_;
    std::vector<RecordBatch> batches_;
    size_t index_ = 0;
};

// FilterStream -- stream.rs:116-163 (predicate = pre-existing Boolean column)
class FilterStream : public DataStream {
  public:
    FilterStream(DataStreamRef input, std::string predicate_column)
        : input_(std::move(input)), predicate_column_(std::move(predicate_column)) {}
    SchemaRef schema() const override { return input_->schema(); }
    std::optional<RecordBatch> next_batch() override {
        auto batch = input_->next_batch();
        if (!batch) return std::nullopt;
        auto idx = batch->schema()->index_of(predicate_column_);
        if (!idx) throw StreamError::execution("Column '" + predicate_column_ + "' not found in schema");
        const auto &pred = batch->column(*idx);
        if (pred->data_type() != DataType::Boolean)
            throw StreamError::execution("Predicate column '" + predicate_column_ + "' is not of boolean type");
        try {
            return batch->filter(pred);  // empty batches are still emitted (:156-158)
        } catch (const Err &e) {
            throw StreamError::execution(e.what());
        }
    }

  private:
    DataStreamRef input_;
    std::string predicate_column_;
};

// SelectStream -- stream.rs:165-213
class SelectStream : public DataStream {
  public:
    SelectStream(DataStreamRef input, std::vector<std::string> column_names)
        : input_(std::move(input)), column_names_(std::move(column_names)) {
        auto in_schema = input_->schema();
        std::vector<Field> fields;
        for (auto &n : column_names_) {
            auto f = in_schema->field_by_name(n);
            if (!f) throw StreamError::execution("Column '" + n + "' not found in schema");
            fields.push_back(*f);
        }
        output_schema_ = std::make_shared<Schema>(fields);
    }
    SchemaRef schema() const override { return output_schema_; }
    std::optional<RecordBatch> next_batch() override {
        auto batch = input_->next_batch();
        if (!batch) return std::nullopt;
        try {
            return batch->select_columns_by_name(column_names_);
        } catch (const Err &e) {
            throw StreamError::execution(e.what());
        }
    }

  private:
    DataStreamRef input_;
    std::vector<std::string> column_names_;
    SchemaRef output_schema_;
};

// LimitStream -- physical_plan/streaming.rs:246-288
class LimitStream : public DataStream {
  public:
    LimitStream(DataStreamRef input, size_t limit) : input_(std::move(input)), limit_(limit) {}
    SchemaRef schema() const override { return input_->schema(); }
    std::optional<RecordBatch> next_batch() override {
        if (rows_returned_ >= limit_) return std::nullopt;
        auto batch = input_->next_batch();
        if (!batch) return std::nullopt;
        size_t remaining = limit_ - rows_returned_;
        if (batch->num_rows() <= remaining) {
            rows_returned_ += batch->num_rows();
            return batch;
        }
        rows_returned_ += remaining;
        return batch->slice(0, remaining);
    }

  private:
    DataStreamRef input_;
    size_t limit_;
    size_t rows_returned_ = 0;
};

}  // namespace rvo
