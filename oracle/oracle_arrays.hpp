// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement (C++17) of the Arrow-style array layer of CleConor/rivulus
// (pure Rust; no Rust toolchain exists in the build image).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  Every type cites the reference lines it follows
// (paths relative to the reference checkout).  Rust `assert!`/`panic!` become
// rvo::Panic, `Err(String)` results become rvo::Err.
//
// Parity pinning: oracle/kat_tests.cpp re-expresses the reference's own inline
// known-answer tests against these classes.
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace rvo {

struct Panic : std::runtime_error {  // Rust panic!/assert!
    using std::runtime_error::runtime_error;
};
struct Err : std::runtime_error {  // Rust Err(String)
    using std::runtime_error::runtime_error;
};

inline void rv_assert(bool cond, const std::string &msg) {
    if (!cond) throw Panic(msg);
}

// ---------------------------------------------------------------------------
// BitMap -- src/execution/array/bitmap.rs:3-113
// LSB-first packed bits; a view is (shared buffer, bit_count, bit offset).
// ---------------------------------------------------------------------------
class BitMap {
  public:
    using Buffer = std::shared_ptr<const std::vector<uint8_t>>;

    BitMap() : buffer_(std::make_shared<std::vector<uint8_t>>()), bit_count_(0), offset_(0) {}
    BitMap(Buffer buf, size_t bit_count, size_t offset)
        : buffer_(std::move(buf)), bit_count_(bit_count), offset_(offset) {}

    // bitmap.rs:11-19
    static BitMap zeros(size_t bit_count) {
        return BitMap(std::make_shared<std::vector<uint8_t>>((bit_count + 7) / 8, 0), bit_count, 0);
    }
    // bitmap.rs:21-38 (tail byte masked)
    static BitMap all_true(size_t bit_count) {
        auto buf = std::make_shared<std::vector<uint8_t>>((bit_count + 7) / 8, 0xFF);
        if (bit_count % 8 != 0 && !buf->empty())
            buf->back() = static_cast<uint8_t>((1u << (bit_count % 8)) - 1);
        return BitMap(buf, bit_count, 0);
    }
    static BitMap all_false(size_t bit_count) { return zeros(bit_count); }  // bitmap.rs:40-42
    // bitmap.rs:44-59
    static BitMap from_bools(const std::vector<bool> &values) {
        auto buf = std::make_shared<std::vector<uint8_t>>((values.size() + 7) / 8, 0);
        for (size_t i = 0; i < values.size(); ++i)
            (*buf)[i / 8] |= static_cast<uint8_t>(values[i] ? 1u : 0u) << (i % 8);
        return BitMap(buf, values.size(), 0);
    }
    // raw bytes + bit offset (used by the C API to adopt caller buffers)
    static BitMap from_bytes(const uint8_t *bytes, size_t bit_count, size_t offset) {
        size_t nbytes = (bit_count + offset + 7) / 8;
        auto buf = std::make_shared<std::vector<uint8_t>>(bytes, bytes + nbytes);
        return BitMap(buf, bit_count, offset);
    }

    // bitmap.rs:61-68
    bool get_bit(size_t index) const {
        rv_assert(index < bit_count_, "assertion failed: index < self.bit_count");
        size_t p = index + offset_;
        return (((*buffer_)[p / 8] >> (p % 8)) & 1u) != 0;
    }
    size_t bit_count() const { return bit_count_; }
    size_t offset() const { return offset_; }
    const Buffer &buffer() const { return buffer_; }

    // bitmap.rs:74-86: absolute positions, one bit at a time
    size_t count(uint8_t value, size_t offset, size_t length) const {
        size_t c = 0;
        for (size_t i = offset; i < offset + length; ++i)
            if ((((*buffer_).at(i / 8) >> (i % 8)) & 1u) == value) ++c;
        return c;
    }
    size_t count_ones() const { return count(1, offset_, bit_count_); }   // bitmap.rs:88-90
    size_t count_zeros() const { return count(0, offset_, bit_count_); }  // bitmap.rs:92-94
    // bitmap.rs:96-102: NOTE absolute positions, self.offset is ignored
    size_t count_ones_range(size_t offset, size_t length) const { return count(1, offset, length); }
    size_t count_zeros_range(size_t offset, size_t length) const { return count(0, offset, length); }

    // bitmap.rs:104-112
    BitMap slice(size_t offset, size_t length) const {
        rv_assert(offset + length <= bit_count_, "assertion failed: offset + length <= self.bit_count");
        return BitMap(buffer_, length, offset + offset_);
    }

    // logical bits re-based to offset 0, tail bits zero (export helper, not in the reference)
    std::vector<uint8_t> to_packed() const {
        std::vector<uint8_t> out((bit_count_ + 7) / 8, 0);
        for (size_t i = 0; i < bit_count_; ++i)
            if (get_bit(i)) out[i / 8] |= static_cast<uint8_t>(1u << (i % 8));
        return out;
    }

  private:
    Buffer buffer_;
    size_t bit_count_;
    size_t offset_;
};

// BitmapBuilder -- bitmap.rs:115-189
class BitmapBuilder {
  public:
    void append(bool value) {  // bitmap.rs:142-155
        if (value) current_byte_ |= static_cast<uint8_t>(1u << current_bit_pos_);
        ++current_bit_pos_;
        ++bit_count_;
        if (current_bit_pos_ == 8) {
            buffer_.push_back(current_byte_);
            current_byte_ = 0;
            current_bit_pos_ = 0;
        }
    }
    bool has_nulls() const {  // bitmap.rs:157-176
        if (bit_count_ == 0) return false;
        for (uint8_t b : buffer_)
            if (b != 0xFF) return true;
        if (current_bit_pos_ > 0) {
            uint8_t expected = static_cast<uint8_t>((1u << current_bit_pos_) - 1);
            if (current_byte_ != expected) return true;
        }
        return false;
    }
    BitMap finish() {  // bitmap.rs:178-188: partial last byte flushed, high bits zero
        if (current_bit_pos_ > 0) buffer_.push_back(current_byte_);
        auto buf = std::make_shared<std::vector<uint8_t>>(std::move(buffer_));
        return BitMap(buf, bit_count_, 0);
    }

  private:
    std::vector<uint8_t> buffer_;
    size_t bit_count_ = 0;
    uint8_t current_byte_ = 0;
    size_t current_bit_pos_ = 0;
};

// ---------------------------------------------------------------------------
// schema.rs:1-76
// ---------------------------------------------------------------------------
enum class DataType { Null, Boolean, Int64, Float64, String };

inline const char *dtype_name(DataType t) {
    switch (t) {
        case DataType::Null: return "Null";
        case DataType::Boolean: return "Boolean";
        case DataType::Int64: return "Int64";
        case DataType::Float64: return "Float64";
        case DataType::String: return "String";
    }
    return "?";
}

struct Field {
    std::string name;
    DataType data_type;
    bool nullable;
    bool operator==(const Field &o) const {
        return name == o.name && data_type == o.data_type && nullable == o.nullable;
    }
};

struct Schema {
    std::vector<Field> fields;
    Schema() = default;
    explicit Schema(std::vector<Field> f) : fields(std::move(f)) {}
    size_t num_fields() const { return fields.size(); }
    bool is_empty() const { return fields.empty(); }
    const Field &field(size_t i) const { return fields.at(i); }
    const Field *field_by_name(const std::string &n) const {  // schema.rs:57-59
        for (auto &f : fields)
            if (f.name == n) return &f;
        return nullptr;
    }
    std::optional<size_t> index_of(const std::string &n) const {  // schema.rs:61-63
        for (size_t i = 0; i < fields.size(); ++i)
            if (fields[i].name == n) return i;
        return std::nullopt;
    }
    bool operator==(const Schema &o) const { return fields == o.fields; }
    bool operator!=(const Schema &o) const { return !(*this == o); }
};
using SchemaRef = std::shared_ptr<const Schema>;

// ---------------------------------------------------------------------------
// trait Array -- array/mod.rs:10-16
// ---------------------------------------------------------------------------
struct Array;
using ArrayRef = std::shared_ptr<const Array>;
struct Array {
    virtual ~Array() = default;
    virtual size_t len() const = 0;
    virtual DataType data_type() const = 0;
    virtual size_t null_count() const = 0;
    virtual ArrayRef slice(size_t offset, size_t length) const = 0;
};

// ---------------------------------------------------------------------------
// PrimitiveArray<i64|f64> -- primitive.rs:20-122
// ---------------------------------------------------------------------------
template <class T>
struct PrimitiveTraits;
template <>
struct PrimitiveTraits<int64_t> {
    static constexpr DataType kType = DataType::Int64;
};
template <>
struct PrimitiveTraits<double> {
    static constexpr DataType kType = DataType::Float64;
};

template <class T>
class PrimitiveArray : public Array {
  public:
    using Values = std::shared_ptr<const std::vector<T>>;

    // primitive.rs:31-42
    PrimitiveArray(std::vector<T> values, std::optional<std::vector<bool>> validity) {
        length_ = values.size();
        if (validity) null_bitmap_ = std::make_shared<BitMap>(BitMap::from_bools(*validity));
        values_ = std::make_shared<std::vector<T>>(std::move(values));
        offset_ = 0;
    }
    static std::shared_ptr<PrimitiveArray> from_values(std::vector<T> values) {  // primitive.rs:44-46
        return std::make_shared<PrimitiveArray>(std::move(values), std::nullopt);
    }
    // internal constructor (slice / builder / C API adoption)
    PrimitiveArray(Values values, std::shared_ptr<const BitMap> bitmap, size_t offset, size_t length)
        : values_(std::move(values)), null_bitmap_(std::move(bitmap)), offset_(offset), length_(length) {}

    // primitive.rs:48-60: validity is tested at offset+index on the UNSLICED bitmap
    std::optional<T> value(size_t index) const {
        rv_assert(index < length_, "Index " + std::to_string(index) + " out of bounds");
        size_t logical = offset_ + index;
        if (null_bitmap_ && !null_bitmap_->get_bit(logical)) return std::nullopt;
        return (*values_)[logical];
    }
    const T *values() const { return values_->data() + offset_; }  // primitive.rs:62-64
    const BitMap *null_bitmap() const { return null_bitmap_.get(); }
    size_t offset() const { return offset_; }
    size_t total_bytes() const { return values_->size() * sizeof(T); }  // primitive.rs:77-79

    size_t len() const override { return length_; }
    DataType data_type() const override { return PrimitiveTraits<T>::kType; }
    size_t null_count() const override {  // primitive.rs:90-105
        return null_bitmap_ ? null_bitmap_->count_zeros_range(offset_, length_) : 0;
    }
    ArrayRef slice(size_t offset, size_t length) const override {  // primitive.rs:107-117
        rv_assert(offset + length <= length_, "assertion failed: offset + length <= self.length");
        return std::make_shared<PrimitiveArray>(values_, null_bitmap_, offset_ + offset, length);
    }

  private:
    Values values_;
    std::shared_ptr<const BitMap> null_bitmap_;
    size_t offset_ = 0;
    size_t length_ = 0;
};

// PrimitiveArrayBuilder -- primitive.rs:150-198
template <class T>
class PrimitiveArrayBuilder {
  public:
    void reserve(size_t n) { values_.reserve(n); }
    void append_value(T v) {  // primitive.rs:169-172
        nulls_.append(true);
        values_.push_back(v);
    }
    void append_null(T placeholder) {  // primitive.rs:174-177
        nulls_.append(false);
        values_.push_back(placeholder);
    }
    std::shared_ptr<PrimitiveArray<T>> finish() {  // primitive.rs:179-196: bitmap dropped when no null
        std::shared_ptr<const BitMap> bm;
        if (nulls_.has_nulls()) bm = std::make_shared<BitMap>(nulls_.finish());
        size_t n = values_.size();
        auto vals = std::make_shared<std::vector<T>>(std::move(values_));
        return std::make_shared<PrimitiveArray<T>>(vals, bm, 0, n);
    }

  private:
    std::vector<T> values_;
    BitmapBuilder nulls_;
};

// ---------------------------------------------------------------------------
// BooleanArray -- boolean.rs:9-180, builder :250-298
// ---------------------------------------------------------------------------
class BooleanArray;
class BooleanArrayBuilder {
  public:
    void append_value(bool v) {  // boolean.rs:270-273
        values_.append(v);
        nulls_.append(true);
    }
    void append_null() {  // boolean.rs:275-278: value bit under a null is false
        values_.append(false);
        nulls_.append(false);
    }
    std::shared_ptr<BooleanArray> finish();  // boolean.rs:280-297

  private:
    BitmapBuilder values_;
    BitmapBuilder nulls_;
};

class BooleanArray : public Array {
  public:
    BooleanArray(std::shared_ptr<const BitMap> values, std::shared_ptr<const BitMap> nulls, size_t offset,
                 size_t length)
        : values_(std::move(values)), null_bitmap_(std::move(nulls)), offset_(offset), length_(length) {}

    // boolean.rs:19-50
    static std::shared_ptr<BooleanArray> make(const std::vector<std::optional<bool>> &booleans) {
        BooleanArrayBuilder b;
        for (auto &o : booleans) {
            if (o) b.append_value(*o);
            else b.append_null();
        }
        auto arr = b.finish();
        // `new` takes length from the input, `finish` from the value bitmap: identical here
        return arr;
    }
    static std::shared_ptr<BooleanArray> from_bools(const std::vector<bool> &v) {  // boolean.rs:52-55
        std::vector<std::optional<bool>> o(v.begin(), v.end());
        return make(o);
    }
    static std::shared_ptr<BooleanArray> new_null(size_t length) {  // boolean.rs:57-68
        return std::make_shared<BooleanArray>(std::make_shared<BitMap>(BitMap::zeros(length)),
                                              std::make_shared<BitMap>(BitMap::all_false(length)), 0, length);
    }
    static std::shared_ptr<BooleanArray> all_true(size_t length) {  // boolean.rs:70-79
        return std::make_shared<BooleanArray>(std::make_shared<BitMap>(BitMap::all_true(length)), nullptr, 0,
                                              length);
    }
    static std::shared_ptr<BooleanArray> all_false(size_t length) {  // boolean.rs:81-90
        return std::make_shared<BooleanArray>(std::make_shared<BitMap>(BitMap::zeros(length)), nullptr, 0,
                                              length);
    }

    // boolean.rs:92-104
    std::optional<bool> value(size_t index) const {
        rv_assert(index < length_, "Index " + std::to_string(index) + " out of bounds");
        size_t logical = offset_ + index;
        if (null_bitmap_ && !null_bitmap_->get_bit(logical)) return std::nullopt;
        return values_->get_bit(logical);
    }
    size_t total_bits() const { return values_->bit_count(); }
    size_t total_bytes() const { return (total_bits() + 7) / 8; }
    const BitMap *values_bitmap() const { return values_.get(); }
    const BitMap *null_bitmap() const { return null_bitmap_.get(); }
    size_t offset() const { return offset_; }

    // boolean.rs:120-135: strict null propagation (false AND null == null)
    std::shared_ptr<BooleanArray> logical_and(const BooleanArray &other) const {
        if (len() != other.len()) throw Err("Array lengths must match for logical operations");
        BooleanArrayBuilder b;
        for (size_t i = 0; i < len(); ++i) {
            auto x = value(i), y = other.value(i);
            if (x && y) b.append_value(*x && *y);
            else b.append_null();
        }
        return b.finish();
    }
    // boolean.rs:137-152 (true OR null == null)
    std::shared_ptr<BooleanArray> logical_or(const BooleanArray &other) const {
        if (len() != other.len()) throw Err("Array lengths must match for logical operations");
        BooleanArrayBuilder b;
        for (size_t i = 0; i < len(); ++i) {
            auto x = value(i), y = other.value(i);
            if (x && y) b.append_value(*x || *y);
            else b.append_null();
        }
        return b.finish();
    }
    // boolean.rs:154-165
    std::shared_ptr<BooleanArray> logical_not() const {
        BooleanArrayBuilder b;
        for (size_t i = 0; i < len(); ++i) {
            auto x = value(i);
            if (x) b.append_value(!*x);
            else b.append_null();
        }
        return b.finish();
    }
    size_t count_true() const {  // boolean.rs:167-172
        size_t c = 0;
        for (size_t i = 0; i < len(); ++i) {
            auto x = value(i);
            if (x && *x) ++c;
        }
        return c;
    }
    size_t count_false() const {  // boolean.rs:174-179
        size_t c = 0;
        for (size_t i = 0; i < len(); ++i) {
            auto x = value(i);
            if (x && !*x) ++c;
        }
        return c;
    }

    size_t len() const override { return length_; }
    DataType data_type() const override { return DataType::Boolean; }
    size_t null_count() const override {  // boolean.rs:191-205
        return null_bitmap_ ? null_bitmap_->count_zeros_range(offset_, length_) : 0;
    }
    ArrayRef slice(size_t offset, size_t length) const override {  // boolean.rs:207-218
        rv_assert(offset + length <= length_, "Slice out of bounds");
        return std::make_shared<BooleanArray>(values_, null_bitmap_, offset_ + offset, length);
    }

  private:
    std::shared_ptr<const BitMap> values_;
    std::shared_ptr<const BitMap> null_bitmap_;
    size_t offset_;
    size_t length_;
};

inline std::shared_ptr<BooleanArray> BooleanArrayBuilder::finish() {
    auto values = std::make_shared<BitMap>(values_.finish());
    std::shared_ptr<const BitMap> nulls;
    if (nulls_.has_nulls()) nulls = std::make_shared<BitMap>(nulls_.finish());
    size_t n = values->bit_count();
    return std::make_shared<BooleanArray>(values, nulls, 0, n);
}

// ---------------------------------------------------------------------------
// StringArray -- string.rs:8-190 (i32 offsets + UTF-8 bytes).  Host-only in the
// product (out of GPU scope); restated so the
// reference's RecordBatch fixtures (record_batch.rs:594-604) can be expressed.
// ---------------------------------------------------------------------------
class StringArray : public Array {
  public:
    explicit StringArray(const std::vector<std::optional<std::string>> &strings) {  // string.rs:19-57
        auto offsets = std::make_shared<std::vector<int32_t>>();
        auto data = std::make_shared<std::vector<uint8_t>>();
        BitmapBuilder nulls;
        offsets->push_back(0);
        for (auto &s : strings) {
            if (s) {
                nulls.append(true);
                data->insert(data->end(), s->begin(), s->end());
            } else {
                nulls.append(false);
            }
            offsets->push_back(static_cast<int32_t>(data->size()));
        }
        if (nulls.has_nulls()) null_bitmap_ = std::make_shared<BitMap>(nulls.finish());
        offsets_ = offsets;
        data_ = data;
        offset_ = 0;
        length_ = strings.size();
    }
    StringArray(std::shared_ptr<const std::vector<uint8_t>> data, std::shared_ptr<const std::vector<int32_t>> offs,
                std::shared_ptr<const BitMap> nulls, size_t offset, size_t length)
        : data_(std::move(data)), offsets_(std::move(offs)), null_bitmap_(std::move(nulls)), offset_(offset),
          length_(length) {}

    std::optional<std::string> value(size_t index) const {  // string.rs:81-99
        rv_assert(index < length_, "Index " + std::to_string(index) + " out of bounds");
        size_t logical = offset_ + index;
        if (null_bitmap_ && !null_bitmap_->get_bit(logical)) return std::nullopt;
        size_t s = static_cast<size_t>((*offsets_)[logical]), e = static_cast<size_t>((*offsets_)[logical + 1]);
        return std::string(data_->begin() + s, data_->begin() + e);
    }
    size_t len() const override { return length_; }
    DataType data_type() const override { return DataType::String; }
    size_t null_count() const override {
        return null_bitmap_ ? null_bitmap_->count_zeros_range(offset_, length_) : 0;
    }
    ArrayRef slice(size_t offset, size_t length) const override {  // string.rs:174-186
        rv_assert(offset + length <= length_, "Slice out of bounds");
        return std::make_shared<StringArray>(data_, offsets_, null_bitmap_, offset_ + offset, length);
    }
    // raw parts (C API export only)
    const std::vector<uint8_t> &data() const { return *data_; }
    const std::vector<int32_t> &offsets() const { return *offsets_; }
    const BitMap *null_bitmap() const { return null_bitmap_.get(); }
    size_t offset() const { return offset_; }

  private:
    std::shared_ptr<const std::vector<uint8_t>> data_;
    std::shared_ptr<const std::vector<int32_t>> offsets_;
    std::shared_ptr<const BitMap> null_bitmap_;
    size_t offset_;
    size_t length_;
};

// NullArray -- null.rs:5-66
class NullArray : public Array {
  public:
    explicit NullArray(size_t length, size_t offset = 0) : length_(length), offset_(offset) {}
    size_t len() const override { return length_; }
    DataType data_type() const override { return DataType::Null; }
    size_t null_count() const override { return length_; }
    ArrayRef slice(size_t offset, size_t length) const override {
        rv_assert(offset + length <= length_, "Slice out of bounds");
        return std::make_shared<NullArray>(length, offset_ + offset);
    }

  private:
    size_t length_;
    size_t offset_;
};

using Int64Array = PrimitiveArray<int64_t>;
using Float64Array = PrimitiveArray<double>;

}  // namespace rvo
