// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the reference's CSV scan, following
//   CsvFileStream::new            src/execution/file_stream.rs:20-41
//   parse_line                    :43-122   split on the delimiter, trim, "" / "null" = null, typed parse per schema field
//   read_batch                    :124-197  first line = header, blank lines skipped (not counted), batch_size rows per batch
//   build_record_batch            :199-327  typed arrays per column
//   calculate_adaptive_batch_size :345-368
// INCLUDING its defect: the Int64 / Float64 branches hand their `nulls` flags (true = null) to PrimitiveArray::new, whose
// second argument is a VALIDITY vector (true = valid, primitive.rs:31-33) -- a column with at least one null comes out
// with every validity bit inverted (:213-249).  The host layer's CsvNulls::AsReference must reproduce these arrays bit for
// bit; CsvNulls::AsIntended differs exactly by that inversion.
// The number parsers restate Rust's str::parse::<i64> / ::<f64> grammars (core::num, core::num::dec2flt) for the inputs
// a CSV field can hold.
#pragma once

#include <cmath>
#include <cstdlib>
#include <fstream>

#include "oracle_batch.hpp"

namespace rvo {

inline size_t csv_adaptive_batch_size(const Schema &schema) {  // file_stream.rs:345-368
    size_t row = 0;
    for (auto &f : schema.fields) {
        switch (f.data_type) {
            case DataType::Int64:
            case DataType::Float64: row += 8; break;
            case DataType::Boolean: row += 1; break;
            case DataType::String: row += 32; break;
            default: break;
        }
    }
    if (row == 0) return 10000;
    const size_t target = 8u * 1024 * 1024 / row;
    return target < 1000 ? 1000 : (target > 100000 ? 100000 : target);
}

// str::parse::<i64>: optional '+' / '-', then one or more ASCII digits, no overflow
inline bool rust_parse_i64(const std::string &s, int64_t &out) {
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) neg = s[i++] == '-';
    if (i == s.size()) return false;
    unsigned __int128 acc = 0;
    const unsigned __int128 limit = neg ? (static_cast<unsigned __int128>(1) << 63) : ((static_cast<unsigned __int128>(1) << 63) - 1);
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        acc = acc * 10 + static_cast<unsigned>(s[i] - '0');
        if (acc > limit) return false;
    }
    out = neg ? static_cast<int64_t>(0 - static_cast<uint64_t>(acc)) : static_cast<int64_t>(acc);
    return true;
}

// str::parse::<f64> (dec2flt): [+-] ( "inf" | "infinity" | "nan" (any case) | digits [. digits] [e [+-] digits] | . digits ... )
inline bool rust_parse_f64(const std::string &s, double &out) {
    size_t i = 0;
    bool neg = false;
    if (i < s.size() && (s[i] == '+' || s[i] == '-')) neg = s[i++] == '-';
    std::string rest = s.substr(i), lower;
    for (char c : rest) lower.push_back(static_cast<char>(c >= 'A' && c <= 'Z' ? c + 32 : c));
    if (lower == "inf" || lower == "infinity") {
        out = neg ? -INFINITY : INFINITY;
        return true;
    }
    if (lower == "nan") {
        out = NAN;
        return true;
    }
    size_t digits = 0, j = 0;
    while (j < rest.size() && rest[j] >= '0' && rest[j] <= '9') ++j, ++digits;
    if (j < rest.size() && rest[j] == '.') {
        ++j;
        while (j < rest.size() && rest[j] >= '0' && rest[j] <= '9') ++j, ++digits;
    }
    if (digits == 0) return false;
    if (j < rest.size() && (rest[j] == 'e' || rest[j] == 'E')) {
        ++j;
        if (j < rest.size() && (rest[j] == '+' || rest[j] == '-')) ++j;
        size_t ed = 0;
        while (j < rest.size() && rest[j] >= '0' && rest[j] <= '9') ++j, ++ed;
        if (ed == 0) return false;
    }
    if (j != rest.size()) return false;
    out = std::strtod(s.c_str(), nullptr);  // correctly rounded, like dec2flt
    return true;
}

class CsvFileStream : public DataStream {
  public:
    CsvFileStream(const std::string &path, SchemaRef schema, std::optional<size_t> batch_size = std::nullopt, std::optional<char> delimiter = std::nullopt)
        : file_(path), schema_(std::move(schema)), delimiter_(delimiter.value_or(',')) {
        if (!file_) throw Err("Failed to open file: " + std::string(std::strerror(errno)));
        batch_size_ = batch_size ? *batch_size : csv_adaptive_batch_size(*schema_);
    }
    SchemaRef schema() const override { return schema_; }
    size_t batch_size() const { return batch_size_; }

    std::optional<RecordBatch> next_batch() override {  // read_batch
        if (finished_) return std::nullopt;
        const size_t ncols = schema_->num_fields();
        std::vector<std::vector<Parsed>> data(ncols);
        std::string line;
        if (current_line_ == 0) {
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
            std::vector<Parsed> values;
            try {
                values = parse_line(line);
            } catch (const Err &e) {
                throw StreamError::execution(std::string("Parse error: ") + e.what());
            }
            for (size_t c = 0; c < ncols; ++c) data[c].push_back(std::move(values[c]));
            ++rows;
        }
        if (rows == 0) return std::nullopt;
        std::vector<ArrayRef> columns;  // build_record_batch
        for (size_t c = 0; c < ncols; ++c) {
            switch (schema_->fields[c].data_type) {
                case DataType::Int64: {
                    std::vector<int64_t> vals;
                    std::vector<bool> nulls;
                    for (auto &p : data[c]) vals.push_back(p.null ? 0 : p.i), nulls.push_back(p.null);
                    const bool any = std::find(nulls.begin(), nulls.end(), true) != nulls.end();
                    // the reference passes `nulls` where PrimitiveArray::new takes validity (:236-243)
                    columns.push_back(std::make_shared<Int64Array>(std::move(vals), any ? std::optional<std::vector<bool>>(nulls) : std::nullopt));
                    break;
                }
                case DataType::Float64: {
                    std::vector<double> vals;
                    std::vector<bool> nulls;
                    for (auto &p : data[c]) vals.push_back(p.null ? 0.0 : p.f), nulls.push_back(p.null);
                    const bool any = std::find(nulls.begin(), nulls.end(), true) != nulls.end();
                    columns.push_back(std::make_shared<Float64Array>(std::move(vals), any ? std::optional<std::vector<bool>>(nulls) : std::nullopt));
                    break;
                }
                case DataType::String: {
                    std::vector<std::optional<std::string>> vals;
                    for (auto &p : data[c]) vals.push_back(p.null ? std::nullopt : std::optional<std::string>(p.s));
                    columns.push_back(std::make_shared<StringArray>(vals));
                    break;
                }
                case DataType::Boolean: {
                    std::vector<std::optional<bool>> vals;
                    for (auto &p : data[c]) vals.push_back(p.null ? std::nullopt : std::optional<bool>(p.b));
                    columns.push_back(BooleanArray::make(vals));
                    break;
                }
                default: columns.push_back(std::make_shared<NullArray>(rows)); break;
            }
        }
        try {
            return RecordBatch::try_new(schema_, std::move(columns));
        } catch (const Err &e) {
            throw StreamError::execution(std::string("Failed to create RecordBatch: ") + e.what());
        }
    }

    // which cells of the last parsed line were null is what CsvNulls::AsIntended keeps: exposed for the differential test
    struct Parsed {
        bool null = false;
        int64_t i = 0;
        double f = 0.0;
        bool b = false;
        std::string s;
    };

  private:
    static std::string trim(const std::string &s) {  // str::trim: Unicode White_Space; the ASCII subset is what CSV holds
        size_t b = 0, e = s.size();
        auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; };
        while (b < e && ws(s[b])) ++b;
        while (e > b && ws(s[e - 1])) --e;
        return s.substr(b, e - b);
    }
    std::vector<Parsed> parse_line(const std::string &line) const {
        std::vector<std::string> fields;
        size_t start = 0;
        for (;;) {
            const size_t p = line.find(delimiter_, start);
            fields.push_back(trim(line.substr(start, p == std::string::npos ? std::string::npos : p - start)));
            if (p == std::string::npos) break;
            start = p + 1;
        }
        if (fields.size() != schema_->num_fields())
            throw Err("Line " + std::to_string(current_line_) + ": Expected " + std::to_string(schema_->num_fields()) + " fields, found " +
                      std::to_string(fields.size()));
        std::vector<Parsed> out(fields.size());
        for (size_t c = 0; c < fields.size(); ++c) {
            const std::string &f = fields[c];
            Parsed &v = out[c];
            auto bad = [&](const char *type) {
                throw Err("Line " + std::to_string(current_line_) + ", field " + std::to_string(c) + ": Cannot parse '" + f + "' as " + type);
            };
            const DataType dt = schema_->fields[c].data_type;
            v.null = dt == DataType::Null || f.empty() || f == "null";
            if (v.null) continue;
            switch (dt) {
                case DataType::Int64:
                    if (!rust_parse_i64(f, v.i)) bad("Int64");
                    break;
                case DataType::Float64:
                    if (!rust_parse_f64(f, v.f)) bad("Float64");
                    break;
                case DataType::String: v.s = f; break;
                default: {
                    std::string l;
                    for (char ch : f) l.push_back(static_cast<char>(ch >= 'A' && ch <= 'Z' ? ch + 32 : ch));
                    if (l == "true" || l == "t" || l == "1") v.b = true;
                    else if (l == "false" || l == "f" || l == "0") v.b = false;
                    else bad("Boolean");
                }
            }
        }
        return out;
    }

    std::ifstream file_;
    SchemaRef schema_;
    size_t batch_size_ = 0;
    char delimiter_;
    size_t current_line_ = 0;
    bool finished_ = false;
};

}  // namespace rvo
