// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// extern "C" surface of the CPU oracle so pytest (ctypes) and bench.py's cpu_baseline
// leg can drive it with the same rv_column / rv_predicate structs the GPU library takes.
// Everything here is host memory; results are owned by an orc_result handle.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../include/rivulus_gpu.h"
#include "oracle_compose.hpp"

using namespace rvo;

namespace {
thread_local std::string g_err;

ArrayRef adopt(const rv_column &c) {
    size_t total = static_cast<size_t>(c.offset + c.length);
    std::shared_ptr<const BitMap> validity;
    if (c.validity) validity = std::make_shared<BitMap>(BitMap::from_bytes(c.validity, total, 0));
    switch (c.dtype) {
        case RV_INT64: {
            auto p = static_cast<const int64_t *>(c.values);
            auto v = std::make_shared<std::vector<int64_t>>(p, p + total);
            return std::make_shared<Int64Array>(v, validity, c.offset, c.length);
        }
        case RV_FLOAT64: {
            auto p = static_cast<const double *>(c.values);
            auto v = std::make_shared<std::vector<double>>(p, p + total);
            return std::make_shared<Float64Array>(v, validity, c.offset, c.length);
        }
        case RV_BOOLEAN: {
            auto vals = std::make_shared<BitMap>(BitMap::from_bytes(static_cast<const uint8_t *>(c.values), total, 0));
            return std::make_shared<BooleanArray>(vals, validity, c.offset, c.length);
        }
        case RV_STRING: {  // string.rs:9-15: shared bytes + int32 offsets + element offset
            auto p = static_cast<const uint8_t *>(c.values);
            auto data = std::make_shared<std::vector<uint8_t>>(p, p + c.data_bytes);
            auto offs = std::make_shared<std::vector<int32_t>>(c.offsets, c.offsets + total + 1);
            return std::make_shared<StringArray>(data, offs, validity, c.offset, c.length);
        }
        case RV_NULL: return std::make_shared<NullArray>(c.length);
        default: throw Err("oracle C API: unsupported dtype");
    }
}

std::vector<ArrayRef> adopt_all(const rv_column *cols, uint32_t n) {
    std::vector<ArrayRef> out;
    for (uint32_t i = 0; i < n; ++i) out.push_back(adopt(cols[i]));
    return out;
}

AnyValue literal_of(const rv_term &t) {
    switch (t.lit_type) {
        case RV_NULL: return AnyValue::null();
        case RV_INT64: return AnyValue(static_cast<int64_t>(t.lit.i));
        case RV_FLOAT64: return AnyValue(t.lit.f);
        case RV_BOOLEAN: return AnyValue(t.lit.i != 0);
        case RV_STRING: return AnyValue(std::string(t.lit.s.ptr ? t.lit.s.ptr : "", static_cast<size_t>(t.lit.s.len)));
        default: throw Err("oracle C API: unsupported literal type");
    }
}

std::vector<Term> terms_of(const rv_predicate *p) {
    std::vector<Term> out;
    for (uint32_t i = 0; i < p->n_terms; ++i) {
        const rv_term &t = p->terms[i];
        out.push_back(Term{t.column, static_cast<TermOp>(t.op), t.op == RV_IS_TRUE ? AnyValue(true) : literal_of(t)});
    }
    return out;
}

std::vector<uint8_t> expr_of(const rv_predicate *p) {
    return p->expr ? std::vector<uint8_t>(p->expr, p->expr + p->n_expr) : std::vector<uint8_t>{};
}

NullPolicy policy_of(const rv_predicate *p) {
    return p->nulls == RV_NULL_IS_LEAST ? NullPolicy::IsLeast : NullPolicy::Drops;
}

struct ExportedColumn {
    rv_dtype dtype;
    std::vector<uint8_t> values;  // raw bytes
    std::vector<uint8_t> validity;
    std::vector<int32_t> offsets;  // RV_STRING: length + 1 entries from 0; values = the logical bytes
    bool has_validity = false;
    uint64_t length = 0;
    uint64_t null_count = 0;
};

ExportedColumn export_array(const ArrayRef &a) {
    ExportedColumn e;
    e.length = a->len();
    e.null_count = a->null_count();
    auto pack_validity = [&](const BitMap *bm, size_t offset) {
        if (!bm) return;
        e.has_validity = true;
        e.validity = bm->slice(offset, a->len()).to_packed();
    };
    switch (a->data_type()) {
        case DataType::Int64: {
            auto p = std::static_pointer_cast<const Int64Array>(a);
            e.dtype = RV_INT64;
            e.values.resize(p->len() * 8);
            if (p->len()) std::memcpy(e.values.data(), p->values(), p->len() * 8);
            pack_validity(p->null_bitmap(), p->offset());
            break;
        }
        case DataType::Float64: {
            auto p = std::static_pointer_cast<const Float64Array>(a);
            e.dtype = RV_FLOAT64;
            e.values.resize(p->len() * 8);
            if (p->len()) std::memcpy(e.values.data(), p->values(), p->len() * 8);
            pack_validity(p->null_bitmap(), p->offset());
            break;
        }
        case DataType::Boolean: {
            auto p = std::static_pointer_cast<const BooleanArray>(a);
            e.dtype = RV_BOOLEAN;
            e.values = p->values_bitmap()->slice(p->offset(), p->len()).to_packed();
            pack_validity(p->null_bitmap(), p->offset());
            break;
        }
        case DataType::String: {
            auto p = std::static_pointer_cast<const StringArray>(a);
            e.dtype = RV_STRING;
            const auto &o = p->offsets();
            const int32_t first = o[p->offset()];
            for (size_t i = 0; i <= p->len(); ++i) e.offsets.push_back(o[p->offset() + i] - first);
            e.values.assign(p->data().begin() + first, p->data().begin() + o[p->offset() + p->len()]);
            pack_validity(p->null_bitmap(), p->offset());
            break;
        }
        default: e.dtype = RV_NULL; break;
    }
    return e;
}
}  // namespace

struct orc_result {
    std::vector<ExportedColumn> cols;
    uint64_t rows = 0;
};

#define ORC_TRY try {
#define ORC_CATCH                      \
    }                                  \
    catch (const std::exception &e) {  \
        g_err = e.what();              \
        return 1;                      \
    }                                  \
    return 0;

extern "C" {

const char *orc_last_error(void) { return g_err.c_str(); }

void orc_result_free(orc_result *r) { delete r; }
uint64_t orc_result_rows(const orc_result *r) { return r->rows; }
uint32_t orc_result_ncols(const orc_result *r) { return static_cast<uint32_t>(r->cols.size()); }
// view of result column j; pointers live until orc_result_free
int orc_result_column(const orc_result *r, uint32_t j, rv_column *out, int *has_validity, uint64_t *null_count) {
    if (j >= r->cols.size()) return 1;
    const ExportedColumn &e = r->cols[j];
    out->dtype = e.dtype;
    out->values = e.values.data();
    out->validity = e.has_validity ? e.validity.data() : nullptr;
    out->offset = 0;
    out->length = e.length;
    out->offsets = e.dtype == RV_STRING ? e.offsets.data() : nullptr;
    out->data_bytes = e.dtype == RV_STRING ? e.values.size() : 0;
    if (has_validity) *has_validity = e.has_validity ? 1 : 0;
    if (null_count) *null_count = e.null_count;
    return 0;
}

static orc_result *make_result(const RecordBatch &b) {
    auto r = new orc_result();
    r->rows = b.num_rows();
    for (auto &c : b.columns()) r->cols.push_back(export_array(c));
    return r;
}

// synthetic generator (the patterns of include/rivulus_gpu.h, rv_synth_spec); validity may be NULL
static inline uint64_t synth_hash(const rv_synth_spec *s, uint64_t g, uint64_t step, uint64_t table) {
    switch (s->pattern) {
        case RV_SYNTH_CLUSTERED: return splitmix64(s->seed + g / s->run_rows);
        case RV_SYNTH_SORTED_ASC: return g * step;
        case RV_SYNTH_SORTED_DESC: return (table - 1 - g) * step;
        default: return splitmix64(s->seed + g);
    }
}
int orc_generate(const rv_synth_spec *s, void *values, uint8_t *validity) {
    ORC_TRY
    if (s->pattern > RV_SYNTH_SORTED_DESC) throw Err("orc_generate: unknown pattern");
    if (s->pattern == RV_SYNTH_CLUSTERED && s->run_rows == 0) throw Err("orc_generate: run_rows is 0");
    const bool sorted = s->pattern == RV_SYNTH_SORTED_ASC || s->pattern == RV_SYNTH_SORTED_DESC;
    const uint64_t table = s->table_rows ? s->table_rows : s->first_row + s->length;
    if (sorted && s->first_row + s->length > table) throw Err("orc_generate: rows past table_rows");
    const uint64_t step = (sorted && table) ? ~0ull / table : 0;
    for (uint64_t i = 0; i < s->length; ++i) {
        uint64_t g = s->first_row + i;
        uint64_t h = synth_hash(s, g, step, table);
        // a sorted pattern's h is a fraction of 2^64 that grows with the row: scaled, not reduced, so the order survives
        if (s->dtype == RV_INT64)
            static_cast<int64_t *>(values)[i] = static_cast<int64_t>(sorted ? static_cast<uint64_t>((static_cast<unsigned __int128>(h) * s->modulus) >> 64) : h % s->modulus);
        else if (s->dtype == RV_FLOAT64) static_cast<double *>(values)[i] = static_cast<double>(h >> 11) * 0x1.0p-53;
        else if (s->dtype == RV_BOOLEAN) {
            if (i % 8 == 0) static_cast<uint8_t *>(values)[i / 8] = 0;
            const uint64_t pct = sorted ? static_cast<uint64_t>((static_cast<unsigned __int128>(h) * 100) >> 64) : h % 100;
            if (pct < s->true_percent) static_cast<uint8_t *>(values)[i / 8] |= static_cast<uint8_t>(1u << (i % 8));
        } else
            throw Err("orc_generate: unsupported dtype");
        if (s->with_validity && validity) {
            if (i % 8 == 0) validity[i / 8] = 0;
            if (splitmix64(s->validity_seed + g) % 100 >= s->null_percent)
                validity[i / 8] |= static_cast<uint8_t>(1u << (i % 8));
        }
    }
    ORC_CATCH
}

// Full-size checker for the synthetic tables of BASELINE configs[3] / [4] (1e10 rows do not fit host memory as a
// column): the generator's Int64 values streamed through `value > literal` (the eager compare, plan.rs:112-130, over
// series.rs:100-117 ordering), COUNT and the wrapping SUM of the survivors, plus a position-dependent checksum
// sum(ordinal_of_survivor * value) mod 2^64 that pins the ORDER of the compacted output (record_batch.rs:235-240:
// ascending).  Row ranges on `threads` host threads; every result is an exact integer, so the split does not matter.
int orc_synth_filter_checksums(uint64_t seed, uint64_t first_row, uint64_t n_rows, uint64_t modulus, int64_t literal,
                               uint32_t threads, int64_t *sum, uint64_t *count, uint64_t *ordered) {
    ORC_TRY
    if (modulus == 0) throw Err("orc_synth_filter_checksums: modulus is 0");
    if (threads == 0) threads = 1;
    struct Part {
        uint64_t sum = 0, count = 0, weighted = 0;  // weighted: sum(local ordinal * value)
    };
    std::vector<Part> parts(threads);
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            const uint64_t b = n_rows / threads * t + std::min<uint64_t>(t, n_rows % threads);
            const uint64_t e = b + n_rows / threads + (t < n_rows % threads ? 1 : 0);
            Part p;
            for (uint64_t i = b; i < e; ++i) {
                const int64_t v = static_cast<int64_t>(splitmix64(seed + first_row + i) % modulus);
                if (v > literal) {
                    p.sum += static_cast<uint64_t>(v);
                    p.weighted += p.count * static_cast<uint64_t>(v);
                    p.count += 1;
                }
            }
            parts[t] = p;
        });
    for (auto &th : pool) th.join();
    uint64_t s = 0, c = 0, w = 0;
    for (const Part &p : parts) {  // global ordinal = survivors before the range + local ordinal
        w += p.weighted + c * p.sum;
        s += p.sum;
        c += p.count;
    }
    if (sum) *sum = static_cast<int64_t>(s);
    if (count) *count = c;
    if (ordered) *ordered = w;
    ORC_CATCH
}

// selection bitmap (ceil(n/8) bytes, tail bits zero) + survivor count
int orc_eval_predicate(const rv_column *cols, uint32_t ncols, const rv_predicate *pred, uint8_t *out_bits,
                       uint64_t *out_count) {
    ORC_TRY
    auto arrays = adopt_all(cols, ncols);
    auto b = evaluate_predicate(arrays, terms_of(pred), policy_of(pred), expr_of(pred));
    size_t n = b->len();
    std::memset(out_bits, 0, (n + 7) / 8);
    uint64_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        auto v = b->value(i);
        if (v && *v) {
            out_bits[i / 8] |= static_cast<uint8_t>(1u << (i % 8));
            ++c;
        }
    }
    if (out_count) *out_count = c;
    ORC_CATCH
}

int orc_compare(const rv_column *col, rv_cmp op, rv_dtype lit_type, int64_t lit_i, double lit_f, orc_result **out) {
    ORC_TRY
    rv_term t{};
    t.column = 0;
    t.op = op;
    t.lit_type = lit_type;
    if (lit_type == RV_FLOAT64) t.lit.f = lit_f;
    else t.lit.i = lit_i;
    auto a = adopt(*col);
    auto b = compare_array(a, static_cast<TermOp>(op), op == RV_IS_TRUE ? AnyValue(true) : literal_of(t));
    auto r = new orc_result();
    r->rows = b->len();
    r->cols.push_back(export_array(b));
    *out = r;
    ORC_CATCH
}

int orc_compare_term(const rv_column *col, const rv_term *term, orc_result **out) {
    ORC_TRY
    auto a = adopt(*col);
    auto b = compare_array(a, static_cast<TermOp>(term->op), term->op == RV_IS_TRUE ? AnyValue(true) : literal_of(*term));
    auto r = new orc_result();
    r->rows = b->len();
    r->cols.push_back(export_array(b));
    *out = r;
    ORC_CATCH
}

// kind: 0 and, 1 or, 2 not (b ignored)
int orc_boolean_op(int kind, const rv_column *a, const rv_column *b, orc_result **out) {
    ORC_TRY
    auto x = std::dynamic_pointer_cast<const BooleanArray>(adopt(*a));
    if (!x) throw Err("orc_boolean_op: not a BooleanArray");
    std::shared_ptr<BooleanArray> res;
    if (kind == 2) res = x->logical_not();
    else {
        auto y = std::dynamic_pointer_cast<const BooleanArray>(adopt(*b));
        if (!y) throw Err("orc_boolean_op: not a BooleanArray");
        res = kind == 0 ? x->logical_and(*y) : x->logical_or(*y);
    }
    auto r = new orc_result();
    r->rows = res->len();
    r->cols.push_back(export_array(res));
    *out = r;
    ORC_CATCH
}

int orc_boolean_count(const rv_column *a, uint64_t *count_true, uint64_t *count_false) {
    ORC_TRY
    auto x = std::dynamic_pointer_cast<const BooleanArray>(adopt(*a));
    if (!x) throw Err("orc_boolean_count: not a BooleanArray");
    *count_true = x->count_true();
    *count_false = x->count_false();
    ORC_CATCH
}

int orc_null_count(const rv_column *a, uint64_t *out) {
    ORC_TRY
    *out = adopt(*a)->null_count();
    ORC_CATCH
}

int orc_filter(const rv_column *cols, uint32_t ncols, const rv_column *predicate, orc_result **out) {
    ORC_TRY
    auto arrays = adopt_all(cols, ncols);
    RecordBatch batch = RecordBatch::try_new(positional_schema(arrays), arrays);
    *out = make_result(batch.filter(adopt(*predicate)));
    ORC_CATCH
}

int orc_take(const rv_column *cols, uint32_t ncols, const uint64_t *indices, uint64_t n, orc_result **out) {
    ORC_TRY
    auto arrays = adopt_all(cols, ncols);
    RecordBatch batch = RecordBatch::try_new(positional_schema(arrays), arrays);
    std::vector<size_t> idx(indices, indices + n);
    *out = make_result(batch.take(idx));
    ORC_CATCH
}

int orc_slice(const rv_column *col, uint64_t offset, uint64_t length, orc_result **out) {
    ORC_TRY
    auto a = adopt(*col)->slice(offset, length);
    auto r = new orc_result();
    r->rows = a->len();
    r->cols.push_back(export_array(a));
    *out = r;
    ORC_CATCH
}

int orc_concat(const rv_column *parts, uint32_t nparts, orc_result **out) {
    ORC_TRY
    auto arrays = adopt_all(parts, nparts);
    auto a = RecordBatch::concat_arrays(arrays);
    auto r = new orc_result();
    r->rows = a->len();
    r->cols.push_back(export_array(a));
    *out = r;
    ORC_CATCH
}

int orc_filter_project(const rv_column *cols, uint32_t ncols, const rv_predicate *pred, const uint32_t *proj,
                       uint32_t nproj, orc_result **out) {
    ORC_TRY
    auto arrays = adopt_all(cols, ncols);
    std::vector<size_t> p(proj, proj + nproj);
    *out = make_result(filter_project(arrays, terms_of(pred), policy_of(pred), p, expr_of(pred)));
    ORC_CATCH
}

// the faithful streaming pipeline (1024-row batches in the reference, streaming_planner.rs:32)
int orc_stream_filter_project(const rv_column *cols, uint32_t ncols, uint64_t batch_rows, const rv_predicate *pred,
                              const uint32_t *proj, uint32_t nproj, orc_result **out) {
    ORC_TRY
    auto arrays = adopt_all(cols, ncols);
    std::vector<size_t> p(proj, proj + nproj);
    *out = make_result(stream_filter_project(arrays, batch_rows, terms_of(pred), policy_of(pred), p, expr_of(pred)));
    ORC_CATCH
}

int orc_filter_agg(const rv_column *cols, uint32_t ncols, const rv_predicate *pred, uint32_t agg_col, int64_t *sum_i,
                   double *sum_f, uint64_t *count) {
    ORC_TRY
    auto arrays = adopt_all(cols, ncols);
    auto r = filter_agg(arrays, terms_of(pred), policy_of(pred), agg_col, expr_of(pred));
    if (sum_i) *sum_i = r.sum_i;
    if (sum_f) *sum_f = r.sum_f;
    if (count) *count = r.count;
    ORC_CATCH
}

// ---------------------------------------------------------------------------
// cpu_baseline legs (bench.py).  Both build their input INSIDE the oracle from the
// synthetic generator, time only the query, and return seconds + surviving rows.
// ---------------------------------------------------------------------------

// (i) eager collect(): LazyFrame::from_dataframe(df).filter(col("x") > lit).select([col("x")]).collect()
//     over a DataFrame of AnyValue cells (builder.rs:96-104 -> plan.rs:97-150, :68-96).
int orc_bench_eager_collect(uint64_t n_rows, uint64_t seed, uint64_t modulus, int64_t literal, double *seconds,
                            uint64_t *out_rows, int64_t *checksum) {
    ORC_TRY
    std::vector<AnyValue> cells;
    cells.reserve(n_rows);
    for (uint64_t i = 0; i < n_rows; ++i) cells.emplace_back(static_cast<int64_t>(splitmix64(seed + i) % modulus));
    DataFrame df({Series("x", std::move(cells))});
    auto t0 = std::chrono::steady_clock::now();
    // from_dataframe clones the frame (builder.rs:30) -- part of the reference's collect() cost
    DataFrame res = LazyFrame::from_dataframe(df).filter(Expr::col("x").gt(Expr::lit(AnyValue(literal))))
                        .select({Expr::col("x")})
                        .collect();
    auto t1 = std::chrono::steady_clock::now();
    *seconds = std::chrono::duration<double>(t1 - t0).count();
    *out_rows = res.height();
    int64_t cs = 0;
    for (auto &v : res.columns()[0].data()) cs = static_cast<int64_t>(static_cast<uint64_t>(cs) + static_cast<uint64_t>(std::get<1>(v.v)));
    *checksum = cs;
    ORC_CATCH
}

// (ii) streaming restatement on typed arrays: batch_rows-row slices -> compare -> filter -> select -> concat
int orc_bench_stream(uint64_t n_rows, uint64_t seed, uint64_t modulus, int64_t literal, uint64_t batch_rows,
                     double *seconds, uint64_t *out_rows, int64_t *checksum) {
    ORC_TRY
    std::vector<int64_t> vals(n_rows);
    for (uint64_t i = 0; i < n_rows; ++i) vals[i] = static_cast<int64_t>(splitmix64(seed + i) % modulus);
    std::vector<ArrayRef> cols{Int64Array::from_values(std::move(vals))};
    std::vector<Term> terms{Term{0, TermOp::Gt, AnyValue(literal)}};
    auto t0 = std::chrono::steady_clock::now();
    RecordBatch res = stream_filter_project(cols, batch_rows, terms, NullPolicy::Drops, {0});
    auto t1 = std::chrono::steady_clock::now();
    *seconds = std::chrono::duration<double>(t1 - t0).count();
    *out_rows = res.num_rows();
    auto out = std::static_pointer_cast<const Int64Array>(res.column(0));
    int64_t cs = 0;
    for (size_t i = 0; i < out->len(); ++i) cs = static_cast<int64_t>(static_cast<uint64_t>(cs) + static_cast<uint64_t>(out->values()[i]));
    *checksum = cs;
    ORC_CATCH
}

// (iii) courtesy figure: the same query written the way a tuned CPU engine would
// -- typed column, branch-free compress loop, one row range per thread, outputs concatenated in range
// order.  Same semantics and row order as (i)/(ii); NOT the reference's algorithm, reported next to it.
int orc_bench_threads(uint64_t n_rows, uint64_t seed, uint64_t modulus, int64_t literal, uint32_t threads,
                      double *seconds, uint64_t *out_rows, int64_t *checksum) {
    ORC_TRY
    if (threads == 0) threads = 1;
    std::vector<int64_t> vals(n_rows);
    {
        std::vector<std::thread> gen;
        for (uint32_t t = 0; t < threads; ++t)
            gen.emplace_back([&, t] {
                const uint64_t b = n_rows * t / threads, e = n_rows * (t + 1) / threads;
                for (uint64_t i = b; i < e; ++i) vals[i] = static_cast<int64_t>(splitmix64(seed + i) % modulus);
            });
        for (auto &g : gen) g.join();
    }
    // scratch buffers are allocated and touched up front, as an engine with a buffer pool would have them
    std::vector<std::vector<int64_t>> part(threads);
    for (uint32_t t = 0; t < threads; ++t) part[t].assign(n_rows * (t + 1) / threads - n_rows * t / threads, 0);
    std::vector<int64_t> result(n_rows / 4 + 1024, 0);
    auto t0 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> work;
        for (uint32_t t = 0; t < threads; ++t)
            work.emplace_back([&, t] {
                const uint64_t b = n_rows * t / threads, e = n_rows * (t + 1) / threads;
                std::vector<int64_t> &out = part[t];
                size_t k = 0;
                for (uint64_t i = b; i < e; ++i) {
                    out[k] = vals[i];
                    k += vals[i] > literal;
                }
                out.resize(k);
            });
        for (auto &w : work) w.join();
    }
    std::vector<size_t> start(threads + 1, 0);
    for (uint32_t t = 0; t < threads; ++t) start[t + 1] = start[t] + part[t].size();
    if (start[threads] > result.size()) result.resize(start[threads]);
    {
        std::vector<std::thread> cat;
        for (uint32_t t = 0; t < threads; ++t)
            cat.emplace_back([&, t] {
                if (!part[t].empty()) std::memcpy(result.data() + start[t], part[t].data(), part[t].size() * 8);
            });
        for (auto &c : cat) c.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    *seconds = std::chrono::duration<double>(t1 - t0).count();
    result.resize(start[threads]);
    *out_rows = result.size();
    int64_t cs = 0;
    for (int64_t v : result) cs = static_cast<int64_t>(static_cast<uint64_t>(cs) + static_cast<uint64_t>(v));
    *checksum = cs;
    ORC_CATCH
}

}  // extern "C"
