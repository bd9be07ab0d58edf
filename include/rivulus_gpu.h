/*
 * rivulus_gpu.h -- C ABI of the MI355X (gfx950) execution backend for the
 * Rivulus filter / project / scan hot path.
 *
 * This header is the drop-in boundary.  The reference (CleConor/rivulus, pure
 * Rust) has no FFI of its own, so every entry point below names the reference
 * seam it stands behind (paths relative to the reference checkout):
 *
 *   S1  trait DataStream::next_batch            src/execution/stream.rs:25-28
 *       FilterStream / SelectStream             src/execution/stream.rs:116-213
 *   S2  RecordBatch::{filter,take,concat,slice} src/execution/record_batch.rs:92-342
 *   S3  PhysicalPlan::{Filter,Select}::execute  src/physical_plan/plan.rs:65-150
 *
 * Conventions
 *   - every function returns rv_status (0 == RV_OK); no exception crosses the ABI;
 *   - rv_last_error() returns a thread-local NUL-terminated message; where the
 *     reference defines the failure text, the text is the reference's;
 *   - plain pointers and sizes only; host pointers are borrowed for the duration
 *     of the call; device objects are opaque handles released with rv_free();
 *   - bit buffers are LSB-first (bit i lives in byte i/8, bit i%8), exactly the
 *     reference BitMap layout (src/execution/array/bitmap.rs:48-52,61-68);
 *     validity bit 1 == valid (src/execution/array/primitive.rs:31-33,53-57);
 *   - one rv_ctx == one device + one HIP stream, not thread-safe (mirrors the
 *     `&mut self` of DataStream::next_batch); different contexts are independent.
 */
#ifndef RIVULUS_GPU_H
#define RIVULUS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RV_ABI_VERSION 4

typedef enum rv_status {
    RV_OK = 0,
    RV_ERR_INVALID_ARG = 1,
    RV_ERR_LENGTH_MISMATCH = 2, /* record_batch.rs:222-228 */
    RV_ERR_TYPE_MISMATCH = 3,   /* record_batch.rs:230-233, stream.rs:147-154 */
    RV_ERR_OUT_OF_BOUNDS = 4,   /* record_batch.rs:109-116, :93 */
    RV_ERR_UNSUPPORTED = 5,
    RV_ERR_DEVICE = 6,
    RV_ERR_OOM = 7,
    RV_ERR_INTERNAL = 8
} rv_status;

/* execution::schema::DataType, src/execution/schema.rs:1-8 */
typedef enum rv_dtype {
    RV_NULL = 0,
    RV_BOOLEAN = 1,
    RV_INT64 = 2,
    RV_FLOAT64 = 3,
    RV_STRING = 4
} rv_dtype;

/* the six compare operators of expressions::BinaryOperator, src/expressions/expr.rs:21-26.
 * RV_IS_TRUE is the predicate form the reference streaming path accepts today: a bare
 * Boolean column kept where value(i) == Some(true) (stream.rs:139-158, record_batch.rs:235-240). */
typedef enum rv_cmp {
    RV_EQ = 0,
    RV_NE = 1,
    RV_LT = 2,
    RV_GT = 3,
    RV_LE = 4,
    RV_GE = 5,
    RV_IS_TRUE = 6
} rv_cmp;

/* How a null cell behaves inside a compare term.
 *   RV_NULL_DROPS    streaming composition: a compare over a null cell is null and
 *                    RecordBatch::filter keeps only Some(true) (record_batch.rs:237,
 *                    boolean.rs:120-135) => the row is dropped.
 *   RV_NULL_IS_LEAST eager path: AnyValue ordering, Null == Null and Null < everything
 *                    (src/datatypes/series.rs:87-117), so <, <=, != keep null rows. */
typedef enum rv_null_policy {
    RV_NULL_DROPS = 0,
    RV_NULL_IS_LEAST = 1
} rv_null_policy;

/* Field-for-field view of PrimitiveArray<i64|f64> (primitive.rs:20-28) or BooleanArray
 * (boolean.rs:9-16).  Used for host memory (rv_upload / rv_download) and for
 * caller-owned device memory (rv_wrap). */
typedef struct rv_column {
    rv_dtype dtype;
    const void *values;      /* int64_t[] / double[]; RV_BOOLEAN: LSB-first bit buffer;
                                RV_STRING: the UTF-8 bytes (StringArray::data)            */
    const uint8_t *validity; /* NULL == no null bitmap; LSB-first, bit 1 == valid        */
    uint64_t offset;         /* element offset into values AND bit offset into bitmaps   */
    uint64_t length;         /* logical number of elements                               */
    /* RV_STRING only (string.rs:9-15): element i of the buffer spans
     * values[offsets[i] .. offsets[i+1]); offset + length + 1 entries are read.  A null element
     * spans zero bytes, as StringArray::new builds it (string.rs:34-38). */
    const int32_t *offsets;
    uint64_t data_bytes;     /* bytes in `values` (RV_STRING)                            */
} rv_column;

/* One `Column <op> Literal` term (planner.rs:134-189).  lit_type == RV_NULL is
 * Literal(AnyValue::Null); RV_BOOLEAN literals carry 0/1 in lit.i. */
typedef struct rv_term {
    uint32_t column; /* index into the cols[] array handed to the call */
    rv_cmp op;
    rv_dtype lit_type;
    union {
        int64_t i;
        double f;
        struct {            /* lit_type == RV_STRING: AnyValue::String literal (UTF-8, not NUL-terminated); */
            const char *ptr; /* String cells order byte-wise like Rust's str (series.rs:100-117)             */
            uint64_t len;
        } s;
    } lit;
} rv_term;

/* A predicate over compare terms.  expr == NULL: the AND of all terms (BinaryOperator::And, expr.rs:27).
 * Otherwise expr[0 .. n_expr) is the predicate in postfix order over
 *   0x00 .. 0x7F  push the truth of terms[code]
 *   RV_EXPR_AND   pop two, push a AND b      BinaryOperator::And (expr.rs:27), BooleanArray::and (boolean.rs:120-135)
 *   RV_EXPR_OR    pop two, push a OR b       BinaryOperator::Or  (expr.rs:28), BooleanArray::or  (boolean.rs:137-152)
 *   RV_EXPR_NOT   pop one, push NOT a        BooleanArray::not   (boolean.rs:154-165)
 * and must leave exactly one value.  Null semantics per policy:
 *   RV_NULL_DROPS     every term is a nullable BooleanArray (null where its cell is null), AND / OR / NOT are the
 *                     reference's strict operators (null if an operand is null, `false AND null = null`), and
 *                     RecordBatch::filter keeps Some(true): a row survives iff every cell the expression reads is
 *                     valid and the expression is true;
 *   RV_NULL_IS_LEAST  every term is the eager mask of plan.rs:112-130 (a definite bool per row, nulls ordered
 *                     lowest); AND = chained filters, OR / NOT the same algebra on those masks.
 * n_terms >= 1, at most 16 terms.  Expressions whose conjunctive normal form (or that of their negation) has at
 * most 16 literals run inside the single fused pass; larger ones are composed from per-term BooleanArrays. */
#define RV_EXPR_AND 0x80
#define RV_EXPR_OR 0x81
#define RV_EXPR_NOT 0x82
typedef struct rv_predicate {
    const rv_term *terms;
    uint32_t n_terms;
    rv_null_policy nulls;
    const uint8_t *expr; /* NULL: AND of all terms */
    uint32_t n_expr;
} rv_predicate;

/* On-device synthetic column, bit-identical to the CPU generator in oracle/
 * (SURVEY.md section 8d):  splitmix64(z): z += 0x9E3779B97F4A7C15;
 *   z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9; z = (z ^ (z >> 27)) * 0x94D049BB133111EB;
 *   return z ^ (z >> 31).
 * Row g = first_row + i (global index, so row-range shards agree):
 *   RV_INT64   value = (int64)(splitmix64(seed + g) % modulus)
 *   RV_FLOAT64 value = (double)(splitmix64(seed + g) >> 11) * 2^-53
 *   RV_BOOLEAN value = splitmix64(seed + g) % 100 < true_percent
 *   validity   bit   = splitmix64(validity_seed + g) % 100 >= null_percent
 * Under a null slot the generated value is kept as is (a null slot may hold anything).
 * `pattern` (ABI 4) replaces the hash h = splitmix64(seed + g) the values are made of -- the reference filters whatever
 * order its source has (plan.rs:112-147, file_stream.rs:122-198: file order), and ids / timestamps come sorted or in runs:
 *   RV_SYNTH_IID          h as above (independent rows: a tile's selectivity is the table's)
 *   RV_SYNTH_CLUSTERED    h = splitmix64(seed + g / run_rows): runs of run_rows equal cells
 *   RV_SYNTH_SORTED_ASC   q = g * ((2^64 - 1) / table_rows), a 64-bit fraction that grows with g:
 *                         RV_INT64 value = (q * modulus) >> 64, RV_FLOAT64 value = (double)(q >> 11) * 2^-53,
 *                         RV_BOOLEAN value = ((q * 100) >> 64) < true_percent   (table_rows 0: first_row + length)
 *   RV_SYNTH_SORTED_DESC  the same with q = (table_rows - 1 - g) * ((2^64 - 1) / table_rows)
 * The validity bits stay independent per row under every pattern. */
typedef enum rv_synth_pattern { RV_SYNTH_IID = 0, RV_SYNTH_CLUSTERED = 1, RV_SYNTH_SORTED_ASC = 2, RV_SYNTH_SORTED_DESC = 3 } rv_synth_pattern;
typedef struct rv_synth_spec {
    rv_dtype dtype;
    uint64_t seed;
    uint64_t first_row;
    uint64_t length;
    uint64_t modulus;      /* RV_INT64 only, > 0                                   */
    uint32_t true_percent; /* RV_BOOLEAN only                                      */
    int32_t with_validity; /* 0: no null bitmap                                    */
    uint64_t validity_seed;
    uint32_t null_percent;
    uint32_t pattern;      /* rv_synth_pattern; 0 = independent rows                */
    uint64_t run_rows;     /* RV_SYNTH_CLUSTERED: rows per run, > 0                 */
    uint64_t table_rows;   /* RV_SYNTH_SORTED_*: rows of the whole table (shards agree) */
} rv_synth_spec;

typedef struct rv_column_info {
    rv_dtype dtype;
    uint64_t length;
    uint64_t offset;
    int32_t has_validity; /* a null bitmap is attached                              */
    int64_t null_count;   /* -1 when not yet known                                  */
    uint64_t data_bytes;  /* RV_STRING: bytes of the logical elements, else 0       */
} rv_column_info;

typedef struct rv_ctx rv_ctx;         /* one device + one stream + scratch arena     */
typedef struct rv_dcolumn rv_dcolumn; /* device-resident array (values + validity)   */
typedef struct rv_comm rv_comm;       /* RCCL communicator, one rank per process     */
typedef struct rv_pending rv_pending; /* a fused launch queued by rv_filter_project_begin */

/* ---- library / context ------------------------------------------------- */
uint32_t rv_abi_version(void);
const char *rv_last_error(void);
const char *rv_status_name(rv_status s);

/* number of HIP devices this process sees (0 without a GPU; never an error: the library reports, it does not fall back) */
int rv_device_count(void);
rv_status rv_ctx_create(int device, rv_ctx **out);
rv_status rv_ctx_destroy(rv_ctx *ctx);
rv_status rv_ctx_synchronize(rv_ctx *ctx);
/* hipStream_t of the context, for callers that interleave their own work. */
void *rv_ctx_stream(rv_ctx *ctx);
/* number of compute units of the context's device (grid sizing, reporting). */
rv_status rv_ctx_device_info(rv_ctx *ctx, int *compute_units, uint64_t *hbm_bytes, char *name,
                             size_t name_len);
/* Tuning / diagnostics.  Keys: "rows_per_lane" (R | waves << 8: geometry of the fused
 * kernel; 0 = default), "vec" (0 auto, 1 force 8-byte loads, 2 force 16-byte loads),
 * "cap_rows" (rows of a wave's LDS slot, 0 = as many as fit), "depth" (iterations between a
 * tile's aggregate and its write-out: 0 auto, 1, 2), "roomy" (1: size the LDS slots as for a dense selection -- one
 * workgroup per CU, two stages; 0 = decided per launch from the selectivity the context last saw with the same predicate:
 * a selectivity the default geometry's slots would not hold takes geometries with fewer rows per lane, results unchanged),
 * "direct" (the register-staged kernel for dense selections, direct_kernel.hpp: 0 = from a selectivity of 52 % (one loaded
 * column) / 60 % (one projected of several) / 22 % (two projected) / 15 % (three, four) known for the predicate, for value columns
 * without an output bitmap, 1 = whenever the launch is eligible, -1 = never; "direct_r" / "direct_waves": diagnostic, a named
 * geometry of it), "sample" (a predicate the context has not run over this data gets its selectivity from a strided sample before
 * its first launch is sized: 0 = for tables of 2^25 rows and more, k > 0 = from k rows on, -1 = never),
 * "skew" (survivors that come in RUNS -- a table sorted or clustered on the predicate's column -- fill some waves' LDS slots while
 * others stay empty; a wave range that outgrows its slot is re-read by the redo kernel at its reserved output offset.  0 = when the
 * share of such ranges the predicate is known to leave (the last staged pass, or the sample's histogram of 1024-row blocks on a first
 * call) would cost more than the direct kernel's slower scan, the direct kernel runs instead; -1 = never: staged pass + redo kernel),
 * "segments" (a table whose survivors sit in a few long STRETCHES -- sorted on the predicate's column: the first call's sample shows its
 * 1024 blocks in table order -- is cut at those edges and filtered stretch by stretch, each with the kernel its own density asks for, all
 * into one set of outputs; 8-byte value columns, queries of one pass.  0 = tables of 2^28 rows and more in which the sample shows two
 * to four such stretches (a stretch more costs a launch and a read-back: below that one pass is ahead); k > 0 = from k rows on; -1 = never.
 * Same rows in the same order either way),
 * "str_tiles_from" (String columns of a filter are copied tile by tile of 512 SOURCE rows, without the survivors' (start, length)
 * lists, when the pass expects at least this share of the rows to survive: 0 = from 50 %, k > 1 = from k %, 1 = always, -1 = never),
 * "groups_by_ranges" (plain 8-byte columns can be compacted AFTER the pass at its wave offsets, by a kernel without a chain between
 * its waves: 0 = the column groups beyond the first four of a wide projection while up to 55 % of the rows survive, and every plain
 * column the predicate does not read when the predicate kept at most a quarter of the rows the last time -- a nullable one at
 * every selectivity; 1 = always; -1 = never),
 * "speculative_batches" (rv_filter_project_batches launches the pass of a window that looks regular before its handles are
 * validated and validates meanwhile: 0 = from 4096 batches on, -1 = never), "wgs_per_cu" (0 = occupancy query),
 * "agg_grid" (rv_filter_agg: workgroups per CU striding over the tiles; 0 = 8192 workgroups whatever the CU count, -1 = one workgroup per tile),
 * "profile_kernels" (0/1), "out_sizing" (capacity of the output buffers: 0, the default = a table of 2^25 rows and more gets outputs
 * for what its predicate is known to keep -- its last pass over these buffers, or the first call's sample -- x 1.2 + 2 % of the rows,
 * smaller tables and unknown predicates outputs for every row; -1 = every row may survive, always: 2x the input in HBM;
 * 1 = the context's last observed selectivity x 1.5 + 1 %; k >= 2 = a caller-given bound of k rows per
 * million.  A launch whose survivors do not fit still counts exactly and is re-run once with outputs of the exact size, so
 * the bound is a hint, never a correctness matter), "bools_in_pass" (1: projected Boolean columns are compacted inside the fused pass instead
 * of by the bit-compaction kernel after it; measured slower, kept selectable).  Diagnostics only, never for results: "stamp" (per-phase cycle
 * counters, printed to stderr) and "debug" (bit 0 skip the value stores, bit 1 skip the
 * output-offset lookup -- both make the output WRONG, timing shares only -- bit 2 print
 * scanner / fallback look-back counts, bit 3 run without the scanner wave: results stay correct, bit 4 fault injection: tile 1 never
 * publishes its count, so the bounded waits behind it give up and the call returns RV_ERR_DEVICE) select the separate FF_STAMP
 * instantiations of the kernel, which exist for three shapes; the production instantiations contain none of it.  "spin_limit": polls a
 * wait for another workgroup's descriptor may take before it gives up (0 = default, 4 Mi polls = seconds).
 * "inject_failure" = k: the next k query calls on this context (rv_filter_project*, rv_filter_agg) return RV_ERR_DEVICE
 * before anything is launched -- fault injection for the failure handling of rv_group_* (tests/test_group_gpu.py).
 * "bool_cap" = k > 0: Boolean columns compacted behind the pass get output bitmaps of at most k rows (they are sized for the expected
 * survivors + 25 %; more survivors than that and the column takes the scan path after the pass: this forces that fallback in tests). */
rv_status rv_ctx_set_option(rv_ctx *ctx, const char *key, int64_t value);
/* current value of an option, or of the read-only counters "overflow_reruns" (launches re-run because speculatively sized
 * outputs were too small), "last_selectivity_ppm" (survivors per million rows of the last fused launch, -1: none yet),
 * "last_redo_ppm" (wave ranges -- 64 x rows-per-lane rows -- per million of that launch whose survivors did not fit the wave's LDS slot and were re-read by the redo kernel) and
 * "batch_counts_in_pass" (rv_filter_project_chunked / _batches calls whose per-batch survivor counts came out of the fused
 * pass itself rather than from a second read of the selection bitmap), "fused_rows_scanned" (input rows of every fused
 * filter launch of the context so far: what a pushed-down Limit keeps small), "samples_taken" (selectivity samples so far),
 * "speculative_batch_passes", "last_rows_in" / "last_rows_out" (rows / survivors of the last fused pass), "hbm_free_bytes"
 * (what the device reports free right now), "segmented_passes" (queries that ran stretch by stretch, option "segments") and
 * "segment_fallbacks" (... that started so and ran as one pass after all: more survivors than the sample's profile promised). */
rv_status rv_ctx_get_option(rv_ctx *ctx, const char *key, int64_t *value);

/* Device time of the hot-path kernel launches (fused filter+compact, filter+aggregate)
 * since the last reset, measured with HIP events recorded on the context stream directly
 * around each launch.  Collected only while option "profile_kernels" is 1. */
rv_status rv_ctx_kernel_stats(rv_ctx *ctx, double *total_ms, uint64_t *launches, int reset);

/* The hot-path kernel instantiation the context launched last, spelled as rocprofv3 prints it without the spaces
 * ("fused_filter_compact<1,16,2,16,32>", "filter_agg_kernel<1,16,2,4,0>"; "" before the first launch): what a bench
 * line's roofline.kernel names, and the key under which profiles/traffic.json holds that kernel's PMC traffic. */
rv_status rv_ctx_last_kernel(rv_ctx *ctx, char *name, size_t name_len);

/* HIP-event timer on the context stream (bench harness; hipEventRecord both ends). */
rv_status rv_timer_start(rv_ctx *ctx);
rv_status rv_timer_stop(rv_ctx *ctx, float *elapsed_ms);

/* ---- arrays: PrimitiveArray / BooleanArray (a1-a3) ---------------------- */
/* copy a host array to the device; offset is preserved (the buffers are copied from
 * element 0 up to offset+length, like the Arc<[T]> an array slice shares). */
rv_status rv_upload(rv_ctx *ctx, const rv_column *host, rv_dcolumn **out);
/* adopt caller-owned device memory (no copy, not freed by rv_free). */
rv_status rv_wrap(rv_ctx *ctx, const rv_column *device, rv_dcolumn **out);
rv_status rv_generate(rv_ctx *ctx, const rv_synth_spec *spec, rv_dcolumn **out);
rv_status rv_free(rv_ctx *ctx, rv_dcolumn *col);
/* Array::slice, zero-copy (primitive.rs:107-117, boolean.rs:208-219). */
rv_status rv_slice(rv_ctx *ctx, const rv_dcolumn *col, uint64_t offset, uint64_t length,
                   rv_dcolumn **out);
rv_status rv_column_info_get(rv_ctx *ctx, const rv_dcolumn *col, rv_column_info *out);
/* Array::null_count (primitive.rs:90-105): popcount of the validity range on device. */
rv_status rv_null_count(rv_ctx *ctx, const rv_dcolumn *col, uint64_t *out);
/* Download the logical range [0,length).  values: length*8 bytes (RV_INT64/RV_FLOAT64) or
 * ceil(length/8) bytes (RV_BOOLEAN), re-based to offset 0, tail bits zero.  validity:
 * ceil(length/8) bytes, may be NULL; *has_validity tells whether the array carries one. */
rv_status rv_download(rv_ctx *ctx, const rv_dcolumn *col, void *values, uint8_t *validity,
                      int *has_validity);
/* raw device pointers of a device column (interop; valid until rv_free). */
rv_status rv_device_ptrs(rv_ctx *ctx, const rv_dcolumn *col, rv_column *out);

/* StringArray (string.rs:8-147): download of the logical elements.  offsets receives length + 1
 * entries starting at 0, data rv_column_info.data_bytes bytes. */
rv_status rv_download_string(rv_ctx *ctx, const rv_dcolumn *col, int32_t *offsets, uint8_t *data,
                             uint8_t *validity, int *has_validity);

/* dataframe_to_batches' treatment of nulls (streaming.rs:135-233): the null cells of an Int64 / Float64 /
 * Boolean array become 0 / 0.0 / false and the array loses its bitmap; String arrays keep their nulls (a
 * shared view is returned), so does an array without a bitmap. */
rv_status rv_fill_nulls(rv_ctx *ctx, const rv_dcolumn *col, rv_dcolumn **out);

/* ---- predicate evaluation (K1) ------------------------------------------ */
/* AND-of-compares over cols -> selection BooleanArray without validity: bit i == 1 iff
 * row i survives RecordBatch::filter under pred->nulls.  *out_count = survivors.
 * Replaces the eager mask loop (plan.rs:112-130) and the index scan
 * (record_batch.rs:235-240). */
rv_status rv_eval_predicate(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols,
                            const rv_predicate *pred, rv_dcolumn **out_selection,
                            uint64_t *out_count);
/* One compare term as a nullable BooleanArray: value = cell <op> literal, null where the
 * cell is null (the composition rule of SURVEY.md section 8c for streaming compares). */
rv_status rv_compare(rv_ctx *ctx, const rv_dcolumn *col, rv_cmp op, rv_dtype lit_type,
                     int64_t lit_i, double lit_f, rv_dcolumn **out_bool);
/* The same with the literal given as a term (term->column is ignored): the form that takes String literals
 * (`name == "Bob"` over a StringArray column, byte-wise str ordering). */
rv_status rv_compare_term(rv_ctx *ctx, const rv_dcolumn *col, const rv_term *term, rv_dcolumn **out_bool);

/* ---- BooleanArray logic (K3, boolean.rs:120-180) -------------------------- */
rv_status rv_boolean_and(rv_ctx *ctx, const rv_dcolumn *a, const rv_dcolumn *b, rv_dcolumn **out);
rv_status rv_boolean_or(rv_ctx *ctx, const rv_dcolumn *a, const rv_dcolumn *b, rv_dcolumn **out);
rv_status rv_boolean_not(rv_ctx *ctx, const rv_dcolumn *a, rv_dcolumn **out);
rv_status rv_boolean_count(rv_ctx *ctx, const rv_dcolumn *a, uint64_t *count_true,
                           uint64_t *count_false);

/* ---- RecordBatch kernels (S2) -------------------------------------------- */
/* RecordBatch::filter (record_batch.rs:221-243): predicate must be RV_BOOLEAN of the
 * batch length; rows with Some(true) are kept in ascending order; every column goes
 * through take_array semantics (null slots -> 0 / 0.0 / false, validity dropped when no
 * null survives, output offset 0).  out[] receives ncols new handles. */
rv_status rv_filter(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols,
                    const rv_dcolumn *predicate, rv_dcolumn **out, uint64_t *out_rows);
/* RecordBatch::take (record_batch.rs:108-178): arbitrary host index list. */
rv_status rv_take(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols,
                  const uint64_t *indices, uint64_t n_indices, rv_dcolumn **out);
/* The same with the index list resident on the device (an Int64 array without nulls): a selection -> indices -> take
 * chain never leaves HBM.  The bounds pre-pass (record_batch.rs:109-116) is a device reduction that finds the FIRST
 * offending position, so the error text is the reference's ("Index 7 out of bounds for 5 rows"). */
rv_status rv_take_device(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, const rv_dcolumn *indices,
                         rv_dcolumn **out);
/* The ascending index list RecordBatch::filter builds from its predicate (record_batch.rs:235-240): the rows of a
 * BooleanArray that are Some(true), as an Int64 device array. */
rv_status rv_selection_indices(rv_ctx *ctx, const rv_dcolumn *selection, rv_dcolumn **out_indices);
/* concat_arrays (record_batch.rs:277-342) for one column position across batches. */
rv_status rv_concat(rv_ctx *ctx, const rv_dcolumn *const *parts, uint32_t nparts,
                    rv_dcolumn **out);

/* ---- fused filter + project (K1+K2, the hot path) ------------------------- */
/* == SelectStream(FilterStream(input)) on one batch (stream.rs:136-158, :202-210) and
 * == PhysicalPlan::Filter followed by Select (plan.rs:97-150, :68-96): evaluate pred
 * over cols, keep surviving rows of the projected columns proj[0..nproj) in ascending
 * row order.  Single pass over HBM.  out[] receives nproj handles; if out_selection is
 * non-NULL the selection bitmap is materialised as well. */
rv_status rv_filter_project(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols,
                            const rv_predicate *pred, const uint32_t *proj, uint32_t nproj,
                            rv_dcolumn **out, uint64_t *out_rows, rv_dcolumn **out_selection);

/* The same in two halves, for a streaming operator that keeps the device busy: begin queues the launch of
 * batch k+1 and returns; finish of batch k (row count, output lengths, null counts) is called afterwards,
 * so the host work and the result read-back of one batch overlap the kernel of the next.  Launches finish
 * in the order they began.  Shapes that need several passes (String columns, more columns than one pass
 * takes) are completed inside begin.  The input columns must stay alive until finish; finish consumes the
 * pending handle (also on error). */
rv_status rv_filter_project_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols,
                                  const rv_predicate *pred, const uint32_t *proj, uint32_t nproj,
                                  rv_pending **out_pending);
rv_status rv_filter_project_finish(rv_ctx *ctx, rv_pending *pending, rv_dcolumn **out, uint64_t *out_rows);

/* Seam S1 at the reference's batch size.  The reference streams 1024-row RecordBatches (streaming_planner.rs:32); one
 * launch per such batch costs ~30 us of fixed launch / read-back time for ~10 ns of work.  This call takes K
 * device-resident input batches of one schema (cols[b * ncols + c] = column c of batch b) and runs
 * SelectStream(FilterStream(batch)) (stream.rs:136-158, :202-210) on ALL of them in one pass:
 *   - batches that are adjacent zero-copy slices of the same buffers (RecordBatch::slice, dataframe_to_batches:
 *     streaming.rs:135-233) are read as they lie in HBM, as one batch; separately allocated batches are joined by one
 *     device concat first;
 *   - out[j] (nproj handles) holds the K output batches back to back, in batch order; output batch b is the rows
 *     [sum(out_rows[0..b)), + out_rows[b]) of every out[j] -- rv_slice_known cuts it out without copying;
 *   - out_rows[K]: surviving rows per batch (ONE read-back for all K); out_nulls[K * nproj] (may be NULL): null count
 *     of every output batch and column, so that a slice drops its bitmap exactly where the reference's builder would
 *     (primitive.rs:179-185); *out_total: all survivors.
 * Result == rv_filter_project on every batch on its own.  A window of 4096 batches and more that looks regular from its first and
 * last batch (adjacent zero-copy slices of one length) is filtered on that assumption WHILE its K x ncols handles are validated; what
 * the validation rejects is dropped and reported as the first offending batch's error -- after an error the contents of out_rows /
 * out_nulls are unspecified (the speculative pass may have written counts there). */
rv_status rv_filter_project_batches(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t nbatches, uint32_t ncols,
                                    const rv_predicate *pred, const uint32_t *proj, uint32_t nproj, rv_dcolumn **out,
                                    uint64_t *out_rows, int64_t *out_nulls, uint64_t *out_total);
/* The same over ONE resident table cut the way the reference's chunker cuts a DataFrame: StreamingPhysicalPlan::execute
 * on DataFrameSource -> Filter -> Select with batch size `chunk_rows` (dataframe_to_batches, streaming.rs:135-233;
 * default 1024, streaming_planner.rs:32), collected.  Batch k is rows [k * chunk_rows, min((k + 1) * chunk_rows, length));
 * there are ceil(length / chunk_rows) of them (none for an empty table), and the caller passes the capacity of out_rows
 * (and of out_nulls / nproj) in `nchunks`.  No per-batch handles to build or walk and no boundary table to upload: at 1024
 * rows per batch this is what keeps the host side of the call under the device time.  Result == rv_filter_project_batches
 * over rv_slice(cols, k * chunk_rows, ...) for every k.
 * When a batch is a whole number of the pass's wave ranges (64 x rows-per-lane rows: 1024 by default, so the reference's
 * 1024-row batches qualify) the per-batch survivor counts come out of the fused pass itself: no selection bitmap is
 * written or re-read.  If out_rows lies in memory from rv_host_alloc / rv_host_register the device writes the counts
 * there directly (8 bytes per batch cross PCIe once and the host copies nothing); a pageable array is filled from a
 * pinned staging block. */
rv_status rv_filter_project_chunked(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, uint64_t chunk_rows,
                                    const rv_predicate *pred, const uint32_t *proj, uint32_t nproj, rv_dcolumn **out,
                                    uint64_t *out_rows, uint64_t nchunks, int64_t *out_nulls, uint64_t *out_total);
/* Seam S1 with TWO WINDOWS IN FLIGHT (ABI 4).  A stream operator pulls batches one at a time (DataStream::next_batch,
 * stream.rs:25-28) and looks ahead by a window of them; begun before the window in front of it is finished, a window's pass runs
 * while the host reads that window's counts and hands its batches on -- the device never waits for the host between windows.
 *   rv_filter_project_chunked_begin  == rv_filter_project_chunked up to the launch; out_rows (capacity nchunks) is written by the
 *                                       device when it lies in pinned memory (rv_host_alloc / rv_host_register) -- on a side stream,
 *                                       beside the next window's pass -- and is complete when finish returns;
 *   rv_filter_project_batches_begin  == rv_filter_project_batches likewise: the pass is launched on what the window's first and
 *                                       last batch say it is (a regular stream's), and the walk over its K x ncols handles
 *                                       validates that on a helper thread until finish -- which drops the result and takes the
 *                                       ordinary path (or reports the first offending batch) when the walk says otherwise;
 *   rv_filter_project_window_finish  waits, hands over out[nproj], out_nulls[K * nproj] (may be NULL), *out_total; consumes the
 *                                       pending handle (also on error).  Windows finish in the order they began.
 * Queued: shapes of one chained pass whose batches it can count (a batch a whole number of its wave ranges: 1024 rows do) with
 * device-writable out_rows.  Everything else (several passes, String columns, the mask path of a Boolean-column predicate, pageable
 * out_rows) completes inside begin.  cols / pred / proj / out_rows must stay valid until finish. */
rv_status rv_filter_project_chunked_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols, uint64_t chunk_rows,
                                          const rv_predicate *pred, const uint32_t *proj, uint32_t nproj, uint64_t *out_rows,
                                          uint64_t nchunks, rv_pending **out_pending);
rv_status rv_filter_project_batches_begin(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t nbatches, uint32_t ncols,
                                          const rv_predicate *pred, const uint32_t *proj, uint32_t nproj, uint64_t *out_rows,
                                          rv_pending **out_pending);
rv_status rv_filter_project_window_finish(rv_ctx *ctx, rv_pending *pending, rv_dcolumn **out, int64_t *out_nulls,
                                          uint64_t *out_total);
/* rv_slice with the null count of the range supplied by the caller (from rv_filter_project_batches): the view drops
 * the bitmap when it is 0 and needs no device pass to answer null_count(). */
rv_status rv_slice_known(rv_ctx *ctx, const rv_dcolumn *col, uint64_t offset, uint64_t length, int64_t null_count,
                         rv_dcolumn **out);

/* ---- host-resident batches: chunked, overlapped upload + filter + project ---- */
/* Pinned host memory for array buffers.  A caller that keeps its Arc<[T]> / Arc<[u8]> backing
 * stores here gets DMA at PCIe rate and truly asynchronous chunk uploads; pageable buffers work
 * too (the copy is then staged by the runtime). */
rv_status rv_host_alloc(rv_ctx *ctx, size_t bytes, void **out);
rv_status rv_host_free(rv_ctx *ctx, void *ptr);
/* StreamingPhysicalPlan::collect() over a host table (streaming.rs:71-133): replaces
 * dataframe_to_batches (:135-233) + the per-batch pull loop + the final concat
 * (collect_stream_batches, :343-352).  The table is cut into chunks of chunk_rows rows
 * (0 = default 32 Mi; rounded up to a multiple of 64); the upload of chunk k+1 runs on a second
 * stream while chunk k is filtered; the per-chunk outputs are concatenated on the device
 * (rv_concat rules: validity kept only if a null survived).  host_cols[i].offset is honoured
 * (primitive.rs:62-64); String columns are cut at element boundaries (offsets rebased per chunk).
 * Result == rv_filter_project on the uploaded whole columns. */
rv_status rv_filter_project_host(rv_ctx *ctx, const rv_column *host_cols, uint32_t ncols,
                                 const rv_predicate *pred, const uint32_t *proj, uint32_t nproj,
                                 uint64_t chunk_rows, rv_dcolumn **out, uint64_t *out_rows);

/* ---- filter + global aggregate (K4) --------------------------------------- */
/* COUNT(*) of surviving rows and SUM(cols[agg_col]) over surviving non-null cells.
 * RV_INT64: two's-complement wrapping sum in *sum_i (order independent => bit exact).
 * RV_FLOAT64: sum in *sum_f, fixed reduction tree: the order of the additions depends on the row count alone (8192
 * workgroups stride over the tiles whatever the device's CU count; option "agg_grid" changes the tree), so a given column
 * sums to the same bits run to run and part to part.  Row-range shards of different sizes sum in a different order:
 * across shardings the Float64 result agrees to rounding (tests: 1e-12 relative), the Int64 result exactly.
 * The reference has no aggregate operator; semantics defined in DESIGN.md. */
rv_status rv_filter_agg(rv_ctx *ctx, const rv_dcolumn *const *cols, uint32_t ncols,
                        const rv_predicate *pred, uint32_t agg_col, int64_t *sum_i,
                        double *sum_f, uint64_t *count);

/* ---- multi-GPU (one process per GPU) --------------------------------------- */
/* Row-range shard of rank `rank` of `world`: [*begin, *end), boundaries at multiples of
 * 64 rows so selection-bitmap words never straddle ranks. */
rv_status rv_shard_range(uint64_t n_rows, uint32_t world, uint32_t rank, uint64_t *begin,
                         uint64_t *end);
#define RV_COMM_ID_BYTES 128
rv_status rv_comm_unique_id(uint8_t id[RV_COMM_ID_BYTES]);
rv_status rv_comm_create(rv_ctx *ctx, const uint8_t id[RV_COMM_ID_BYTES], uint32_t world,
                         uint32_t rank, rv_comm **out);
/* ncclAllReduce(count = 2, ncclInt64, ncclSum) over xGMI: {sum, count} in place. */
rv_status rv_comm_allreduce_sum_count(rv_comm *comm, int64_t *sum, uint64_t *count);
rv_status rv_comm_destroy(rv_comm *comm);

/* ---- multi-GPU, ONE process: a group of contexts (SURVEY.md sections 8b, 8e) ---------- */
/* The caller the north star names is a single Rust process driving the 8 GPUs of a node.  A group owns one
 * context and one host worker thread per entry of `devices`; a table is sharded by row range (shard r holds
 * rows rv_shard_range(length, n, r)), every device runs the same single-pass kernels on its shard, and there is
 * no collective on the data path.  The same device may be listed more than once (contexts are independent):
 * that is how the protocol is rehearsed on a one-GPU box.  Calls on one group are serialised by the caller
 * (`&mut self`), like calls on one context. */
typedef struct rv_group rv_group;
typedef struct rv_gather rv_gather; /* a query result gathered on the host, in rank order == row order */

rv_status rv_group_create(const int *devices, uint32_t n, rv_group **out);
rv_status rv_group_destroy(rv_group *group);
uint32_t rv_group_size(const rv_group *group);
rv_ctx *rv_group_ctx(rv_group *group, uint32_t rank);

/* Sharded columns: shards[r] lives on rank r's device.  rv_group_generate: rank r generates rows
 * [begin_r, end_r) of the spec with the GLOBAL row index (first_row + begin_r), so the shards agree with the
 * unsharded column without moving data.  rv_group_upload: rank r receives rows [begin_r, end_r) of the host
 * array (offset honoured).  rv_group_free releases the n handles. */
rv_status rv_group_generate(rv_group *group, const rv_synth_spec *spec, rv_dcolumn **shards);
rv_status rv_group_upload(rv_group *group, const rv_column *host, rv_dcolumn **shards);
rv_status rv_group_free(rv_group *group, rv_dcolumn **shards);

/* BASELINE configs[3]: row-range partitioned filter + project, no collective, concat on the host.
 * shards[r * ncols + c] = column c of rank r.  Every rank runs rv_filter_project on its shard (all devices
 * at once); the N survivor counts are prefix-summed on the host and every device copies its output into its
 * slice of ONE pinned host buffer per column -- the gather the reference does with
 * collect_stream_batches -> RecordBatch::concat (streaming.rs:343-352, record_batch.rs:245-342), with the
 * shards as the batches: rank order == row order, validity kept only if a null survived somewhere. */
/* The same for a HOST-resident table (ABI 4): StreamingPhysicalPlan::collect() over N devices (streaming.rs:71-133; dataframe_to_batches
 * :135-233 is the source it replaces, collect_stream_batches -> concat :343-352 the gather).  The table is cut into N row ranges
 * (rv_shard_range), every range streamed through its device's own double-buffered chunk pipeline (rv_filter_project_host) -- all
 * devices at once, each over its own PCIe link -- and the survivors gathered in rank order == row order.  host_cols as for
 * rv_filter_project_host (pinned memory uploads at link rate); chunk_rows 0 = 32 Mi.  rank_upload_gbs[n] (may be NULL): what every
 * rank's link carried, GB/s of its range's input bytes over its filter time. */
rv_status rv_group_filter_project_host(rv_group *group, const rv_column *host_cols, uint32_t ncols, const rv_predicate *pred,
                                       const uint32_t *proj, uint32_t nproj, uint64_t chunk_rows, rv_gather **out,
                                       uint64_t *out_rows, double *rank_upload_gbs);
rv_status rv_group_filter_project(rv_group *group, const rv_dcolumn *const *shards, uint32_t ncols,
                                  const rv_predicate *pred, const uint32_t *proj, uint32_t nproj,
                                  rv_gather **out, uint64_t *out_rows);
/* The same in its two phases, for a caller that keeps the result in HBM (and for measuring them apart):
 *   rv_group_filter_project_resident  every rank's rv_filter_project, all devices at once; outs[r * nproj + j] = output
 *                                     column j of rank r, resident on rank r's device (freed with rv_free on
 *                                     rv_group_ctx(group, r)); rank_rows[n] (may be NULL) the survivors per rank.  If any
 *                                     rank fails, every output is released and the first failure is returned -- after ALL
 *                                     ranks have finished, so nothing is still running on a device when the call returns;
 *   rv_group_gather                   the collect leg over such outputs: prefix sum of the N lengths, one pinned host
 *                                     buffer per column, rank order == row order (streaming.rs:343-352). */
rv_status rv_group_filter_project_resident(rv_group *group, const rv_dcolumn *const *shards, uint32_t ncols,
                                           const rv_predicate *pred, const uint32_t *proj, uint32_t nproj,
                                           rv_dcolumn **outs, uint64_t *rank_rows, uint64_t *out_rows);
rv_status rv_group_gather(rv_group *group, const rv_dcolumn *const *outs, uint32_t nproj, rv_gather **out);
/* column j of the result as a host rv_column (pinned memory owned by the result; offset 0). */
rv_status rv_gather_column(const rv_gather *result, uint32_t j, rv_column *view, int64_t *null_count);
/* surviving rows per rank (n entries) and the two phases' wall times: filter_ms = slowest rank's
 * rv_filter_project (launch to row count), gather_ms = prefix sum + device-to-host copies + bitmap merge. */
rv_status rv_gather_stats(const rv_gather *result, uint64_t *rank_rows, double *filter_ms, double *gather_ms);
rv_status rv_gather_free(rv_gather *result);

/* BASELINE configs[4]: rv_filter_agg per rank, then ONE all-reduce of {SUM, COUNT} -- ncclAllReduce(count = 2,
 * ncclInt64, ncclSum) (+ 1 x ncclFloat64 for a Float64 SUM) on communicators made by ncclCommInitAll over the
 * group's devices; every rank ends with the same 16 bytes (checked).  A group that lists a device twice cannot
 * form an RCCL communicator: its partials are summed on the host in rank order instead. */
rv_status rv_group_filter_agg(rv_group *group, const rv_dcolumn *const *shards, uint32_t ncols,
                              const rv_predicate *pred, uint32_t agg_col, int64_t *sum_i, double *sum_f,
                              uint64_t *count);
/* Failure contract of the collective: the all-reduce is entered only after EVERY rank's rv_filter_agg has returned
 * successfully (a rank that fails -- out of memory on one device, a device fault -- makes the call return that rank's
 * error; no rank is left waiting inside RCCL), it is issued for all ranks by the calling thread inside one
 * ncclGroupStart / ncclGroupEnd, and the wait for it is bounded (environment RV_GROUP_TIMEOUT_MS, default 120 000):
 * on a failure or timeout the communicators are aborted (ncclCommAbort) and re-made by the next call.
 * Counters of a group: "rccl_ranks" (ranks of the communicator ncclCommInitAll formed, 0: none yet / host sum),
 * "allreduce_calls", "comm_aborts", "distinct_devices" (0 / 1), "last_agg_filter_us", "last_allreduce_us" (wall time of
 * the two phases of the last rv_group_filter_agg). */
rv_status rv_group_stat(rv_group *group, const char *key, int64_t *value);

/* Pin caller-owned host memory (e.g. a shared-memory segment several one-GPU processes gather into) so that
 * rv_download / rv_upload move it by DMA. */
rv_status rv_host_register(rv_ctx *ctx, void *ptr, size_t bytes);
rv_status rv_host_unregister(rv_ctx *ctx, void *ptr);

#ifdef __cplusplus
}
#endif
#endif /* RIVULUS_GPU_H */
