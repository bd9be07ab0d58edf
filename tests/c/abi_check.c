/* C99 consumer of include/rivulus_gpu.h, independent of the ctypes mirror (rivulus_amd/capi.py) and of the
 * generated Rust declarations (rust_shim/ffi.rs): it pins the struct layouts all three rely on and drives one
 * query through the ABI the way a foreign caller would.
 *
 *   gcc -std=c99 -pedantic -Wall -Wextra -c tests/c/abi_check.c                    (CPU suite: layout only)
 *   gcc -std=c99 tests/c/abi_check.c -Lrivulus_amd/csrc -lrivulus_gpu -o abi_check    (GPU suite: ./abi_check)
 *
 * The query is BASELINE configs[1] in small: filter(x > 899).select([x]) over x = splitmix64(42 + i) % 1000
 * (SURVEY.md section 8d), checked against the same integer arithmetic done here in plain C. */
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rivulus_gpu.h"

#define LAYOUT(cond, msg) _Static_assert(cond, msg)
LAYOUT(sizeof(rv_column) == 56, "rv_column size");
LAYOUT(offsetof(rv_column, dtype) == 0 && offsetof(rv_column, values) == 8 && offsetof(rv_column, validity) == 16, "rv_column head");
LAYOUT(offsetof(rv_column, offset) == 24 && offsetof(rv_column, length) == 32, "rv_column offset/length");
LAYOUT(offsetof(rv_column, offsets) == 40 && offsetof(rv_column, data_bytes) == 48, "rv_column string part");
LAYOUT(sizeof(rv_term) == 32, "rv_term size");
LAYOUT(offsetof(rv_term, column) == 0 && offsetof(rv_term, op) == 4 && offsetof(rv_term, lit_type) == 8 && offsetof(rv_term, lit) == 16, "rv_term fields");
LAYOUT(sizeof(((rv_term *)0)->lit) == 16, "rv_term literal union");
LAYOUT(sizeof(rv_predicate) == 32, "rv_predicate size");
LAYOUT(offsetof(rv_predicate, terms) == 0 && offsetof(rv_predicate, n_terms) == 8 && offsetof(rv_predicate, nulls) == 12, "rv_predicate head");
LAYOUT(offsetof(rv_predicate, expr) == 16 && offsetof(rv_predicate, n_expr) == 24, "rv_predicate expression");
LAYOUT(sizeof(rv_synth_spec) == 80, "rv_synth_spec size");
LAYOUT(offsetof(rv_synth_spec, seed) == 8 && offsetof(rv_synth_spec, first_row) == 16 && offsetof(rv_synth_spec, length) == 24, "rv_synth_spec 1");
LAYOUT(offsetof(rv_synth_spec, modulus) == 32 && offsetof(rv_synth_spec, true_percent) == 40 && offsetof(rv_synth_spec, with_validity) == 44, "rv_synth_spec 2");
LAYOUT(offsetof(rv_synth_spec, validity_seed) == 48 && offsetof(rv_synth_spec, null_percent) == 56, "rv_synth_spec 3");
LAYOUT(offsetof(rv_synth_spec, pattern) == 60 && offsetof(rv_synth_spec, run_rows) == 64 && offsetof(rv_synth_spec, table_rows) == 72, "rv_synth_spec 4");
LAYOUT(sizeof(rv_column_info) == 48, "rv_column_info size");
LAYOUT(offsetof(rv_column_info, length) == 8 && offsetof(rv_column_info, offset) == 16 && offsetof(rv_column_info, has_validity) == 24, "rv_column_info 1");
LAYOUT(offsetof(rv_column_info, null_count) == 32 && offsetof(rv_column_info, data_bytes) == 40, "rv_column_info 2");
LAYOUT(sizeof(rv_status) == 4 && sizeof(rv_dtype) == 4 && sizeof(rv_cmp) == 4 && sizeof(rv_null_policy) == 4, "enums are ints");
LAYOUT(RV_OK == 0 && RV_ERR_INTERNAL == 8 && RV_STRING == 4 && RV_IS_TRUE == 6 && RV_NULL_IS_LEAST == 1, "enum values");

#ifndef RV_ABI_LAYOUT_ONLY
static uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#define CHECK(call)                                                                             \
    do {                                                                                        \
        rv_status st_ = (call);                                                                 \
        if (st_ != RV_OK) {                                                                     \
            fprintf(stderr, "FAIL %s -> %s: %s\n", #call, rv_status_name(st_), rv_last_error()); \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

int main(void) {
    const uint64_t n = 1000003;
    rv_ctx *ctx = NULL;
    rv_dcolumn *x = NULL, *out = NULL;
    rv_synth_spec spec;
    rv_term term;
    rv_predicate pred;
    rv_column_info info;
    uint32_t proj = 0;
    uint64_t rows = 0, want = 0, i, k;
    int64_t *got;
    int has_validity = -1;

    if (rv_abi_version() != RV_ABI_VERSION) {
        fprintf(stderr, "FAIL abi version %u != %u\n", rv_abi_version(), (unsigned)RV_ABI_VERSION);
        return 1;
    }
    CHECK(rv_ctx_create(0, &ctx));
    memset(&spec, 0, sizeof spec);
    spec.dtype = RV_INT64;
    spec.seed = 42;
    spec.length = n;
    spec.modulus = 1000;
    CHECK(rv_generate(ctx, &spec, &x));
    memset(&term, 0, sizeof term);
    term.column = 0;
    term.op = RV_GT;
    term.lit_type = RV_INT64;
    term.lit.i = 899;
    memset(&pred, 0, sizeof pred);
    pred.terms = &term;
    pred.n_terms = 1;
    pred.nulls = RV_NULL_DROPS;
    CHECK(rv_filter_project(ctx, (const rv_dcolumn *const *)&x, 1, &pred, &proj, 1, &out, &rows, NULL));
    CHECK(rv_column_info_get(ctx, out, &info));
    if (info.dtype != RV_INT64 || info.length != rows || info.offset != 0 || info.has_validity != 0 || info.null_count != 0) {
        fprintf(stderr, "FAIL column info\n");
        return 1;
    }
    got = (int64_t *)malloc((size_t)(rows ? rows : 1) * 8);
    CHECK(rv_download(ctx, out, got, NULL, &has_validity));
    for (i = 0, k = 0; i < n; ++i) {
        const int64_t v = (int64_t)(splitmix64(42 + i) % 1000);
        if (v > 899) {
            if (k >= rows || got[k] != v) {
                fprintf(stderr, "FAIL row %llu: survivor %llu differs\n", (unsigned long long)i, (unsigned long long)k);
                return 1;
            }
            ++k;
        }
    }
    want = k;
    if (want != rows || has_validity != 0) {
        fprintf(stderr, "FAIL rows %llu != %llu\n", (unsigned long long)rows, (unsigned long long)want);
        return 1;
    }
    /* an error crosses the ABI as a status + the reference's text (record_batch.rs:223-227) */
    {
        rv_dcolumn *bad = NULL, *sl = NULL;
        uint64_t r2 = 0;
        CHECK(rv_slice(ctx, x, 0, 3, &sl));
        if (rv_filter(ctx, (const rv_dcolumn *const *)&sl, 1, x, &bad, &r2) != RV_ERR_LENGTH_MISMATCH ||
            strcmp(rv_last_error(), "Predicate length 1000003 doesn't match batch length 3") != 0) {
            fprintf(stderr, "FAIL error text: %s\n", rv_last_error());
            return 1;
        }
        CHECK(rv_free(ctx, sl));
    }
    /* seam S1 with a window in flight (ABI 4): the table as 1024-row RecordBatches, the per-batch survivor counts written by the
     * device into memory from rv_host_alloc; a sorted table (rv_synth_spec::pattern) through the same call */
    {
        const uint64_t nb = (n + 1023) / 1024;
        void *pinned = NULL;
        uint64_t *counts, total = 0, sum = 0;
        rv_pending *pending = NULL;
        rv_dcolumn *wout = NULL, *xs = NULL;
        CHECK(rv_host_alloc(ctx, (size_t)nb * 8, &pinned));
        counts = (uint64_t *)pinned;
        CHECK(rv_filter_project_chunked_begin(ctx, (const rv_dcolumn *const *)&x, 1, 1024, &pred, &proj, 1, counts, nb, &pending));
        CHECK(rv_filter_project_window_finish(ctx, pending, &wout, NULL, &total));
        for (i = 0; i < nb; ++i) sum += counts[i];
        if (total != rows || sum != rows) {
            fprintf(stderr, "FAIL window: %llu survivors, per-batch counts add up to %llu, expected %llu\n", (unsigned long long)total,
                    (unsigned long long)sum, (unsigned long long)rows);
            return 1;
        }
        CHECK(rv_free(ctx, wout));
        spec.pattern = RV_SYNTH_SORTED_ASC;   /* x grows with the row index: x > 899 keeps exactly the last tenth */
        CHECK(rv_generate(ctx, &spec, &xs));
        CHECK(rv_filter_project(ctx, (const rv_dcolumn *const *)&xs, 1, &pred, &proj, 1, &wout, &total, NULL));
        if (total < n / 10 - 1 || total > n / 10 + 1) {
            fprintf(stderr, "FAIL sorted table: %llu survivors of %llu\n", (unsigned long long)total, (unsigned long long)n);
            return 1;
        }
        CHECK(rv_free(ctx, wout));
        CHECK(rv_free(ctx, xs));
        CHECK(rv_host_free(ctx, pinned));
    }
    free(got);
    CHECK(rv_free(ctx, out));
    CHECK(rv_free(ctx, x));
    CHECK(rv_ctx_destroy(ctx));
    printf("ok abi_check: %llu of %llu rows survive, identical to the C restatement of the generator\n", (unsigned long long)rows,
           (unsigned long long)n);
    return 0;
}
#endif
