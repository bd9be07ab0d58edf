// UNCOMPILED reference-side glue (the image has no rustc): how `impl DataStream` (src/execution/stream.rs:25-28) stands on
// the C ABI declared in ffi.rs (generated from include/rivulus_gpu.h).  It is the Rust twin of
// `execution::GpuFilterProjectStream` in rivulus_amd/host/rivulus_host.hpp, which IS compiled and tested here
// (tests/cpp/host_tests.cpp: gpu_filter_project_stream_matches_composed_oracle, ..._or_of_compares, ..._empty_and_errors).
//
// Placement in the reference: src/execution/gpu/{ffi.rs, gpu_stream.rs}; `pub mod gpu;` in src/execution/mod.rs; the
// Filter arm of logical_to_streaming (src/physical_plan/streaming_planner.rs:70-74) routes here when `lower_predicate`
// accepts the expression.
use std::collections::VecDeque;
use std::ffi::CStr;
use std::ptr::{null, null_mut};
use std::sync::Arc;

use super::ffi::*;
use crate::execution::{DataStream, DataStreamRef, RecordBatch, Result, Schema, StreamError};

fn last_error() -> String {
    unsafe { CStr::from_ptr(rv_last_error()) }.to_string_lossy().into_owned()
}

/// One device + one stream (`rv_ctx`); `&mut self` everywhere, like `DataStream::next_batch`.
#[derive(Debug)]
pub struct GpuCtx { pub raw: *mut RvCtx }
unsafe impl Send for GpuCtx {}
unsafe impl Sync for GpuCtx {}
impl GpuCtx {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = null_mut();
        if unsafe { rv_ctx_create(device, &mut raw) } != RV_OK { return Err(StreamError::Execution { message: last_error() }); }
        Ok(GpuCtx { raw })
    }
}
impl Drop for GpuCtx { fn drop(&mut self) { unsafe { rv_ctx_destroy(self.raw); } } }

/// SelectStream(FilterStream(input)) with the predicate lowered to compare terms (+ AND / OR), a window of batches per launch.
#[derive(Debug)]
pub struct GpuFilterProjectStream {
    input: DataStreamRef,
    ctx: Arc<GpuCtx>,
    terms: Vec<RvTerm>,        // `column` = slot in `columns` below
    expr: Vec<u8>,             // postfix RV_EXPR_AND / RV_EXPR_OR over the terms; empty = AND of all
    nulls: i32,                // RV_NULL_DROPS (streaming composition)
    columns: Vec<usize>,       // batch column index of every slot the predicate / projection reads
    projection: Vec<u32>,      // slots, in output order
    output_schema: Arc<Schema>,
    ready: VecDeque<RecordBatch>,
    window_batches: usize,     // e.g. 4096
    window_rows: usize,        // e.g. 1 << 28
}

impl DataStream for GpuFilterProjectStream {
    fn schema(&self) -> Arc<Schema> { self.output_schema.clone() }

    fn next_batch(&mut self) -> Result<Option<RecordBatch>> {
        if self.ready.is_empty() { self.refill()?; }
        Ok(self.ready.pop_front())
    }
}

impl GpuFilterProjectStream {
    fn refill(&mut self) -> Result<()> {
        let mut window: Vec<RecordBatch> = Vec::new();
        let mut rows = 0usize;
        while window.len() < self.window_batches && rows < self.window_rows {
            match self.input.next_batch()? {
                Some(b) => { rows += b.num_rows(); window.push(b); }
                None => break,
            }
        }
        if window.is_empty() { return Ok(()); }
        // device handles of the referenced columns, batch-major: cols[b * ncols + c].  `device_column` uploads an array once
        // (rv_upload of its Arc<[T]> / Arc<[u8]> buffers, offset preserved) and caches the handle next to the Arc.
        let ncols = self.columns.len();
        let cols: Vec<*const RvDcolumn> = window.iter()
            .flat_map(|b| self.columns.iter().map(move |&c| device_column(&self.ctx, b.column(c))))
            .collect();
        let pred = RvPredicate {
            terms: self.terms.as_ptr(), n_terms: self.terms.len() as u32, nulls: self.nulls,
            expr: if self.expr.is_empty() { null() } else { self.expr.as_ptr() }, n_expr: self.expr.len() as u32,
        };
        let (k, np) = (window.len(), self.projection.len());
        let mut out = vec![null_mut(); np.max(1)];
        let mut out_rows = vec![0u64; k];
        let mut out_nulls = vec![0i64; k * np.max(1)];
        let mut total = 0u64;
        let rc = unsafe {
            rv_filter_project_batches(self.ctx.raw, cols.as_ptr(), k as u32, ncols as u32, &pred, self.projection.as_ptr(), np as u32,
                                      out.as_mut_ptr(), out_rows.as_mut_ptr(), out_nulls.as_mut_ptr(), &mut total)
        };
        if rc != RV_OK { return Err(StreamError::Execution { message: last_error() }); }   // stream.rs:156-157
        let mut at = 0u64;
        for b in 0..k {   // every input batch yields its output batch, empty ones included (stream.rs:156-158)
            let mut arrays = Vec::with_capacity(np);
            for j in 0..np {
                let mut piece = null_mut();
                let rc = unsafe { rv_slice_known(self.ctx.raw, out[j], at, out_rows[b], out_nulls[b * np + j], &mut piece) };
                if rc != RV_OK { return Err(StreamError::Execution { message: last_error() }); }
                arrays.push(device_array(&self.ctx, piece));   // an ArrayRef backed by the device handle (downloaded on demand)
            }
            self.ready.push_back(RecordBatch::new_unchecked(self.output_schema.clone(), arrays, out_rows[b] as usize));
            at += out_rows[b];
        }
        for h in out { unsafe { rv_free(self.ctx.raw, h); } }   // the slices share the buffers
        Ok(())
    }
}

// ---- round 3 additions (C++ twins compiled and tested: GpuChunkedFilterProjectStream, LimitWindow in rivulus_host.hpp) ----
//
// * `Limit` downstream (src/physical_plan/streaming.rs:246-288).  `trait DataStream` gains one defaulted method,
//       fn limit_hint(&mut self, _rows: usize) {}
//   LimitStream::new calls `input.limit_hint(limit)`, SelectStream forwards it, and the device streams size the window
//   they filter ahead by it: rows still owed / selectivity seen so far (before the first window: `rv_ctx_get_option(ctx,
//   "last_selectivity_ppm")`, else 1/256) * 1.5, at least twice the previous window after a miss.  A hint, not a contract.
//
// * A resident table as the source (DataFrameSource -> Filter -> Select): one `rv_filter_project_chunked` call per window,
//   with the per-batch counts in memory the device can write:
//       let mut counts: *mut c_void = null_mut();
//       rv_host_alloc(ctx.raw, nb * 8, &mut counts);                       // kept across refills
//       rv_filter_project_chunked(ctx.raw, cols.as_ptr(), ncols, batch_size as u64, &pred, proj.as_ptr(), np,
//                                 out.as_mut_ptr(), counts as *mut u64, nb as u64, out_nulls.as_mut_ptr(), &mut total);
//   When `batch_size` is a whole number of the pass's 1024-row wave ranges (the reference's default batch size is), the
//   counts come out of the fused pass itself and are written into `counts` by the device.

// ---- round 5 additions (C++ twin compiled and tested: GpuChunkedFilterProjectStream keeps two windows in flight) ----
//
// * Two windows in flight.  `rv_filter_project_chunked_begin` / `rv_filter_project_batches_begin` queue a window's pass and return an
//   `RvPending`; `rv_filter_project_window_finish` waits, hands over the outputs, the per-batch null counts and the total.  refill()
//   finishes the window begun by the previous refill and at once begins the next one -- before next_batch() hands out this window's
//   batches -- so the device never waits for the host between windows (INTEGRATION.md section 3 has the sketch).  The argument arrays
//   (handles, RvPredicate and its terms, projection, the pinned counts block) live in the operator until finish has consumed the
//   pending handle; two pinned counts blocks are used in turn.  Under a `limit_hint` no window is begun ahead: how far to look follows
//   from what the window in front of it kept.
//
// * The reference's own streaming filter (streaming_planner.rs:137-168: a Boolean column) over windows of 2^24 rows and more takes no
//   chained pass: the same two calls, the per-batch counts come from mask_select_kernel.
