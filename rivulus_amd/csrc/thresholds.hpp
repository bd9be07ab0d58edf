// EVERY switch between the kernels / paths of the backend, in one place: the value, what it decides, and the measurement that set it
// (files under profiles/, tools that re-measure it).  The launch code (fused_launch.hip, query.hip, strings.hip) reads these constants
// and nothing else; tests/test_paths_gpu.py asserts the path taken on either side of every row.  Boxes differ by +- 4 % and several
// crossovers were measured inside that noise: the values are the middle of the measured band, not a sharp edge.
#pragma once
#include <cstdint>

namespace rvt {
// ---- which kernel runs the pass (fused_launch.hip, fused_begin) -----------------------------------------------------------------
// A predicate nobody has run over these buffers is SAMPLED before its first launch is sized (1024 blocks of 1024 rows, ~10 us):
// below this many rows a mis-sized pass costs less than the sample.                     profiles/README.md "the first call", tools/dense_one.py
constexpr uint64_t kSampleFromRows = uint64_t{1} << 25;
// The direct (register-staged) kernel instead of the staged pass, from this selectivity on, by what the launch loads / projects:
//                                                              profiles/r04_dense_sweep.txt, r05d_dense_sweep.txt, tools/dense_sweep.py
constexpr double kDirectFromOneColumn = 0.52;            // one loaded column (r05d: 1.80 / 2.01 against the staged 1.77 / 2.03 ms at 30 / 50 %, 2.22 against 3.06 at 70 %)
constexpr double kDirectFromOneProjectedOfSeveral = 0.60;  // several loaded, one projected (the staged slots hold every survivor there)
constexpr double kDirectFromTwoProjected = 0.22;
constexpr double kDirectFromThreeProjected = 0.15;       // three or four
constexpr double kDirectFromTwoProjectedNullable = 0.35;   // columns that keep nulls carry a validity byte through the LDS slot: later
constexpr double kDirectFromThreeProjectedNullable = 0.22; //                                         tools/dense_nullable.py
// ... with ONE loaded column, 16 rows per lane (8192-row tiles) instead of 12 while fewer than this survive: 1e9 rows at 30 / 50 / 70 /
// 84 % kept: 1.80 / 2.01 / 2.30 / 2.43 ms against 2.18 / 2.20 / 2.22 / 2.33      tools/dense_sweep.py, tools/direct_geometry.py, profiles/r05d_dense_sweep.txt, r05d_direct_geometry.txt
constexpr double kDirectTallBelow = 0.62;
// Staged geometries: a wave whose expected survivors x 1.1 + 3 sigma (binomial) pass its LDS slot walks down to geometries whose
// slots hold a larger share of a wave's rows.                                            tools/roomy_ab.py, profiles/README.md
constexpr double kCrowdedMargin = 1.1, kCrowdedSigmas = 3.0;
// Survivors that come in RUNS (sorted / clustered tables): the redo kernel costs ~kRedoMsPerShare ms per 1e9 rows x the share of wave
// ranges it re-reads; the direct kernel costs kDirectPenaltyAt0 - kDirectPenaltySlope x selectivity ms more than the staged pass.  The
// direct kernel runs when the first exceeds the second (never below kDirectPenaltyFloor).   profiles/r05_skew_sweep.txt, tools/skew_sweep.py
// (Both sides got cheaper since -- the redo kernel's whole-line stores: ~2.2-2.8 ms per share; the direct kernel's 16-row tiles: 0.46 /
// 0.27 / 0.02 ms more than the staged pass at 10 / 20 / 30 % -- by about the same factor: the rule still picks the faster launch on
// either side of its crossover at ~15 % kept.                                   profiles/r05d_skew_rule_check.txt, tools/skew_rule_check.py)
constexpr double kRedoMsPerShare = 3.5, kDirectPenaltyAt0 = 0.9, kDirectPenaltySlope = 1.3, kDirectPenaltyFloor = 0.1;
// ... a measured redo share is trusted while the selectivity stayed within this of what it was then, and for slots no roomier than
// this much of a wave's rows beyond the slots it was measured with
constexpr double kRedoMemorySelectivityBand = 0.15, kRedoMemorySlotBand = 0.05;
// Output buffers from the predicate's known selectivity x kOutSizingFactor + kOutSizingSlack of the rows, for tables this big (the
// sample's 1024 blocks are off by 1 % of the rows, one sigma, over runs of 1e5 rows); smaller tables: every row.   tests/test_skew_gpu.py
constexpr uint64_t kOutSizingFromRows = uint64_t{1} << 25;
constexpr double kOutSizingFactor = 1.2, kOutSizingSlack = 0.02;
// A table whose survivors sit in a few LONG STRETCHES (sorted on the predicate's column) is cut at the edges the sample's profile shows
// (its 1024 blocks in table order) and filtered stretch by stretch, each with the kernel its own density asks for (run_segmented_pass):
// a block counts as sparse up to / dense from these shares of its 1024 sampled rows; a plan is at most kStretchesMost stretches of at
// least kStretchLeastBlocks blocks (every stretch costs a launch and a ~30 us read-back: runs of 1e7 rows in 1e9 are 100 stretches), all
// sparse or dense; a selectivity counted by a later pass outranks a profile that promised something else by more than kStretchKnownBand.
//                                                                                       profiles/r05c_skew_sweep.txt, tools/skew_sweep.py
constexpr double kStretchSparseUpTo = 0.30, kStretchDenseFrom = 0.55, kStretchKnownBand = 0.05;
constexpr int kStretchLeastBlocks = 24, kStretchesMost = 4;
// ... for tables this big: a stretch more costs a launch and a read-back, ~55 us per query over one pass -- 4e7 / 1.3e8 / 2.7e8 / 5e8 sorted
// rows at 10 % kept: 0.156 / 0.249 / 0.461 / 0.763 ms against 0.106 / 0.246 / 0.458 / 0.786 as one pass (50 %: 0.170 / 0.344 / 0.604 /
// 1.046 against 0.128 / 0.326 / 0.629 / 1.140)                                             profiles/r05d_stretches_by_rows.txt, tools/stretch_rows.py
constexpr uint64_t kStretchFromRows = uint64_t{1} << 28;

// ---- which columns the pass carries (query.hip, filter_by_groups) ---------------------------------------------------------------
// Columns compacted AFTER the pass at its wave offsets (compact_ranges_kernel) instead of inside it, for tables this big:
constexpr uint64_t kRangesFromRows = uint64_t{1} << 24;
// plain columns the predicate does not read, while the predicate keeps at most this share (10-15 % faster at 10 and 20 % kept, a
// wash from 30 % on; nullable ones always: a pass with output bitmaps is the weakest launch there is)       profiles/r04d_*, tools/wide_ab.py
constexpr double kDeferPlainUpTo = 0.25;
// the column groups beyond the first of a wide projection, while at most this share survives (past it the direct kernel's whole-line
// stores are 5 % ahead for plain columns): rows x 20 <= length x 11                                           tools/wide_ab.py
constexpr uint64_t kRangesSparseNum = 11, kRangesSparseDen = 20;
// RecordBatch::filter by a BooleanArray without a chained pass (mask_select_kernel + scan + compact_ranges_kernel): plain columns of a
// selection denser than this go back to the direct kernel's pass                                              profiles/r04_bool_x_*, r05_batch_bool*
constexpr double kMaskPathPlainUpTo = 0.55;
// ... its outputs sized from the predicate's last selectivity (the compaction queued behind the scan, no host round trip) up to:
constexpr double kMaskPathAssumeUpTo = 0.5;

// ---- windows of RecordBatches (query.hip) ------------------------------------------------------------------------------------------
constexpr uint32_t kSpeculateFromBatches = 4096;   // the query runs on the assumed (regular) window while the walk validates it
constexpr uint32_t kWalkThreadsFromBatches = 16384;  // the handle walk is split over host threads            profiles/r04_batch_sweep.json
constexpr uint32_t kWalkThreads = 8;               // (16 measured slower: profiles/r05_batch_sweep.txt)

// ---- String / Boolean columns behind the pass (strings.hip) -------------------------------------------------------------------------
constexpr double kStrTilesFrom = 0.50;             // source-tile order instead of (start, length) lists            tools/str_sweep.py, profiles/r04c_*
constexpr double kBoolCapFactor = 1.25;            // Boolean outputs sized for the expected survivors x this + kBoolCapSlack rows
constexpr uint64_t kBoolCapSlack = 65536;
}  // namespace rvt
