// Explicit instantiations of the direct (register-staged) filter + compact kernel for dense selections: one per split of the
// loaded 8-byte columns into predicate columns (NP, two register sets each) and payload columns (NQ, one set each).  Rows per
// lane are what 128 VGPRs hold of those sets and what 72 KiB of LDS hold of a tile (a third stage: one slot per wave); 8 waves,
// two workgroups per CU (out of phase: one's load latency under the other's stores), a tile of 512 R rows.  The first entry of a (NP, NQ, flags) is the default; the others are reachable with
// the options "direct_r" / "direct_waves" (tools/dense_sweep.py).
#include "direct_kernel.hpp"
#include "fused_table.hpp"
namespace rvk {
#define RV_DIRECT(NP, NQ, R, W, F) DirectEntry{NP, NQ, R, W, F, &fused_direct_compact<NP, NQ, R, W, F, 4>}
#define RV_DIRECT3(NP, NQ, R, W) RV_DIRECT(NP, NQ, R, W, 0), RV_DIRECT(NP, NQ, R, W, FF_VALIDITY), RV_DIRECT(NP, NQ, R, W, FF_VALIDITY | FF_BOOL)
const DirectEntry *direct_entries_a(size_t *n) {
    static const DirectEntry t[] = {
        RV_DIRECT3(1, 0, 12, 8), RV_DIRECT3(1, 1, 8, 8), RV_DIRECT3(1, 2, 6, 8), RV_DIRECT3(1, 3, 4, 8),
        RV_DIRECT3(2, 0, 6, 8), RV_DIRECT3(2, 0, 4, 8),
        // wave ranges that tile the 4096-row steps of the kernels compacting String / Boolean columns behind the pass, and the
        // reference's 1024-row batches (FusedParams::wave_offsets, wave_counts)
        RV_DIRECT3(1, 0, 16, 8), RV_DIRECT3(1, 2, 4, 8),
        // a projected column keeps its nulls: validity bytes ride along in the LDS slot (9 bytes per row and column: fewer rows per lane)
        RV_DIRECT(1, 0, 12, 8, FF_VALIDITY | FF_OUTVALID), RV_DIRECT(1, 1, 8, 8, FF_VALIDITY | FF_OUTVALID),
        RV_DIRECT(1, 2, 6, 8, FF_VALIDITY | FF_OUTVALID), RV_DIRECT(1, 2, 4, 8, FF_VALIDITY | FF_OUTVALID),  // 6 rows per lane while the bitmaps fit the LDS
        RV_DIRECT(1, 3, 4, 8, FF_VALIDITY | FF_OUTVALID), RV_DIRECT(2, 0, 6, 8, FF_VALIDITY | FF_OUTVALID), RV_DIRECT(1, 0, 16, 8, FF_VALIDITY | FF_OUTVALID),
        // alternatives (diagnostic)
        RV_DIRECT(1, 0, 8, 8, 0), RV_DIRECT(1, 0, 16, 16, 0), RV_DIRECT(1, 0, 16, 4, 0),
        RV_DIRECT(1, 2, 4, 16, 0), RV_DIRECT(1, 2, 8, 4, 0),
        RV_DIRECT(1, 0, 12, 8, FF_STAMP), RV_DIRECT(1, 2, 6, 8, FF_STAMP),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
