#include "fused_table.hpp"
namespace rvk {
// 2..4 eight-byte columns.  The first entry of a (flags, load width) class is the default geometry
// (rows per lane, 16 waves): measured on config 3 (tools/sweep3.py) 12 rows 1.49 ms, 16 rows 1.50, 8 rows 1.58.
const FusedEntry *fused_entries_multi(size_t *n) {
    static const FusedEntry t[] = {
        // predicate column not projected, ... : per-column checks stay in the staging loop
        RV_FUSED(2, 8, 1, 16, 0), RV_FUSED(2, 8, 2, 16, 0), RV_FUSED(2, 8, 1, 16, FF_VALIDITY), RV_FUSED(2, 8, 2, 16, FF_VALIDITY),
        // every loaded column projected
        RV_FUSED(2, 12, 2, 16, FF_PROJALL), RV_FUSED(2, 8, 2, 16, FF_PROJALL), RV_FUSED(2, 8, 1, 16, FF_PROJALL),
        RV_FUSED(2, 12, 2, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(2, 8, 2, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(2, 8, 1, 16, FF_VALIDITY | FF_PROJALL),
        // ... and no null can survive (BASELINE config 3)
        // 8-byte loads first choice here (fused_begin): with them a slot is one word of every bitmap and the whole
        // predicate runs in lane form (fused_kernel.hpp); measured on config 3, 5e8 rows, same box: (16,1,16) 1.435 ms,
        // (12,2,16) 1.562, (16,1,8) 1.497, (16,2,8) 1.629, (24,1,8) 1.627
        RV_FUSED(2, 16, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL), RV_FUSED(2, 12, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(2, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(2, 12, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL), RV_FUSED(2, 16, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(2, 8, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(2, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_STAMP),  // diagnostic (option "stamp")
        // three and four columns.  First match wins among equal flag sets, so the default geometry is listed first; the ones
        // after it are taken by option "rows_per_lane", and the one with the fewest rows per lane (whose LDS slots hold the
        // largest share of a wave's rows) when the last pass's selectivity would crowd the default's slots (fused_begin).
        // Measured on 5e8 rows at 10 %, same box (tools/geometry_ab.py, ms):
        //   x > 899 -> [xn, fn]      (flags VALIDITY, x not projected)   R=8 2.44 | R=12 2.49 | R=4 2.94
        //   x > 899 -> [y, fn, xn]   (four columns, x not projected)     R=8 3.38 | R=4 3.67
        //   x > 899 -> [y, f]        (no bitmap, x not projected)        (8, 16-byte loads) 2.09 | (8, 8-byte) 2.13
        RV_FUSED(3, 8, 1, 16, FF_VALIDITY), RV_FUSED(3, 12, 1, 16, FF_VALIDITY), RV_FUSED(3, 4, 1, 16, FF_VALIDITY),
        RV_FUSED(4, 8, 1, 16, FF_VALIDITY), RV_FUSED(4, 4, 1, 16, FF_VALIDITY),
        RV_FUSED(3, 8, 1, 16, 0), RV_FUSED(3, 8, 2, 16, 0),
        //   x > 899 -> [x, y, f]     (PROJALL)            (8, 8-byte) 2.16 | (8, 16-byte) 2.18 | R=12 2.30
        //   x > 899 -> [x, y, fn]    (VALIDITY | PROJALL) R=12 2.39 | R=16 2.65 | R=8 2.68 (16-byte loads 2.65)
        //   fn > 0.5 AND xn < 200 -> [fn, xn, y] (… | NONULL)  R=8 2.16 | R=12 2.28 | (8, 16-byte) 2.46
        //   x > 899 -> [x, y, f, z]  (four, PROJALL)      R=4 3.02 | (8, 16-byte) 3.08 | R=8 3.13
        //   x > 899 -> [x, y, fn, xn] (four, VALIDITY | PROJALL)  R=8 3.43 | R=4 3.93
        RV_FUSED(3, 8, 2, 16, FF_PROJALL), RV_FUSED(3, 8, 2, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(3, 8, 2, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(3, 8, 1, 16, FF_PROJALL), RV_FUSED(3, 12, 1, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(3, 8, 1, 16, FF_VALIDITY | FF_PROJALL),
        RV_FUSED(3, 8, 1, 16, FF_VALIDITY | FF_PROJALL | FF_NONULL),
        RV_FUSED(3, 4, 1, 16, FF_PROJALL), RV_FUSED(3, 4, 1, 16, FF_VALIDITY | FF_PROJALL),
        RV_FUSED(4, 4, 1, 16, FF_PROJALL), RV_FUSED(4, 8, 1, 16, FF_VALIDITY | FF_PROJALL), RV_FUSED(4, 4, 1, 16, FF_VALIDITY | FF_PROJALL),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
}  // namespace rvk
