// Explicit instantiations of the direct (unstaged) filter + compact kernel for dense selections, one per column count.  A tile is
// 512 R rows = 48 to 64 KiB of input (with 256 R the launch ran at the pace of the prefix chain: 76 tiles per microsecond).
#include "direct_kernel.hpp"
#include "fused_table.hpp"
namespace rvk {
const FusedEntry *direct_entry(int ncols) {
    static const FusedEntry t[] = {
        FusedEntry{1, 16, 1, 8, FF_DIRECT, &fused_direct_compact<1, 16>}, FusedEntry{2, 8, 1, 8, FF_DIRECT, &fused_direct_compact<2, 8>},
        FusedEntry{3, 4, 1, 8, FF_DIRECT, &fused_direct_compact<3, 4>},   FusedEntry{4, 4, 1, 8, FF_DIRECT, &fused_direct_compact<4, 4>},
    };
    return ncols >= 1 && ncols <= 4 ? &t[ncols - 1] : nullptr;
}
}  // namespace rvk
