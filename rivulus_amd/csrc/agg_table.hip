#include "fused_table.hpp"
namespace rvk {
const AggEntry *agg_entries(size_t *n) {
    constexpr int F = FF_VALIDITY | FF_BOOL;
    static const AggEntry t[] = {
        RV_AGG(1, 16, 1, 4, 0), RV_AGG(1, 16, 2, 4, 0), RV_AGG(1, 16, 1, 4, F), RV_AGG(1, 16, 2, 4, F),
        RV_AGG(2, 8, 1, 4, F),  RV_AGG(2, 8, 2, 4, F),  RV_AGG(3, 4, 1, 4, F),  RV_AGG(3, 4, 2, 4, F),
        RV_AGG(4, 4, 1, 4, F),  RV_AGG(4, 4, 2, 4, F),
    };
    *n = sizeof(t) / sizeof(t[0]);
    return t;
}
SampleFn sample_kernel(int ncols) {
    static const SampleFn t[] = {&sample_count_kernel<0>, &sample_count_kernel<1>, &sample_count_kernel<2>, &sample_count_kernel<3>, &sample_count_kernel<4>};
    return ncols >= 0 && ncols <= 4 ? t[ncols] : nullptr;
}
}  // namespace rvk
