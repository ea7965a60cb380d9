// psdr_variant.hip -- one kernel variant of libpsdr_hip.so: every kernel of psdr_kernels.h instantiated for the
// scene flag set PSDR_VARIANT_FLAGS (bit 0 environment map, bit 1 rough conductor, bit 2 two-level tree, bit 3 no tree at all).  Compiled
// eight times: 0..3, 4, 6 (a two-level tree is never built under an environment map), 8, 10 (tiny scenes without an environment map).
#include "psdr_kernels.h"

#ifndef PSDR_VARIANT_FLAGS
#error "compile with -DPSDR_VARIANT_FLAGS=0|1|2|3|4|6|8|10"
#endif
#define PSDR_CAT2(a, b) a##b
#define PSDR_CAT(a, b) PSDR_CAT2(a, b)

namespace psdr_host {
const VariantOps *PSDR_CAT(variant_ops_, PSDR_VARIANT_FLAGS)() {
    constexpr int FL = PSDR_VARIANT_FLAGS;
    static const VariantOps ops{&render_c_launch<FL>, &render_fwd_launch<FL>, &render_rev<FL>, &guide_launch<FL>};
    return &ops;
}
}  // namespace psdr_host
