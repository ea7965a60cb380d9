// psdr_variant.hip -- one kernel variant of libpsdr_hip.so: every kernel of psdr_kernels.h instantiated for the
// scene flag set PSDR_VARIANT_FLAGS (bit 0 environment map, bit 1 rough conductor, bit 2 two-level tree, bit 3 no tree at all).  Compiled
// eight times: 0..3, 4, 6 (a two-level tree is never built under an environment map), 8, 10 (tiny scenes without an environment map).
// Which tree the kernels of this flag set walk (psdr_device.h PSDR_WIDE_TREE; the host makes the same choice per scene, psdr_hip.hip use_wide_tree):
// the 4-wide quantised tree where it measured ahead -- rough-conductor two-level scenes (flag set 6: the 50 k-triangle interior, 5-10 %) --, the
// BVH2 elsewhere (level on the 5 k-triangle bunny in a room, and a kernel that carries BOTH walks loses up to 9 % to register pressure there:
// profiles/r03_bvh4_ab.txt).
#ifndef PSDR_WIDE_TREE
#if PSDR_VARIANT_FLAGS == 6
#define PSDR_WIDE_TREE 1
#else
#define PSDR_WIDE_TREE 0
#endif
#endif
// Leaf triangles fetched two at a time (psdr_device.h leaf_from_memory): same flag set, same reason (registers to spare).
#ifndef PSDR_LEAF_PAIR
#if PSDR_VARIANT_FLAGS == 6
#define PSDR_LEAF_PAIR 1
#else
#define PSDR_LEAF_PAIR 0
#endif
#endif
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
