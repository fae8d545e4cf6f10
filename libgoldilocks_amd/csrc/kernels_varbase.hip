// kernels_varbase.hip -- the variable-base kernel with digit-addressed tables (public scalars, or a caller who opted
// into GOLDILOCKS_AMD_TABLES_FAST) that is still faster than two ladders: (s1*B, s2*B); body in varbase_bodies.hpp.
#include "varbase_bodies.hpp"

namespace gd {

GD_KERNEL k_point_dual_scalarmul(uint64_t *out1, uint64_t *out2, const uint64_t *base,
                                 const uint64_t *__restrict__ s1, const uint64_t *__restrict__ s2, uint32_t n,
                                 uint4 *__restrict__ workspace) {
    point_dual_scalarmul_body(out1, out2, base, s1, s2, n, workspace);
}

}  // namespace gd
