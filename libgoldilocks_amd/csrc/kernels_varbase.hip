// kernels_varbase.hip -- the variable-base kernels with digit-addressed tables (public scalars, or a
// caller who opted into GOLDILOCKS_AMD_TABLES_FAST); bodies in varbase_bodies.hpp.
#include "varbase_bodies.hpp"

namespace gd {

GD_KERNEL k_point_scalarmul(uint64_t *out, const uint64_t *base, const uint64_t *__restrict__ scalar, uint32_t n,
                            uint4 *__restrict__ workspace) {
    point_scalarmul_body(out, base, scalar, n, workspace);
}

GD_KERNEL k_direct_scalarmul(uint8_t *__restrict__ scaled, int32_t *__restrict__ status,
                             const uint8_t *__restrict__ base, const uint64_t *__restrict__ scalar, uint32_t n,
                             int allow_identity, int short_circuit, uint4 *__restrict__ workspace,
                             const uint64_t *__restrict__ point_base_abi) {
    direct_scalarmul_body(scaled, status, base, scalar, n, allow_identity, short_circuit, workspace,
                                 point_base_abi);
}

GD_KERNEL k_point_dual_scalarmul(uint64_t *out1, uint64_t *out2, const uint64_t *base,
                                 const uint64_t *__restrict__ s1, const uint64_t *__restrict__ s2, uint32_t n,
                                 uint4 *__restrict__ workspace) {
    point_dual_scalarmul_body(out1, out2, base, s1, s2, n, workspace);
}

}  // namespace gd
