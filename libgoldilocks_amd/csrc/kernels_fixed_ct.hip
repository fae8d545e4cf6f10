// kernels_fixed_ct.hip -- the index-independent variants (comb in LDS + wavefront-shuffle gather) of
// the kernels that multiply the base point by a secret scalar; see fixed_bodies.hpp and
// goldilocks_amd_set_table_access() in include/goldilocks_amd.h.
#include "fixed_bodies.hpp"

namespace gd {

GD_KERNEL k_ed448_derive_public_key_ct(uint8_t *__restrict__ pk, const uint8_t *__restrict__ sk, uint32_t n,
                                       const uint4 *__restrict__ comb, uint4 *__restrict__ workspace) {
    derive_body<true>(pk, sk, n, comb, workspace);
}

GD_KERNEL k_ed448_sign_ct(uint8_t *__restrict__ sig, const uint8_t *__restrict__ sk, const uint8_t *__restrict__ pk,
                          const uint8_t *__restrict__ msgs, const uint64_t *__restrict__ msg_offsets,
                          uint32_t msg_len, uint32_t prehashed, const uint8_t *__restrict__ ctx, uint32_t ctx_len,
                          uint32_t n, const uint4 *__restrict__ comb, uint4 *__restrict__ workspace) {
    sign_body<true>(sig, sk, pk, msgs, msg_offsets, msg_len, prehashed, ctx, ctx_len, n, comb, workspace);
}

GD_KERNEL k_x448_derive_ct(uint8_t *__restrict__ shared, const uint8_t *__restrict__ scalar, uint32_t n,
                           const uint4 *__restrict__ comb, uint4 *__restrict__ workspace) {
    x448_body<true>(shared, nullptr, nullptr, scalar, n, comb, workspace);
}

}  // namespace gd
