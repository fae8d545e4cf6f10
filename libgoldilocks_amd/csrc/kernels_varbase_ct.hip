// kernels_varbase_ct.hip -- the index-independent variants of the variable-base kernels: no table at all, a
// Montgomery ladder of selects (montgomery.hpp); the library's default for every entry point whose scalar may be
// secret.  Bodies in varbase_bodies.hpp.
// The ladder's steps run on gf28s.hpp (signed limbs, pair-wise additions where the type allows them); what is left for
// gf28.hpp here -- the shared inversions and the recovery of the point -- is compiled WITHOUT its pair-wise additions: the
// aligned pairs change the allocation of the 256 registers the step loop lives in, and the kernels measure 0.7 % slower
// with them (same-box A/B, profiles/r05/ab_pairs.txt).
#if !defined(GD_PAIRS_EVERYWHERE)   // (A/B builds: tools/build_variants.py)
#define GD_NO_PAIRED_ADDS 1
#endif
#include "varbase_bodies.hpp"

namespace gd {

// (A/B: GD_TWO_LADDER_WAVES=1 gives the two-ladder and the decode-ladder-encode kernels 512 registers per lane -- their
// spills land in AGPRs instead of scratch -- at one wave per SIMD; round 6 measured it, profiles/r06/ab_launch_bounds.txt)
#ifndef GD_TWO_LADDER_WAVES
#define GD_TWO_LADDER_WAVES WAVES_PER_SIMD
#endif
#define GD_KERNEL_2L extern "C" __global__ void __launch_bounds__(BLOCK, GD_TWO_LADDER_WAVES)

GD_KERNEL k_point_scalarmul_ct(uint64_t *out, const uint64_t *base, const uint64_t *__restrict__ scalar,
                               uint32_t n, uint4 *__restrict__ workspace) {
    point_scalarmul_ladder_body(out, base, scalar, n, workspace);
}

GD_KERNEL_2L k_direct_scalarmul_ct(uint8_t *__restrict__ scaled, int32_t *__restrict__ status,
                                const uint8_t *__restrict__ base, const uint64_t *__restrict__ scalar, uint32_t n,
                                int allow_identity, int short_circuit, const uint64_t *__restrict__ point_base_abi) {
    direct_scalarmul_ladder_body(scaled, status, base, scalar, n, allow_identity, short_circuit, point_base_abi);
}

GD_KERNEL_2L k_point_dual_scalarmul_ct(uint64_t *out1, uint64_t *out2, const uint64_t *base,
                                    const uint64_t *__restrict__ s1, const uint64_t *__restrict__ s2, uint32_t n,
                                    uint4 *__restrict__ workspace) {
    point_dual_scalarmul_ladder_body(out1, out2, base, s1, s2, n, workspace);
}

GD_KERNEL_2L k_double_scalarmul_ct(uint64_t *out, const uint64_t *b1, const uint64_t *__restrict__ s1,
                                const uint64_t *b2, const uint64_t *__restrict__ s2, uint32_t n,
                                uint4 *__restrict__ workspace, const uint64_t *__restrict__ point_base_abi) {
    double_scalarmul_ladder_body(out, b1, s1, b2, s2, n, workspace, point_base_abi);
}

}  // namespace gd
