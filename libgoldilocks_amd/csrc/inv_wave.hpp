// inv_wave.hpp -- ONE field inversion per WAVE for the kernels whose lanes each end a Montgomery chain with one
// (InvChain, fixed_bodies.hpp): key-comb verification's finish, the ladder's u(P), signing, key derivation, X448.
//
// A lane's own inversion is fe_invert: 447 squarings + 14 multiplications of 207 - 278 instructions = 95 K
// instructions -- which the WAVE issues whether one lane needs the result or all 64.  Here the 64 lanes put their
// values into 4 KiB of LDS, the wave multiplies them up in the one-element-per-row arrangement of wave_coop.hpp (four
// running products, one per row: 16 vector multiplications), inverts the four products at once (wc::invert: the same
// addition chain on 82-instruction vector multiplications), walks back (32 vector multiplications) and every lane reads
// its own inverse: (16 + 32 + 463) x 82 = 42 K instructions per wave instead of 95 K.
//
// Every lane of the wave must call (the values of lanes without work are 1).  A zero -- InvChain::push never hands one
// over, but a lane's state may come from memory -- would annihilate the products of its whole row: it is replaced by 1
// on the way in and comes back as 0, which is what fe_invert(0) gives (gf_invert(0) = 0, src/goldilocks.c:69-80).
// In: mag <= 2.  Out: limbs < 2^28 + 8 (mag 1).
#pragma once
#include "wave_coop.hpp"

namespace gd {

constexpr int INV_WAVE_LDS_WORDS = 64 * 16;   // per wave

__device__ __forceinline__ fe wave_shared_invert(const fe &x, uint32_t *lds /* INV_WAVE_LDS_WORDS of this wave */) {
    const uint32_t l = threadIdx.x & 63u;
    const wc::Lane L = wc::make_lane();
    const bool zero = fe_is_zero(x);
    const fe xs = fe_select(x, fe_one(), zero);
    wave_sync();                                   // (whatever the region held before has been read)
#pragma unroll
    for (int i = 0; i < 16; i++) lds[l * 16 + i] = xs.v[i];
    wave_sync();
    // row r takes the elements r, 4 + r, 8 + r, ...: prefix[k] = the product of its elements before the k-th
    wc::wfe prefix[16];
    wc::wfe run = wc::one(L);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const wc::wfe v = lds[(4 * k + L.row) * 16 + L.i];
        prefix[k] = run;
        run = k == 0 ? wc::weak(L, v) : wc::mul(L, run, v);
    }
    wc::wfe inv = wc::invert(L, run);              // four inversions at once
    // inverse of element k = (inverse of the product up to and including it) x (product before it)
#pragma unroll
    for (int k = 15; k >= 0; k--) {
        uint32_t *slot = lds + (4 * k + L.row) * 16 + L.i;
        const wc::wfe v = *slot;
        *slot = k == 0 ? inv : wc::mul(L, inv, prefix[k]);
        if (k) inv = wc::mul(L, inv, v);
    }
    wave_sync();
    fe r;
#pragma unroll
    for (int i = 0; i < 16; i++) r.v[i] = lds[l * 16 + i];
    wave_sync();                                   // (before the caller reuses the region)
    return fe_select(r, fe_zero(), zero);
}

// (One inversion per BLOCK -- the waves' row totals meeting in LDS, wave 0 inverting for all four -- was built and measured in
// round 6: config 4 and signing within the noise, the headline 1.3 % SLOWER (32.0 against 31.6 ms: two more barriers in a
// kernel whose waves otherwise never wait for each other).  Not adopted: profiles/r06/ab_block_invert.txt.)

}  // namespace gd
