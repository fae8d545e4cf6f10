// kernels_wave.hip -- one operation per wavefront: the small-batch / single-call kernels (wave_coop.hpp).
#include "kernels.hpp"
#include "wave_coop.hpp"

namespace gd {

// scaled[i] = scalar[i] * base[i], one operation per wave   (ref: goldilocks_448_point_scalarmul)
// out may alias base.  Index-independent table access whatever the table mode (the LDS scan is cheap).
extern "C" __global__ void __launch_bounds__(BLOCK) k_point_scalarmul_wave(uint64_t *out, const uint64_t *base,
                                                                           const uint64_t *__restrict__ scalar,
                                                                           uint32_t n) {
    __shared__ uint32_t s_tab[BLOCK / 64][wc::TABLE_WORDS];
    __shared__ uint32_t s_bits[BLOCK / 64][16];
    const wc::Lane L = wc::make_lane();
    const uint32_t w = threadIdx.x >> 6;
    const uint32_t nwaves = gridDim.x * (BLOCK / 64);
    const wc::WaveTable tab{s_tab[w]};
    for (uint32_t op = blockIdx.x * (BLOCK / 64) + w; op < n; op += nwaves) {   // wave-uniform
        const wc::wfe B = wc::load_point(L, base + 32 * (size_t)op);
        const sc k = sc_load_abi(scalar + 7 * (size_t)op);
        const wc::wfe r = wc::scalarmul(L, tab, s_bits[w], B, k);
        wc::store_point(L, out + 32 * (size_t)op, r);
    }
    // the recoded scalar does not stay behind in LDS (the table holds multiples of the caller's point)
    if ((threadIdx.x & 63u) < 16) s_bits[w][threadIdx.x & 15u] = 0;
}

}  // namespace gd
