// kernels_varbase.hip -- kernel definitions (see kernels.hpp for the memory plan and policies).
#include "kernels.hpp"

namespace gd {

// config 2: scaled[i] = scalar[i] * base[i]   (ref: goldilocks_448_point_scalarmul)
GD_KERNEL k_point_scalarmul(uint64_t *__restrict__ out, const uint64_t *__restrict__ base,
                            const uint64_t *__restrict__ scalar, uint32_t n, uint4 *__restrict__ workspace) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    __shared__ uint4 s_stage[(BLOCK / 64) * WAVE_STAGE_U4];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t l = threadIdx.x & 63u;
    uint4 *stage = s_stage + (threadIdx.x >> 6) * WAVE_STAGE_U4;
    LaneTable tab{workspace + (size_t)lane * TABLE_U4};
    // wave-uniform loop: the 64 lanes of a wave own 64 consecutive operations per round
    for (uint32_t i0 = lane - l; i0 < n; i0 += stride) {
        const uint32_t m = n - i0 < 64u ? n - i0 : 64u;
        pt b = wave_load_points(stage, base, i0, m, l);
        const sc k = wave_load_scalars(stage, scalar, i0, m, l);
        pt r = b;
        if (l < m) {
            LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(k));
            build_window_table(tab, b);
            r = ladder_varbase(bits, tab);
        }
        wave_store_points(stage, out, i0, m, l, r);
    }
}

// "next" row f2: wire-format scalarmul, 56 bytes in / 56 bytes out   (ref: goldilocks_448_direct_scalarmul)
GD_KERNEL k_direct_scalarmul(uint8_t *__restrict__ scaled, int32_t *__restrict__ status,
                             const uint8_t *__restrict__ base, const uint64_t *__restrict__ scalar, uint32_t n,
                             int allow_identity, int short_circuit, uint4 *__restrict__ workspace,
                             const uint64_t *__restrict__ point_base_abi) {
    __shared__ uint32_t s_bits[15 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    LaneTable tab{workspace + (size_t)lane * TABLE_U4};
    for (uint32_t i = lane; i < n; i += stride) {
        uint32_t w[14];
        const uint32_t *src = reinterpret_cast<const uint32_t *>(base + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) w[k] = src[k];
        pt b;
        bool ok = pt_decode_words(b, w, allow_identity != 0);
        status[i] = ok ? -1 : 0;
        if (!ok && short_circuit) continue;
        if (!ok) b = pt_load_abi(point_base_abi);   // src/goldilocks.c:898: multiply the base point instead
        LdsBits bits = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_load_abi(scalar + 7 * (size_t)i)));
        build_window_table(tab, b);
        pt r = ladder_varbase(bits, tab);
        pt_encode_words(w, r);
        uint32_t *dst = reinterpret_cast<uint32_t *>(scaled + 56 * (size_t)i);
#pragma unroll
        for (int k = 0; k < 14; k++) dst[k] = w[k];
    }
}

// "next" row f4: (s1*B, s2*B) for one base   (ref: goldilocks_448_point_dual_scalarmul)
GD_KERNEL k_point_dual_scalarmul(uint64_t *__restrict__ out1, uint64_t *__restrict__ out2,
                                 const uint64_t *__restrict__ base, const uint64_t *__restrict__ s1,
                                 const uint64_t *__restrict__ s2, uint32_t n, uint4 *__restrict__ workspace) {
    __shared__ uint32_t s_bits[30 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    LaneTable tab{workspace + (size_t)lane * TABLE_U4};
    for (uint32_t i = lane; i < n; i += stride) {
        LdsBits b1 = lds_put_bits(s_bits + threadIdx.x, sc_recode_signed(sc_load_abi(s1 + 7 * (size_t)i)));
        LdsBits b2 = lds_put_bits(s_bits + 15 * BLOCK + threadIdx.x, sc_recode_signed(sc_load_abi(s2 + 7 * (size_t)i)));
        build_window_table(tab, pt_load_abi(base + 32 * (size_t)i));
        pt r1, r2;
        ladder_dual(r1, r2, b1, b2, tab);
        pt_store_abi(out1 + 32 * (size_t)i, r1);
        pt_store_abi(out2 + 32 * (size_t)i, r2);
    }
}

}  // namespace gd
