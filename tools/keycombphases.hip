// keycombphases.hip -- where does k_ed448_verify_keycomb spend its time?
//
// The product kernel's first pass (kernels_verify.hip, eddsa.hpp ed448_verify_keycomb_begin) re-stated phase by phase
// with a clock read (s_memtime) between the phases, then the shared inversion and the second pass as one phase each;
// the same launch shape (256 CUs x 2 blocks x 256 lanes, 8 signatures per lane at 2^20), 2^10 combs of random limbs,
// the signatures already in the order of their keys (position t uses comb (t / 1024) % 1024 -- what the counting sort
// produces).  Inputs are random bytes: every arithmetic phase runs whatever the verdict is.  Printed: each phase's
// share of the lanes' total, next to its multiply-accumulate count, i.e. clocks per MAC by phase (tools/verifyphases
// does the same for k_ed448_verify).
//
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -Ilibgoldilocks_amd/csrc -o tools/keycombphases tools/keycombphases.hip
// Diagnostic builds (round 4, profiles/r04/experiments.md):
//   -DKC_TEETH=8   the 4 x 8 x 14 comb of keys with hundreds of signatures (default 7: 4 x 7 x 16)
//   -DKC_MODE=1    every lane of a wave reads the SAME entry (lane 0's): what the gathers' divergence costs
//   -DKC_MODE=2    the block's key's comb staged in LDS (7 teeth, 52-word stride), re-staged every round
//   -DKC_XCD=1     a block's positions follow its XCD (blocks b, b+8, ... of one XCD take neighbouring positions, so
//                  that one key's comb is fetched into one L2 instead of four)
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "varbase_bodies.hpp"

using namespace gd;

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

#ifndef KC_TEETH
#define KC_TEETH 7
#endif
#ifndef KC_MODE
#define KC_MODE 0
#endif
#ifndef KC_XCD
#define KC_XCD 0
#endif
#if KC_TEETH == 9
using kc_plan = comb_xwide;     // 5 x 9 x 10: what 2^20 signatures of 2^10 keys run (BASELINE config 4)
#elif KC_TEETH == 8
using kc_plan = comb_wide;
#else
using kc_plan = comb_big;
#endif
#ifndef KC_BWT_BITS
#define KC_BWT_BITS 20          // the library's default base table
#endif
static_assert(KC_MODE != 2 || KC_TEETH == 7, "only the 48-KiB comb fits LDS twice per CU");
constexpr int NPH = 7;
static const char *PHASE[NPH] = {"hash + scalar decoding", "the key's comb: doublings + adds", "the base point's adds (one per digit)",
                                 "R's test (L, K, L^2 v == K^2 u)", "park + chain", "the lane's inversion", "second pass"};
static const double MACS[NPH] = {0, (kc_plan::SPACING - 1) * 1312.0 + (kc_plan::SPACING * kc_plan::COMBS - 1) * 1344 + 576, bwt_windows(KC_BWT_BITS) * 1344.0,
                                 3 * 136 + 8 * 192.0 + 16, 192, 63616.0 / 8, 3 * 192};

// KC_MODE 1: lane 0's entry for the whole wave
struct UniformComb {
    using plan = kc_plan;
    const uint4 *p;
    __device__ __forceinline__ niels load(int j, uint32_t idx) const {
        const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        const uint4 *q = p + 12 * (plan::PER_COMB * j + u);
        niels e;
        e.a = fe_load(q);
        e.b = fe_load(q + 4);
        e.cn = fe_load(q + 8);
        return e;
    }
};
// KC_MODE 2: the comb in LDS, one entry every KC_LDS_STRIDE uint4 (13: 52 words, so that the entries spread over all banks)
constexpr int KC_LDS_STRIDE = 13;
struct LdsKeyComb {
    using plan = kc_plan;
    const uint4 *s;
    __device__ __forceinline__ niels load(int j, uint32_t idx) const {
        const uint4 *q = s + KC_LDS_STRIDE * (plan::PER_COMB * j + idx);
        niels e;
        e.a = fe_load(q);
        e.b = fe_load(q + 4);
        e.cn = fe_load(q + 8);
        return e;
    }
};

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

extern "C" __global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_phases(int32_t *__restrict__ status, const uint8_t *__restrict__ sig, const uint8_t *__restrict__ pk,
         const uint8_t *__restrict__ msgs, uint32_t msg_len, uint32_t n, const uint4 *__restrict__ bwt,
         const uint4 *__restrict__ combs, uint4 *__restrict__ park, unsigned long long *__restrict__ totals) {
    __shared__ uint32_t s_bits[VERIFY_LDS_WORDS * BLOCK];
#if KC_MODE == 2
    __shared__ uint4 s_comb[KC_LDS_STRIDE * kc_plan::ENTRIES];
#endif
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> fb{bwt_tab};
    LdsStage stage{s_bits + threadIdx.x};
    LdsMkBitsVerify mkbits{s_bits + threadIdx.x};
    uint64_t acc[NPH];
    for (int k = 0; k < NPH; k++) acc[k] = 0;
    uint64_t t0, t1;
#define MARK(k) t1 = now(); acc[k] += t1 - t0; t0 = t1
    InvChain ch;
    ch.begin();
#if KC_XCD
    // blocks are dealt to the 8 XCDs round robin: block b of XCD b % 8 takes virtual block (b % 8) * (grid / 8) + b / 8
    const uint32_t vblock = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#else
    const uint32_t vblock = blockIdx.x;
#endif
    const uint32_t stride_all = gridDim.x * BLOCK, rounds_all = (n + stride_all - 1) / stride_all;
    for (uint32_t round = 0; round < rounds_all; round++) {
        const uint32_t pos = vblock * BLOCK + threadIdx.x + round * stride_all;
        const bool live = pos < n;
        const uint32_t i = live ? pos : n - 1;
        t0 = now();
        const Ed448Msg m = ed448_challenge_string(sig + 114 * (size_t)i, pk + 57 * (size_t)i, msgs + (size_t)msg_len * i,
                                                  msg_len, 0, nullptr, 0);
        const uint4 *key_comb = combs + (size_t)kc_plan::ENTRIES * 12 * ((i >> 10) & 1023u);
#if KC_MODE == 0
        const GlobalCombOf<kc_plan> comb{key_comb};
#elif KC_MODE == 1
        const UniformComb comb{key_comb};
#else
        __syncthreads();   // (the block's 256 positions are one key's: 1 024 signatures per key, blocks aligned)
        for (uint32_t e = threadIdx.x; e < (uint32_t)kc_plan::ENTRIES; e += BLOCK)
#pragma unroll
            for (int k = 0; k < 12; k++) s_comb[KC_LDS_STRIDE * e + k] = key_comb[12 * e + k];
        __syncthreads();
        const LdsKeyComb comb{s_comb};
#endif
        uint32_t w[29];
        shake256_114(w, m, m.total(), stage);
        const sc challenge = sc_sub(sc_zero(), sc_decode_long_words<114>(w));
        load_bytes_as_words(w, m.a + 57, 57, 15);
        const sc response = sc_decode_long_words<57>(w);
#if KC_MODE == 0      // the product's walk: digits transposed once, the next entry requested an addition ahead
        auto dig = mkbits.template digits<kc_plan>(kc_plan::recode(challenge));
        MARK(0);
        pt P = ladder_comb_digits(dig, comb);
#else
        auto bits = mkbits(kc_plan::recode(challenge), 0);
        MARK(0);
        pt P = ladder_comb(bits, comb);
#endif
        MARK(1);
        fb.add_to(P, response, mkbits);
        MARK(2);
        load_bytes_as_words(w, m.a, 57, 15);
        const uint32_t last = w[14] & 0xff;
        const bool sign = (last & 0x80) != 0;
        bool ok = (last & 0x7f) == 0;
        fe y;
        ok = fe_deserialize_words(y, w) && ok;
        const fe y2 = fe_sqr(y);
        const fe u = fe_weak(fe_sub<2>(fe_one(), y2));
        const fe v = fe_weak(fe_add(fe_one(), fe_mulw(y2, NEG_EDWARDS_D)));
        ok = ok && !fe_is_zero(u) && !fe_is_zero(v);
        const fe w1 = fe_mul(y2, v);
        const fe lf = fe_mul(fe_weak(fe_sub<2>(w1, u)), fe_add(u, w1));
        const fe L = fe_mul(P.x, lf);
        const fe ef = fe_weak(fe_sub<4>(fe_add(v, v), fe_add(u, w1)));
        fe K = fe_mul(P.y, fe_mul(fe_mul(ef, v), y));
        K = fe_weak(fe_add(K, K));
        const bool poly = fe_eq(fe_mul(fe_sqr(L), v), fe_mul(fe_sqr(K), u));
        MARK(3);
        uint4 *slot = park + (size_t)KEYCOMB_SLOT_U4 * i;
        if (live) {
            fe_store(slot + 8, L);
            slot[12] = make_uint4(ok && poly ? 1u : 0u, sign ? 1u : 0u, 0u, 0u);
        }
        ch.push(slot, K, live);
        MARK(4);
    }
    t0 = now();
    ch.invert();
    MARK(5);
    for (uint32_t round = rounds_all; round-- > 0;) {
        const uint32_t i = vblock * BLOCK + threadIdx.x + round * stride_all;
        if (i >= n) continue;
        const uint4 *slot = park + (size_t)KEYCOMB_SLOT_U4 * i;
        const fe inv_k = ch.pop(slot);
        const uint4 flags = slot[12];
        status[i] = flags.x && (fe_lobit(fe_mul(fe_load(slot + 8), inv_k)) == (flags.y != 0)) ? -1 : 0;
    }
    MARK(6);
    for (int k = 0; k < NPH; k++) atomicAdd(totals + k, (unsigned long long)acc[k]);
}

int main() {
    const uint32_t n = 1u << 20, msg_len = 32, nkeys = 1024;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * WAVES_PER_SIMD;
    uint8_t *sig, *pk, *msg;
    int32_t *status;
    uint4 *bwt, *combs, *park;
    unsigned long long *totals;
    const uint32_t bwt_bits = KC_BWT_BITS;   // the table's digits (timing only: random entries behind a valid header)
    const size_t bwt_bytes = ((size_t)bwt_entries(bwt_bits) * 12 + BWT_HEADER_U4) * sizeof(uint4), comb_bytes = (size_t)nkeys * kc_plan::ENTRIES * 12 * sizeof(uint4);
    CHECK(hipMalloc(&sig, 114 * (size_t)n));
    CHECK(hipMalloc(&pk, 57 * (size_t)n));
    CHECK(hipMalloc(&msg, msg_len * (size_t)n));
    CHECK(hipMalloc(&status, 4 * (size_t)n));
    CHECK(hipMalloc(&bwt, bwt_bytes));
    CHECK(hipMalloc(&combs, comb_bytes));
    CHECK(hipMalloc(&park, (size_t)n * KEYCOMB_SLOT_U4 * sizeof(uint4)));
    CHECK(hipMalloc(&totals, NPH * sizeof(unsigned long long)));
    {   // random bytes everywhere (table entries: limbs below 2^28)
        const size_t nb = bwt_bytes > comb_bytes ? bwt_bytes : comb_bytes;
        uint32_t *h = (uint32_t *)malloc(nb);
        uint64_t x = 0x9e3779b97f4a7c15ull;
        for (size_t i = 0; i < nb / 4; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[i] = (uint32_t)x & 0x0fffffffu;
        }
        h[0] = bwt_bits;
        h[1] = bwt_windows(bwt_bits);
        CHECK(hipMemcpy(bwt, h, bwt_bytes, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(combs, h + 12345, comb_bytes - 4 * 12345, hipMemcpyHostToDevice));
        for (size_t i = 0; i < 114 * (size_t)n / 4; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[i] = (uint32_t)x;
        }
        CHECK(hipMemcpy(sig, h, 114 * (size_t)n, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(pk, h + 1000, 57 * (size_t)n, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(msg, h + 5000, msg_len * (size_t)n, hipMemcpyHostToDevice));
        free(h);
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipMemset(totals, 0, NPH * sizeof(unsigned long long)));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_phases, dim3(grid), dim3(BLOCK), 0, 0, status, sig, pk, msg, msg_len, n, bwt, combs, park, totals);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[NPH];
        CHECK(hipMemcpy(t, totals, sizeof(t), hipMemcpyDeviceToHost));
        double sum = 0;
        for (int k = 0; k < NPH; k++) sum += (double)t[k];
        printf("run %d: %.3f ms per 2^20 (instrumented; teeth %d, mode %d, xcd %d)\n", rep, ms, KC_TEETH, KC_MODE, KC_XCD);
        if (rep < 2) continue;
        printf("%-42s %8s %12s %10s %14s\n", "phase", "share", "ms of total", "MACs", "clocks / MAC");
        for (int k = 0; k < NPH; k++) {
            char b[32] = "-";
            if (MACS[k] > 0) snprintf(b, sizeof b, "%.3f", (double)t[k] / n / MACS[k]);
            printf("%-42s %7.2f%% %12.3f %10.0f %14s\n", PHASE[k], 100.0 * t[k] / sum, ms * t[k] / sum, MACS[k], b);
        }
    }
    return 0;
}
