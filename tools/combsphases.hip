// combsphases.hip -- where does k_verify_key_combs spend its time?
//
// The per-key comb entries of config 4 (2^10 keys, 5 x 9 x 10: 1 280 entries per key) take 1.06 ms of the step's 7.0 - 7.3
// for 0.12e9 instructions -- 0.19 of the issue rate -- and everything behind them waits.  The product kernel's body
// (kernels_verify.hip) re-stated with a clock read between its phases, alone on the device, at the segment lengths the
// product may pick (8 ... 64 entries per lane), in four orders of memory operations:
//   ORDER 0   round 5's: entry stores, chain push, then the next doubled tooth's load and its addition
//   ORDER 1   the product's: the next doubled tooth requested BEFORE the stores of the entry (memory operations return in
//             order: a load behind 20 scattered stores waits for their acknowledgements)
//   ORDER 3   1 + the entries through LDS (a wave's 64 records moved by coalesced instructions), the chain lane-interleaved,
//             the second pass's requests ahead of the chain's multiplications (docs/history/r06_key_combs_through_lds.hpp):
//             59 spilled registers, and a reload from scratch waits behind the stores like any other load: SLOWER
//   ORDER 2   1 + walking back, a step's entry and the next step's chain slot requested before the chain's two
//             multiplications of the step, which cover their latency; nothing requested behind a store
// Teeth are random limbs: every arithmetic phase runs whatever the points are.
//
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -Ilibgoldilocks_amd/csrc -o tools/combsphases tools/combsphases.hip
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "varbase_bodies.hpp"
#include "../docs/history/r06_key_combs_through_lds.hpp"   // ORDER 3: built, measured, not adopted

using namespace gd;

#define CHECK(x)                                                       \
    do {                                                               \
        hipError_t e = (x);                                            \
        if (e != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));     \
            exit(1);                                                   \
        }                                                              \
    } while (0)

constexpr int NPH = 5;
static const char *PHASE[NPH] = {"first entry: the teeth's signed sum", "Gray walk: store, push, load, add", "the wave's inversion",
                                 "second pass: 3 multiplications per entry", "(whole kernel, per wave)"};

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

template <int ORDER>
__global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void k_combs(uint4 *__restrict__ combs, const uint4 *__restrict__ teeth,
                                                                  uint32_t combed, uint32_t teeth_per, uint32_t SEG,
                                                                  uint4 *__restrict__ chain, unsigned long long *totals) {
    const uint32_t NT = key_comb_combs(teeth_per) * teeth_per, per_comb = 1u << (teeth_per - 1), entries = key_comb_entries(teeth_per);
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t segs = (per_comb + SEG - 1) / SEG, per_key = key_comb_combs(teeth_per) * segs, total = combed * per_key;
    __shared__ uint32_t s_inv[(BLOCK / 64) * INV_WAVE_LDS_WORDS];
    uint32_t *const inv_region = s_inv + (threadIdx.x >> 6) * INV_WAVE_LDS_WORDS;
    uint64_t acc[NPH];
    for (int k = 0; k < NPH; k++) acc[k] = 0;
    uint64_t t0, t1;
    const uint64_t start = now();
#define MARK(k) t1 = now(); acc[k] += t1 - t0; t0 = t1
    for (uint32_t tt = blockIdx.x * BLOCK + threadIdx.x; tt - (threadIdx.x & 63u) < total; tt += stride) {
        const bool live = tt < total;
        const uint32_t t = live ? tt : total - 1;
        const uint32_t k = t / per_key, j = (t % per_key) / segs, g0 = (t % segs) * SEG;
        const uint32_t cnt = per_comb - g0 < SEG ? per_comb - g0 : SEG;   // (the comb's last segment may be shorter)
        const TeethAt tooth{teeth + (size_t)KEY_TEETH_U4 * k}, twice{teeth + (size_t)KEY_TEETH_U4 * k + 16 * NT};
        uint4 *const comb = combs + (size_t)key_comb_u4(teeth_per) * k + 12 * per_comb * j;
        uint4 *const slots = chain + ((size_t)entries * k + per_comb * j) * 8;
        uint32_t idx = g0 ^ (g0 >> 1);
        t0 = now();
        pt p = pniels_to_pt(tooth.load(teeth_per - 1 + teeth_per * j), false);
#pragma unroll 1
        for (uint32_t b = 0; b + 1 < teeth_per; b++)
            pt_add_pniels(p, tooth.load(b + teeth_per * j), ((idx >> b) & 1u) == 0, true);
        MARK(0);
        InvChain ch;
        ch.begin();
        if (ORDER == 0) {
#pragma unroll 1
            for (uint32_t s = 0;; s++) {
                uint4 *q = comb + 12 * idx;
                if (live) {
                    fe_store(q, fe_weak(fe_sub<2>(p.y, p.x)));
                    fe_store(q + 4, fe_weak(fe_add(p.x, p.y)));
                    fe_store(q + 8, fe_mulw(p.t, TWO_EFF_D));
                }
                ch.push(slots + 8 * idx, fe_add(p.z, p.z), live);
                if (s + 1 == cnt) break;
                const uint32_t g = g0 + s + 1, b = (uint32_t)__builtin_ctz(g);
                idx ^= 1u << b;
                pt_add_pniels(p, twice.load(b + teeth_per * j), ((idx >> b) & 1u) == 0, true);
            }
        } else {
#pragma unroll 1
            for (uint32_t s = 0;; s++) {
                uint4 *q = comb + 12 * idx;
                const bool more = s + 1 < cnt;
                const uint32_t g = g0 + s + 1, b = more ? (uint32_t)__builtin_ctz(g) : 0u;
                const pniels e = twice.load(b + teeth_per * j);      // requested ahead of the stores
                gd_keep_order();
                if (live) {
                    fe_store(q, fe_weak(fe_sub<2>(p.y, p.x)));
                    fe_store(q + 4, fe_weak(fe_add(p.x, p.y)));
                    fe_store(q + 8, fe_mulw(p.t, TWO_EFF_D));
                }
                ch.push(slots + 8 * idx, fe_add(p.z, p.z), live);
                if (!more) break;
                idx ^= 1u << b;
                pt_add_pniels(p, e, ((idx >> b) & 1u) == 0, true);
            }
        }
        MARK(1);
        ch.invert_wave(inv_region, false);
        MARK(2);
        if (live) {
            if (ORDER < 2) {
#pragma unroll 1
                for (uint32_t s = cnt; s-- > 0;) {
                    const fe zi = ch.pop(slots + 8 * idx);
                    uint4 *q = comb + 12 * idx;
                    fe_store(q, fe_mul(fe_load(q), zi));
                    fe_store(q + 4, fe_mul(fe_load(q + 4), zi));
                    fe_store(q + 8, fe_mul(fe_load(q + 8), zi));
                    if (s) idx ^= 1u << (uint32_t)__builtin_ctz(g0 + s);
                }
            } else {
                // the chain's two multiplications cover the entry's loads; nothing is requested behind a store
                fe ze = fe_load(slots + 8 * idx), pre = fe_load(slots + 8 * idx + 4);
#pragma unroll 1
                for (uint32_t s = cnt; s-- > 0;) {
                    uint4 *q = comb + 12 * idx;
                    const fe a = fe_load(q), b = fe_load(q + 4), cn = fe_load(q + 8);
                    const uint32_t prev = s ? idx ^ (1u << (uint32_t)__builtin_ctz(g0 + s)) : idx;
                    const fe ze_prev = fe_load(slots + 8 * prev), pre_prev = fe_load(slots + 8 * prev + 4);
                    gd_keep_order();
                    const fe zi = fe_mul(ch.acc, pre);
                    ch.acc = fe_mul(ch.acc, ze);
                    fe_store(q, fe_mul(a, zi));
                    fe_store(q + 4, fe_mul(b, zi));
                    fe_store(q + 8, fe_mul(cn, zi));
                    ze = ze_prev;
                    pre = pre_prev;
                    idx = prev;
                }
            }
        }
        MARK(3);
    }
    acc[4] = now() - start;
    if ((threadIdx.x & 63u) == 0)
        for (int k = 0; k < NPH; k++) atomicAdd(totals + k, (unsigned long long)acc[k]);
}

struct PhaseClock {
    uint64_t acc[NPH], t0;
    __device__ __forceinline__ void start() { t0 = now(); }
    __device__ __forceinline__ void mark(int k) { const uint64_t t1 = now(); acc[k] += t1 - t0; t0 = t1; }
};
__global__ __launch_bounds__(BLOCK, WAVES_PER_SIMD) void k_combs_product(uint4 *__restrict__ combs, const uint4 *__restrict__ teeth,
                                                                          uint32_t combed, uint32_t teeth_per, uint32_t SEG,
                                                                          uint4 *__restrict__ chain, unsigned long long *totals) {
    __shared__ uint4 s_rows[(BLOCK / 64) * WAVE_ENTRIES_LDS_U4];
    PhaseClock clock;
    for (int k = 0; k < NPH; k++) clock.acc[k] = 0;
    const uint64_t start = now();
    verify_key_combs_body(combs, teeth, combed, teeth_per, SEG, chain, s_rows, clock);
    clock.acc[4] = now() - start;
    if ((threadIdx.x & 63u) == 0)
        for (int k = 0; k < NPH; k++) atomicAdd(totals + k, (unsigned long long)clock.acc[k]);
}

template <int ORDER>
static void run(uint4 *combs, const uint4 *teeth, uint4 *chain, unsigned long long *totals, uint32_t keys, uint32_t SEG, int grid_cap) {
    const uint32_t teeth_per = 9, entries = key_comb_entries(teeth_per);
    const uint32_t lanes = keys * key_comb_combs(teeth_per) * ((256 + SEG - 1) / SEG);
    int grid = (int)((lanes + BLOCK - 1) / BLOCK);
    if (grid > grid_cap) grid = grid_cap;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    unsigned long long t[NPH];
    for (int rep = 0; rep < 4; rep++) {
        CHECK(hipMemset(totals, 0, NPH * sizeof(unsigned long long)));
        CHECK(hipEventRecord(e0));
        if (ORDER == 3)
            hipLaunchKernelGGL(k_combs_product, dim3(grid), dim3(BLOCK), 0, 0, combs, teeth, keys, teeth_per, SEG, chain, totals);
        else
            hipLaunchKernelGGL(k_combs<(ORDER < 3 ? ORDER : 0)>, dim3(grid), dim3(BLOCK), 0, 0, combs, teeth, keys, teeth_per, SEG, chain, totals);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CHECK(hipMemcpy(t, totals, sizeof(t), hipMemcpyDeviceToHost));
        }
    }
    const double waves = (double)((lanes + 63) / 64);
    printf("order %d, %2u entries per lane, %6.0f waves in %4d blocks: %.3f ms;  per wave, K cycles:", ORDER, SEG, waves, grid, best);
    const double resident = waves < grid * 4.0 ? waves : grid * 4.0;   // (waves that ran the loop: the clock totals are per resident wave)
    for (int k = 0; k < NPH; k++) printf("  %s%.0f", k == 4 ? "| " : "", (double)t[k] / resident / 1e3);
    printf("\n");
}

int main() {
    const uint32_t keys = 1024, teeth_per = 9, entries = key_comb_entries(teeth_per);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cap = prop.multiProcessorCount * WAVES_PER_SIMD;
    uint4 *teeth, *combs, *chain;
    unsigned long long *totals;
    const size_t teeth_bytes = (size_t)keys * KEY_TEETH_U4 * sizeof(uint4), comb_bytes = (size_t)keys * entries * 12 * sizeof(uint4),
                 chain_bytes = key_comb_chain_u4((size_t)keys * entries) * sizeof(uint4);
    CHECK(hipMalloc(&teeth, teeth_bytes));
    CHECK(hipMalloc(&combs, comb_bytes));
    CHECK(hipMalloc(&chain, chain_bytes));
    CHECK(hipMalloc(&totals, NPH * sizeof(unsigned long long)));
    {
        uint32_t *h = (uint32_t *)malloc(teeth_bytes);
        uint64_t x = 0x9e3779b97f4a7c15ull;
        for (size_t i = 0; i < teeth_bytes / 4; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[i] = (uint32_t)x & 0x0fffffffu;
        }
        CHECK(hipMemcpy(teeth, h, teeth_bytes, hipMemcpyHostToDevice));
        free(h);
    }
    printf("k_verify_key_combs restated, %u keys x %u entries, alone on the device; phases:", keys, entries);
    for (int k = 0; k < NPH; k++) printf("  [%d] %s", k, PHASE[k]);
    printf("\n");
    for (uint32_t SEG : {8u, 11u, 13u, 16u, 22u, 32u, 64u}) {
        run<0>(combs, teeth, chain, totals, keys, SEG, cap);
        run<1>(combs, teeth, chain, totals, keys, SEG, cap);
        run<3>(combs, teeth, chain, totals, keys, SEG, cap);
    }
    // one block per CU (what is left beside k_verify_base_part's persistent blocks)
    printf("at one block per CU:\n");
    for (uint32_t SEG : {11u, 16u, 22u, 32u}) {
        run<1>(combs, teeth, chain, totals, keys, SEG, cap / 2);
        run<3>(combs, teeth, chain, totals, keys, SEG, cap / 2);
    }
    return 0;
}
