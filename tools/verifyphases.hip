// verifyphases.hip -- where does k_ed448_verify spend its time?  
//
// The product kernel's body (kernels_verify.hip, eddsa.hpp ed448_verify_lattice) for a lane that works for itself
// (no pooled key table: with one, the phases "decode A" and "table A" are not the lane's) re-stated phase by phase
// with a clock read (s_memtime) between the phases; the same launch shape (256 CUs x 2 blocks x 256 lanes,
// grid-stride over 2^20 signatures), the same tables and workspace layout.  Inputs are random bytes: every
// arithmetic phase runs whatever the verdict is (there is no early exit), so timing needs no valid signatures.
// Every lane adds its per-phase cycles into 16 global counters; printed: each phase's share of the lanes'
// total, next to its multiply-accumulate count (hostsim figures), i.e. cycles per MAC by phase.  With two waves
// per SIMD a wave's clock also runs while its partner issues, so the shares are of wall time per wave --
// which is what the kernel's duration is made of.
//
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -Ilibgoldilocks_amd/csrc -o tools/verifyphases tools/verifyphases.hip
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

constexpr int NPH = 10;
static const char *PHASE[NPH] = {"hash + scalar decoding", "short pair (lattice) + tau*S", "decode A", "table A",
                                 "decode R",              "table R",                      "joint ladder (45 windows)",
                                 "two correcting adds",   "28 base-point adds",           "test + store"};
// multiply-accumulates per phase (tests/hostsim counters): decode 64.6 K each, table 26.1 K each, ...
static const double MACS[NPH] = {0, 0, 64728, 24624, 64728, 24624, 225 * 1120.0 + 45 * 192 + 90 * 1728, 2 * 1728, 28 * 1344, 0};

__device__ __forceinline__ uint64_t now() { return __builtin_readcyclecounter(); }

extern "C" __global__ void __launch_bounds__(BLOCK, WAVES_PER_SIMD)
k_phases(int32_t *__restrict__ status, const uint8_t *__restrict__ sig, const uint8_t *__restrict__ pk,
         const uint8_t *__restrict__ msgs, uint32_t msg_len, uint32_t n, uint4 *__restrict__ workspace,
         const uint4 *__restrict__ bwt, unsigned long long *__restrict__ totals, int stagger_units, int stagger_mod) {
    __shared__ uint32_t s_bits[16 * BLOCK];
    __shared__ uint32_t s_stage[34 * BLOCK];
    const uint32_t lane = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t stride = gridDim.x * BLOCK;
    GlobalBwt bwt_tab{bwt};
    FixedBwt<GlobalBwt> fb{bwt_tab};
    LdsStage stage{s_stage + threadIdx.x};
    LdsMkBitsVerify mkbits{s_bits + threadIdx.x};
    __shared__ uint4 s_step[STEP_LDS_U4];
    LdsStepTable<> a_tab{lane_table_at(workspace, 0, 2).p, s_step + threadIdx.x},
                   r_tab{lane_table_at(workspace, 1, 2).p, s_step + threadIdx.x};
    uint64_t acc[NPH];
    for (int k = 0; k < NPH; k++) acc[k] = 0;
    // experiment: blocks start at different times so that their table-building phases (bursts of stores) do not coincide
    if (stagger_units)
        for (int k = 0; k < stagger_units * (int)(blockIdx.x % (unsigned)stagger_mod); k++) __builtin_amdgcn_s_sleep(127);
    for (uint32_t i = lane; i < n; i += stride) {
        uint64_t t0 = now(), t1;
#define MARK(k) t1 = now(); acc[k] += t1 - t0; t0 = t1
        const Ed448Msg m = ed448_challenge_string(sig + 114 * (size_t)i, pk + 57 * (size_t)i, msgs + (size_t)msg_len * i,
                                                  msg_len, 0, nullptr, 0);
        LatticePair pr;
        uint32_t w[29];
        shake256_114(w, m, m.total(), stage);
        const sc h = sc_decode_long_words<114>(w);
        load_bytes_as_words(w, m.a + 57, 57, 15);
        const sc response = sc_decode_long_words<57>(w);
        MARK(0);
        wide15 rho;
        int8w tau;
        half_size_pair(rho, tau, h);
        pr.tau_pos = !is_negative(tau);
        const sc tau_mag = magnitude_as_scalar(tau);
        pr.ts = sc_mul(tau_mag, response);
        wide15 tw;
#pragma unroll
        for (int k = 0; k < 15; k++) tw.w[k] = k < 14 ? tau_mag.w[k] : 0u;
        pr.rho_even = (rho.w[0] & 1u) == 0;
        pr.tau_even = (tw.w[0] & 1u) == 0;
        rho.w[0] |= 1u;
        tw.w[0] |= 1u;
        recode_odd_base(pr.b1, rho);
        recode_odd_base(pr.b2, tw);
        constexpr int TOP = 5 * LATTICE_WINDOWS - 1;
        pr.b1[TOP >> 5] |= 1u << (TOP & 31);
        pr.b2[TOP >> 5] |= 1u << (TOP & 31);
        auto bits1 = mkbits.words(pr.b1, 0);
        auto bits2 = mkbits.words(pr.b2, 1);
        MARK(1);
        bool ok;
        {
            pt A;
            load_bytes_as_words(w, m.b, 57, 15);
            ok = pt_decode_eddsa_words(A, w);
            MARK(2);
            build_window_table(a_tab, A);
            MARK(3);
        }
        {
            pt R;
            load_bytes_as_words(w, m.a, 57, 15);
            ok = pt_decode_eddsa_words(R, w) && ok;
            MARK(4);
            build_window_table(r_tab, pt_negate(R));
            MARK(5);
        }
        pt V = ladder_double_var(bits1, a_tab, pr.tau_pos, bits2, r_tab, LATTICE_WINDOWS);
        MARK(6);
        lattice_subtract_once(V, a_tab, pr.rho_even, pr.tau_pos);
        lattice_subtract_once(V, r_tab, pr.tau_even, false);
        MARK(7);
        fb.add_to(V, pr.ts, mkbits);
        MARK(8);
        status[i] = ok && fe_is_zero(V.x) ? -1 : 0;
        MARK(9);
    }
    for (int k = 0; k < NPH; k++) atomicAdd(totals + k, (unsigned long long)acc[k]);
}

int main(int argc, char **argv) {
    const int stagger_units = argc > 1 ? atoi(argv[1]) : 0, stagger_mod = argc > 2 ? atoi(argv[2]) : 8;
    printf("stagger: block b starts %d x (b mod %d) x 8128 cycles late\n", stagger_units, stagger_mod);
    const uint32_t n = 1u << 20, msg_len = 32;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount * WAVES_PER_SIMD;
    uint8_t *sig, *pk, *msg;
    int32_t *status;
    uint4 *ws, *bwt;
    unsigned long long *totals;
    CHECK(hipMalloc(&sig, 114 * (size_t)n));
    CHECK(hipMalloc(&pk, 57 * (size_t)n));
    CHECK(hipMalloc(&msg, msg_len * (size_t)n));
    CHECK(hipMalloc(&status, 4 * (size_t)n));
    CHECK(hipMalloc(&ws, (size_t)grid * BLOCK * 2 * TABLE_U4 * sizeof(uint4)));
    const uint32_t bwt_bits = 16;   // the table's digits (timing only: random entries behind a valid header)
    const size_t bwt_bytes = ((size_t)bwt_entries(bwt_bits) * 12 + BWT_HEADER_U4) * sizeof(uint4);
    CHECK(hipMalloc(&bwt, bwt_bytes));
    CHECK(hipMalloc(&totals, NPH * sizeof(unsigned long long)));
    // random bytes everywhere (the window table's entries: limbs below 2^28)
    {
        const size_t nb = bwt_bytes;
        uint32_t *h = (uint32_t *)malloc(nb);
        uint64_t x = 0x9e3779b97f4a7c15ull;
        for (size_t i = 0; i < nb / 4; i++) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            h[i] = (uint32_t)x & 0x0fffffffu;
        }
        h[0] = bwt_bits;
        h[1] = bwt_windows(bwt_bits);
        CHECK(hipMemcpy(bwt, h, nb, hipMemcpyHostToDevice));
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
        hipLaunchKernelGGL(k_phases, dim3(grid), dim3(BLOCK), 0, 0, status, sig, pk, msg, msg_len, n, ws, bwt, totals, stagger_units, stagger_mod);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t[NPH];
        CHECK(hipMemcpy(t, totals, sizeof(t), hipMemcpyDeviceToHost));
        double sum = 0;
        for (int k = 0; k < NPH; k++) sum += (double)t[k];
        printf("run %d: %.3f ms per 2^20 (instrumented)\n", rep, ms);
        if (rep < 2) continue;
        printf("%-34s %8s %12s %16s\n", "phase", "share", "ms of total", "clocks / MAC");
        for (int k = 0; k < NPH; k++)
            printf("%-34s %7.2f%% %12.3f %16s\n", PHASE[k], 100.0 * t[k] / sum, ms * t[k] / sum,
                   MACS[k] > 0 ? ([&] { static char b[32]; snprintf(b, sizeof b, "%.3f", (double)t[k] / n / MACS[k]); return b; })() : "-");
    }
    return 0;
}
