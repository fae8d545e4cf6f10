// eddsa.hpp -- per-lane Ed448 verification (RFC 8032), restating
//   goldilocks_ed448_verify                  src/eddsa.c:253-306
//   hash_init_with_dom ("SigEd448" dom2)     src/eddsa.c:51-74
//   SHAKE256 (rate 136, pad 0x1f .. 0x80)    src/shake.c:60-162, 211-213
//   scalar_decode_long                       src/scalar.c:257-293
// The reference finishes with the variable-time wNAF double-base multiply
// (src/goldilocks.c:1260-1330); only accept/reject is observable, so the lanes run
// the lane-uniform fixed-window double-base ladder instead (no divergence).
#pragma once
#include "lattice.hpp"
#include "scalarmul.hpp"

namespace gd {

// ------------------------------------------------------------------ Keccak-f[1600]
// State words are only ever indexed with compile-time constants so they stay in VGPRs.

GD_FN uint64_t rotl64(uint64_t x, int s) { return s ? (x << s) | (x >> (64 - s)) : x; }

GD_CONST uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 form of one round on 32-bit halves: the 5-input column parities and chi are three-input
// boolean functions, one v_bitop3_b32 each (truth tables in the a = 0xF0, b = 0xCC, c = 0xAA
// convention: a^b^c = 0x96, a ^ (~b & c) = 0xD2); rotations are v_alignbit_b32 pairs.  About 190
// instructions per round where the generic form below compiles to about 310.
struct k32 {
    uint32_t lo, hi;
};
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
template <int R>
__device__ __forceinline__ k32 rotl_k32(k32 v) {   // rotate the 64-bit value {hi, lo} left by R
    if (R == 0) return v;
    if (R == 32) return k32{v.hi, v.lo};
    if (R < 32) return k32{__builtin_amdgcn_alignbit(v.lo, v.hi, 32 - R), __builtin_amdgcn_alignbit(v.hi, v.lo, 32 - R)};
    return k32{__builtin_amdgcn_alignbit(v.hi, v.lo, 64 - R), __builtin_amdgcn_alignbit(v.lo, v.hi, 64 - R)};
}
template <int X, int Y>
__device__ __forceinline__ void keccak_rho_pi(k32 (&b)[25], const uint64_t (&a)[25], const k32 (&d)[5]) {
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43,
                             25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    const k32 t{(uint32_t)a[X + 5 * Y] ^ d[X].lo, (uint32_t)(a[X + 5 * Y] >> 32) ^ d[X].hi};
    b[Y + 5 * ((2 * X + 3 * Y) % 5)] = rotl_k32<RHO[X + 5 * Y]>(t);
}
template <int X>
__device__ __forceinline__ void keccak_rho_pi_column(k32 (&b)[25], const uint64_t (&a)[25], const k32 (&d)[5]) {
    keccak_rho_pi<X, 0>(b, a, d);
    keccak_rho_pi<X, 1>(b, a, d);
    keccak_rho_pi<X, 2>(b, a, d);
    keccak_rho_pi<X, 3>(b, a, d);
    keccak_rho_pi<X, 4>(b, a, d);
}
GD_FN void keccak_round(uint64_t (&a)[25], uint64_t rc) {
    k32 c[5], d[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        c[x].lo = xor3(xor3((uint32_t)a[x], (uint32_t)a[x + 5], (uint32_t)a[x + 10]), (uint32_t)a[x + 15],
                       (uint32_t)a[x + 20]);
        c[x].hi = xor3(xor3((uint32_t)(a[x] >> 32), (uint32_t)(a[x + 5] >> 32), (uint32_t)(a[x + 10] >> 32)),
                       (uint32_t)(a[x + 15] >> 32), (uint32_t)(a[x + 20] >> 32));
    }
#pragma unroll
    for (int x = 0; x < 5; x++) {
        const k32 r = rotl_k32<1>(c[(x + 1) % 5]);
        d[x].lo = c[(x + 4) % 5].lo ^ r.lo;
        d[x].hi = c[(x + 4) % 5].hi ^ r.hi;
    }
    keccak_rho_pi_column<0>(b, a, d);
    keccak_rho_pi_column<1>(b, a, d);
    keccak_rho_pi_column<2>(b, a, d);
    keccak_rho_pi_column<3>(b, a, d);
    keccak_rho_pi_column<4>(b, a, d);
#pragma unroll
    for (int y = 0; y < 25; y += 5)
#pragma unroll
        for (int x = 0; x < 5; x++) {
            const k32 b0 = b[y + x], b1 = b[y + (x + 1) % 5], b2 = b[y + (x + 2) % 5];
            const uint32_t lo = __builtin_amdgcn_bitop3_b32(b0.lo, b1.lo, b2.lo, 0xd2);
            const uint32_t hi = __builtin_amdgcn_bitop3_b32(b0.hi, b1.hi, b2.hi, 0xd2);
            a[y + x] = (uint64_t)lo | (uint64_t)hi << 32;
        }
    a[0] ^= rc;
}
#else
GD_FN void keccak_round(uint64_t (&a)[25], uint64_t rc) {
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43,
                             25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    uint64_t c[5], b[25];
#pragma unroll
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        uint64_t d = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
#pragma unroll
        for (int y = 0; y < 25; y += 5) a[y + x] ^= d;
    }
#pragma unroll
    for (int x = 0; x < 5; x++)
#pragma unroll
        for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y], RHO[x + 5 * y]);
#pragma unroll
    for (int y = 0; y < 25; y += 5)
#pragma unroll
        for (int x = 0; x < 5; x++) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
    a[0] ^= rc;
}
#endif
GD_FN void keccak_f1600(uint64_t (&a)[25]) {
#pragma unroll 1
    for (int r = 0; r < 24; r++) keccak_round(a, KECCAK_RC[r]);
}

constexpr int SHAKE256_RATE = 136;

// Host-side convenience sponge (tests only; dynamic indexing would spill on the GPU).
#if !defined(__HIPCC__)
struct shake256 {
    uint64_t st[25];
    unsigned pos;
    void init() {
        for (int i = 0; i < 25; i++) st[i] = 0;
        pos = 0;
    }
    void absorb(const uint8_t *in, size_t len) {
        for (size_t i = 0; i < len; i++) {
            st[pos / 8] ^= (uint64_t)in[i] << (8 * (pos % 8));
            if (++pos == SHAKE256_RATE) { keccak_f1600(st); pos = 0; }
        }
    }
    void finish() {
        st[pos / 8] ^= (uint64_t)0x1f << (8 * (pos % 8));
        st[16] ^= 0x8000000000000000ull;
        keccak_f1600(st);
        pos = 0;
    }
    void squeeze(uint8_t *out, size_t len) {
        for (size_t i = 0; i < len; i++) {
            if (pos == SHAKE256_RATE) { keccak_f1600(st); pos = 0; }
            out[i] = (uint8_t)(st[pos / 8] >> (8 * (pos % 8)));
            pos++;
        }
    }
};
#endif

// ------------------------------------------------------------------ challenge hash
// The hashed string is  "SigEd448" | ph | ctxlen | ctx | R(57) | A(57) | msg.
// MSG policy: src.block(w, blk) fills w with bytes 136 blk .. 136 blk + 135 of that virtual string as 34 little-endian
// words, zero beyond its end.  A block is absorbed as 34 such words with compile-time indices, so the Keccak state
// stays in registers and the 34 reads of a block are in flight together (read byte by byte, each read waiting for
// the one before, the two blocks of a verification cost 4 % of the kernel: tools/verifyphases).
// Returns the first 114 output bytes as 29 words (top 2 bytes of word 28 zero).
template <class MSG, class STAGE>
GD_FN void shake256_114(uint32_t out[29], const MSG &src, uint32_t total, STAGE &stage) {
    (void)stage;
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    const uint32_t nblocks = total / SHAKE256_RATE + 1;  // padding always adds >= 1 byte
#pragma unroll 1
    for (uint32_t blk = 0; blk < nblocks; blk++) {
        const uint32_t base = blk * SHAKE256_RATE;
        uint32_t w[SHAKE256_RATE / 4];
        src.block(w, blk);
#pragma unroll
        for (int i = 0; i < SHAKE256_RATE / 4; i++) {
            const uint32_t j = base + 4 * i;
            // SHAKE's domain byte right behind the string (by a mask, not under a branch: 34 branches per block)
            uint32_t here = total - j < 4u ? ~0u : 0u;
#if defined(__HIPCC__)
            asm("" : "+v"(here));
#endif
            w[i] |= (0x1fu << ((8 * (total - j)) & 31u)) & here;
        }
        if (blk == nblocks - 1) w[SHAKE256_RATE / 4 - 1] ^= 0x80000000u;        // the last byte of the last block
#pragma unroll
        for (int k = 0; k < SHAKE256_RATE / 8; k++) st[k] ^= (uint64_t)w[2 * k] | (uint64_t)w[2 * k + 1] << 32;
        keccak_f1600(st);
    }
#pragma unroll
    for (int k = 0; k < 14; k++) {
        out[2 * k] = (uint32_t)st[k];
        out[2 * k + 1] = (uint32_t)(st[k] >> 32);
    }
    out[28] = (uint32_t)st[14] & 0xffffu;
}

// ------------------------------------------------------------------ scalar_decode_long
// NBYTES-byte little-endian integer (packed in words, unused high bytes zero) mod q: the value of the
// reference's scalar_decode_long (src/scalar.c:257-293, which folds 56 bytes at a time with Montgomery products:
// six of them for the 114 bytes of a challenge), computed by folding at the modulus' own size: q = 2^446 - c with
// c of 224 bits, so x = lo + 2^446 hi == lo + c hi (mod q) (sc14.hpp sc_fold).  114 bytes shrink 912 -> 691 -> 470 ->
// 447 bits in three folds (105 + 56 + 7 word products), 57 bytes in one; one conditional subtraction makes the
// result canonical.
template <int NBYTES>
GD_FN sc sc_decode_long_words(const uint32_t *w) {
    static_assert(NBYTES == 57 || NBYTES == 72 || NBYTES == 114, "only the shapes EdDSA needs");
    uint32_t y[15];
    if constexpr (NBYTES == 114) {
        uint32_t x[29], y1[23], y2[16];
#pragma unroll
        for (int i = 0; i < 29; i++) x[i] = w[i];
        sc_fold<29, 15, 23>(y1, x);       // 912 bits -> < 2^691
        sc_fold<23, 8, 16>(y2, y1);       // -> < 2^470
        sc_fold<16, 1, 15>(y, y2);        // -> < 2^446 + 2^248
    } else if constexpr (NBYTES == 72) {
        uint32_t x[18], y1[16];
#pragma unroll
        for (int i = 0; i < 18; i++) x[i] = w[i];
        sc_fold<18, 5, 16>(y1, x);        // 576 bits -> < 2^446 + 2^354
        sc_fold<16, 1, 15>(y, y1);
    } else {
        uint32_t x[15];
#pragma unroll
        for (int i = 0; i < 15; i++) x[i] = w[i];
        sc_fold<15, 1, 15>(y, x);         // 456 bits -> < 2^446 + 2^234
    }
    return sc_final(y);   // y < 2^446 + 2^355 < 2 q: one conditional subtraction
}

#if !defined(__HIPCC__)
static inline sc sc_decode_long_bytes(const uint8_t *in, size_t len) {  // host tests, generic length
    uint32_t w[64] = {0};
    for (size_t i = 0; i < len && i < 256; i++) w[i / 4] |= (uint32_t)in[i] << (8 * (i % 4));
    if (len == 114) return sc_decode_long_words<114>(w);
    if (len == 57) return sc_decode_long_words<57>(w);
    if (len == 72) return sc_decode_long_words<72>(w);
    return sc_zero();
}
#endif

// ------------------------------------------------------------------ hashed strings
// Every string EdDSA hashes is  [dom] | A | B | msg  with
//   dom = "SigEd448" | ph | ctxlen | ctx   (src/eddsa.c:51-74; absent for key expansion)
// verify:     dom | R(57) | pk(57) | msg          sign (challenge): the same
// sign nonce: dom | seed(57) | msg                key expansion:    sk(57)
struct Ed448Msg {
    const uint8_t *a, *b, *msg, *ctx;
    uint32_t alen, blen, msglen, ctxlen;
    uint32_t ph;
    bool dom;
    GD_MFN uint32_t total() const { return (dom ? 10 + ctxlen : 0) + alen + blen + msglen; }
    GD_MFN uint32_t byte(uint32_t j) const {
        if (dom) {
            // "SigEd448" as two little-endian words
            if (j < 8) return ((j < 4 ? 0x45676953u : 0x38343464u) >> (8 * (j & 3))) & 0xffu;
            if (j == 8) return ph;
            if (j == 9) return ctxlen;
            j -= 10;
            if (j < ctxlen) return ctx[j];
            j -= ctxlen;
        }
        if (j < alen) return a[j];
        j -= alen;
        if (j < blen) return b[j];
        return msg[j - blen];
    }
    // bytes j .. j+3 as a little-endian word, zero beyond the end of the string.  Four bytes inside one of
    // the caller's buffers are ONE read (global memory takes unaligned 32-bit reads); only the words that
    // straddle two pieces -- at most four per string -- are put together byte by byte.
    static GD_MFN uint32_t load32(const uint8_t *p) {
        uint32_t v;
        __builtin_memcpy(&v, p, 4);
        return v;
    }
    GD_MFN uint32_t word(uint32_t j) const {
        const uint32_t off_a = dom ? 10 + ctxlen : 0, off_b = off_a + alen, off_m = off_b + blen, end = off_m + msglen;
        if (j >= off_m && j + 4 <= end) return load32(msg + (j - off_m));
        if (j >= off_b && j + 4 <= off_m) return load32(b + (j - off_b));
        if (j >= off_a && j + 4 <= off_b) return load32(a + (j - off_a));
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            if (j + k < end) v |= byte(j + k) << (8 * k);
        return v;
    }
    // The string of a verification or a signature's challenge with an EMPTY context -- "SigEd448" | ph | 0 | R(57) |
    // pk(57) | msg -- has one layout up to byte 124, where the message starts, word-aligned.  word() decides for each
    // of a block's 34 words which piece it lies in, with offsets that are run-time values and lengths that may differ
    // from lane to lane: 7 K instructions, a fifth of them reloads of spilled scalar registers, and 160 waits per
    // signature in k_ed448_verify_keycomb (round 6's ISA count) for what is 31 reads at fixed offsets and a message.
    GD_MFN bool fixed_head() const { return dom && ctxlen == 0 && alen == 57 && blen == 57; }
    // bytes m .. m+3 of the message (m a multiple of 4), zero beyond its end, without a branch: a word that reaches
    // beyond the end is cut out of the message's LAST four bytes; a message shorter than that is `shortw` (its bytes,
    // read one by one by the caller) and the read goes to R instead, whose 57 bytes are always there.
    GD_MFN uint32_t msg_word(uint32_t m, uint32_t shortw) const {
        const bool longm = msglen >= 4;
        const uint32_t last = longm ? msglen - 4 : 0u, at = m < last ? m : last;
        const uint32_t v = load32((longm ? msg : a) + at);
        const uint32_t cut = v >> ((8 * (m - at)) & 31u);
        // (a mask the compiler cannot see through: as a selection it puts each read under a branch of its own and waits
        // for it there, 37 times per signature)
        uint32_t keep = longm && m < msglen ? ~0u : 0u;
#if defined(__HIPCC__)
        asm("" : "+v"(keep));
#endif
        return (cut & keep) | (!longm && m == 0 ? shortw : 0u);
    }
    GD_MFN void block(uint32_t (&w)[34], uint32_t blk) const {
        if (!fixed_head()) {
#pragma unroll
            for (int i = 0; i < 34; i++) w[i] = word(blk * 136 + 4 * i);
            return;
        }
        uint32_t shortw = 0;
        if (msglen < 4) {
#pragma unroll
            for (uint32_t k = 0; k < 3; k++)
                if (k < msglen) shortw |= (uint32_t)msg[k] << (8 * k);
        }
        if (blk == 0) {
            w[0] = 0x45676953u;                                        // "SigE"
            w[1] = 0x38343464u;                                        // "d448"
            w[2] = ph | (uint32_t)a[0] << 16 | (uint32_t)a[1] << 24;   // ph, ctxlen = 0, R[0..1]
#pragma unroll
            for (int k = 0; k < 13; k++) w[3 + k] = load32(a + 2 + 4 * k);           // R[2..53]
            w[16] = load32(a + 53) >> 8 | (uint32_t)b[0] << 24;                      // R[54..56], pk[0]
#pragma unroll
            for (int k = 0; k < 14; k++) w[17 + k] = load32(b + 1 + 4 * k);          // pk[1..56]
#pragma unroll
            for (int k = 0; k < 3; k++) w[31 + k] = msg_word(4 * k, shortw);         // the message starts at byte 124
        } else {
#pragma unroll
            for (int i = 0; i < 34; i++) w[i] = msg_word(blk * 136 + 4 * i - 124, shortw);
        }
    }
};
GD_FN Ed448Msg ed448_challenge_string(const uint8_t *r57, const uint8_t *pk57, const uint8_t *msg, uint32_t msglen,
                                      uint32_t ph, const uint8_t *ctx, uint32_t ctxlen) {
    Ed448Msg m;
    m.a = r57; m.alen = 57; m.b = pk57; m.blen = 57; m.msg = msg; m.msglen = msglen;
    m.ctx = ctx; m.ctxlen = ctxlen; m.ph = ph ? 1u : 0u; m.dom = true;
    return m;
}

// nbytes bytes at p (any alignment) as little-endian words; the unused high bytes of the last word are zero.
// Whole words are ONE read each (global memory takes unaligned 32-bit reads): byte by byte, 57 reads each
// waiting for the one before, a decoding spent a tenth of its time fetching its input (tools/verifyphases).
GD_FN void load_bytes_as_words(uint32_t *w, const uint8_t *p, int nbytes, int nwords) {
    for (int i = 0; i < nwords; i++) {
        uint32_t x = 0;
        if (4 * i + 4 <= nbytes) {
            __builtin_memcpy(&x, p + 4 * i, 4);
        } else {
            for (int b = 0; b < 4; b++)
                if (4 * i + b < nbytes) x |= (uint32_t)p[4 * i + b] << (8 * b);
        }
        w[i] = x;
    }
}

// ------------------------------------------------------------------ verification with half-size scalars
// (lattice.hpp)  Accept iff  V = (|tau| S)*B + rho*PA + |tau|*PR  is the identity (V.x == 0: all of this
// happens in the subgroup of prime order q), with (rho, tau) the short pair of the challenge,
// PA = -+A by the sign of tau and PR = -R: both variable points share one ladder of 45 windows instead of A
// alone taking 90.  Signed odd digits represent odd integers only: an even rho or |tau| is walked as the
// next odd number and one copy of its point is subtracted afterwards.
// Two phases with little live state in common (the split costs nothing and spills less).
struct LatticePair {
    uint32_t b1[15], b2[15];   // signed-window words of rho and |tau| (each made odd) for a 45-window ladder
    sc ts;                     // |tau| * S mod q, the base point's scalar
    bool tau_pos, rho_even, tau_even;
};
// phase 1: challenge hash, the short pair, everything the walk needs except the points
template <class STAGE>
GD_FN LatticePair ed448_verify_lattice_pair(const Ed448Msg &m, STAGE &stage) {
    LatticePair pr;
    uint32_t w[29];
    shake256_114(w, m, m.total(), stage);
    const sc h = sc_decode_long_words<114>(w);                                // the challenge, mod q
    load_bytes_as_words(w, m.a + 57, 57, 15);
    const sc response = sc_decode_long_words<57>(w);                          // S mod q, no range check
    wide15 rho;
    int8w tau;
    half_size_pair(rho, tau, h);
    pr.tau_pos = !is_negative(tau);
    const sc tau_mag = magnitude_as_scalar(tau);
    pr.ts = sc_mul(tau_mag, response);
    wide15 tw;
#pragma unroll
    for (int i = 0; i < 15; i++) tw.w[i] = i < 14 ? tau_mag.w[i] : 0u;
    pr.rho_even = (rho.w[0] & 1u) == 0;
    pr.tau_even = (tw.w[0] & 1u) == 0;
    rho.w[0] |= 1u;                                                           // the next odd number; the walk fixes it up
    tw.w[0] |= 1u;
    recode_odd_base(pr.b1, rho);
    recode_odd_base(pr.b2, tw);
    constexpr int TOP = 5 * LATTICE_WINDOWS - 1;                              // completes the recoding
    pr.b1[TOP >> 5] |= 1u << (TOP & 31);
    pr.b2[TOP >> 5] |= 1u << (TOP & 31);
    return pr;
}
// V - [doit] * (entry 0 of the table)
// V -+ [doit] * (entry 0 of the table): minus, or plus if the table holds the other sign's multiples
template <class AT>
GD_FN void lattice_subtract_once(pt &V, const AT &tab, bool doit, bool flip) {
    pt W = V;
    pt_add_pniels(W, tab.load(0), !flip, true);
    V.x = fe_select(V.x, W.x, doit);
    V.y = fe_select(V.y, W.y, doit);
    V.z = fe_select(V.z, W.z, doit);
    V.t = fe_select(V.t, W.t, doit);
}
// One encoded point (R, or the key) decoded and its window table of odd multiples built; negate: the table of -P.
// A table type may bring an out-of-line overload of this (kernels.hpp does for the verification kernel's tables:
// called once for the key and once for R, the two copies of 16 K instructions become one).
template <class AT>
GD_FN bool decode_into_table(AT &tab, const uint8_t *enc, bool negate) {
    uint32_t w[15];
    pt P;
    load_bytes_as_words(w, enc, 57, 15);
    const bool ok = pt_decode_eddsa_words(P, w);
    build_window_table(tab, negate ? pt_negate(P) : P);
    return ok;
}
// phase 2: decode A and R, walk, add the base point's part, test.
// The key's table holds the multiples of +A whatever the sign of tau (the digits' signs are flipped at the lookups
// instead), so that it can be SHARED: a batch's signatures of one key need its decoding and its table once
// (kernels_verify.hip: k_verify_dedupe, k_verify_key_tables).  shared_key: a_tab already holds A's table and
// key_ok says whether A decoded; otherwise this lane decodes the key and fills a_tab itself.
template <class FB, class AT, class BITS, class MKBITS>
GD_FN bool ed448_verify_lattice_walk(const Ed448Msg &m, const LatticePair &pr, const BITS &bits1, const BITS &bits2,
                                     const FB &fb, AT &a_tab, AT &r_tab, MKBITS &mkbits, bool shared_key, bool key_ok) {
    bool ok = key_ok;
    if (!shared_key) ok = decode_into_table(a_tab, m.b, false);               // the public key
    ok = decode_into_table(r_tab, m.a, true) && ok;                           // PR = -R, R = sig[0:57]
    // PA = -+A by the sign of tau: the table of +A with every digit of rho flipped when tau is positive
    pt V = ladder_double_var(bits1, a_tab, pr.tau_pos, bits2, r_tab, LATTICE_WINDOWS);
    lattice_subtract_once(V, a_tab, pr.rho_even, pr.tau_pos);
    lattice_subtract_once(V, r_tab, pr.tau_even, false);
    fb.add_to(V, pr.ts, mkbits);                                              // + (|tau| S)*B
    return ok && fe_is_zero(V.x);
}
// MKBITS: mkbits.words(w15, slot) turns 15 words into a BITS reader.
template <class FB, class AT, class STAGE, class MKBITS>
GD_FN bool ed448_verify_lattice(const Ed448Msg &m, const FB &fb, AT &a_tab, AT &r_tab, STAGE &stage, MKBITS &mkbits,
                                bool shared_key = false, bool key_ok = true) {
    const LatticePair pr = ed448_verify_lattice_pair(m, stage);
    auto bits1 = mkbits.words(pr.b1, 0);
    auto bits2 = mkbits.words(pr.b2, 1);
    return ed448_verify_lattice_walk(m, pr, bits1, bits2, fb, a_tab, r_tab, mkbits, shared_key, key_ok);
}

// Verification against a key that has a fixed-base comb (kernels_verify.hip: keys that sign many of a batch's
// signatures): src/eddsa.c:253-306 as it stands -- P = (-h)*A + S*B, accept iff P equals the decoded R up to
// 2-torsion (goldilocks_448_point_eq, src/goldilocks.c:644-653) -- with (-h)*A from the key's comb instead of a
// ladder, and WITHOUT decoding R: that exponentiation would be a third of what is left.  With u = 1 - y^2,
// v = 1 - d y^2 (so that x_R^2 = u / v), the reference's test "P == isogeny(R) up to 2-torsion", X_P Y' == Y_P X'
// (src/goldilocks.c:644-653 after :949-1004), multiplied through by v^2 reads
//       L == K * x_R,     L = X_P (y^2 v - u)(u + y^2 v),     K = 2 Y_P (2 v - u - y^2 v) v y.
// So (K != 0, u v != 0):  L^2 v == K^2 u  says that L / K is a square root of u / v (if u / v has none the reference
// rejects R, and the equation cannot hold); and then x_R == L / K iff the low bit of L / K is R's sign bit, because
// the decoder picks the root with that low bit.  The one division left, 1 / K, is an INVERSION, which the signatures
// a lane handles share (Montgomery's trick along the lane, as key derivation and signing share theirs): begin()
// returns what waits for 1 / K, finish() the verdict.  u v == 0 is the reference's isr(0) failure; K == 0 (y = 0,
// Y_P = 0 or 2 - x^2 - y^2 = 0) decodes R after all and compares points (decided = true).  Round 2 verified
// full-length ladders this way (the derivation and its tests: profiles/r02/experiments.md C); fixtures F3 and F7
// (torsion-shifted R, which the reference accepts) and RFC 8032 pass through it on the host checker and on the device.
// COMB: comb.load(j, idx) -> niels, COMB::plan its geometry.  The caller ANDs the key's own decoding in.
struct KeycombPending {
    fe K, L;          // R's x-coordinate is L / K
    bool ok;          // everything else about the signature held (or, decided: the verdict)
    bool sign;        // R's sign bit
    bool decided;     // the slow path ran: ok is final, K is 1
};
// QSRC: S*B may have been computed AHEAD of this call -- it needs nothing but the signature, so a batch's first
// signatures get theirs from a kernel that runs while the keys' combs are still being built (kernels_verify.hip
// k_verify_base_part) -- q.have() says so (uniformly for a wave) and q.load() is S*B as a projective niels.
struct NoParkedBase {
    GD_MFN bool have() const { return false; }
    GD_MFN pniels load() const { return pniels(); }
};
// S*B for that purpose: what fb.add_to would have added (S = sig[57:114] mod q, no range check)
template <class FB, class MKBITS>
GD_FN pniels ed448_verify_base_part(const uint8_t *sig114, const FB &fb, MKBITS &mkbits) {
    uint32_t w[15];
    load_bytes_as_words(w, sig114 + 57, 57, 15);
    return pt_to_pniels(fb.mul_ahead(sc_decode_long_words<57>(w), mkbits));
}
template <class FB, class COMB, class STAGE, class MKBITS, class QSRC>
GD_FN KeycombPending ed448_verify_keycomb_begin(const Ed448Msg &m, const FB &fb, const COMB &comb, STAGE &stage, MKBITS &mkbits,
                                                const QSRC &q) {
    uint32_t w[29];
    shake256_114(w, m, m.total(), stage);
    const sc challenge = sc_sub(sc_zero(), sc_decode_long_words<114>(w));     // -h mod q
    // (public digits, a comb read by the digit: transposed once, the next entry requested an addition ahead)
    auto dig = mkbits.template digits<typename COMB::plan>(COMB::plan::recode(challenge));
    pt P = ladder_comb_digits(dig, comb);                                     // -h*A, T included
    if (q.have()) {
        pt_add_pniels(P, q.load(), false, true);                              // + S*B, computed ahead
    } else {
        load_bytes_as_words(w, m.a + 57, 57, 15);
        const sc response = sc_decode_long_words<57>(w);                      // S mod q, no range check
        fb.add_to(P, response, mkbits);                                       // + S*B
    }
    KeycombPending pend;
    load_bytes_as_words(w, m.a, 57, 15);                                      // R = sig[0:57]: only what the equation needs
    const uint32_t last = w[14] & 0xff;
    pend.sign = (last & 0x80) != 0;
    bool ok = (last & 0x7f) == 0;
    fe y;
    ok = fe_deserialize_words(y, w) && ok;
    const fe y2 = fe_sqr(y);
    const fe u = fe_weak(fe_sub<2>(fe_one(), y2));                            // 1 - y^2
    const fe v = fe_weak(fe_add(fe_one(), fe_mulw(y2, NEG_EDWARDS_D)));       // 1 - d y^2
    ok = ok && !fe_is_zero(u) && !fe_is_zero(v);                              // the reference's isr(0) failure
    const fe w1 = fe_mul(y2, v);                                              // y^2 v
    const fe lf = fe_mul(fe_weak(fe_sub<2>(w1, u)), fe_add(u, w1));           // (y^2 v - u)(u + y^2 v)
    pend.L = fe_mul(P.x, lf);
    const fe ef = fe_weak(fe_sub<4>(fe_add(v, v), fe_add(u, w1)));            // 2 v - u - y^2 v
    fe K = fe_mul(P.y, fe_mul(fe_mul(ef, v), y));
    K = fe_weak(fe_add(K, K));
    const bool poly = fe_eq(fe_mul(fe_sqr(pend.L), v), fe_mul(fe_sqr(K), u)); // L^2 v == K^2 u
    pend.decided = fe_is_zero(K);
    pend.ok = ok && poly;
    if (pend.decided) {   // rare: decode R after all and compare points
        pt R;
        const bool okr = pt_decode_eddsa_words(R, w);
        pend.ok = ok && okr && fe_eq(fe_mul(P.y, R.x), fe_mul(R.y, P.x));
        K = fe_one();
    }
    pend.K = K;
    return pend;
}
template <class FB, class COMB, class STAGE, class MKBITS>
GD_FN KeycombPending ed448_verify_keycomb_begin(const Ed448Msg &m, const FB &fb, const COMB &comb, STAGE &stage, MKBITS &mkbits) {
    return ed448_verify_keycomb_begin(m, fb, comb, stage, mkbits, NoParkedBase());
}
GD_FN bool ed448_verify_keycomb_finish(const KeycombPending &pend, const fe &inv_k) {
    return pend.ok && (pend.decided || fe_lobit(fe_mul(pend.L, inv_k)) == pend.sign);
}

// ------------------------------------------------------------------ key derivation and signing
// ("next" row f1 of SURVEY.md section 8; restates src/eddsa.c:34-48, 98-230)

GD_FN void ed448_clamp_words(uint32_t w[15]) {   // src/eddsa.c:34-48 on 57 bytes
    w[0] &= ~3u;              // byte 0 &= -COFACTOR
    w[13] |= 0x80000000u;     // byte 55 |= 0x80
    w[14] = 0;                // byte 56 = 0
}
GD_FN void store_words_as_bytes(uint8_t *p, const uint32_t *w, int nbytes) {
    for (int i = 0; i < nbytes; i++) p[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
}

// pk = encode_like_eddsa( (clamp(SHAKE256(sk)[0:57]) / 4) * B )      src/eddsa.c:98-147
// Two halves around the one field inversion of the encoding: begin() returns its denominator.
struct Ed448DeriveState {
    fe xn, yn;
};
template <class FB, class STAGE, class MKBITS>
GD_FN fe ed448_derive_begin(Ed448DeriveState &st, const uint8_t *sk57, const FB &fb, STAGE &stage, MKBITS &mkbits) {
    Ed448Msg m;
    m.a = sk57; m.alen = 57; m.b = sk57; m.blen = 0; m.msg = sk57; m.msglen = 0;
    m.ctx = sk57; m.ctxlen = 0; m.ph = 0; m.dom = false;
    uint32_t w[29];
    shake256_114(w, m, 57, stage);
    ed448_clamp_words(w);
    sc secret = sc_decode_long_words<57>(w);
    secret = sc_halve(sc_halve(secret));                       // ENCODE_RATIO = 4
    pt p = fb.mul(secret, mkbits);
    fe zn;
    pt_eddsa_isogeny(st.xn, st.yn, zn, p);
    return zn;
}
GD_FN void ed448_derive_finish(uint8_t *pk57, const Ed448DeriveState &st, const fe &zi) {
    uint32_t e[15];
    eddsa_finish_words(e, st.xn, st.yn, zi);
    store_words_as_bytes(pk57, e, 57);
}
template <class FB, class STAGE, class MKBITS>
GD_FN void ed448_derive_core(uint8_t *pk57, const uint8_t *sk57, const FB &fb, STAGE &stage, MKBITS &mkbits) {
    Ed448DeriveState st;
    fe zn = ed448_derive_begin(st, sk57, fb, stage, mkbits);
    ed448_derive_finish(pk57, st, fe_invert(zn));
}

// RFC 8032 signing (src/eddsa.c:149-230).  scratch: 64 bytes of lane-private memory for the
// hashed-key seed.  sig114 doubles as the place R is read back from for the challenge hash.
// begin(): everything up to the point R = (nonce/4)*B and its isogeny image; returns the denominator
// of R's encoding.  finish(): encode R with the inverse, hash the challenge, write R | S.
struct Ed448SignState {
    fe xn, yn;
    sc nonce, secret;
};
template <class FB, class STAGE, class MKBITS>
GD_FN fe ed448_sign_begin(Ed448SignState &st, const uint8_t *sk57, const uint8_t *msg, uint32_t msglen, uint32_t ph,
                          const uint8_t *ctx, uint32_t ctxlen, uint8_t *scratch, const FB &fb, STAGE &stage,
                          MKBITS &mkbits) {
    Ed448Msg m;
    m.a = sk57; m.alen = 57; m.b = sk57; m.blen = 0; m.msg = sk57; m.msglen = 0;
    m.ctx = ctx; m.ctxlen = 0; m.ph = 0; m.dom = false;
    uint32_t w[29];
    shake256_114(w, m, 57, stage);                             // expanded = secret(57) | seed(57)
    // seed = bytes 57..113: park them in lane-private memory to be hashed next
    for (int i = 0; i < 57; i++) scratch[i] = (uint8_t)(w[(57 + i) >> 2] >> (8 * ((57 + i) & 3)));
    uint32_t sw[15];
#pragma unroll
    for (int i = 0; i < 15; i++) sw[i] = w[i];
    ed448_clamp_words(sw);
    st.secret = sc_decode_long_words<57>(sw);

    m.a = scratch; m.alen = 57; m.blen = 0; m.msg = msg; m.msglen = msglen;
    m.ctx = ctx; m.ctxlen = ctxlen; m.ph = ph ? 1u : 0u; m.dom = true;
    shake256_114(w, m, m.total(), stage);
    st.nonce = sc_decode_long_words<114>(w);
    pt rp = fb.mul(sc_halve(sc_halve(st.nonce)), mkbits);
    fe zn;
    pt_eddsa_isogeny(st.xn, st.yn, zn, rp);
    return zn;
}
template <class STAGE>
GD_FN void ed448_sign_finish(uint8_t *sig114, const Ed448SignState &st, const fe &zi, const uint8_t *pk57,
                             const uint8_t *msg, uint32_t msglen, uint32_t ph, const uint8_t *ctx, uint32_t ctxlen,
                             STAGE &stage) {
    uint32_t e[15];
    eddsa_finish_words(e, st.xn, st.yn, zi);
    store_words_as_bytes(sig114, e, 57);

    uint32_t w[29];
    Ed448Msg c = ed448_challenge_string(sig114, pk57, msg, msglen, ph, ctx, ctxlen);
    shake256_114(w, c, c.total(), stage);
    sc challenge = sc_decode_long_words<114>(w);
    sc resp = sc_add(sc_mul(challenge, st.secret), st.nonce);
    uint32_t rw[15];
#pragma unroll
    for (int i = 0; i < 14; i++) rw[i] = resp.w[i];
    rw[14] = 0;
    store_words_as_bytes(sig114 + 57, rw, 57);
}
template <class FB, class STAGE, class MKBITS>
GD_FN void ed448_sign_core(uint8_t *sig114, const uint8_t *sk57, const uint8_t *pk57, const uint8_t *msg,
                           uint32_t msglen, uint32_t ph, const uint8_t *ctx, uint32_t ctxlen, uint8_t *scratch,
                           const FB &fb, STAGE &stage, MKBITS &mkbits) {
    Ed448SignState st;
    fe zn = ed448_sign_begin(st, sk57, msg, msglen, ph, ctx, ctxlen, scratch, fb, stage, mkbits);
    ed448_sign_finish(sig114, st, fe_invert(zn), pk57, msg, msglen, ph, ctx, ctxlen, stage);
}

#if !defined(__HIPCC__)
struct HostStage {
    uint8_t b[136];
    void put(uint32_t i, uint32_t v) { b[i] = (uint8_t)v; }
    uint64_t get64(int k) const {
        uint64_t x = 0;
        for (int i = 0; i < 8; i++) x |= (uint64_t)b[8 * k + i] << (8 * i);
        return x;
    }
};
struct HostBitsV {
    uint32_t w[15];
    uint32_t word(int k) const { return w[k]; }
};
struct HostMkBits {
    HostBitsV words(const uint32_t (&w)[15], int) const {
        HostBitsV b;
        for (int i = 0; i < 15; i++) b.w[i] = w[i];
        return b;
    }
    HostBitsV operator()(const sc &s, int) const {
        HostBitsV b;
        for (int i = 0; i < 14; i++) b.w[i] = s.w[i];
        b.w[14] = 0;
        return b;
    }
    struct HostDigits {     // scalarmul.hpp comb_digits: two 16-bit digits per word
        uint32_t w[32];
        void put(int k, uint32_t two) { w[k] = two; }
        uint32_t get(int t) const { return (w[t >> 1] >> (16 * (t & 1))) & 0xffffu; }
    };
    template <class PLAN>
    HostDigits digits(const sc &recoded) const {
        HostDigits d;
        comb_digits<PLAN>::store(d, recoded);
        return d;
    }
};
#endif

}  // namespace gd
