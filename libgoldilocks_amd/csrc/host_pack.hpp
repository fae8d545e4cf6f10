// host_pack.hpp -- the host side of the host-array verification / signing entry points: messages given as a table of
// pointers and lengths (the reference's calling convention, one message per goldilocks_ed448_verify call:
// src/public_include/goldilocks/ed448.h:157-165) gathered into one packed buffer + n + 1 offsets, chunk by chunk, by a
// few threads.  Plain C++ (no HIP): goldilocks_amd.hip uses it, and tools/hostfeed.cpp times it without a device --
// what one host can pack and stage per second bounds how many GPUs it can feed.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#ifndef GOLDILOCKS_AMD_MAX_MESSAGE_BYTES
#define GOLDILOCKS_AMD_MAX_MESSAGE_BYTES 0x7fffff00ull   /* include/goldilocks_amd.h */
#endif

namespace gd_host {

// Messages given as host pointer/length tables -> one packed buffer + n+1 offsets.  The offsets are
// built up front; the bytes of lanes [lo, lo+m) are gathered by pack() so that a pipelined caller
// can pack chunk i+1 while the GPU works on chunk i.
struct PackedMessages {
    static constexpr size_t THREADS = 4;   // at most; `threads` of them are used (a sharded call on a host with few cores asks for fewer)
    size_t threads = THREADS;
    std::vector<uint64_t> &off;        // the caller's buffers, kept between calls (the device context's)
    std::vector<uint8_t> &store;
    struct Bytes {                     // (what the unique_ptr this replaced offered)
        uint8_t *p;
        uint8_t *get() const { return p; }
    } bytes{nullptr};
    const uint8_t *const *message;
    const size_t *message_len;
    size_t count = 0;
    bool too_long = false;   // some message is >= GOLDILOCKS_AMD_MAX_MESSAGE_BYTES: the kernels count in 32 bits
    // (construction is free; index() does the work, so that a caller may run it beside something else -- the upload of
    // the public keys, goldilocks_amd.hip verify_pipelined)
    PackedMessages(std::vector<uint64_t> &off_, std::vector<uint8_t> &store_, const uint8_t *const *message_,
                   const size_t *message_len_, size_t n, bool index_now = true, size_t threads_ = THREADS)
        : threads(threads_ < 1 ? 1 : threads_ > THREADS ? THREADS : threads_), off(off_), store(store_), message(message_),
          message_len(message_len_), count(n) {
        if (index_now) index();
    }
    // the n + 1 offsets and room for the packed bytes
    void index() {
        const size_t n = count;
        if (off.size() < n + 1) off.resize(n + 1);
        off[0] = 0;
        if (n < ((size_t)1 << 16)) {
            for (size_t i = 0; i < n; i++) {
                if (message_len[i] >= GOLDILOCKS_AMD_MAX_MESSAGE_BYTES) too_long = true;
                off[i + 1] = off[i] + (too_long ? 0 : message_len[i]);
            }
        } else {   // the offsets of a million messages by four threads: sums of quarters, then the quarters' offsets
            const size_t per = (n + threads - 1) / threads;
            uint64_t sum[THREADS] = {0};
            bool bad[THREADS] = {false};
            const auto range = [&](size_t t, size_t &a, size_t &b) { a = t * per < n ? t * per : n; b = a + per < n ? a + per : n; };
            const auto each = [&](const auto &fn) {
                std::thread th[THREADS];
                for (size_t t = 1; t < threads; t++) th[t] = std::thread(fn, t);
                fn((size_t)0);
                for (size_t t = 1; t < threads; t++) th[t].join();
            };
            each([&](size_t t) {
                size_t a, b;
                range(t, a, b);
                uint64_t s = 0;
                for (size_t i = a; i < b; i++) {
                    if (message_len[i] >= GOLDILOCKS_AMD_MAX_MESSAGE_BYTES) bad[t] = true;
                    s += message_len[i];
                }
                sum[t] = s;
            });
            for (size_t t = 0; t < threads; t++) too_long = too_long || bad[t];
            if (too_long) {
                for (size_t i = 0; i < n; i++) off[i + 1] = 0;
            } else {
                uint64_t start[THREADS];
                uint64_t acc = 0;
                for (size_t t = 0; t < threads; t++) { start[t] = acc; acc += sum[t]; }
                each([&](size_t t) {
                    size_t a, b;
                    range(t, a, b);
                    uint64_t o = start[t];
                    for (size_t i = a; i < b; i++) { o += message_len[i]; off[i + 1] = o; }
                });
            }
        }
        if (store.size() < off[n] + 1) store.resize(off[n] + 1);
        bytes.p = store.data();
    }
    size_t size() const { return off[count]; }
    void pack_range(size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++)
            if (message_len[i]) memcpy(bytes.get() + off[i], message[i], message_len[i]);
    }
    // (a million 32-byte messages behind a pointer table are 10 ms of one core: large ranges go to four threads)
    std::thread workers[THREADS];
    size_t nworkers = 0;
    void pack_start(size_t lo, size_t m) {     // ... which work while the caller uploads what needs no packing
        if (m < ((size_t)1 << 16)) return pack_range(lo, lo + m);
        const size_t per = (m + threads - 1) / threads, hi = lo + m;
        for (size_t t = 0; t < threads; t++) {
            const size_t a = lo + t * per < hi ? lo + t * per : hi, b = a + per < hi ? a + per : hi;
            workers[nworkers++] = std::thread([this, a, b] { pack_range(a, b); });
        }
    }
    void pack_join() {
        for (size_t t = 0; t < nworkers; t++) workers[t].join();
        nworkers = 0;
    }
    void pack(size_t lo, size_t m) {
        pack_start(lo, m);
        pack_join();
    }
    ~PackedMessages() { pack_join(); }
};

}  // namespace gd_host
