// hostfeed.cpp -- how many signatures per second can ONE HOST pack and stage for goldilocks_ed448_verify_batch_ex,
// with no GPU in the loop?
//
// BASELINE config 5 (2^24 verifications over 8 GPUs) through the host-array entry point is
// goldilocks_ed448_verify_batch_ex(..., devices = NULL, device_count = 8): the batch is cut into 8 contiguous shards, one
// host thread per shard, and every shard does what the single-device path does on the host (goldilocks_amd.hip
// ed448_verify_batch_1dev / verify_pipelined): the offsets of its messages (prefix sums, 4 threads), then chunk by chunk
// (2^18 signatures) the messages gathered from the caller's pointer table into one packed buffer (4 threads) while the
// signatures are handed to hipMemcpyAsync, then the packed messages and their offsets.  This program runs exactly that
// host code -- csrc/host_pack.hpp is the library's own -- with the device calls replaced by a stub:
//     --stage none     the copy is dropped (what the host does besides the copies)
//     --stage memcpy   the copy is a memcpy into a per-shard buffer (a pageable hipMemcpyAsync stages through the runtime's
//                      pinned buffers; one memcpy of the same bytes on the calling thread is its host-side cost)
// and prints signatures per second for 1 / 2 / 4 / 8 shards next to the cores this process may use.  One GPU verifies
// 105 - 125 M signatures/s: a host feeds G GPUs end to end only if its rate with G shards is G times that.
//
//   g++ -O2 -std=c++17 -pthread -Ilibgoldilocks_amd/csrc -o tools/hostfeed tools/hostfeed.cpp
//   tools/hostfeed [--log2n 22] [--msg 32] [--stage none|memcpy] [--order sequential|scattered] [--reps 3]
// --order: where the caller's messages lie -- one after the other (bench.py's end_to_end arrays) or scattered over an
// arena (every message a cache miss: the pointer table's worst case).
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "host_pack.hpp"

using gd_host::PackedMessages;

struct Shard {   // what a device context keeps between calls
    std::vector<uint64_t> pack_off;
    std::vector<uint8_t> pack_bytes;
    std::vector<uint8_t> device;   // stand-in for the io buffer on the device
};

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the host side of ed448_verify_batch_1dev for signatures [0, n) of the arrays given
static void shard_body(Shard &sh, const uint8_t *sig, const uint8_t *pk, const uint8_t *const *message, const size_t *message_len,
                       size_t n, bool stage) {
    PackedMessages pm(sh.pack_off, sh.pack_bytes, message, message_len, n);
    const std::vector<uint64_t> &off = pm.off;
    const size_t o_sig = 0, o_pk = 114 * n, o_msg = o_pk + 57 * n, o_off = o_msg + ((pm.size() + 255) & ~(size_t)255),
                 total = o_off + 8 * (n + 1);
    if (stage && sh.device.size() < total) sh.device.resize(total);
    uint8_t *d = sh.device.data();
    const auto h2d = [&](size_t at, const void *src, size_t bytes) {
        if (stage) memcpy(d + at, src, bytes);
    };
    h2d(o_pk, pk, 57 * n);                          // the keys first (verify_pipelined)
    const size_t chunk = (size_t)1 << 18;
    for (size_t lo = 0; lo < n; lo += chunk) {
        const size_t m = n - lo < chunk ? n - lo : chunk;
        pm.pack_start(lo, m);
        h2d(o_sig + 114 * lo, sig + 114 * lo, 114 * m);
        pm.pack_join();
        h2d(o_msg + off[lo], pm.bytes.get() + off[lo], off[lo + m] - off[lo]);
        h2d(o_off + 8 * lo, off.data() + lo, 8 * (m + 1));
    }
}

int main(int argc, char **argv) {
    int log2n = 22, reps = 3;
    size_t msg_len = 32;
    bool stage = true, scattered = false;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--log2n" && i + 1 < argc) log2n = atoi(argv[++i]);
        else if (a == "--msg" && i + 1 < argc) msg_len = (size_t)atol(argv[++i]);
        else if (a == "--reps" && i + 1 < argc) reps = atoi(argv[++i]);
        else if (a == "--stage" && i + 1 < argc) stage = std::string(argv[++i]) == "memcpy";
        else if (a == "--order" && i + 1 < argc) scattered = std::string(argv[++i]) == "scattered";
        else {
            fprintf(stderr, "usage: hostfeed [--log2n N] [--msg BYTES] [--stage none|memcpy] [--order sequential|scattered] [--reps R]\n");
            return 2;
        }
    }
    const size_t n = (size_t)1 << log2n;
    cpu_set_t set;
    CPU_ZERO(&set);
    int usable = (int)std::thread::hardware_concurrency();
    if (sched_getaffinity(0, sizeof(set), &set) == 0) usable = CPU_COUNT(&set);
    printf("host feed rate of goldilocks_ed448_verify_batch_ex, device calls stubbed (--stage %s)\n", stage ? "memcpy" : "none");
    printf("signatures: 2^%d, %zu-byte messages behind a pointer table (%s); cores usable by this process: %d (hardware threads: %u)\n",
           log2n, msg_len, scattered ? "scattered" : "sequential", usable, std::thread::hardware_concurrency());
    // the caller's arrays: signatures, keys, messages scattered in one arena, the pointer and length tables
    std::vector<uint8_t> sig(114 * n), pk(57 * n), arena((msg_len + 8) * n);
    std::vector<const uint8_t *> message(n);
    std::vector<size_t> message_len(n, msg_len);
    uint64_t x = 0x9e3779b97f4a7c15ull;
    for (size_t i = 0; i < sig.size(); i += 8) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        memcpy(&sig[i], &x, sig.size() - i < 8 ? sig.size() - i : 8);
    }
    memset(pk.data(), 0x5a, pk.size());
    memset(arena.data(), 0xa5, arena.size());
    for (size_t i = 0; i < n; i++) message[i] = arena.data() + (msg_len + 8) * (scattered ? (i * 2654435761ull) % n : i);
    printf("%8s %14s %16s %10s\n", "shards", "ms per batch", "signatures/s", "threads");
    for (int G : {1, 2, 4, 8}) {
        std::vector<Shard> shards(G);
        double best = 0;
        for (int r = 0; r < reps + 1; r++) {       // the first pass sizes the shards' buffers (untimed, as a first call would)
            const double t0 = now();
            std::vector<std::thread> th;
            for (int g = 0; g < G; g++)
                th.emplace_back([&, g] {
                    const size_t lo = (size_t)g * n / G, hi = (size_t)(g + 1) * n / G;
                    shard_body(shards[g], sig.data() + 114 * lo, pk.data() + 57 * lo, message.data() + lo, message_len.data() + lo,
                               hi - lo, stage);
                });
            for (std::thread &t : th) t.join();
            const double dt = now() - t0;
            if (r && (best == 0 || dt < best)) best = dt;
        }
        printf("%8d %14.2f %16.3e %10d\n", G, best * 1e3, n / best, G * (1 + (int)PackedMessages::THREADS));
    }
    return 0;
}
