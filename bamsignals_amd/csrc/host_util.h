// Small host-side helpers shared by the runtime and the BAM I/O code.
#ifndef BSIG_HOST_UTIL_H
#define BSIG_HOST_UTIL_H
#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <string>
#include <utility>
#include <vector>

namespace bsig {

extern thread_local std::string g_last_error;
// records the message for bsig_last_error() and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Exact unsigned division of 0 <= n < 2^31 by d >= 2:  n / d == __umulhi(n, magic) >> shift.
// (round-up method: s = ceil(log2 d), magic = ceil(2^(31+s) / d) < 2^32, shift = s - 1.)
// d == 1 is handled by the callers (no division).
inline void magic_u31(int32_t d, uint32_t *magic, int32_t *shift)
{
    if (d < 2) { *magic = 0; *shift = 0; return; }
    int s = 0;
    while ((1ll << s) < (long long)d) ++s;
    const unsigned __int128 num = (unsigned __int128)1 << (31 + s);
    *magic = (uint32_t)((num + (unsigned)d - 1) / (unsigned)d);
    *shift = s - 1;
}

// order[k] = index of the k-th range in (rid, loc) order, ties in the caller's order (the order the
// reference sorts its ranges in, ref: src/bamsignals.cpp:222-226,246).  Ranges that arrive sorted cost one
// pass; otherwise (key, index) pairs are radix-sorted 16 bits at a time (1M ranges: ~40 ms, a comparator
// sort chasing two arrays: ~150 ms).
inline void sort_ranges(int64_t n, const int32_t *rid, const int32_t *loc, std::vector<int64_t> &order)
{
    order.resize((size_t)n);
    std::iota(order.begin(), order.end(), (int64_t)0);
    bool sorted = true;
    for (int64_t i = 1; i < n && sorted; ++i)
        sorted = rid[i - 1] < rid[i] || (rid[i - 1] == rid[i] && loc[i - 1] <= loc[i]);
    if (sorted) return;
    std::vector<std::pair<uint64_t, int64_t>> keyed((size_t)n);
    for (int64_t i = 0; i < n; ++i)
        keyed[(size_t)i] = {(uint64_t)(uint32_t)rid[i] << 32 | (uint32_t)(loc[i] ^ INT32_MIN), i};
    if (n < 4096) {
        std::sort(keyed.begin(), keyed.end());    // index as tie-break = stable
    } else {
        // LSD radix sort, 16 bits per pass; passes whose digit is constant are skipped
        std::vector<std::pair<uint64_t, int64_t>> tmp2((size_t)n);
        std::vector<uint32_t> hist(65536);
        for (int pass = 0; pass < 4; ++pass) {
            const int sh = 16 * pass;
            std::fill(hist.begin(), hist.end(), 0u);
            for (int64_t i = 0; i < n; ++i) ++hist[(keyed[(size_t)i].first >> sh) & 0xFFFF];
            if (hist[(keyed[0].first >> sh) & 0xFFFF] == (uint32_t)n) continue;
            uint32_t acc = 0;
            for (uint32_t &h : hist) { const uint32_t c = h; h = acc; acc += c; }
            for (int64_t i = 0; i < n; ++i) tmp2[hist[(keyed[(size_t)i].first >> sh) & 0xFFFF]++] = keyed[(size_t)i];
            keyed.swap(tmp2);
        }
    }
    for (int64_t i = 0; i < n; ++i) order[(size_t)i] = keyed[(size_t)i].second;
}

}  // namespace bsig
#endif
