// Small host-side helpers shared by the runtime and the BAM I/O code.
#ifndef BSIG_HOST_UTIL_H
#define BSIG_HOST_UTIL_H
#include <stdint.h>

#include <string>

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

}  // namespace bsig
#endif
