// Raw DEFLATE (RFC 1951) decoder for ONE work-item: every active lane of a wave inflates its own
// BGZF block (a BAM holds tens of thousands of independent blocks of <= 64 KiB: that, not the bits
// inside a block, is where the parallelism is).  Written for lanes that march together:
//
//   * Huffman decoding: an 8-bit (literal/length, 16-bit entries) and a 6-bit (distance, byte
//     entries) first-level table answer nearly every symbol with one lookup; longer codes take
//     the canonical walk over the code lengths (per length the NUMBER of codes, packed two per
//     register, and the symbols sorted by (length, value)) from the first length the table does not
//     cover, as a chain of compares WITHOUT branches, then one lookup.  704 bytes of LDS per lane on
//     the device (6 bits of distance table cost no resident lane and save 2 % on literal-heavy blocks),
//     where zlib-style two-level tables need 5.7 KB;
//   * a block's work is split over TWO lanes in two waves (round 5, see produce / consume below): one decodes
//     symbols into tokens, the other turns tokens into bytes; a 12-byte mailbox in LDS lies between them;
//   * the consumer waits for memory ONCE per turn (see BSIG_VM_DRAIN in consume): the compiler's waits are
//     all-or-nothing (s_waitcnt vmcnt(0)), so every load whose value is needed at once stalls the wave for a whole
//     trip and drains what else is in flight;
//   * per turn one token (up to six literals the first-level table knows and/or the match behind them) OR a slice
//     of a pending match: a lane that copies a 258-byte match does not hold the other lanes of its wave for 258 turns;
//   * the LZ77 window is the output itself (global memory): matches read back what the lane wrote.
//
// The same source compiles for the host (tests/test_inflate_lane.py runs it against zlib) and for
// gfx950 (devdecode.hip: k_inflate).
#ifndef BSIG_INFLATE_LANE_H
#define BSIG_INFLATE_LANE_H
#include <stdint.h>
#include <string.h>
#if !defined(__HIP_DEVICE_COMPILE__)
#include <vector>
#endif

#if defined(__HIPCC__)
#define BSIG_HD __host__ __device__ __forceinline__
#else
#define BSIG_HD inline
#endif

namespace bsig_inflate {

#ifndef BSIG_LFAST
#define BSIG_LFAST 8
#endif
#ifndef BSIG_DFAST
#define BSIG_DFAST 6
#endif
constexpr int kLFast = BSIG_LFAST, kDFast = BSIG_DFAST;   // first-level table bits (literal/length: 16-bit entries; distance: 8-bit, 6 or 7 bits)
#ifndef BSIG_MULTI_LIT
#define BSIG_MULTI_LIT 1
#endif
#ifndef BSIG_LIT_LOOP_BRANCHES
#define BSIG_LIT_LOOP_BRANCHES 0
#endif
constexpr bool kMultiLit = BSIG_MULTI_LIT != 0;
constexpr uint32_t kTurn = 64;     // bytes of a pending match copied per turn of the main loop
// (diagnostic builds only, scripts/inflate_bench.py: what a turn costs WITHOUT part of its memory traffic -- the
// control flow of the decode does not depend on the output, so the kernel runs the same turns and writes garbage:
// 1 = no match loads, 2 = neither match loads nor match stores, 3 = no output traffic at all, 4 = everything but the
// literals' store, 5 = the literals' store alone, 6 / 7 = see consume)
#ifndef BSIG_ABLATE
#define BSIG_ABLATE 0
#endif
#define BSIG_ABL_NO_MLOAD (BSIG_ABLATE == 1 || BSIG_ABLATE == 2 || BSIG_ABLATE == 3 || BSIG_ABLATE == 5)
#define BSIG_ABL_NO_MSTORE (BSIG_ABLATE == 2 || BSIG_ABLATE == 3 || BSIG_ABLATE == 5)
#define BSIG_ABL_NO_LSTORE (BSIG_ABLATE == 3 || BSIG_ABLATE == 4)

// per-lane working storage that nearly every symbol touches (LDS on the device): 704 bytes = 224 lanes per CU
// (160 with the cold tables in LDS as well).  More would not help a big file: a lane's turn takes 6,000-7,000
// cycles however many waves share its SIMD, but each resident lane also keeps three or four 128-byte lines live in
// L2 (the output line it is writing, its input line, the line its matches copy from) and 4 MB per XCD is 32,768
// lines: from ~160 lanes per CU on the L2 evicts half-written lines (r03 counters at 285 lanes per CU: TCC busy all
// the time, one HBM write per two stores), and 12-bit first-level entries + 64 hot symbols (480 bytes, 320 lanes)
// decoded the north star's file no faster and a small file 7 % slower.
#ifndef BSIG_HOT_SYMS
#define BSIG_HOT_SYMS 128
#endif
constexpr int kHotSyms = BSIG_HOT_SYMS;      // (a multiple of 64) literal/length symbols with codes longer than the first-level table kept in LDS
struct alignas(4) LaneTables {
    uint16_t lfast[1 << kLFast];   // literal/length: (symbol << 4) | code length, 0 = longer code
    uint8_t dfast[1 << kDFast];    // distance: (symbol << 3) | code length (<= kDFast), 0 = longer code
    // the first kHotSyms sorted symbols whose codes are LONGER than the first-level table (the shortest of the
    // long codes, i.e. the most frequent of the rare symbols): what the walk looks up.  With all sorted symbols
    // in global memory nearly every turn of a wave waited for one such load (32 lanes x up to six literals: some
    // lane meets a long code), and a load that must be waited for drains the deferred match stores with it.
    uint8_t lsym_hot[kHotSyms];
};
// ... and what only the table construction and the walk for codes longer than the first-level tables
// touch: it lives behind the code-length scratch (global memory on the device)
struct ColdTables {
    uint32_t lhi[9];               // bit 8 of the literal/length symbols below
    uint8_t lsym[288];             // literal/length symbols sorted by (code length, symbol), low 8 bits
};

// keeps the compiler from folding the two homes of a sorted symbol (LDS, global memory) into ONE load through a
// generic pointer: such a load waits for LDS and memory alike, and with it for every store in flight
#if defined(__HIP_DEVICE_COMPILE__)
#define BSIG_KEEP_APART() asm volatile("" ::: "memory")
#define BSIG_ALL_LANES(c) __all(c)          // true for every lane of the wave that is here
#define BSIG_ANY_LANES(c) __any(c)
#else
#define BSIG_KEEP_APART() do { } while (0)
#define BSIG_ALL_LANES(c) (c)
#define BSIG_ANY_LANES(c) (c)
#endif

// the sorted symbols of the literal/length code: the low 8 bits as bytes, bit 8 as a bit
struct LSyms {
    uint8_t *lo;            // ColdTables::lsym / lhi: every sorted position outside the hot window
    uint32_t *hi;
    uint8_t *hot;           // LaneTables::lsym_hot: sorted positions [base, base + kHotSyms)
    uint64_t hot_hi[kHotSyms / 64];     // their bit 8 (registers)
    int base;               // first sorted position whose code is longer than the first-level table
    BSIG_HD void clear()
    {
        for (int k = 0; k < 9; ++k) hi[k] = 0;
        for (int k = 0; k < kHotSyms / 64; ++k) hot_hi[k] = 0;
    }
    BSIG_HD void begin_long(int first_long_idx) { base = first_long_idx; }
    BSIG_HD void put(int idx, int s)
    {
        if (idx < base) return;             // a code the first-level table holds: nobody looks its symbol up here
        const unsigned h = (unsigned)(idx - base);
        if (h < (unsigned)kHotSyms) {
            hot[h] = (uint8_t)s;
            const uint64_t bit = (uint64_t)(s >> 8) << (h & 63);
            for (int k = 0; k < kHotSyms / 64; ++k) hot_hi[k] |= (int)(h >> 6) == k ? bit : 0ull;
            return;
        }
        lo[idx] = (uint8_t)s;
        if (s >> 8) hi[idx >> 5] |= 1u << (idx & 31);
    }
    BSIG_HD int get(int idx) const
    {
        const unsigned h = (unsigned)(idx - base);
        int r;
        if (h < (unsigned)kHotSyms) {
            uint64_t w = hot_hi[0];
            for (int k = 1; k < kHotSyms / 64; ++k) w = (int)(h >> 6) == k ? hot_hi[k] : w;
            r = (int)hot[h] | (int)((w >> (h & 63)) & 1ull) << 8;
        } else {
            BSIG_KEEP_APART();
            r = (int)lo[idx] | (int)(((hi[idx >> 5] >> (idx & 31)) & 1u) << 8);
            BSIG_KEEP_APART();
        }
        return r;
    }
};
// the sorted symbols of the distance code (30) and of the code-length code (19): 5 bits each, twelve per
// 64-bit word -- registers, so a distance code longer than its first-level table costs no trip to memory
struct DSyms {
    uint64_t w[3];
    BSIG_HD void clear() { w[0] = w[1] = w[2] = 0; }
    BSIG_HD void begin_long(int) {}
    BSIG_HD void put(int idx, int s)
    {
        const int q = idx >= 24 ? 2 : idx >= 12 ? 1 : 0;
        const uint64_t v = (uint64_t)(s & 31) << (5 * (idx - 12 * q));
        w[0] |= q == 0 ? v : 0ull;
        w[1] |= q == 1 ? v : 0ull;
        w[2] |= q == 2 ? v : 0ull;
    }
    BSIG_HD int get(int idx) const
    {
        const int q = idx >= 24 ? 2 : idx >= 12 ? 1 : 0;
        const uint64_t x = q == 0 ? w[0] : q == 1 ? w[1] : w[2];
        return (int)((x >> (5 * (idx - 12 * q))) & 31ull);
    }
};

// number of codes of every length 1..15, two 16-bit counts per word (index len >> 1)
struct Counts {
    uint32_t w[8];
};

struct BitIn {
    const uint8_t *p;      // next input byte that is not in buf yet
    const uint8_t *end;    // end of this block's deflate data
    uint64_t buf;          // bit buffer, next bit = bit 0
    uint64_t ahead;        // the 8 bytes at p, loaded a turn before they are needed
    uint64_t amask;        // ... and which of them lie inside the input
    int cnt;               // valid bits in buf
};

enum { OK = 0, ERR_INPUT = 1, ERR_CODE = 2, ERR_OUTPUT = 3, ERR_DIST = 4, ERR_STORED = 5, ERR_TABLE = 6 };

BSIG_HD uint32_t load32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
BSIG_HD uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
BSIG_HD void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }
struct W16 {
    uint64_t a, b;
};
BSIG_HD W16 load128(const uint8_t *p) { W16 v; memcpy(&v, p, 16); return v; }
BSIG_HD void store128(uint8_t *p, const W16 &v) { memcpy(p, &v, 16); }

// the 8 input bytes at p; zeros behind the end of the input (a well-formed stream never consumes
// them, a damaged one runs into ERR_INPUT)
BSIG_HD uint64_t peek64(const uint8_t *p, const uint8_t *end)
{
    if (p + 8 <= end) return load64(p);
    uint64_t v = 0;
    for (int k = 0; k < 8; ++k)
        if (p + k < end) v |= (uint64_t)p[k] << (8 * k);
    return v;
}

// requests the 8 bytes at in.p for the NEXT refill.
// On the device this is ONE unconditional load whose result nobody touches before that refill: the address is
// clamped to `end` (the 8 bytes there are readable -- a BGZF block's trailer follows its deflate data, see
// produce) and the bytes behind the input are masked off when they are taken.  With peek64's two paths
// (a whole word, or the last bytes one by one) the value went through a register copy where the paths join,
// and the compiler waits for a load before it copies it: every refill waited for the load it had just issued
// (r03: 9,500 cycles per turn of a wave, profiles/r03_inflate_prof.txt).
BSIG_HD void request(BitIn &in)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint8_t *q = in.p < in.end ? in.p : in.end;
    const uint32_t avail = (uint32_t)(in.end - q);
    in.amask = avail >= 8u ? ~0ull : (1ull << (8u * avail)) - 1ull;
    in.ahead = load64(q);
#else
    in.amask = ~0ull;
    in.ahead = peek64(in.p, in.end);
#endif
}

// at least 56 valid bits afterwards.  The bytes come from `ahead`, which was loaded when the
// previous refill finished: the load of the next 8 bytes is issued here and has a whole turn of
// the main loop to arrive.
BSIG_HD void refill(BitIn &in)
{
    const int nb = (64 - in.cnt) >> 3;           // whole bytes that fit (0: the same bytes are requested again)
    const uint64_t take_bits = in.ahead & in.amask & (nb >= 8 ? ~0ull : (1ull << (8 * nb)) - 1ull);
    in.buf |= nb > 0 ? take_bits << in.cnt : 0ull;
    in.cnt += 8 * nb;
    in.p += nb;
    request(in);
}

BSIG_HD uint32_t take(BitIn &in, int n)
{
    const uint32_t v = (uint32_t)(in.buf & ((1ull << n) - 1));
    in.buf >>= n;
    in.cnt -= n;
    return v;
}

BSIG_HD int count_of(const Counts &c, int len) { return (int)((c.w[len >> 1] >> ((len & 1) * 16)) & 0xFFFFu); }

// one symbol of a canonical code by the walk over the code lengths; needs 15 valid bits.  -1: no such code
// (the published algorithm of zlib's contrib/puff/puff.c -- first code, count and index of every
// length -- with the per-length counts packed two per register)
template <typename Syms>
BSIG_HD int decode_walk(BitIn &in, const Counts &c, const Syms &sym)
{
    int code = 0, first = 0, index = 0;
    uint64_t b = in.buf;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(b & 1);
        b >>= 1;
        const int count = count_of(c, len);
        if (code - count < first) {
            in.buf = b;
            in.cnt -= len;
            return sym.get(index + (code - first));
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

BSIG_HD uint32_t bit_reverse(uint32_t v, int n)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32 - n);
#else
    uint32_t r = 0;
    for (int k = 0; k < n; ++k) r |= ((v >> k) & 1u) << (n - 1 - k);
    return r;
#endif
}

// where the walk stands after the lengths the first-level table covers (1..FAST): a code that table does not
// hold is longer, so its walk starts here
struct WalkStart {
    int first, index;
};
template <int FAST>
BSIG_HD WalkStart walk_start(const Counts &c)
{
    WalkStart w{0, 0};
    for (int len = 1; len <= FAST; ++len) {
        const int count = count_of(c, len);
        w.index += count;
        w.first += count;
        w.first <<= 1;
    }
    return w;
}

// a symbol whose code is longer than FAST bits: the walk of decode_walk from length FAST + 1 on, WITHOUT
// branches -- the lanes of a wave meet codes of different lengths, and an exit per length made the wave run
// the rest of the chain once per distinct length, each exit with its own symbol lookup -- and ONE lookup.
template <int FAST, typename Syms>
BSIG_HD int decode_long(BitIn &in, const Counts &c, const Syms &sym, const WalkStart &ws)
{
    int code = (int)(bit_reverse((uint32_t)in.buf & ((1u << FAST) - 1u), FAST) << 1);
    int first = ws.first, index = ws.index;
    uint32_t b = (uint32_t)(in.buf >> FAST);
    int hit_len = 0, hit_idx = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int len = FAST + 1; len <= 15; ++len) {
        code |= (int)(b & 1u);
        b >>= 1;
        const int count = count_of(c, len);
        const bool hit = hit_len == 0 && code - count < first;
        hit_idx = hit ? index + (code - first) : hit_idx;
        hit_len = hit ? len : hit_len;
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
        // (codes of 13-15 bits are rare: once every lane that is here has its symbol the rest of the chain is skipped)
        if (BSIG_ALL_LANES(hit_len != 0)) break;
    }
    if (hit_len == 0) return -1;
    in.buf >>= hit_len;
    in.cnt -= hit_len;
    return sym.get(hit_idx);
}

// first-level entry of the literal/length code for the next bits: (symbol << 4) | code length, 0 = longer code
BSIG_HD uint32_t lfast_at(const LaneTables &T, uint64_t buf)
{
    return T.lfast[(uint32_t)buf & ((1u << kLFast) - 1u)];
}

template <int FAST>
BSIG_HD int decode(BitIn &in, const LaneTables &T, const Counts &c, const LSyms &sym, const WalkStart &ws)
{
    const uint32_t e = lfast_at(T, in.buf);
    if (e) {
        const int len = (int)(e & 15u);
        in.buf >>= len;
        in.cnt -= len;
        return (int)(e >> 4);
    }
    return decode_long<FAST>(in, c, sym, ws);
}

// a distance symbol: the first-level table, else the walk
BSIG_HD int decode_dist(BitIn &in, const uint8_t *fast, const Counts &c, const DSyms &sym, const WalkStart &ws)
{
    const uint32_t e = fast[in.buf & ((1u << kDFast) - 1)];
    if (e) {
        const int len = (int)(e & 7u);
        in.buf >>= len;
        in.cnt -= len;
        return (int)(e >> 3);
    }
    return decode_long<kDFast>(in, c, sym, ws);
}

// counts + sorted symbols + first-level table from n code lengths (0: symbol unused).  Returns
// false for an over-subscribed set; incomplete sets are accepted (their unused codes decode to -1).
// first-level tables: 16-bit entries (symbol << 4 | length) or 8-bit ones (symbol << 3 | length)
struct Fast16 {
    uint16_t *t;
    BSIG_HD bool on() const { return t != nullptr; }
    BSIG_HD void put(uint32_t k, int s, int l) const { t[k] = (uint16_t)((s << 4) | l); }
    BSIG_HD void zero(uint32_t k) const { t[k] = 0; }
};
struct Fast8 {
    uint8_t *t;
    BSIG_HD bool on() const { return t != nullptr; }
    BSIG_HD void put(uint32_t k, int s, int l) const { t[k] = (uint8_t)((s << 3) | l); }
    BSIG_HD void zero(uint32_t k) const { t[k] = 0; }
};

// sixteen 16-bit slots in eight registers (the per-length offsets and next codes of a table under construction:
// they used to live in the lane's global scratch, where every `slot[len]++` was a load, a wait that drained every
// store in flight, and a store -- two trips to memory per symbol, 1.6 million cycles per code table, a tenth of a
// lane's time on blocks that hold six deflate blocks (zlib level 6) and a third of its wave's, because the lanes
// of a wave reach their tables at different times)
// (every loop over the words is unrolled by hand: left to itself the compiler kept `w` in scratch memory and indexed
// it there -- a trip to memory per access, which is what the registers were for)
struct Slots16 {
    uint32_t w0, w1, w2, w3, w4, w5, w6, w7;
    // all-ones if word k is the one slot l lives in (masks and ORs, not selects: a chain of selects over the
    // eight words is what the compiler turns back into an array in scratch memory)
    static BSIG_HD uint32_t is(int l, int k) { return 0u - (uint32_t)((l >> 1) == k); }
    BSIG_HD void clear() { w0 = w1 = w2 = w3 = w4 = w5 = w6 = w7 = 0; }
    BSIG_HD uint32_t get(int l) const
    {
        const uint32_t x = (w0 & is(l, 0)) | (w1 & is(l, 1)) | (w2 & is(l, 2)) | (w3 & is(l, 3)) |
                           (w4 & is(l, 4)) | (w5 & is(l, 5)) | (w6 & is(l, 6)) | (w7 & is(l, 7));
        return (x >> ((l & 1) * 16)) & 0xFFFFu;
    }
    BSIG_HD void add(int l, uint32_t d)                    // (values stay below 65,536: no carry into the neighbour)
    {
        const uint32_t v = d << ((l & 1) * 16);
        w0 += v & is(l, 0); w1 += v & is(l, 1); w2 += v & is(l, 2); w3 += v & is(l, 3);
        w4 += v & is(l, 4); w5 += v & is(l, 5); w6 += v & is(l, 6); w7 += v & is(l, 7);
    }
    BSIG_HD void set(int l, uint32_t val)
    {
        const int sh = (l & 1) * 16;
        const uint32_t field = 0xFFFFu << sh, put = (val & 0xFFFFu) << sh;
        w0 = (w0 & ~(field & is(l, 0))) | (put & is(l, 0)); w1 = (w1 & ~(field & is(l, 1))) | (put & is(l, 1));
        w2 = (w2 & ~(field & is(l, 2))) | (put & is(l, 2)); w3 = (w3 & ~(field & is(l, 3))) | (put & is(l, 3));
        w4 = (w4 & ~(field & is(l, 4))) | (put & is(l, 4)); w5 = (w5 & ~(field & is(l, 5))) | (put & is(l, 5));
        w6 = (w6 & ~(field & is(l, 6))) | (put & is(l, 6)); w7 = (w7 & ~(field & is(l, 7))) | (put & is(l, 7));
    }
};

// ... or in sixteen 16-bit words of LDS that nobody else needs meanwhile (the distance table's, while the
// literal/length table is built: a slot is then one read and one write instead of two dozen selects)
struct SlotsMem {
    uint16_t *p;
    BSIG_HD void clear() { for (int k = 0; k < 16; ++k) p[k] = 0; }
    BSIG_HD uint32_t get(int l) const { return p[l]; }
    BSIG_HD void add(int l, uint32_t d) { p[l] = (uint16_t)(p[l] + d); }
    BSIG_HD void set(int l, uint32_t val) { p[l] = (uint16_t)val; }
};

// the code lengths a dynamic block's header left in the scratch, read eight at a time
struct LenReader {
    const uint8_t *p;
    int base;
    uint64_t w;
    BSIG_HD explicit LenReader(const uint8_t *q) : p(q), base(-8), w(0) {}
    BSIG_HD int operator()(int i)
    {
        const int b = i & ~7;
        if (b != base) { base = b; w = load64(p + b); }
        return (int)((w >> (8 * (i & 7))) & 0xFFu);
    }
};

template <int FAST, typename FastT, typename Syms, typename LenAt, typename Slots>
BSIG_HD bool construct(Counts &c, const FastT &fast, Syms &sym, int n, LenAt len_at, Slots offs, Slots next)
{
    sym.clear();
    offs.clear();
    next.clear();
    for (int i = 0; i < n; ++i) offs.add(len_at(i), 1u);                  // offs = counts for now (slot 0: unused symbols)
    for (int k = 0; k < 8; ++k) c.w[k] = 0;
    int left = 1, acc = 0;
    uint32_t code = 0, shorter = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int len = 1; len <= 15; ++len) {
        const uint32_t cnt = offs.get(len);
        left <<= 1;
        left -= (int)cnt;
        if (left < 0) return false;
        c.w[len >> 1] |= cnt << ((len & 1) * 16);
        code = (code + shorter) << 1;                // first canonical code of this length
        next.set(len, code);
        shorter = cnt;
        offs.set(len, (uint32_t)acc);                // ... and where its symbols begin in the sorted array
        acc += (int)cnt;
    }
    const bool use_fast = FAST > 0 && fast.on();
    // (sorted positions from here on belong to codes the first-level table does not hold: offs[FAST + 1] is where
    // length FAST + 1 begins, the total if there is no longer code)
    sym.begin_long(FAST > 0 && FAST < 15 ? (int)offs.get(FAST + 1) : acc);
    if (use_fast)
        for (uint32_t k = 0; k < (1u << FAST); ++k) fast.zero(k);
    for (int i = 0; i < n; ++i) {
        const int l = len_at(i);
        if (!l) continue;
        sym.put((int)offs.get(l), i);
        offs.add(l, 1u);
        const uint32_t cd = next.get(l);
        next.add(l, 1u);
        if (use_fast && l <= FAST)
            for (uint32_t k = bit_reverse(cd, l); k < (1u << FAST); k += 1u << l) fast.put(k, i, l);
    }
    return true;
}

template <int FAST, typename FastT, typename Syms, typename LenAt>
BSIG_HD bool construct(Counts &c, const FastT &fast, Syms &sym, int n, LenAt len_at)
{
    Slots16 a, b;
    a.clear();
    b.clear();
    return construct<FAST>(c, fast, sym, n, len_at, a, b);
}

// position k of the code-length code's lengths in the stream -> symbol (RFC 1951, 3.2.7:
// 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15), 5 bits per entry
BSIG_HD int cl_order(int k)
{
    const uint64_t lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                        10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    return (int)((k < 12 ? lo >> (5 * k) : hi >> (5 * (k - 12))) & 31);
}

constexpr int kLensCodes = 352;   // scratch of produce: 32 for the code-length code + 316 lengths
constexpr int kLensBytes = 784;   // ... + ColdTables (324 bytes) behind it; the scratch must be 8-byte aligned
static_assert(kLensCodes % 4 == 0 && kLensCodes + (int)sizeof(ColdTables) <= kLensBytes, "scratch layout");

// true once more bits were consumed than the input holds
BSIG_HD bool overrun(const BitIn &in) { return in.p - (in.cnt >> 3) > in.end; }

BSIG_HD int len_base(int s)     // length symbols 257..285 -> s = 0..28
{
    return s < 8 ? 3 + s : s == 28 ? 258 : 3 + ((4 + (s & 3)) << ((s >> 2) - 1));
}
BSIG_HD int len_extra(int s) { return s < 8 || s == 28 ? 0 : (s >> 2) - 1; }
BSIG_HD int dist_base(int s)    // distance symbols 0..29
{
    return s < 4 ? 1 + s : 1 + ((2 + (s & 1)) << ((s >> 1) - 1));
}
BSIG_HD int dist_extra(int s) { return s < 4 ? 0 : (s >> 1) - 1; }

// (diagnostic build -DBSIG_INFLATE_PROF, scripts/inflate_prof.py: what a lane's time goes to)
struct LaneProf {
    uint32_t turns, hdrs, walks, match_turns, lit_turns, lits;
    uint64_t hdr_cycles;
    uint64_t sec_cycles[8];     // (rounds 3-4: shader-clock cycles per section of the one-lane turn; unused since the split)
    uint32_t sec_n[8];
    uint32_t short_period;      // matches with a distance below 8
    uint32_t long_dist;         // distance codes longer than the first-level table
    uint32_t pad_[2];
};
// device: wait until every load and store this lane's wave has issued is done (s_waitcnt vmcnt(0), gfx9 encoding)
#if defined(__HIP_DEVICE_COMPILE__)
#define BSIG_VM_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define BSIG_VM_DRAIN() do { } while (0)
#endif
#if defined(BSIG_INFLATE_PROF) && defined(__HIP_DEVICE_COMPILE__)
#define BSIG_PROF(x) do { if (prof) { x; } } while (0)
#define BSIG_PROF_CLOCK() ((uint64_t)clock64())
#define BSIG_SEC_BEGIN(k) const uint64_t sec_t##k = prof ? (uint64_t)clock64() : 0ull
#define BSIG_SEC_END(k) do { if (prof) { prof->sec_cycles[k] += (uint64_t)clock64() - sec_t##k; prof->sec_n[k]++; } } while (0)
#else
#define BSIG_SEC_BEGIN(k) do { } while (0)
#define BSIG_SEC_END(k) do { } while (0)
#define BSIG_PROF(x) do { } while (0)
#define BSIG_PROF_CLOCK() 0ull
#endif

// ---- a code's tables built by the WAVE for one of its lanes (device) -------------------------------------------------
// A lane that builds its tables by itself (construct above) goes through ~4,000 LDS operations one behind the other,
// and its wave stands still meanwhile: a round of headers costs a launch 3-4 ms -- a sixth of a level-1 launch, half of a
// level-6 one (six deflate blocks per BGZF block; profiles/NOTES_r05.md, section 8).  The canonical code is all counting
// and ranking, which a wave does in a few ballots: the W lanes of the producer wave build the tables of ONE lane
// together, each holding the code lengths of the symbols hx, hx + W, hx + 2W, ...  Per length: how many symbols have it
// (ballot + count), the first code and the first sorted position (a 15-step scan, the same in every lane), then per
// symbol its rank among the symbols of its length (the lanes below it in the ballot + what earlier rounds counted) --
// code = first + rank -- and every lane writes its symbols' first-level entries and sorted positions.  No LDS
// read-modify-write, no chain: about 1,500 instructions of the wave per literal/length code.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t lanes_below(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
template <int W>
__device__ __forceinline__ uint32_t wave_or(uint32_t v)          // OR over the W lanes (W a power of two; lanes >= W are not here)
{
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) v |= (uint32_t)__shfl_xor((int)v, d);
    return v;
}
template <int W>
__device__ __forceinline__ uint64_t wave_or64(uint64_t v)
{
    return (uint64_t)wave_or<W>((uint32_t)v) | (uint64_t)wave_or<W>((uint32_t)(v >> 32)) << 32;
}
// first-level tables as the wave fills them: ENTRY bytes per entry, 1 << FAST entries at `t` (LDS of the target lane)
template <int FAST, typename Entry, int SHIFT>
struct CoopFast {
    Entry *t;
    template <int W> __device__ __forceinline__ void zero(int hx) const
    {
        constexpr int bytes = (int)sizeof(Entry) << FAST;
        static_assert(bytes % 16 == 0, "whole 16-byte vectors");
        for (int o = 16 * hx; o < bytes; o += 16 * W) {
            const W16 z{0, 0};
            __builtin_memcpy(reinterpret_cast<uint8_t *>(t) + o, &z, 16);
        }
    }
    __device__ __forceinline__ void put(uint32_t k, uint32_t sym, uint32_t l) const { t[k] = (Entry)((sym << SHIFT) | l); }
};
// ml[r]: the code length of symbol hx + W * r (0: unused or beyond the code).  on_symbol(sorted position, symbol, length)
// is called by the symbol's lane for every used symbol.  c: the counts per length (the same in every lane); base: the
// first sorted position of a code longer than FAST.  false: over-subscribed (the same in every lane).
template <int FAST, int W, int ROUNDS, typename FastT, typename OnSymbol>
__device__ __forceinline__ bool coop_construct(const uint32_t (&ml)[ROUNDS], int hx, const FastT &fast, Counts &c, int &base,
                                               OnSymbol on_symbol)
{
    uint32_t cnt[16], first[16], offs[16], run[16];
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        uint32_t k = 0;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) k += (uint32_t)__popcll(__ballot(ml[r] == (uint32_t)len));
        cnt[len] = k;
        run[len] = 0;
    }
    int left = 1;
    uint32_t code = 0, shorter = 0, acc = 0;
    bool over = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) c.w[k] = 0;
#pragma unroll
    for (int len = 1; len <= 15; ++len) {
        left <<= 1;
        left -= (int)cnt[len];
        over = over | (left < 0);
        c.w[len >> 1] |= cnt[len] << ((len & 1) * 16);
        code = (code + shorter) << 1;                // first canonical code of this length
        first[len] = code;
        shorter = cnt[len];
        offs[len] = acc;                             // ... and where its symbols begin in the sorted order
        acc += cnt[len];
    }
    if (over) return false;
    base = FAST < 15 ? (int)offs[FAST + 1] : (int)acc;
    fast.template zero<W>(hx);
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t l = ml[r];
        uint32_t cd = 0, ix = 0;
#pragma unroll
        for (int len = 1; len <= 15; ++len) {
            if (cnt[len] == 0u) continue;                 // (uniform: a code uses eight to ten of the fifteen lengths)
            const bool mine = l == (uint32_t)len;
            const uint64_t m = __ballot(mine);
            const uint32_t rk = run[len] + lanes_below(m);
            cd = mine ? first[len] + rk : cd;
            ix = mine ? offs[len] + rk : ix;
            run[len] += (uint32_t)__popcll(m);
        }
        const uint32_t sym = (uint32_t)(hx + W * r);
        if (l != 0u) {
            if (l <= (uint32_t)FAST)
                for (uint32_t k = bit_reverse(cd, (int)l); k < (1u << FAST); k += 1u << l) fast.put(k, sym, l);
            on_symbol(ix, sym, l);
        }
    }
    return true;
}

// the code-length code of lane L (its 19 lengths, 3 bits each, in clw): a 7-bit table in L's hot-symbol LDS
template <int W>
__device__ __forceinline__ bool coop_cl_table(uint64_t clw, int hx, LaneTables *TL)
{
    constexpr int R = (19 + W - 1) / W;
    uint32_t ml[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = hx + W * r;
        ml[r] = i < 19 ? (uint32_t)((clw >> (3 * i)) & 7ull) : 0u;
    }
    Counts cc;
    int base;
    return coop_construct<7, W, R>(ml, hx, CoopFast<7, uint8_t, 3>{TL->lsym_hot}, cc, base, [](uint32_t, uint32_t, uint32_t) {});
}

// the literal/length and distance tables of lane L from the code lengths staged in L's literal table storage (nlen
// literal/length lengths, then ndist distance lengths; `fixed`: the fixed code of RFC 1951, 3.2.6 instead).  What
// lives in registers of the target lane -- counts, the hot symbols' ninth bits, the distance symbols -- comes back
// through lc / dc / hot_hi / dsw / base, valid in EVERY lane (lane L keeps them).  false: a code is over-subscribed.
template <int W>
__device__ __forceinline__ bool coop_tables(int hx, LaneTables *TL, uint8_t *lensL, int nlen, int ndist, bool fixed,
                                            Counts &lc, Counts &dc, uint64_t (&hot_hi)[kHotSyms / 64], uint64_t (&dsw)[3], int &base)
{
    constexpr int RL = (288 + W - 1) / W, RD = (32 + W - 1) / W;
    uint32_t ml[RL], md[RD];
    const uint8_t *stage = reinterpret_cast<const uint8_t *>(TL->lfast);
#pragma unroll
    for (int r = 0; r < RL; ++r) {
        const int i = hx + W * r;
        ml[r] = fixed ? (i < 144 ? 8u : i < 256 ? 9u : i < 280 ? 7u : i < 288 ? 8u : 0u) : i < nlen ? (uint32_t)stage[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RD; ++r) {
        const int i = hx + W * r;
        md[r] = fixed ? (i < 30 ? 5u : 0u) : i < ndist ? (uint32_t)stage[nlen + i] : 0u;
    }
    ColdTables &Cd = *reinterpret_cast<ColdTables *>(lensL + kLensCodes);
    for (int k = hx; k < 9; k += W) Cd.lhi[k] = 0;
    uint64_t hh[kHotSyms / 64];
#pragma unroll
    for (int k = 0; k < kHotSyms / 64; ++k) hh[k] = 0;
    int lbase = 0;
    // (base is known inside coop_construct before any symbol is placed: the callback reads it through the reference)
    const bool ok_l = coop_construct<kLFast, W, RL>(ml, hx, CoopFast<kLFast, uint16_t, 4>{TL->lfast}, lc, lbase,
        [&](uint32_t ix, uint32_t sym, uint32_t l) {
            if (l <= (uint32_t)kLFast) return;                         // the first-level table answers it
            const uint32_t h = ix - (uint32_t)lbase;
            if (h < (uint32_t)kHotSyms) {
                TL->lsym_hot[h] = (uint8_t)sym;
                const uint64_t bit = (uint64_t)(sym >> 8) << (h & 63u);
#pragma unroll
                for (int k = 0; k < kHotSyms / 64; ++k) hh[k] |= (int)(h >> 6) == k ? bit : 0ull;
            } else {
                Cd.lsym[ix] = (uint8_t)sym;
                if (sym >> 8) atomicOr(&Cd.lhi[ix >> 5], 1u << (ix & 31u));
            }
        });
#pragma unroll
    for (int k = 0; k < kHotSyms / 64; ++k) hot_hi[k] = wave_or64<W>(hh[k]);
    base = lbase;
    uint64_t dw[3] = {0, 0, 0};
    int dbase = 0;
    const bool ok_d = coop_construct<kDFast, W, RD>(md, hx, CoopFast<kDFast, uint8_t, 3>{TL->dfast}, dc, dbase,
        [&](uint32_t ix, uint32_t sym, uint32_t) {
            const uint32_t q = ix >= 24u ? 2u : ix >= 12u ? 1u : 0u;
            const uint64_t v = (uint64_t)(sym & 31u) << (5u * (ix - 12u * q));
            dw[0] |= q == 0u ? v : 0ull;
            dw[1] |= q == 1u ? v : 0ull;
            dw[2] |= q == 2u ? v : 0ull;
        });
#pragma unroll
    for (int k = 0; k < 3; ++k) dsw[k] = wave_or64<W>(dw[k]);
    return ok_l && ok_d;
}
#endif

// ---- the two halves of a block's work and the channel between them ------------------------------------------------
// Huffman decoding and LZ77 copying are two dependent chains of their own -- bits -> table -> bits ..., and
// load -> wait -> store ... -- and one lane that runs both pays their SUM per turn (r05 ablation, one round of 57,344
// blocks: bare records 16.6 ms of which 8.2 decode with all output traffic removed; real-shaped 24.5 of which 18.0).
// So a block is worked on by TWO lanes, one in each of two waves of a workgroup: the PRODUCER decodes symbols into
// tokens -- up to six literal bytes and/or one match (length, distance) -- and the CONSUMER turns tokens into bytes
// (what used to be the second half of a turn: the deferred match stores, the literals' store, the match loads).  The
// two waves issue side by side and neither waits for the other's memory; between them a one-token mailbox in LDS
// (12 bytes per block: the tables leave no more at seven workgroups per CU).
//
// token: a = literal bytes (bits 0..47) | number of literals << 48 (0..6) | match length << 51 (0 = none, 3..258)
//        b = valid (bit 0) | distance << 1 (1..32768) | end of stream (bit 17) | error code << 18
// b == 0: the mailbox is empty.  The producer writes a, then b; the consumer reads b, then a, then clears b; LDS
// operations of one wave are carried out in order, so no fence is needed, only that the compiler keeps the order
// (volatile).  The host build pushes the tokens of a whole block into a vector and runs the consumer over it: the
// same two functions, against zlib in tests/test_inflate_lane.py and under the sanitizers.
struct Token {
    uint64_t a;
    uint32_t b;
};
constexpr uint32_t kTokValid = 1u, kTokEnd = 1u << 17;
constexpr int kTokErrShift = 18;

#if defined(__HIP_DEVICE_COMPILE__)
#ifndef BSIG_IDLE_SLEEP
#define BSIG_IDLE_SLEEP 2
#endif
#define BSIG_SLEEP_IF_ALL(c) do { if (__all(c)) __builtin_amdgcn_s_sleep(BSIG_IDLE_SLEEP); } while (0)
#else
#define BSIG_SLEEP_IF_ALL(c) do { } while (0)
#endif

// the device's mailbox: two words of LDS per block.  (Pointers INTO LDS by type: through generic pointers the compiler
// issues flat_ loads and stores, which wait for LDS and memory alike and may complete out of order.)
#if defined(__HIP_DEVICE_COMPILE__)
#define BSIG_LDS __attribute__((address_space(3)))
#else
#define BSIG_LDS
#endif
struct ChanLds {
    BSIG_LDS volatile uint64_t *a;
    BSIG_LDS volatile uint32_t *b;
    template <typename A, typename B>
    BSIG_HD ChanLds(A *a_, B *b_) : a((BSIG_LDS volatile uint64_t *)a_), b((BSIG_LDS volatile uint32_t *)b_) {}
    BSIG_HD bool ready() const { return *b == 0u; }              // (producer) may the next token be sent?
    BSIG_HD void send(const Token &t) const { *a = t.a; *b = t.b; }
    BSIG_HD bool take(Token &t) const                             // (consumer)
    {
        const uint32_t f = *b;
        if (f == 0u) return false;
        t.a = *a;
        t.b = f;
        *b = 0u;
        return true;
    }
};
#if !defined(__HIP_DEVICE_COMPILE__)
// the host's channel: every token of the block, then the consumer
struct ChanVec {
    std::vector<Token> *v;
    size_t rd;
    bool ready() const { return true; }
    void send(const Token &t) const { v->push_back(t); }
    bool take(Token &t)
    {
        if (rd >= v->size()) return false;
        t = (*v)[rd++];
        return true;
    }
};
#endif

// PRODUCER: decodes one raw DEFLATE stream of in_len bytes that must inflate to exactly out_len bytes into tokens.
// lens: kLensBytes of scratch, 8-byte aligned (global memory on the device).  On the device the 8 bytes behind
// in_p + in_len must be readable (never used: see request()); the host build reads nothing behind the input
// (tests/asan checks that).  Every token is checked before it is sent (distance within what is out, length within
// out_len), so the consumer needs no checks of its own.  Ends the stream with an end token that carries the return
// value: OK or an ERR_ code.
// Device: W = the lanes of the producer wave (lane ids 0 .. W-1, ALL of them here: a lane without a block comes with
// present = false and only helps with the others' tables); lane_stride / lens of neighbouring lanes lie
// (their lane - mine) * lane_stride / * kLensBytes bytes from T / lens.
template <int W = 1, typename Chan>
BSIG_HD int produce(const uint8_t *in_p, uint32_t in_len, uint32_t out_len, LaneTables &T, uint8_t *lens, Chan &ch,
                    LaneProf *prof = nullptr, bool present = true, int hx = 0, uint32_t lane_stride = 0)
{
    (void)prof; (void)hx; (void)lane_stride;
    BitIn in;
    in.p = in_p; in.end = in_p + in_len; in.buf = 0; in.cnt = 0;
    if (present) request(in); else { in.ahead = 0; in.amask = 0; }
    uint32_t op = 0;                   // bytes the tokens sent so far make
    int err = OK;
    Counts lc, dc;
    for (int k = 0; k < 8; ++k) lc.w[k] = dc.w[k] = 0;
    ColdTables &Cd = *reinterpret_cast<ColdTables *>(lens + kLensCodes);
    LSyms ls{Cd.lsym, Cd.lhi, T.lsym_hot, {}, 0};
    DSyms ds;
    ds.clear();
    // Per lane, one loop over the states of a deflate block:
    //   kHeader    its first bits; a stored block is sent off as literals at once
    //   kWantCl    (dynamic code) the 19 lengths of the code-length code are read: its table is to be built
    //   kLens      ... with which the literal/length and distance code lengths are decoded into the literal table's
    //              storage
    //   kWantTabs  the two codes' tables are to be built
    //   kSyms      one token per turn
    // On the device the two builds are the WAVE's work, one lane's tables at a time (coop_* above); the host, one lane,
    // builds them with construct().
    enum { kHeader = 0, kWantCl = 1, kLens = 2, kWantTabs = 3, kSyms = 4, kEnd = 5 };
    int st = present ? kHeader : kEnd;
    uint32_t last = 0;
    uint64_t clw = 0;                  // the code-length code's lengths, 3 bits each
    int nlen = 0, ndist = 0;
    bool fixed = false;
    WalkStart lws{0, 0}, dws{0, 0};
    uint8_t *ll_near = reinterpret_cast<uint8_t *>(T.lfast);
    static_assert(sizeof(T.lfast) >= 320, "the code lengths are staged in the literal table's storage");
    for (;;) {
        // ---- a block's first bits
        if (st == kHeader) do {
            refill(in);
            last = take(in, 1);
            const uint32_t type = take(in, 2);
            if (type == 0) {
                // stored: skip to the byte boundary, LEN / NLEN, raw bytes -- sent as literals, six per token (rare in a
                // BAM: only data that does not compress)
                take(in, in.cnt & 7);
                refill(in);
                const uint32_t len = take(in, 16), nl = take(in, 16);
                if ((len ^ 0xFFFFu) != nl) { err = ERR_STORED; break; }
                // bytes still in the bit buffer belong to the raw data
                const uint8_t *src = in.p - (in.cnt >> 3);
                if (src + len > in.end) { err = ERR_INPUT; break; }
                if (op + len > out_len) { err = ERR_OUTPUT; break; }
                for (uint32_t k = 0; k < len;) {
                    if (!ch.ready()) { BSIG_SLEEP_IF_ALL(true); continue; }
                    const uint32_t m = len - k < 6u ? len - k : 6u;
                    uint64_t w = 0;
                    for (uint32_t q = 0; q < m; ++q) w |= (uint64_t)src[k + q] << (8 * q);
                    ch.send(Token{w | (uint64_t)m << 48, kTokValid});
                    k += m;
                }
                op += len;
                in.p = src + len; in.buf = 0; in.cnt = 0;
                request(in);
                if (overrun(in)) { err = ERR_INPUT; break; }
                st = last ? kEnd : kHeader;
            } else if (type == 3) {
                err = ERR_CODE;
            } else if (type == 1) {
                fixed = true;
                nlen = 288; ndist = 30;
                st = kWantTabs;
            } else {
                fixed = false;
                nlen = (int)take(in, 5) + 257; ndist = (int)take(in, 5) + 1;
                const int ncode = (int)take(in, 4) + 4;
                if (nlen > 286 || ndist > 30) { err = ERR_TABLE; break; }
                // the code-length code: its 19 lengths (3 bits each, in the order of RFC 1951, 3.2.7) in one register
                clw = 0;
                for (int k = 0; k < ncode; ++k) {
                    if (in.cnt < 8) refill(in);
                    clw |= (uint64_t)take(in, 3) << (3 * cl_order(k));
                }
                st = kWantCl;
            }
        } while (0);
        if (err) st = kEnd;
        // ---- the code-length code's table: 7 bits answer every one of its codes; it is built in the hot symbols' LDS
        // (128 bytes, rewritten with the literal/length table below)
        static_assert(kHotSyms >= 128, "the code-length code's 7-bit table lives in the hot symbols' storage");
#if defined(__HIP_DEVICE_COMPILE__)
        for (uint64_t need = __ballot(st == kWantCl); need; need &= need - 1) {
            const int L = __ffsll((unsigned long long)need) - 1;
            const uint64_t clwL = (uint64_t)(uint32_t)__shfl((int)(uint32_t)clw, L) | (uint64_t)(uint32_t)__shfl((int)(uint32_t)(clw >> 32), L) << 32;
            LaneTables *TL = reinterpret_cast<LaneTables *>(reinterpret_cast<uint8_t *>(&T) + (L - hx) * (int)lane_stride);
            const bool ok = coop_cl_table<W>(clwL, hx, TL);
            if (hx == L) { if (ok) st = kLens; else { err = ERR_TABLE; st = kEnd; } }
        }
#else
        if (st == kWantCl) {
            Counts cc;
            if (construct<7>(cc, Fast8{T.lsym_hot}, ds, 19, [&](int i) { return (int)((clw >> (3 * i)) & 7ull); })) st = kLens;
            else { err = ERR_TABLE; st = kEnd; }
        }
#endif
        if (st == kLens) do {
            // literal/length + distance code lengths, run-length coded.  They are collected in the first-level
            // literal table's LDS (free until that table is rebuilt below): a store per length to global memory made
            // every wait of this loop -- each refill's -- a wait for the stores before it (2,000 cycles per code length).
            int idx = 0, prev_len = 0;
            // (zeroed first: the runs of unused symbols -- codes 17 and 18, up to 138 lengths each, and every
            // lane of a wave waits for the longest -- then only move the index)
            for (int k = 0; k < 320 / 4; ++k) reinterpret_cast<uint32_t *>(T.lfast)[k] = 0;
            while (idx < nlen + ndist) {
                if (in.cnt < 24) refill(in);                  // 15 bits of code + 7 of repeat count
                const uint32_t e = T.lsym_hot[in.buf & 127u];
                if (!e) { err = ERR_CODE; break; }
                in.buf >>= (e & 7u);
                in.cnt -= (int)(e & 7u);
                const int s = (int)(e >> 3);
                if (s < 16) {
                    ll_near[idx++] = (uint8_t)s;
                    prev_len = s;
                } else {
                    int prev = 0, rep;
                    if (s == 16) {
                        if (idx == 0) { err = ERR_TABLE; break; }
                        prev = prev_len;
                        rep = 3 + (int)take(in, 2);
                    } else if (s == 17) {
                        rep = 3 + (int)take(in, 3);
                    } else {
                        rep = 11 + (int)take(in, 7);
                    }
                    if (idx + rep > nlen + ndist) { err = ERR_TABLE; break; }
                    if (prev) {
                        while (rep--) ll_near[idx++] = (uint8_t)prev;      // (code 16: at most six)
                    } else {
                        idx += rep;
                    }
                    prev_len = prev;
                }
                if (overrun(in)) { err = ERR_INPUT; break; }
            }
            if (err) break;
            if (ll_near[256] == 0) { err = ERR_TABLE; break; }            // no end-of-block code
            st = kWantTabs;
        } while (0);
        if (err) st = kEnd;
        // ---- the literal/length and the distance code's tables
#if defined(__HIP_DEVICE_COMPILE__)
        for (uint64_t need = __ballot(st == kWantTabs); need; need &= need - 1) {
            const int L = __ffsll((unsigned long long)need) - 1;
            const uint64_t prof_t0 = BSIG_PROF_CLOCK();
            (void)prof_t0;
            LaneTables *TL = reinterpret_cast<LaneTables *>(reinterpret_cast<uint8_t *>(&T) + (L - hx) * (int)lane_stride);
            uint8_t *lensL = lens + (L - hx) * kLensBytes;
            Counts c1, c2;
            uint64_t hh[kHotSyms / 64], dw[3];
            int base = 0;
            const bool ok = coop_tables<W>(hx, TL, lensL, __shfl(nlen, L), __shfl(ndist, L), __shfl((int)fixed, L) != 0, c1, c2, hh, dw, base);
            if (hx == L) {
                if (ok) {
                    lc = c1; dc = c2;
                    for (int k = 0; k < kHotSyms / 64; ++k) ls.hot_hi[k] = hh[k];
                    ls.base = base;
                    ds.w[0] = dw[0]; ds.w[1] = dw[1]; ds.w[2] = dw[2];
                    lws = walk_start<kLFast>(lc);
                    dws = walk_start<kDFast>(dc);
                    st = kSyms;
                    BSIG_PROF(prof->hdrs++; prof->hdr_cycles += BSIG_PROF_CLOCK() - prof_t0);
                } else { err = ERR_TABLE; st = kEnd; }
            }
        }
#else
        if (st == kWantTabs) {
            // (the slots of the literal/length table's construction: the distance table's 64 bytes, rebuilt right after it)
            static_assert(sizeof(T.dfast) >= 64, "the distance table's LDS doubles as the literal table's construction slots");
            const SlotsMem lit_offs{reinterpret_cast<uint16_t *>(T.dfast)}, lit_next{reinterpret_cast<uint16_t *>(T.dfast) + 16};
            bool ok;
            if (fixed) {
                // fixed code: lengths 8 (0..143), 9 (144..255), 7 (256..279), 8 (280..287); 30 distances of 5 bits
                auto fl = [](int i) { return i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8; };
                auto fd = [](int) { return 5; };
                ok = construct<kLFast>(lc, Fast16{T.lfast}, ls, 288, fl, lit_offs, lit_next) && construct<kDFast>(dc, Fast8{T.dfast}, ds, 30, fd);
            } else {
                // (the lengths leave the literal table's storage before that table is built in it)
                uint8_t *ll = lens + 32;                          // up to 286 + 30 entries (kLensBytes)
                for (int k = 0; k < nlen + ndist; k += 8) store64(ll + k, load64(ll_near + k));      // (<= 320 bytes either side)
                ok = construct<kLFast>(lc, Fast16{T.lfast}, ls, nlen, LenReader(ll), lit_offs, lit_next) &&
                     construct<kDFast>(dc, Fast8{T.dfast}, ds, ndist, LenReader(ll + nlen));
            }
            if (ok) { lws = walk_start<kLFast>(lc); dws = walk_start<kDFast>(dc); st = kSyms; }
            else { err = ERR_TABLE; st = kEnd; }
        }
#endif
        // ---- the compressed data of a block: per turn ONE token -- a run of up to six literals the first-level table
        // knows and/or the match behind them.  The only memory a turn touches is the input word requested a turn ahead
        // (and, rarely, a cold symbol), so a turn is the decode chain and nothing else. ----
        if (st == kSyms) do {
            // Is the mailbox empty?  Asked NOW, needed only when the token is ready: the answer arrives beside the
            // first table lookup instead of standing in front of the chain, and the token is decoded while the
            // consumer may still be busy with the previous one.
            const bool room = ch.ready();
            uint64_t lit = 0;
            uint32_t nlit = 0, mlen = 0, mdist = 0;
            bool eob = false;
            BSIG_PROF(prof->turns++);
            refill(in);
            BSIG_PROF(if (lfast_at(T, in.buf) == 0) prof->walks++);
            int s = decode<kLFast>(in, T, lc, ls, lws);
            uint32_t opx = op;                 // where the next symbol's bytes will go
            if (s < 256) {
                if (s < 0) { err = ERR_CODE; }
                else if (op >= out_len) { err = ERR_OUTPUT; }
                else {
                    // a literal; up to five more if the first-level table says the next symbols
                    // are literals too (<= 8 bits each: the 56 bits of the refill cover 15 + 5 x 8)
                    lit = (uint64_t)s;
                    nlit = 1;
                    s = -2;                    // nothing more this turn, unless a match follows
                    if (kLFast > 0 && kMultiLit) {
#if BSIG_LIT_LOOP_BRANCHES
#if defined(__HIPCC__)
#pragma unroll
#endif
                        for (int q = 1; q < 6; ++q) {
                            const uint32_t e = lfast_at(T, in.buf);
                            if (e == 0 || (e >> 4) >= 256u || in.cnt < 2 * kLFast || op + (uint32_t)q >= out_len) break;
                            const int len = (int)(e & 15u);
                            in.buf >>= len;
                            in.cnt -= len;
                            lit |= (uint64_t)(e >> 4) << (8 * q);
                            nlit = (uint32_t)q + 1;
                        }
#else
                        // (no branch per literal: the lanes of a wave leave such a loop at five different places, and
                        // every exit is an exec-mask dance the whole wave pays -- a literal cost 500 cycles of which
                        // the table lookup is 100.  A lane whose run has ended goes through the remaining steps
                        // consuming zero bits.)
                        bool run = true;
#if defined(__HIPCC__)
#pragma unroll
#endif
                        for (int q = 1; q < 6; ++q) {
                            const uint32_t e = lfast_at(T, in.buf);
                            run = run & (e != 0u) & ((e >> 4) < 256u) & (in.cnt >= 2 * kLFast) & (op + (uint32_t)q < out_len);
                            const int len = run ? (int)(e & 15u) : 0;
                            in.buf >>= len;
                            in.cnt -= len;
                            lit |= (uint64_t)(run ? e >> 4 : 0u) << (8 * q);
                            nlit += run ? 1u : 0u;
                            if (!BSIG_ANY_LANES(run)) break;          // (uniform: nobody's run goes on)
                        }
#endif
                        // a match right behind the literals rides in the same token (its bits come
                        // with a second refill; the word it needs was requested a turn ago)
                        if (in.cnt >= kLFast) {
                            const uint32_t e = lfast_at(T, in.buf);
                            if (e && (e >> 4) > 256u) {
                                refill(in);
                                s = decode<kLFast>(in, T, lc, ls, lws);
                            }
                        }
                    }
                    opx = op + nlit;
                }
            }
            if (err == OK) {
                if (s == 256) {
                    eob = true;
                } else if (s > 256) {
                    s -= 257;
                    if (s >= 29) { err = ERR_CODE; }
                    else {
                        const uint32_t len = (uint32_t)len_base(s) + take(in, len_extra(s));
                        BSIG_PROF(if (T.dfast[in.buf & ((1u << kDFast) - 1)] == 0) prof->long_dist++);
                        const int d = decode_dist(in, T.dfast, dc, ds, dws);             // <= 20 + 28 of the 56 bits
                        if (d < 0 || d >= 30) { err = ERR_CODE; }
                        else {
                            const uint32_t dist = (uint32_t)dist_base(d) + take(in, dist_extra(d));
                            if (dist > opx) { err = ERR_DIST; }
                            else if (opx + len > out_len) { err = ERR_OUTPUT; }
                            else if (overrun(in)) { err = ERR_INPUT; }
                            else { mlen = len; mdist = dist; }
                        }
                    }
                }
            }
            if (err) { st = kEnd; break; }
            BSIG_PROF(if (nlit) { prof->lit_turns++; prof->lits += nlit; } if (mlen) prof->match_turns++);
            if (nlit | mlen) {
                // (the consumer still holds the previous token: a long match, or its memory is slow)
                if (!room)
                    while (!ch.ready()) BSIG_SLEEP_IF_ALL(true);
                ch.send(Token{lit | (uint64_t)nlit << 48 | (uint64_t)mlen << 51, kTokValid | mdist << 1});
                op = opx + mlen;
            }
            if (eob) {
                if (overrun(in)) { err = ERR_INPUT; st = kEnd; }
                else st = last ? kEnd : kHeader;
            }
        } while (0);
#if defined(__HIP_DEVICE_COMPILE__)
        if (__all(st == kEnd)) break;          // (a lane that is done stays: the others' tables need it)
#else
        if (st == kEnd) break;
#endif
    }
    if (!present) return OK;
    if (err == OK && op != out_len) err = ERR_OUTPUT;
    while (!ch.ready()) BSIG_SLEEP_IF_ALL(true);
    ch.send(Token{0, kTokValid | kTokEnd | (uint32_t)err << kTokErrShift});
    return err;
}

// CONSUMER: the tokens of one block -> its out_len bytes at `out` (nothing behind out + out_len is touched).  Per
// turn one token, OR a slice of the pending match: a lane that copies a 258-byte match does not hold the other lanes
// of its wave for 258 turns.  The LZ77 window is the output itself (global memory on the device): matches read back
// what the lane wrote.  The bytes of a match are LOADED in the turn that meets it and taken up in the next one: the trip
// to memory runs beside the next token's handling, and memory is waited for ONCE per turn (BSIG_VM_DRAIN: the
// compiler's own waits are all-or-nothing -- s_waitcnt vmcnt(0) wherever a loaded value is first used, the exact counts
// being beyond it in a loop whose loads and stores sit under lane-dependent branches -- so every load whose value is
// needed at once stalls the wave for a whole trip and drains what else is in flight).
//
// What bounds a full chip is the number of per-lane memory accesses, not their bytes (r05: TCP accesses per second
// are the same at half and at full occupancy, and k_inflate is no faster with all lanes than with half of them): every
// lane works in its own block, so each lane of each load or store is an access of its own to the CU's address
// pipeline.  The output therefore goes through a 16-byte ACCUMULATOR in registers and reaches memory in whole
// 16-byte stores, each byte once: literals (up to six bytes a token) and match bytes are shifted in behind each other;
// a token costs about out_bytes / 16 stores instead of one per literal run plus one per 16 bytes of match, each
// overshooting into bytes that were then written again.
struct OutAcc {
    uint64_t a, b;       // the bytes [op - fill, op), low byte first; zero beyond them
    uint64_t prev;       // the eight bytes in front of them (the upper half of the chunk stored last)
    uint32_t fill;       // 0..15
};
// the low n bytes of x (n >= 8: all of it)
BSIG_HD uint64_t low_bytes(uint64_t x, uint32_t n) { return n >= 8u ? x : x & ((1ull << (8u * n)) - 1ull); }
// x << s and x >> (64 - s) for s in 0..63, in bits (the second is 0 for s = 0)
BSIG_HD uint64_t shl_bits(uint64_t x, uint32_t s) { return x << s; }
BSIG_HD uint64_t carry_bits(uint64_t x, uint32_t s) { return s ? x >> (64u - s) : 0ull; }

// k bytes (0..16: the low bytes of (x0, x1), the rest already zero) behind the accumulator's; a full chunk goes to
// memory at out + (op - fill) and what is over starts the next one
BSIG_HD void acc_append(OutAcc &A, uint8_t *out, uint32_t &op, uint64_t x0, uint64_t x1, uint32_t k)
{
    const uint32_t s = 8u * (A.fill & 7u);
    const uint64_t p0 = shl_bits(x0, s), p1 = shl_bits(x1, s) | carry_bits(x0, s), p2 = carry_bits(x1, s);
    const bool up = A.fill >= 8u;                       // the new bytes begin in the upper half
    const uint64_t c0 = A.a | (up ? 0ull : p0), c1 = A.b | (up ? p0 : p1), c2 = up ? p1 : p2, c3 = up ? p2 : 0ull;
    const uint32_t nf = A.fill + k;
    const bool full = nf >= 16u;
    if (full) {
        store128(out + (op - A.fill), W16{c0, c1});
        A.prev = c1;
    }
    A.a = full ? c2 : c0;
    A.b = full ? c3 : c1;
    A.fill = nf & 15u;
    op += k;
}

template <typename Chan>
BSIG_HD void consume(uint8_t *out, uint32_t out_len, Chan &ch)
{
    constexpr uint32_t kTail = 160;    // the last bytes of a block are written one by one (16-byte moves may not pass its end)
    uint32_t op = 0;                   // bytes out, those in the accumulator included
    OutAcc A{0, 0, 0, 0};
    bool tail = false;                 // byte by byte from here on (the accumulator is empty)
    uint32_t pend = 0, pdist = 0;
    uint32_t pper = 0, pdone = 0;      // the match's period (its distance as coded) and how much of it is out
    uint32_t dn = 0;                   // bytes of the pending match loaded last turn (v), to be taken up in this one
    W16 v[kTurn / 16];
    for (uint32_t k = 0; k < kTurn / 16; ++k) v[k] = W16{0, 0};
    for (;;) {
        uint64_t lit = 0;
        uint32_t nlit = 0;
        bool stop = false;
        if (pend == 0) {
            Token t;
            if (ch.take(t)) {
                lit = t.a & 0xFFFFFFFFFFFFull;
                nlit = (uint32_t)(t.a >> 48) & 7u;
                const uint32_t len = (uint32_t)(t.a >> 51) & 0x1FFu;
                stop = (t.b & kTokEnd) != 0u;
                if (len) { pend = len; pdist = (t.b >> 1) & 0xFFFFu; pper = pdist; pdone = 0; }
            } else {
                BSIG_SLEEP_IF_ALL(dn == 0);
            }
        }
        // the previous turn's match bytes
        BSIG_VM_DRAIN();
        if (dn) {
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (uint32_t k = 0; k < kTurn / 16; ++k) {
                if (16 * k < dn) {
                    const uint32_t cnt = dn - 16 * k < 16u ? dn - 16 * k : 16u;
#if !BSIG_ABL_NO_MSTORE
                    acc_append(A, out, op, low_bytes(v[k].a, cnt), cnt > 8u ? low_bytes(v[k].b, cnt - 8u) : 0ull, cnt);
#else
                    op += cnt;
#endif
                }
            }
            dn = 0;
        }
        if (!tail && out_len - op < kTail) {
            // the rest of the block byte by byte: what the accumulator holds first
            for (uint32_t q = 0; q < A.fill; ++q) out[op - A.fill + q] = (uint8_t)((q < 8u ? A.a >> (8u * q) : A.b >> (8u * (q - 8u))) & 0xFFu);
            A.a = A.b = 0; A.fill = 0;
            tail = true;
        }
        if (stop) break;
        if (tail) {
            for (uint32_t q = 0; q < nlit; ++q) out[op + q] = (uint8_t)(lit >> (8 * q));
            op += nlit;
            if (pend) {
                const uint8_t *from = out + op - pdist;
                for (uint32_t k = 0; k < pend; ++k) out[op + k] = from[k];      // byte by byte: overlaps repeat
                op += pend;
                pend = 0;
            }
            continue;
        }
#if !BSIG_ABL_NO_LSTORE
        acc_append(A, out, op, lit, 0ull, nlit);
#else
        op += nlit;
#endif
        if (pend) {
            // up to kTurn bytes of the match per turn, in 16-byte loads (every lane's load is an access of its own)
            uint32_t n = pend < kTurn ? pend : kTurn;
            if (pdist < 8) {
                // short period (runs, 2- and 3-byte patterns): the first 8 bytes come from a pattern built in
                // registers out of the last bytes out -- they are at hand: the accumulator and the upper half of
                // the chunk in front of it -- ; behind them the same bytes repeat at a distance that is a multiple of
                // the period and >= 8, so the rest is ordinary moves.
                const uint32_t at = 8u + A.fill - pdist;                      // byte offset in (prev, a, b): 1..22
                const uint32_t sh = 8u * (at & 7u);
                const uint64_t w0 = at < 8u ? A.prev : at < 16u ? A.a : A.b, w1 = at < 8u ? A.a : at < 16u ? A.b : 0ull;
                const uint64_t w = (w0 >> sh) | (sh ? w1 << (64u - sh) : 0ull);   // its low `pdist` bytes are the period
                uint64_t pat = 0;
                uint32_t j = 0;
#if defined(__HIPCC__)
#pragma unroll
#endif
                for (int i = 0; i < 8; ++i) {
                    pat |= ((w >> (8 * j)) & 0xFFull) << (8 * i);
                    j = j + 1 == pdist ? 0 : j + 1;
                }
                const uint32_t m = n < 8u ? n : 8u;
#if !BSIG_ABL_NO_LSTORE
                acc_append(A, out, op, low_bytes(pat, m), 0ull, m);
#else
                op += m;
#endif
                pend -= m; n -= m;
                // smallest multiple of the period >= 8 (periods 1..7: 8 8 9 8 10 12 14)
                pdist = (0xECA8988u >> (4u * (pdist - 1u))) & 15u;
                pper = pdist;
                pdone = 0;
            }
            // a source that overlaps the destination (period < the slice): the bytes repeat with
            // that period, so once enough of them are out the SAME bytes lie further back as
            // well -- the distance doubles (it stays a multiple of the period) -- and until then
            // the slice stops where the source would run into it.  Every slice is then loads
            // from bytes that are all there, taken up next turn: no load-wait-store chain, which
            // for the bare reads of the north star's file (a 36-byte record repeats most of the
            // previous one: distances below 64 are the rule) was a quarter of a turn.
            if (pdist < kTurn && pdone + pper >= 2u * pdist) pdist *= 2u;
            // (what a load reads past the slice may not be written yet: it is masked off when it is taken up)
            if (pdist < n) n = pdist;                                    // (>= 8: pdist >= 8 here)
            // the source must be in MEMORY: its last bytes may still be in the accumulator, which then goes out
            // as it is (the same chunk is stored again when it is full)
            if (n && pdist - n < A.fill) store128(out + (op - A.fill), W16{A.a, A.b});
#if BSIG_ABLATE == 6 || BSIG_ABLATE == 7
            // (6 / 7: the match loads read 8 KB / 256 B FURTHER BACK than they should -- as many loads, but never from a
            // cache line that is still being written)
            const uint32_t back_ = BSIG_ABLATE == 6 ? 8192u : 256u, src_ = op - pdist;
            const uint8_t *from = out + (src_ > back_ ? src_ - back_ : 0u);
#else
            const uint8_t *from = out + op - pdist;
#endif
#if defined(__HIPCC__)
#pragma unroll
#endif
            for (uint32_t k = 0; k < kTurn / 16; ++k) {
#if !BSIG_ABL_NO_MLOAD
                if (16 * k < n) v[k] = load128(from + 16 * k);
#else
                if (16 * k < n) v[k] = W16{(uint64_t)(from - out), k};
#endif
            }
            dn = n;
            pdone += n;
            pend -= n;
        }
    }
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host: inflates one raw DEFLATE stream of in_len bytes into exactly out_len bytes -- the producer's tokens of the
// whole block, then the consumer over them (the device runs the same two functions side by side, devdecode.hip:
// k_inflate).  T: first-level tables; lens: kLensBytes of scratch, 8-byte aligned.  Returns OK or an ERR_ code; on
// an error the bytes the tokens before it describe are written.
inline int inflate_block(const uint8_t *in_p, uint32_t in_len, uint8_t *out, uint32_t out_len, LaneTables &T, uint8_t *lens,
                         LaneProf *prof = nullptr)
{
    std::vector<Token> toks;
    ChanVec ch{&toks, 0};
    const int rc = produce(in_p, in_len, out_len, T, lens, ch, prof);
    consume(out, out_len, ch);
    return rc;
}
#endif

}  // namespace bsig_inflate
#endif
