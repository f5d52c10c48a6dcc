// gfx950 (MI355X, CDNA4) kernels for the bamsignals interval-counting hot path.
//
// Replaces the per-read loop of overlapAndPileup<T> (reference src/bamsignals.cpp:240-291)
// with one workgroup per *tile* of a range: bins staged in LDS, 16-B coalesced loads of the
// read columns, LDS integer atomics for the histogram, 16-B coalesced stores of the result.
// Pure 32-bit integer work, HBM-bound: no MFMA anywhere.
//
//   k_profile   Pileupper::setRead/pileup       src/bamsignals.cpp:326-363  (binsize >= 1)
//   k_count     the same with binsize <= 0      src/bamsignals.cpp:148-169, 349-363
//   k_coverage  Coverager::setRead/pileup+cumsum src/bamsignals.cpp:392-438, 464-470
//   k_cigar_end bam_endpos - 1 from packed CIGAR (htslib; call site src/bamsignals.cpp:16-18)
//   k_span_hist / k_scatter / k_build_idx       one-time layout of the reads in HBM (bsig_types.h)
//   k_visits    counts read visits for the roofline figure
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "bsig_types.h"
#include "kernels.h"

#ifdef BSIG_STAMPS
// Diagnostic build only (libbamsignals_hip_stamps.so, scripts/stamps.py): per-workgroup
// s_memrealtime stamps (100 MHz, one counter for the whole chip) of k_profile's phases.  The shipped library has no stamp code.
__device__ unsigned long long *g_stamp_buf = nullptr;
__device__ int g_ablate = 0;     // bit 0: skip the read streaming; bit 1: skip the global stores
#define BSIG_STAMP(k)                                                                       \
    do {                                                                                    \
        if (g_stamp_buf && threadIdx.x == 0) {                                              \
            unsigned long long t_;                                                          \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
            g_stamp_buf[(size_t)blockIdx.x * 8 + (k)] = t_;                                 \
        }                                                                                   \
    } while (0)
#define BSIG_ABLATE(bit) (g_ablate & (bit))
#else
#define BSIG_STAMP(k) do { } while (0)
#define BSIG_ABLATE(bit) 0
#endif

namespace {

constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------

// The filter of Pileupper::setRead (src/bamsignals.cpp:328-333) == Coverager::setRead (:394-399), in two parts:
// what only reads flag and mapq (:328-331) -- a property of the (flag, mapq) pair, evaluated per read for the
// span classes and once per code and launch for the packed class (build_ptab) -- ...
__device__ __forceinline__ bool fm_rejected(const BsigKParams &P, uint32_t fm)
{
    const uint32_t nf = ~(fm & 0xFFFFu);                 // ~flag after int promotion
    const int mapq = (int)((fm >> 16) & 0xFFu);
    return (mapq < P.mapqual) | ((P.requiredF & nf) != 0u) | ((P.filteredF & nf) == 0u);
}
// ... and what reads the template length (:332-333)
// bamCount's whole per-read work without a branch (ref: src/bamsignals.cpp:326-363 with binsize = the range): `acc`
// counts the read in its low half if it is one, passes the filters and has its 5' end in [glo, glo + gn), and in its
// high half too if it lies on the reverse strand (a lane sees at most 32,768 reads of a tile, the heavy-tile ceiling)
__device__ __forceinline__ void count_one(const BsigKParams &P, int glo, int gn, int p, int e, bool neg, bool rej, int tl,
                                          bool valid, uint32_t &acc)
{
    bool ok = valid & !rej;
    int offset = P.shift;
    if (P.has_tlen_filter | P.midpoint) {                  // (uniform)
        const int a = tl < 0 ? -tl : tl;
        if (P.has_tlen_filter) ok = ok & (a >= P.tf0) & (a <= P.tf1);
        if (P.midpoint) offset += a >> 1;
    }
    const int p5 = neg ? e - offset : p + offset;
    ok = ok & ((unsigned)(p5 - glo) < (unsigned)gn);
    acc += ok ? (neg ? 0x10001u : 1u) : 0u;
}

__device__ __forceinline__ bool tlen_rejected(const BsigKParams &P, int32_t tl)
{
    if (!P.has_tlen_filter) return false;
    const int a = tl < 0 ? -tl : tl;
    return (a < P.tf0) | (a > P.tf1);
}

// bamProfile's per-read work (ref: src/bamsignals.cpp:326-363) on a tile image of 16-bit cells, two per LDS dword.
template <bool SS>
struct ProfileOne {
    const BsigKParams &P;
    uint32_t *cnt;
    int loc, len, c0, nc, sh;
    bool neg_range;
    __device__ __forceinline__ void operator()(int p, int e, bool neg /* isNegStrand, :11-13 */, bool rej, int tl, bool valid) const
    {
        constexpr int S = SS ? 2 : 1;
        if (!valid || rej || tlen_rejected(P, tl)) return;             // :328-333
        const int a = tl < 0 ? -tl : tl;
        const int offset = P.midpoint ? (a >> 1) + P.shift : P.shift;  // :339
        const int p5 = neg ? e - offset : p + offset;                  // :340-344
        int rel = p5 - loc;                                            // :351
        if ((unsigned)rel >= (unsigned)len) return;                    // :353
        int anti = neg ? 1 : 0;
        if (neg_range) { rel = len - rel - 1; anti ^= 1; }             // :356-359
        const int cell = P.binsize == 1 ? rel
                                        : (int)(__umulhi((uint32_t)rel, P.div_magic) >> P.div_shift);
        const int lc = cell - c0;
        if ((unsigned)lc < (unsigned)nc) {
            const int k = sh + lc * S + (SS ? anti : 0);               // :361-362
            atomicAdd(&cnt[k >> 1], 1u << ((k & 1) << 4));
        }
    }
    // Four reads of the packed class at once for bins of one base: straight arithmetic on the packed word, one masked LDS add at the end.  With d = (word - base) & mask the 5' end relative to the range is
    // d + cp on the forward strand and d + span + cp - 2 shift on the reverse one; the tile's cell is that minus c0,
    // or counted from the range's other end on a reverse-strand range (a tile lies inside its range, so the cell
    // test is the range test, :351-353).
    __device__ __forceinline__ void quad(const uint4 &w, const int4 &t, uint32_t dj, uint32_t nj, int base,
                                         const uint8_t *__restrict__ ptab) const
    {
        const uint32_t b0 = ptab[w.x >> 23], b1 = ptab[w.y >> 23], b2 = ptab[w.z >> 23], b3 = ptab[w.w >> 23];
        if (P.binsize != 1) {                                          // (uniform)
            auto dec = [&](uint32_t x, uint32_t b, int tl, bool valid) {
                const int pos = base + (int)((x - (uint32_t)base) & (((uint32_t)1 << BSIG_PACK_POS_BITS) - 1u));
                (*this)(pos, pos + (int)((x >> BSIG_PACK_POS_BITS) & 0xFFu), (b & 2u) != 0u, (b & 1u) != 0u, tl, valid);
            };
            dec(w.x, b0, t.x, dj < nj);
            dec(w.y, b1, t.y, dj + 1u < nj);
            dec(w.z, b2, t.z, dj + 2u < nj);
            dec(w.w, b3, t.w, dj + 3u < nj);
            return;
        }
        const bool tl_rule = (P.has_tlen_filter | P.midpoint) != 0;    // (uniform)
        if (neg_range) {
            const int K = len - 1 - c0 - (base - loc + P.shift);
            if (tl_rule) four<true, true>(w, t, b0, b1, b2, b3, dj, nj, base, K);
            else four<true, false>(w, t, b0, b1, b2, b3, dj, nj, base, K);
        } else {
            const int K = base - loc + P.shift - c0;
            if (tl_rule) four<false, true>(w, t, b0, b1, b2, b3, dj, nj, base, K);
            else four<false, false>(w, t, b0, b1, b2, b3, dj, nj, base, K);
        }
    }
    // TL: a template-length rule applies (:332-333 the filter, :339 the midpoint: the 5' end moves by |tlen| / 2)
    template <bool REV, bool TL>
    __device__ __forceinline__ void four(const uint4 &w, const int4 &t, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t dj,
                                         uint32_t nj, int base, int K) const
    {
        const uint32_t cd = (uint32_t)(-2 * P.shift);
        auto rd = [&](uint32_t x, uint32_t b, int tl, uint32_t k) {
            const uint32_t d = (x - (uint32_t)base) & (((uint32_t)1 << BSIG_PACK_POS_BITS) - 1u);
            const uint32_t sp = (x >> BSIG_PACK_POS_BITS) & 0xFFu;
            const uint32_t nm = (uint32_t)((int32_t)(b << 30) >> 31);
            uint32_t rj = (uint32_t)((int32_t)(b << 31) >> 31), h = 0;
            if (TL) {
                const int a = tl < 0 ? -tl : tl;
                if (P.has_tlen_filter) rj |= ((a < P.tf0) | (a > P.tf1)) ? 0xFFFFFFFFu : 0u;
                if (P.midpoint) h = (uint32_t)(a >> 1);
            }
            // 5' end from base + shift: pos + h on the forward strand, pos + span - 2 shift - h on the reverse one
            const uint32_t fwd = d + h + (nm & (sp + cd - 2u * h));
            const uint32_t lc = (REV ? (uint32_t)K - fwd : (uint32_t)K + fwd) | rj;       // (rejected: beyond every tile)
            const bool ok = (dj + k < nj) & (lc < (uint32_t)nc);
            // antisense: reverse-strand read on a forward range, forward read on a reverse one
            const uint32_t cell = SS ? (uint32_t)sh + 2u * lc + ((REV ? ~nm : nm) & 1u) : (uint32_t)sh + lc;
            // (masked, not redirected: the many reads of a window that miss the tile would all meet in one dummy
            // cell, and LDS atomics on one address take their turns -- config 5's 1-kb tiles: 0.16 -> 0.30 ms)
            if (ok) atomicAdd(&cnt[cell >> 1], 1u << ((cell << 4) & 31u));     // (odd cell: the dword's high half)
        };
        rd(w.x, b0, t.x, 0u);
        rd(w.y, b1, t.y, 1u);
        rd(w.z, b2, t.z, 2u);
        rd(w.w, b3, t.w, 3u);
    }
};

// bamProfile's per-read work on the wide-bin image of k_profile_small: 32-bit cells in the lane's own replica.
template <bool SS>
struct SmallOne {
    const BsigKParams &P;
    int32_t *mine;
    int loc, len, c0, nc;
    bool neg_range;
    __device__ __forceinline__ void operator()(int p, int e, bool neg, bool rej, int tl, bool valid) const
    {
        constexpr int S = SS ? 2 : 1;
        if (!valid || rej || tlen_rejected(P, tl)) return;
        const int a = tl < 0 ? -tl : tl;
        const int offset = P.midpoint ? (a >> 1) + P.shift : P.shift;
        const int p5 = neg ? e - offset : p + offset;
        int rel = p5 - loc;
        if ((unsigned)rel >= (unsigned)len) return;
        int anti = neg ? 1 : 0;
        if (neg_range) { rel = len - rel - 1; anti ^= 1; }
        const int cell = P.binsize == 1 ? rel : (int)(__umulhi((uint32_t)rel, P.div_magic) >> P.div_shift);
        const int lc = cell - c0;
        if ((unsigned)lc < (unsigned)nc) atomicAdd(&mine[lc * S + (SS ? anti : 0)], 1);
    }
    // four packed reads at once, as ProfileOne::quad, with the bin of the range-oriented position by an exact magic
    // multiply.  The launch is bound by its vector instructions (0.98 of it once the replicas were right: `SQ_INSTS_VALU`),
    // so the read body is counted out: the position is taken relative to the TILE's first base (K carries
    // -c0 * binsize), which makes "inside the range" and "one of this tile's cells" ONE unsigned compare against the
    // tile's length in bases and the cell its own index; the table byte is used as it is (bit 0 rejected, bit 1 reverse
    // strand: two v_bfe_i32 make the masks); the strand's half of the address is a mask and an AND; for ranges
    // shorter than 32,768 bases the bin is a 24-bit multiply at the vector unit's full rate instead of the
    // quarter-rate v_mul_hi_u32.  19 vector instructions a read where the round began with 33 and its middle had 26.
    __device__ __forceinline__ void quad(const uint4 &w, const int4 &t, uint32_t dj, uint32_t nj, int base,
                                         const uint8_t *__restrict__ ptab) const
    {
        const uint32_t b0 = ptab[w.x >> 23], b1 = ptab[w.y >> 23], b2 = ptab[w.z >> 23], b3 = ptab[w.w >> 23];
        // (uniform, like everything these branches ask.  Three forms only -- the orientation of the range is a sign and
        // a strand mask, not a fourth template argument: the kernel inlines this at four places, and 32 copies of the
        // read body were 35 KB of code that ran a third SLOWER than round 4's 16)
        const bool tl_rule = (P.has_tlen_filter | P.midpoint) != 0;
        const bool narrow = len < 32768 && P.div_s15 != 0;
        const int A = base - loc + P.shift;
        const int first = c0 * P.binsize;                                  // the tile's first base in range orientation
        const int K = (neg_range ? len - 1 - A : A) - first, sgn = neg_range ? -1 : 1;
        const int rest = len - first, mine_bases = nc * P.binsize;
        const uint32_t tile_bases = (uint32_t)(rest < mine_bases ? rest : mine_bases);
        if (tl_rule) four<true, false>(w, t, b0, b1, b2, b3, dj, nj, base, K, sgn, tile_bases);
        else if (narrow) four<false, true>(w, t, b0, b1, b2, b3, dj, nj, base, K, sgn, tile_bases);
        else four<false, false>(w, t, b0, b1, b2, b3, dj, nj, base, K, sgn, tile_bases);
    }
    template <bool TL, bool NARROW>
    __device__ __forceinline__ void four(const uint4 &w, const int4 &t, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t dj,
                                         uint32_t nj, int base, int K, int sgn, uint32_t tile_bases) const
    {
        const int cd = -2 * P.shift;                                   // (binsize > 1 here: bins of one base run k_profile)
        const int flip = sgn < 0 ? -1 : 0;
        int Kv = K;
        asm volatile("" : "+v"(Kv));                                   // (in a vector register once: v_mad_i32_i24 takes one scalar operand)
        char *const image = reinterpret_cast<char *>(mine);
        auto rd = [&](uint32_t x, uint32_t b, int tl, uint32_t k) {
            const int d = (int)((x - (uint32_t)base) & (((uint32_t)1 << BSIG_PACK_POS_BITS) - 1u));
            int spcd = (int)((x >> BSIG_PACK_POS_BITS) & 0xFFu) + cd, h = 0;
            const int rev = __builtin_amdgcn_sbfe((int)b, 1, 1);       // all ones: reverse strand
            int rj = __builtin_amdgcn_sbfe((int)b, 0, 1);              // all ones: rejected by flag / mapq
            if (TL) {
                const int a = tl < 0 ? -tl : tl;
                if (P.has_tlen_filter) rj |= ((a < P.tf0) | (a > P.tf1)) ? -1 : 0;
                if (P.midpoint) { h = a >> 1; spcd -= 2 * h; }
            }
            // (|spcd| and fwd < 2^23: a window that takes this path is narrower than 32,768 bases, and so are shift and h)
            const int fwd = (rev & spcd) + d + h;
            const uint32_t rel = (uint32_t)(__mul24(fwd, sgn) + Kv) | (uint32_t)rj;     // from the tile's first base, range orientation
            const uint32_t cell = NARROW ? __umul24(rel, P.div_m15) >> P.div_s15 : __umulhi(rel, P.div_magic) >> P.div_shift;
            const bool ok = (dj + k < nj) & (rel < tile_bases);
            if (ok) atomicAdd(reinterpret_cast<int32_t *>(image + (SS ? (cell << 3) + (uint32_t)((rev ^ flip) & 4) : cell << 2)), 1);
        };
        rd(w.x, b0, t.x, 0u);
        rd(w.y, b1, t.y, 1u);
        rd(w.z, b2, t.z, 2u);
        rd(w.w, b3, t.w, 3u);
    }
};

// bamCoverage's per-read work (ref: src/bamsignals.cpp:392-438): +1 where the read begins to cover the tile, -1 behind
// its last covered cell, in the tile's difference array of signed 16-bit cells (two per LDS dword: see k_coverage).
struct CoverOne {
    const BsigKParams &P;
    int32_t *lds;
    int loc, c0, nc, sh;
    bool neg_range;
    int rend1;                              // last base of the range
    __device__ __forceinline__ void operator()(int p, int e, bool neg, bool rej, int tl, bool valid) const
    {
        if (!valid || rej || tlen_rejected(P, tl)) return;            // :394-399
        int start = p, end = e;                                       // :401-403
        if (P.tspan) {                                                // :404-413
            if (neg && tl < 0) start = end + tl + 1;
            else if (!neg && tl > 0) end = start + tl - 1;
        }
        // covered cells [ra, rb] in range orientation (:423-436), then relative to the tile
        const int ra = neg_range ? rend1 - end : start - loc;
        const int rb = neg_range ? rend1 - start : end - loc;
        const int la = ra - c0, lb = rb - c0;
        if (la >= nc || lb < 0) return;                               // :420
        const int ka = sh + (la > 0 ? la : 0);
        atomicAdd(&lds[ka >> 1], (ka & 1) ? 65536 : 1);
        if (lb + 1 < nc) {
            const int kb = sh + lb + 1;
            atomicAdd(&lds[kb >> 1], (kb & 1) ? -65536 : -1);
        }
    }
    // Four reads of the packed class at once (the launch is bound by its vector instructions: PMC, config 3).
    // Straight arithmetic on the packed word -- first covered cell
    // la = d + A (or B - d - span on a reverse-strand range), last lb = la + span, d = (word - base) & mask -- and two
    // masked LDS adds at the end.
    __device__ __forceinline__ void quad(const uint4 &w, const int4 &t, uint32_t dj, uint32_t nj, int base,
                                         const uint8_t *__restrict__ ptab) const
    {
        const uint32_t b0 = ptab[w.x >> 23], b1 = ptab[w.y >> 23], b2 = ptab[w.z >> 23], b3 = ptab[w.w >> 23];
        // (sh rides in K: k = cell index in the image, first covered cell max(ka, sh), one past the last kb)
        const bool tl_rule = (P.has_tlen_filter | P.tspan) != 0;       // (uniform)
        if (neg_range) {
            const int K = rend1 - c0 - base + sh;
            if (tl_rule) four<true, true>(w, t, b0, b1, b2, b3, dj, nj, base, K);
            else four<true, false>(w, t, b0, b1, b2, b3, dj, nj, base, K);
        } else {
            const int K = base - loc - c0 + sh;
            if (tl_rule) four<false, true>(w, t, b0, b1, b2, b3, dj, nj, base, K);
            else four<false, false>(w, t, b0, b1, b2, b3, dj, nj, base, K);
        }
    }
    // TL: a template-length rule applies (:398-399 the filter; :404-413 paired.end = "extend": a forward read with
    // tlen > 0 covers tlen bases from its start, a reverse one with tlen < 0 covers -tlen bases up to its end)
    template <bool REV, bool TL>
    __device__ __forceinline__ void four(const uint4 &w, const int4 &t, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, uint32_t dj,
                                         uint32_t nj, int base, int K) const
    {
        const int hi = sh + nc;
        auto rd = [&](uint32_t x, uint32_t b, int tl, uint32_t k) {
            const int d = (int)((x - (uint32_t)base) & (((uint32_t)1 << BSIG_PACK_POS_BITS) - 1u));
            const int sp = (int)((x >> BSIG_PACK_POS_BITS) & 0xFFu);
            int rj = (int32_t)(b << 31) >> 31;                             // all ones: rejected
            int s0 = d, e0 = d + sp;                                       // first and last covered base, from `base`
            if (TL) {
                const int a = tl < 0 ? -tl : tl;
                if (P.has_tlen_filter) rj |= ((a < P.tf0) | (a > P.tf1)) ? -1 : 0;
                if (P.tspan) {
                    const int nm = (int32_t)(b << 30) >> 31;               // all ones: reverse strand
                    const int s1 = e0 + tl + 1, e1 = d + tl - 1;
                    s0 = (nm & (tl < 0 ? -1 : 0)) ? s1 : s0;
                    e0 = (~nm & (tl > 0 ? -1 : 0)) ? e1 : e0;
                }
            }
            const int ka = REV ? K - e0 : K + s0;
            const int kb = ((REV ? K - s0 : K + e0) | rj) + 1;             // (a rejected read ends before the tile)
            const bool ok = (dj + k < nj) & (ka < hi) & (kb > sh);
            const bool ok2 = ok & (kb < hi);
            const int ca = ka > sh ? ka : sh;
            // an odd cell is the high half of its dword: the shifter takes (k << 4) & 31 = 16 for odd k
            // (and -1 is all ones: shifted by 16 it is -65536 in the dword's arithmetic mod 2^32).  The adds are
            // masked, not redirected to a dummy cell: see ProfileOne.
            uint32_t *img = reinterpret_cast<uint32_t *>(lds);
            if (ok) atomicAdd(&img[ca >> 1], 1u << (((uint32_t)ca << 4) & 31u));
            if (ok2) atomicAdd(&img[kb >> 1], 0xFFFFFFFFu << (((uint32_t)kb << 4) & 31u));
        };
        rd(w.x, b0, t.x, 0u);
        rd(w.y, b1, t.y, 1u);
        rd(w.z, b2, t.z, 2u);
        rd(w.w, b3, t.w, 3u);
    }
};

struct CountOne {
    const BsigKParams &P;
    int glo, gn;
    uint32_t &acc;
    __device__ __forceinline__ void operator()(int p, int e, bool neg, bool rej, int tl, bool valid) const
    {
        count_one(P, glo, gn, p, e, neg, rej, tl, valid, acc);
    }
    // four reads of the packed class (words w; read k is one of the window's iff dj + k < nj, unsigned -- not asked
    // at all on an INNER pass, one that lies inside the window with all its reads).  The count family has a table of
    // its own, one DWORD per code (build_ctab: bit 0 reverse strand, bit 31 rejected), which the arithmetic takes as
    // it comes: as a 24-bit factor it is the strand bit, and-ed with the sign bit it is the rejection.  Without a
    // template-length rule the 5' end relative to the interval comes straight out of the word:
    // pos - glo + shift = d + cp and end - glo - shift = d + span + cp + cd with d = (word - base) & mask:
    // rel = d + cp + strand * (span + cd), 13 vector instructions a read (11 on an inner pass; 19 in round 4).
    __device__ __forceinline__ void quad(const uint4 &w, const int4 &t, uint32_t dj, uint32_t nj, int base,
                                         const uint32_t *__restrict__ ctab, bool inner) const
    {
        const uint32_t b0 = ctab[w.x >> 23], b1 = ctab[w.y >> 23], b2 = ctab[w.z >> 23], b3 = ctab[w.w >> 23];
        if (P.has_tlen_filter | P.midpoint) {                  // (uniform)
            auto dec = [&](uint32_t x, uint32_t b, int tl, bool valid) {
                const int pos = base + (int)((x - (uint32_t)base) & (((uint32_t)1 << BSIG_PACK_POS_BITS) - 1u));
                count_one(P, glo, gn, pos, pos + (int)((x >> BSIG_PACK_POS_BITS) & 0xFFu), (b & 1u) != 0u, (b >> 31) != 0u, tl, valid, acc);
            };
            dec(w.x, b0, t.x, dj < nj);
            dec(w.y, b1, t.y, dj + 1u < nj);
            dec(w.z, b2, t.z, dj + 2u < nj);
            dec(w.w, b3, t.w, dj + 3u < nj);
            return;
        }
        const int cp = base - glo + P.shift, cd = -2 * P.shift;
        auto rel_of = [&](uint32_t x, uint32_t b) {
            const int d = (int)((x - (uint32_t)base) & (((uint32_t)1 << BSIG_PACK_POS_BITS) - 1u));
            const int spcd = (int)((x >> BSIG_PACK_POS_BITS) & 0xFFu) + cd;
            // (|spcd| < 2^23: a window that takes this path is narrower than 32,768 bases, and so is the shift)
            return ((uint32_t)(__mul24((int)b, spcd) + d + cp)) | (b & 0x80000000u);
        };
        if (inner) {
            acc += rel_of(w.x, b0) < (uint32_t)gn ? (b0 << 16 | 1u) : 0u;
            acc += rel_of(w.y, b1) < (uint32_t)gn ? (b1 << 16 | 1u) : 0u;
            acc += rel_of(w.z, b2) < (uint32_t)gn ? (b2 << 16 | 1u) : 0u;
            acc += rel_of(w.w, b3) < (uint32_t)gn ? (b3 << 16 | 1u) : 0u;
        } else {
            acc += ((dj < nj) & (rel_of(w.x, b0) < (uint32_t)gn)) ? (b0 << 16 | 1u) : 0u;
            acc += ((dj + 1u < nj) & (rel_of(w.y, b1) < (uint32_t)gn)) ? (b1 << 16 | 1u) : 0u;
            acc += ((dj + 2u < nj) & (rel_of(w.z, b2) < (uint32_t)gn)) ? (b2 << 16 | 1u) : 0u;
            acc += ((dj + 3u < nj) & (rel_of(w.w, b3) < (uint32_t)gn)) ? (b3 << 16 | 1u) : 0u;
        }
    }
};


constexpr int kPackChunk = 1 << BSIG_PACK_POS_BITS;        // bases a packed word's position bits span
constexpr uint32_t kPackPosMask = (uint32_t)kPackChunk - 1u;

// Per launch and workgroup: what the flag/mapq filter says about every code of the packed class, one byte per
// code in LDS (bit 0: rejected, bit 1: reverse strand).  The pair table is 2 KB and stays in L2.
// The packed class's filter table (per code: bit 0 rejected by mapq / flag masks, bit 1 reverse strand) depends on
// the file's pair table and the call's parameters only: it is made ONCE PER PLAN (k_make_ptab, 512 bytes on the
// device, BsigKParams::ptab) and every workgroup copies it into its LDS -- one 16-byte load for half the lanes.
// Every workgroup used to build it for itself: 90 vector instructions per wave, 13 % of what a north-star tile
// issues, in launches that are bound by exactly those (NOTES_r04.md, section 9).
template <int NT>
__device__ __forceinline__ void build_ptab(uint8_t *ptab, const BsigReadsDev &R, const BsigKParams &P, int tid)
{
    (void)R;
    const uint4 *src = reinterpret_cast<const uint4 *>(P.ptab);
    uint4 *dst = reinterpret_cast<uint4 *>(ptab);
    for (int v = tid; v < BSIG_PACK_CODES / 16; v += NT) dst[v] = src[v];
}
// ... and the count family's form of it: one dword per code, bit 0 reverse strand, bit 31 rejected (CountOne::quad)
template <int NT>
__device__ __forceinline__ void build_ctab(uint32_t *ctab, const BsigKParams &P, int tid)
{
    const uint32_t *src = reinterpret_cast<const uint32_t *>(P.ptab);
    for (int v = tid; v < BSIG_PACK_CODES / 4; v += NT) {
        const uint32_t b = src[v];
        auto e = [](uint32_t x) { return ((x >> 1) & 1u) | (x << 31); };
        reinterpret_cast<uint4 *>(ctab)[v] = make_uint4(e(b & 0xFFu), e((b >> 8) & 0xFFu), e((b >> 16) & 0xFFu), e(b >> 24));
    }
}
__global__ __launch_bounds__(128) void k_make_ptab(const BsigReadsDev R, const BsigKParams P, uint8_t *__restrict__ out)
{
    const int v = threadIdx.x;                              // four codes each: BSIG_PACK_CODES = 4 x 128
    uint32_t word = 0;
    if (4 * v < R.n_codes) {
        const uint4 f = reinterpret_cast<const uint4 *>(R.fmtab)[v];
        auto b = [&](uint32_t fm) { return (uint32_t)fm_rejected(P, fm) | ((fm >> 3) & 2u); };     // 0x10 >> 3
        word = b(f.x) | b(f.y) << 8 | b(f.z) << 16 | b(f.w) << 24;
    }
    reinterpret_cast<uint32_t *>(out)[v] = word;
}
static_assert(BSIG_PACK_CODES == 4 * 128, "k_make_ptab: four codes per thread of one 128-thread workgroup");

// The packed class's window of a tile: the reads whose pos lies in the bucket-rounded window [rlo, rhi) of the
// reference, walked in chunks of kPackChunk bases (nearly always one): inside a chunk that starts at `base`,
// pos = base + ((word - base) & kPackPosMask).
struct PackedWin {
    int base;          // start of the first chunk (reference coordinate, a multiple of the bucket width)
    int n_chunks;      // 0: nothing to read
};

// Reads of class C that can touch the genomic interval [tlo, thi) lie in [j_lo, j_hi).
// pos in [tlo - ext - maxspan + 1, thi + ext), rounded outwards to index buckets.
__device__ __forceinline__ bool class_window(const BsigClassCols &C, const BsigWorkItem &w,
                                             int64_t tlo, int64_t thi, int ext,
                                             uint32_t &j_lo, uint32_t &j_hi)
{
    int64_t wlo = tlo - ext - C.maxspan + 1;
    int64_t whi = thi + ext;
    const int64_t ref_bp = (int64_t)(w.units_strand & BSIG_ITEM_UNITS_MASK) << BSIG_REF_UNIT_SHIFT;
    if (wlo < 0) wlo = 0;
    if (whi > ref_bp) whi = ref_bp;
    if (wlo >= whi) return false;
    const uint64_t g0 = (uint64_t)w.ref_unit0 << BSIG_REF_UNIT_SHIFT;
    const uint64_t b_lo = (g0 + (uint64_t)wlo) >> C.kshift;
    const uint64_t b_hi = (g0 + (uint64_t)whi - 1) >> C.kshift;
    j_lo = C.idx[b_lo];
    j_hi = C.idx[b_hi + 1];
    return j_lo < j_hi;
}

// genomic interval covered by a profile/coverage tile (cells [c0, c0+nc) in range orientation)
__device__ __forceinline__ void tile_interval(const BsigWorkItem &w, int binsize, bool neg_range,
                                              int64_t &tlo, int64_t &thi)
{
    const int64_t a = (int64_t)w.c0 * binsize;
    int64_t b = ((int64_t)w.c0 + w.nc) * binsize;
    if (b > w.len) b = w.len;
    if (neg_range) { tlo = (int64_t)w.loc + w.len - b; thi = (int64_t)w.loc + w.len - a; }
    else           { tlo = (int64_t)w.loc + a;         thi = (int64_t)w.loc + b; }
}

// Inclusive prefix sum over the 64 lanes of a wave in six DPP adds (no LDS traffic): a
// Hillis-Steele scan inside each row of 16 lanes (row_shr 1,2,4,8; lanes shifted in from outside
// the row read 0), then row 0 -> row 1 and row 2 -> row 3 (row_bcast:15), then lanes 0-31 ->
// lanes 32-63 (row_bcast:31).
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);    // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);    // row_bcast:31 -> rows 2, 3
    return v;
}

// Pileup kernels take (items, n_tiles, out, windows) first: with -amdgpu-kernarg-preload-count=8
// (Makefile) those arrive in SGPRs, so the work-item load is issued with the first instruction
// instead of behind a kernarg fetch (config 2: 24.9 -> 24.6 us).
// Workgroups are dealt round-robin to the 8 XCDs (blocks b and b+8 share an L2), tiles are sorted
// by position: give every XCD one contiguous run of the tile list, so that neighbouring tiles,
// whose read windows overlap, meet in the same L2.  A bijection of [0, n); any mapping would be
// correct, this one is only faster (no assumption about WHICH XCD a block lands on).
__device__ __forceinline__ uint32_t tile_of_block(uint32_t b, uint32_t n)
{
    const uint32_t q = n >> 3, r = n & 7u, x = b & 7u, k = b >> 3;
    return x * q + (x < r ? x : r) + k;
}

template <int NT>
__device__ __forceinline__ void block_sync()
{
    // a 64-thread workgroup is one wave: LDS operations of a wave execute in order
    if (NT > kWave) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Store nv values that sit at lds[sh .. sh+nv) to out[out_off .. out_off+nv): the LDS image is
// shifted by sh = out_off & 3 so that 16-B LDS vectors line up with 16-B aligned global addresses.
__device__ __forceinline__ void store_vec(int32_t *__restrict__ gbase, int v, int4 x, int sh, int nv)
{
    const int e0 = 4 * v;
    if (e0 >= sh && e0 + 4 <= sh + nv) {
        // the result is written once and never re-read by this launch: non-temporal stores keep
        // it from displacing the read columns in L2 (measured -6 % step time at config 2)
        typedef int v4i_t __attribute__((ext_vector_type(4)));
        v4i_t xv = {x.x, x.y, x.z, x.w};
        __builtin_nontemporal_store(xv, reinterpret_cast<v4i_t *>(gbase + e0));
    } else {
        const int lo = sh, hi = sh + nv;
        if (e0 + 0 >= lo && e0 + 0 < hi) gbase[e0 + 0] = x.x;
        if (e0 + 1 >= lo && e0 + 1 < hi) gbase[e0 + 1] = x.y;
        if (e0 + 2 >= lo && e0 + 2 < hi) gbase[e0 + 2] = x.z;
        if (e0 + 3 >= lo && e0 + 3 < hi) gbase[e0 + 3] = x.w;
    }
}

// genomic interval a work item needs reads for
__device__ __forceinline__ void item_interval(const BsigWorkItem &w, const BsigKParams &P, int mode,
                                              int64_t &tlo, int64_t &thi)
{
    const bool neg_range = (w.units_strand & BSIG_ITEM_NEG) != 0u;
    if (mode == BSIG_MODE_COUNT) { tlo = (int64_t)w.loc + w.c0; thi = tlo + w.nc; }
    else tile_interval(w, mode == BSIG_MODE_COVERAGE ? 1 : P.binsize, neg_range, tlo, thi);
}

// Index lookup for every (tile, class): windows[BSIG_MAX_CLASSES*t + c] = [j_lo, j_hi) of class c's reads
// that can touch tile t.  The device-side counterpart of bam_itr_queryi (src/bamsignals.cpp:267).
// Runs once per plan (bsig_plan_create probes the window sizes for heavy tiles); the pileup kernels look
// their windows up themselves (load_windows).
__global__ void k_resolve(const BsigReadsDev R, const BsigKParams P, int mode,
                          const BsigWorkItem *__restrict__ items, int64_t n_items,
                          uint2 *__restrict__ windows)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = g / BSIG_MAX_CLASSES;
    const int c = (int)(g - t * BSIG_MAX_CLASSES);
    if (t >= n_items) return;
    const BsigWorkItem w = items[t];
    int64_t tlo, thi;
    item_interval(w, P, mode, tlo, thi);
    uint32_t j_lo = 0, j_hi = 0;
    const BsigClassCols &C = R.cls[c];
    if (C.n == 0 || !class_window(C, w, tlo, thi, P.ext, j_lo, j_hi)) { j_lo = 0; j_hi = 0; }
    windows[g] = make_uint2(j_lo, j_hi);
}

// how many tiles hold more reads in their windows than `heavy_reads` (plan-time probe)
__global__ void k_count_heavy(const uint2 *__restrict__ windows, int64_t n_items, int64_t heavy_reads,
                              unsigned long long *__restrict__ count)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_items) return;
    int64_t total = 0;
#pragma unroll
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
        const uint2 w = windows[t * BSIG_MAX_CLASSES + c];
        total += (int64_t)w.y - w.x;
    }
    if (total > heavy_reads) atomicAdd(count, 1ull);
}

// class 1's word (flag (12 bits) | mapq << 12 | (span - 1) << 20, see read_class below) in the layout the
// filters read: flag | mapq << 16
__device__ __forceinline__ uint32_t fm_of_class1(uint32_t w) { return (w & 0xFFFu) | ((w >> 12) & 0xFFu) << 16; }

// four consecutive packed words (one lane's 16-B load) of a chunk that starts at `base`
// a per-read functor may bring its own treatment of four packed reads at once (`quad`): bamCount does
template <typename F, typename = void>
struct has_quad : std::false_type {};
template <typename F>
struct has_quad<F, std::void_t<decltype(&std::remove_reference_t<F>::quad)>> : std::true_type {};

template <typename Tab, typename F>
__device__ __forceinline__ void four_packed(const uint4 &w, const int4 &t, uint32_t j, uint32_t j_lo, uint32_t nj, int base,
                                            const Tab *__restrict__ ptab, F &&one, bool inner = false)
{
    if constexpr (has_quad<F>::value) {
        if constexpr (std::is_same_v<Tab, uint32_t>) one.quad(w, t, j - j_lo, nj, base, ptab, inner);      // (the count family)
        else one.quad(w, t, j - j_lo, nj, base, ptab);
        return;
    }
    // the four table bytes are requested before the first one is used
    const uint32_t b0 = ptab[w.x >> 23], b1 = ptab[w.y >> 23], b2 = ptab[w.z >> 23], b3 = ptab[w.w >> 23];
    const uint32_t dj = j - j_lo;
    auto dec = [&](uint32_t x, uint32_t b, int tl, bool valid) {
        const int pos = base + (int)((x - (uint32_t)base) & kPackPosMask);
        one(pos, pos + (int)((x >> BSIG_PACK_POS_BITS) & 0xFFu), (b & 2u) != 0u, (b & 1u) != 0u, tl, valid);
    };
    dec(w.x, b0, t.x, dj < nj);
    dec(w.y, b1, t.y, dj + 1u < nj);
    dec(w.z, b2, t.z, dj + 2u < nj);
    dec(w.w, b3, t.w, dj + 3u < nj);
}

// Stream the reads of all class windows of one tile through `one(pos, end, neg, rejected, tlen, valid)`
// (`rejected`: by the flag/mapq part of the filter).  Everything a typical tile needs is requested before
// anything is consumed, so the workgroup pays ONE memory round trip for its reads: the first kPre passes
// (kPre * 4 * NT reads) of the packed class, where nearly all reads live, and the first pass of classes 0
// and 1.  Longer windows and the two long-span classes continue in plain loops.
// (The packed class's later chunks -- windows wider than kPackChunk bases -- are walked by packed_later_chunks.)
template <int NT, int kPre = 2, typename Tab, typename F>
__device__ __forceinline__ void for_each_read(const BsigReadsDev &R, const BsigKParams &P,
                                              const uint2 (&win)[BSIG_MAX_CLASSES], int pbase,
                                              const Tab *__restrict__ ptab, int tid, F &&one)
{
    uint4 w0[kPre];
    int4 t0[kPre];
    const BsigClassCols &CP = R.cls[BSIG_CLASS_PACKED];
    const uint32_t jbp = (win[BSIG_CLASS_PACKED].x & ~3u) + 4u * tid;
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const uint32_t j = jbp + 4u * NT * k;
        // (t0[k] is read under P.use_tlen only, which is when it is loaded: setting it to zero otherwise was four vector
        // instructions a pass in launches that are bound by exactly those)
        if (j < win[BSIG_CLASS_PACKED].y) {
            w0[k] = *reinterpret_cast<const uint4 *>(CP.fm + j);
            if (P.use_tlen) t0[k] = *reinterpret_cast<const int4 *>(CP.tlen + j);
        }
    }
    // The rare classes' windows are short -- the north star's tiles see some 20 reads of class 1 (the 5 % of reads
    // with a skipped region) -- and the launches are bound by their vector instructions, which a wave issues for all
    // its lanes or none: four reads per lane made such a window cost a whole pass of four `one`s (160 of a tile's 690
    // vector instructions).  A window of up to NT reads is taken ONE read per lane, requested here, ahead of the
    // packed class's work; a longer one is loaded where it is walked (the registers of a four-read prefetch held
    // across the packed class cost a wave per SIMD).
    const bool small0 = win[0].y - win[0].x <= (uint32_t)NT, small1 = win[1].y - win[1].x <= (uint32_t)NT;      // (uniform)
    int pa = 0, ta = 0, pb = 0, tb = 0;
    uint32_t fa = 0, fb = 0;
    const uint32_t jb0 = win[0].x + (uint32_t)tid, jb1 = win[1].x + (uint32_t)tid;
    if (small0 && jb0 < win[0].y) {
        pa = R.cls[0].pos[jb0];
        fa = R.cls[0].fm[jb0];
        if (P.use_tlen) ta = R.cls[0].tlen[jb0];
    }
    if (small1 && jb1 < win[1].y) {
        pb = R.cls[1].pos[jb1];
        fb = R.cls[1].fm[jb1];
        if (P.use_tlen) tb = R.cls[1].tlen[jb1];
    }
    // The 16-B aligned loads may start before j_lo (possibly on the previous reference) and end
    // after j_hi: only reads in [j_lo, j_hi) count -> `dj < nj` with unsigned wrap-around.
    {   // ---- packed class (span <= 256, a frequent flag/mapq pair): one word per read -------------------
        const uint32_t j_lo = win[BSIG_CLASS_PACKED].x, j_hi = win[BSIG_CLASS_PACKED].y, nj = j_hi - j_lo;
        // a pass whose 4 * NT reads all belong to the window (uniform): nobody has to ask read by read
        const uint32_t jp0 = j_lo & ~3u;
        auto inner_pass = [&](uint32_t first) { return first >= j_lo && first + 4u * NT <= j_hi; };
#pragma unroll
        for (int k = 0; k < kPre; ++k) {
            const uint32_t j = jbp + 4u * NT * k;
            if (j < j_hi) four_packed(w0[k], t0[k], j, j_lo, nj, pbase, ptab, one, inner_pass(jp0 + 4u * NT * k));
        }
        // deeper windows: two passes per trip, both requested before either is consumed
        for (uint32_t j = jbp + 4u * NT * kPre, jf = jp0 + 4u * NT * kPre; j < j_hi; j += 8u * NT, jf += 8u * NT) {
            const uint32_t j2 = j + 4u * NT;
            const uint4 wa = *reinterpret_cast<const uint4 *>(CP.fm + j);
            int4 xa, xb;                                                   // (as t0: read under P.use_tlen only)
            uint4 wb = make_uint4(0, 0, 0, 0);
            if (P.use_tlen) xa = *reinterpret_cast<const int4 *>(CP.tlen + j);
            if (j2 < j_hi) {
                wb = *reinterpret_cast<const uint4 *>(CP.fm + j2);
                if (P.use_tlen) xb = *reinterpret_cast<const int4 *>(CP.tlen + j2);
            }
            four_packed(wa, xa, j, j_lo, nj, pbase, ptab, one, inner_pass(jf));
            if (j2 < j_hi) four_packed(wb, xb, j2, j_lo, nj, pbase, ptab, one, inner_pass(jf + 4u * NT));
        }
    }
    {   // ---- class 0 (span <= 256, a rare pair): no end column, end = pos + (fm >> 24) ------------------
        const BsigClassCols &C = R.cls[0];
        const uint32_t j_lo = win[0].x, j_hi = win[0].y, nj = j_hi - j_lo;
        if (small0) {
            if (nj) one(pa, pa + (int)(fa >> 24), (fa & 0x10u) != 0u, fm_rejected(P, fa), ta, jb0 < j_hi);
        } else
        for (uint32_t j = (j_lo & ~3u) + 4u * tid; j < j_hi; j += 4u * NT) {
            const int4 p = *reinterpret_cast<const int4 *>(C.pos + j);
            const uint4 f = *reinterpret_cast<const uint4 *>(C.fm + j);
            int4 t = make_int4(0, 0, 0, 0);
            if (P.use_tlen) t = *reinterpret_cast<const int4 *>(C.tlen + j);
            const uint32_t dj = j - j_lo;
            one(p.x, p.x + (int)(f.x >> 24), (f.x & 0x10u) != 0u, fm_rejected(P, f.x), t.x, dj < nj);
            one(p.y, p.y + (int)(f.y >> 24), (f.y & 0x10u) != 0u, fm_rejected(P, f.y), t.y, dj + 1u < nj);
            one(p.z, p.z + (int)(f.z >> 24), (f.z & 0x10u) != 0u, fm_rejected(P, f.z), t.z, dj + 2u < nj);
            one(p.w, p.w + (int)(f.w >> 24), (f.w & 0x10u) != 0u, fm_rejected(P, f.w), t.w, dj + 3u < nj);
        }
    }
    {   // ---- class 1 (span <= 4096, 12-bit flags): no end column either, end = pos + (word >> 20) -----------
        const BsigClassCols &C = R.cls[1];
        const uint32_t j_lo = win[1].x, j_hi = win[1].y, nj = j_hi - j_lo;
        if (small1) {
            const uint32_t gx = fm_of_class1(fb);
            if (nj) one(pb, pb + (int)(fb >> 20), (gx & 0x10u) != 0u, fm_rejected(P, gx), tb, jb1 < j_hi);
        } else
        for (uint32_t j = (j_lo & ~3u) + 4u * tid; j < j_hi; j += 4u * NT) {
            const int4 p = *reinterpret_cast<const int4 *>(C.pos + j);
            const uint4 f = *reinterpret_cast<const uint4 *>(C.fm + j);
            int4 t = make_int4(0, 0, 0, 0);
            if (P.use_tlen) t = *reinterpret_cast<const int4 *>(C.tlen + j);
            const uint32_t dj = j - j_lo;
            const uint32_t gx = fm_of_class1(f.x), gy = fm_of_class1(f.y), gz = fm_of_class1(f.z), gw = fm_of_class1(f.w);
            one(p.x, p.x + (int)(f.x >> 20), (gx & 0x10u) != 0u, fm_rejected(P, gx), t.x, dj < nj);
            one(p.y, p.y + (int)(f.y >> 20), (gy & 0x10u) != 0u, fm_rejected(P, gy), t.y, dj + 1u < nj);
            one(p.z, p.z + (int)(f.z >> 20), (gz & 0x10u) != 0u, fm_rejected(P, gz), t.z, dj + 2u < nj);
            one(p.w, p.w + (int)(f.w >> 20), (gw & 0x10u) != 0u, fm_rejected(P, gw), t.w, dj + 3u < nj);
        }
    }
#pragma unroll
    for (int c = 2; c < BSIG_SPAN_CLASSES; ++c) {   // ---- classes 2-3: pos, end, fm columns ---------
        const BsigClassCols &C = R.cls[c];
        const uint32_t j_lo = win[c].x, j_hi = win[c].y, nj = j_hi - j_lo;
        if (nj <= (uint32_t)NT) {                          // (one read per lane, as above)
            const uint32_t j = j_lo + (uint32_t)tid;
            if (nj) {
                int p = 0, e = 0, t = 0;
                uint32_t f = 0;
                if (j < j_hi) {
                    p = C.pos[j]; f = C.fm[j]; e = C.end[j];
                    if (P.use_tlen) t = C.tlen[j];
                }
                one(p, e, (f & 0x10u) != 0u, fm_rejected(P, f), t, j < j_hi);
            }
        } else
        for (uint32_t j = (j_lo & ~3u) + 4u * tid; j < j_hi; j += 4u * NT) {
            const int4 p = *reinterpret_cast<const int4 *>(C.pos + j);
            const uint4 f = *reinterpret_cast<const uint4 *>(C.fm + j);
            const int4 e = *reinterpret_cast<const int4 *>(C.end + j);
            int4 t = make_int4(0, 0, 0, 0);
            if (P.use_tlen) t = *reinterpret_cast<const int4 *>(C.tlen + j);
            const uint32_t dj = j - j_lo;
            one(p.x, e.x, (f.x & 0x10u) != 0u, fm_rejected(P, f.x), t.x, dj < nj);
            one(p.y, e.y, (f.y & 0x10u) != 0u, fm_rejected(P, f.y), t.y, dj + 1u < nj);
            one(p.z, e.z, (f.z & 0x10u) != 0u, fm_rejected(P, f.z), t.z, dj + 2u < nj);
            one(p.w, e.w, (f.w & 0x10u) != 0u, fm_rejected(P, f.w), t.w, dj + 3u < nj);
        }
    }
}

// the bucket-rounded window [rlo, rhi) of the packed class for the genomic interval [tlo, thi) of an item
__device__ __forceinline__ bool packed_window(const BsigClassCols &C, const BsigWorkItem &w, int64_t tlo, int64_t thi, int ext,
                                              int64_t &rlo, int64_t &rhi)
{
    int64_t wlo = tlo - ext - C.maxspan + 1, whi = thi + ext;
    const int64_t ref_bp = (int64_t)(w.units_strand & BSIG_ITEM_UNITS_MASK) << BSIG_REF_UNIT_SHIFT;
    if (wlo < 0) wlo = 0;
    if (whi > ref_bp) whi = ref_bp;
    rlo = (wlo >> C.kshift) << C.kshift;
    rhi = whi > 0 ? (((whi - 1) >> C.kshift) + 1) << C.kshift : 0;
    return C.n != 0 && wlo < whi;
}

// The packed class's chunks behind the first (a window wider than kPackChunk bases: a shift or a template
// length filter of tens of kilobases -- rare, so plain loops): every chunk is looked up in the index by
// itself and clipped to `clip` (the read range of a slice of a heavy tile; everything otherwise).
template <int NT, typename Tab, typename F>
__device__ __forceinline__ void packed_later_chunks(const BsigReadsDev &R, const BsigKParams &P, int mode, const BsigWorkItem &w,
                                                    int n_chunks, uint2 clip, const Tab *__restrict__ ptab, int tid, F &&one)
{
    const BsigClassCols &C = R.cls[BSIG_CLASS_PACKED];
    int64_t tlo, thi, rlo, rhi;
    item_interval(w, P, mode, tlo, thi);
    if (!packed_window(C, w, tlo, thi, P.ext, rlo, rhi)) return;
    const uint64_t g0 = (uint64_t)w.ref_unit0 << BSIG_REF_UNIT_SHIFT;
    for (int c = 1; c < n_chunks; ++c) {
        const int64_t a = rlo + (int64_t)c * kPackChunk;
        const int64_t b = a + kPackChunk < rhi ? a + kPackChunk : rhi;
        uint32_t j_lo = C.idx[(g0 + (uint64_t)a) >> C.kshift], j_hi = C.idx[(g0 + (uint64_t)b) >> C.kshift];
        j_lo = j_lo > clip.x ? j_lo : clip.x;
        j_hi = j_hi < clip.y ? j_hi : clip.y;
        if (j_lo >= j_hi) continue;
        const uint32_t nj = j_hi - j_lo;
        for (uint32_t j = (j_lo & ~3u) + 4u * tid; j < j_hi; j += 4u * NT) {
            const uint4 x = *reinterpret_cast<const uint4 *>(C.fm + j);
            int4 t = make_int4(0, 0, 0, 0);
            if (P.use_tlen) t = *reinterpret_cast<const int4 *>(C.tlen + j);
            four_packed(x, t, j, j_lo, nj, (int)a, ptab, one);
        }
    }
}

// The read windows of a tile, looked up here with the index loads of all classes issued back to back.
// `windows` (slices of heavy tiles): fixed read ranges instead -- for the span classes as they are, for the
// packed class as a clip of what the index says (the chunk's start is needed as well).
// P.resolved (large launches): the windows as k_resolve_tiles wrote them in front of this launch; a kernel
// instantiated with RES = true has nothing but that form in it (fewer registers).
template <bool RES = false>
__device__ __forceinline__ void load_windows(const BsigReadsDev &R, const BsigKParams &P, int mode,
                                             const BsigWorkItem &w, const BsigWorkItem *__restrict__ items,
                                             const uint2 *__restrict__ windows, uint2 (&win)[BSIG_MAX_CLASSES],
                                             uint32_t tile, PackedWin &pk, uint2 &clip)
{
    if (RES || P.resolved) {
        // the launch in front of this one looked the windows up (k_resolve_tiles): this load does not depend
        // on the work item's
        const BsigResolved r = reinterpret_cast<const BsigResolved *>(windows)[tile];
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) win[c] = make_uint2(r.win[2 * c], r.win[2 * c + 1]);
        pk.base = r.pbase;
        pk.n_chunks = r.pchunks;
        clip = make_uint2(0u, 0xFFFFFFFFu);
        return;
    }
    int64_t tlo, thi;
    item_interval(w, P, mode, tlo, thi);
    const uint32_t *alo[BSIG_MAX_CLASSES], *ahi[BSIG_MAX_CLASSES];
    bool live[BSIG_MAX_CLASSES];
    const int64_t ref_bp = (int64_t)(w.units_strand & BSIG_ITEM_UNITS_MASK) << BSIG_REF_UNIT_SHIFT;
    const uint64_t g0 = (uint64_t)w.ref_unit0 << BSIG_REF_UNIT_SHIFT;
    // dead classes read a harmless valid word instead of branching around the load
    const uint32_t *dummy = reinterpret_cast<const uint32_t *>(items);
#pragma unroll
    for (int c = 0; c < BSIG_SPAN_CLASSES; ++c) {
        const BsigClassCols &C = R.cls[c];
        int64_t wlo = tlo - P.ext - C.maxspan + 1, whi = thi + P.ext;
        if (wlo < 0) wlo = 0;
        if (whi > ref_bp) whi = ref_bp;
        live[c] = C.n != 0 && wlo < whi && !windows;
        alo[c] = live[c] ? C.idx + ((g0 + (uint64_t)wlo) >> C.kshift) : dummy;
        ahi[c] = live[c] ? C.idx + (((g0 + (uint64_t)whi - 1) >> C.kshift) + 1) : dummy;
    }
    {
        const BsigClassCols &C = R.cls[BSIG_CLASS_PACKED];
        int64_t rlo, rhi;
        const bool on = packed_window(C, w, tlo, thi, P.ext, rlo, rhi);
        live[BSIG_CLASS_PACKED] = on;
        const int64_t first_end = rlo + kPackChunk < rhi ? rlo + kPackChunk : rhi;
        alo[BSIG_CLASS_PACKED] = on ? C.idx + ((g0 + (uint64_t)rlo) >> C.kshift) : dummy;
        ahi[BSIG_CLASS_PACKED] = on ? C.idx + ((g0 + (uint64_t)first_end) >> C.kshift) : dummy;
        pk.base = (int)rlo;
        pk.n_chunks = on ? (int)((rhi - rlo + kPackChunk - 1) >> BSIG_PACK_POS_BITS) : 0;
    }
    uint32_t lo[BSIG_MAX_CLASSES], hi[BSIG_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) { lo[c] = *alo[c]; hi[c] = *ahi[c]; }
#pragma unroll
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) win[c] = live[c] && lo[c] < hi[c] ? make_uint2(lo[c], hi[c]) : make_uint2(0u, 0u);
    clip = make_uint2(0u, 0xFFFFFFFFu);
    if (windows) {
        const uint2 *wp = windows + (size_t)BSIG_MAX_CLASSES * tile;
#pragma unroll
        for (int c = 0; c < BSIG_SPAN_CLASSES; ++c) win[c] = wp[c];
        clip = wp[BSIG_CLASS_PACKED];
        uint2 &q = win[BSIG_CLASS_PACKED];
        q.x = q.x > clip.x ? q.x : clip.x;
        q.y = q.y < clip.y ? q.y : clip.y;
        if (q.x >= q.y) q = make_uint2(0u, 0u);
    }
    // a heavy tile only zero-fills its cells here; its reads come through slice items afterwards
    if (w.units_strand & BSIG_ITEM_HEAVY) {
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) win[c] = make_uint2(0u, 0u);
        pk.n_chunks = 0;
    }
}

// The index lookups of a large launch as a launch of their own, one lane per tile: the pileup workgroups,
// which hold LDS and registers for their whole life, then start with their windows one load away.
__global__ void k_resolve_tiles(const BsigReadsDev R, const BsigKParams P, int mode,
                                const BsigWorkItem *__restrict__ items, uint32_t n_items,
                                BsigResolved *__restrict__ out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_items) return;
    const BsigWorkItem w = items[t];
    uint2 win[BSIG_MAX_CLASSES], clip;
    PackedWin pk;
    load_windows(R, P, mode, w, items, nullptr, win, t, pk, clip);
    BsigResolved r;
#pragma unroll
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) { r.win[2 * c] = win[c].x; r.win[2 * c + 1] = win[c].y; }
    r.pbase = pk.base;
    r.pchunks = pk.n_chunks;
    out[t] = r;
}

// the accumulating form of store_vec: slices of a heavy tile add their (partial) image
__device__ __forceinline__ void add_vec(int32_t *__restrict__ gbase, int v, int4 x, int sh, int nv)
{
    const int e0 = 4 * v, lo = sh, hi = sh + nv;
    if (x.x && e0 + 0 >= lo && e0 + 0 < hi) atomicAdd(gbase + e0 + 0, x.x);
    if (x.y && e0 + 1 >= lo && e0 + 1 < hi) atomicAdd(gbase + e0 + 1, x.y);
    if (x.z && e0 + 2 >= lo && e0 + 2 < hi) atomicAdd(gbase + e0 + 2, x.z);
    if (x.w && e0 + 3 >= lo && e0 + 3 < hi) atomicAdd(gbase + e0 + 3, x.w);
}

// ------------------------------------------------------------------------------------------
// bamProfile: per-bin counts of 5' ends
// ------------------------------------------------------------------------------------------
// The tile image holds 16-bit counters, two per LDS dword: a tile that is not split into slices has
// at most 32,768 reads in its windows (kMaxTileReads; bsig_plan_create cuts anything above into
// slices of a quarter of that), so no counter can reach 65,536 and an add of 1 << 16 into the upper
// half never sees a carry from the lower one.  Half the LDS per workgroup: 24 instead of 18
// single-wave workgroups per CU (LDS is allocated in 1,280-byte granules on gfx950), which puts
// config 2's 10,000 tiles into two rounds of resident workgroups instead of two and a sparse third.
template <int NT, bool SS, int PRE, int WAVES, bool RES>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void k_profile(const BsigWorkItem *__restrict__ items, uint32_t n_tiles,
                                                int32_t *__restrict__ out,
                                                const uint2 *__restrict__ windows,
                                                const BsigReadsDev R, const BsigKParams P)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    constexpr int S = SS ? 2 : 1;
    const int tid = threadIdx.x;
    BSIG_STAMP(0);
    const uint32_t tile = tile_of_block(blockIdx.x, n_tiles);
    const BsigWorkItem w = items[tile];
    uint2 win[BSIG_MAX_CLASSES], clip;
    PackedWin pk;
    load_windows<RES>(R, P, BSIG_MODE_PROFILE, w, items, windows, win, tile, pk, clip);
    int4 *lds4 = reinterpret_cast<int4 *>(lds);
    // clear the whole tile image: this needs nothing from the work item, so it overlaps its load
    const int img_vec = (P.tile_cells * S + 8 + 7) / 8;
    for (int v = tid; v < img_vec; v += NT) lds4[v] = make_int4(0, 0, 0, 0);
    // ... and so does the packed class's filter table, which lives behind the image
    uint8_t *ptab = reinterpret_cast<uint8_t *>(lds4 + img_vec);
    build_ptab<NT>(ptab, R, P, tid);
    const int nv = w.nc * S;
    const int sh = (int)(w.out_off & 3);
    const int nvec = (sh + nv + 3) >> 2;
    block_sync<NT>();
    BSIG_STAMP(1);

    const bool neg_range = (w.units_strand & BSIG_ITEM_NEG) != 0u;
    uint32_t *cnt = reinterpret_cast<uint32_t *>(lds);

    const ProfileOne<SS> one{P, cnt, w.loc, w.len, w.c0, w.nc, sh, neg_range};
    if (!BSIG_ABLATE(1)) {
        for_each_read<NT, PRE>(R, P, win, pk.base, ptab, tid, one);
        if (pk.n_chunks > 1) packed_later_chunks<NT>(R, P, BSIG_MODE_PROFILE, w, pk.n_chunks, clip, ptab, tid, one);
    }
    block_sync<NT>();
    BSIG_STAMP(2);

    int32_t *gbase = out + (w.out_off - sh);
    const uint2 *lds2 = reinterpret_cast<const uint2 *>(lds);
    auto widen = [](uint2 d) {
        return make_int4((int)(d.x & 0xFFFFu), (int)(d.x >> 16), (int)(d.y & 0xFFFFu), (int)(d.y >> 16));
    };
    if (P.accumulate) {
        for (int v = tid; v < nvec; v += NT) add_vec(gbase, v, widen(lds2[v]), sh, nv);
    } else if (!BSIG_ABLATE(2)) {
        for (int v = tid; v < nvec; v += NT) store_vec(gbase, v, widen(lds2[v]), sh, nv);
    }
    BSIG_STAMP(3);
#ifdef BSIG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BSIG_STAMP(4);
    if (g_stamp_buf && threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_stamp_buf[(size_t)blockIdx.x * 8 + 5] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

// k_profile for T consecutive tiles per wave (large launches, windows from k_resolve_tiles): lane t fetches the work
// item and the windows of tile t of its group -- ONE round trip for the T tiles -- and parks them in LDS; the wave then
// works the tiles off one after the other through one image, which the store loop of a tile clears for the next.  From
// its second tile on a wave pays neither a workgroup launch, nor the item / windows trip, nor the filter table.
template <bool SS, int PRE, int WAVES, int T>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void k_profile_multi(
    const BsigWorkItem *__restrict__ items, uint32_t n_tiles, int32_t *__restrict__ out, const uint2 *__restrict__ windows,
    const BsigReadsDev R, const BsigKParams P)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    constexpr int S = SS ? 2 : 1;
    constexpr int kStage = 20;                               // dwords per tile: 8 of the item, 12 of its windows
    const int tid = threadIdx.x;
    const uint32_t n_groups = (n_tiles + T - 1) / T;
    const uint32_t first = tile_of_block(blockIdx.x, n_groups) * T;
    const uint32_t n_here = n_tiles - first < (uint32_t)T ? n_tiles - first : (uint32_t)T;      // uniform
    int4 *lds4 = reinterpret_cast<int4 *>(lds);
    const int img_vec = (P.tile_cells * S + 8 + 7) / 8;
    uint8_t *ptab = reinterpret_cast<uint8_t *>(lds4 + img_vec);
    uint32_t *stage = reinterpret_cast<uint32_t *>(ptab + BSIG_PACK_CODES);
    if ((uint32_t)tid < n_here) {
        const uint4 *ip = reinterpret_cast<const uint4 *>(items + first + tid);
        const uint4 *rp = reinterpret_cast<const uint4 *>(reinterpret_cast<const BsigResolved *>(windows) + first + tid);
        const uint4 a = ip[0], b = ip[1], c = rp[0], d = rp[1], e = rp[2];
        uint4 *sp = reinterpret_cast<uint4 *>(stage + kStage * tid);
        sp[0] = a; sp[1] = b; sp[2] = c; sp[3] = d; sp[4] = e;
    }
    for (int v = tid; v < img_vec; v += kWave) lds4[v] = make_int4(0, 0, 0, 0);
    build_ptab<kWave>(ptab, R, P, tid);
    block_sync<kWave>();
    uint32_t *cnt = reinterpret_cast<uint32_t *>(lds);
    uint2 *lds2 = reinterpret_cast<uint2 *>(lds);
#pragma unroll 1
    for (uint32_t t = 0; t < n_here; ++t) {
        const uint32_t *sg = stage + kStage * t;
        auto sc = [&](int k) { return __builtin_amdgcn_readfirstlane((int)sg[k]); };
        BsigWorkItem w;
        w.loc = sc(0); w.len = sc(1); w.c0 = sc(2); w.nc = sc(3);
        w.out_off = (int64_t)(((uint64_t)(uint32_t)sc(5) << 32) | (uint32_t)sc(4));
        w.ref_unit0 = (uint32_t)sc(6); w.units_strand = (uint32_t)sc(7);
        uint2 win[BSIG_MAX_CLASSES];
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) win[c] = make_uint2((uint32_t)sc(8 + 2 * c), (uint32_t)sc(9 + 2 * c));
        const int pbase = sc(18), pchunks = sc(19);
        const int nv = w.nc * S;
        const int sh = (int)(w.out_off & 3);
        const int nvec = (sh + nv + 3) >> 2;
        const bool neg_range = (w.units_strand & BSIG_ITEM_NEG) != 0u;
        const ProfileOne<SS> one{P, cnt, w.loc, w.len, w.c0, w.nc, sh, neg_range};
        for_each_read<kWave, PRE>(R, P, win, pbase, ptab, tid, one);
        if (pchunks > 1) packed_later_chunks<kWave>(R, P, BSIG_MODE_PROFILE, w, pchunks, make_uint2(0u, 0xFFFFFFFFu), ptab, tid, one);
        block_sync<kWave>();
        int32_t *gbase = out + (w.out_off - sh);
        for (int v = tid; v < nvec; v += kWave) {
            const uint2 d = lds2[v];
            lds2[v] = make_uint2(0u, 0u);                    // the image is the next tile's
            store_vec(gbase, v, make_int4((int)(d.x & 0xFFFFu), (int)(d.x >> 16), (int)(d.y & 0xFFFFu), (int)(d.y >> 16)), sh, nv);
        }
        block_sync<kWave>();
    }
}

// Wide bins (binsize >~ 8): a tile has few cells and thousands of reads, and the position-sorted
// reads of one wave instruction fall into one or two bins, so plain LDS atomics take turns on one or
// two addresses.  32-bit cells in LDS, a few replicas of the image (odd stride: the same cell of
// different replicas lies in different banks), summed at the end.
constexpr int kSmallCells = 256;     // at most this many values (cells * S) per tile
// Replicas of the image (lanes add into replica `tid & (r - 1)`, the replicas are summed at the end): FEW, and chosen by
// the TILE's own values.  Rounds 2-4 kept up to 32 (8 KiB) so that no two lanes of a wave would meet on a cell; clearing
// and summing r x values dwords per tile and the LDS they take cost more than lanes taking turns (100,000 x 2 kb ranges,
// 1e8 reads, one box; 32 / 4 / 2 / 1 replicas: binsize 50 with strands (80 values) 0.167 / 0.117 / 0.114 / 0.107 ms,
// binsize 16 (125 values) 0.174 / 0.116 / 0.113 / 0.107): four replicas for images of up to 32 values, two up to 64, one
// beyond.  With a handful of cells they matter most -- one replica instead of four: binsize 2000 with strands (two
// values) 0.107 -> 0.172 ms, binsize 500 (four values) 0.105 -> 0.24 -- and until late in round 5 the choice went by the
// PLAN's tile_cells, which is never below 64: a 2-kb range in bins of 200 bases (10 cells, 20 values with strands) ran
// with one replica.  By the tile's values: binsize 200 with strands 0.1235 -> 0.110 ms, binsize 500 0.139 -> 0.105.
// (Spreading the lanes over the window instead -- a lane takes consecutive vectors, so that an instruction's reads come
// from all over the tile -- does the same for 20 values and nothing once the replicas are right: not kept.)
__host__ __device__ inline int small_replicas(int stride)
{
    return stride <= 32 ? 4 : stride <= 64 ? 2 : 1;
}
// dwords of the largest image (replicas x odd stride) a tile of at most `stride_max` values can ask for
__host__ __device__ inline int small_image_dwords(int stride_max)
{
    int most = stride_max * small_replicas(stride_max);          // (strides are odd: 31 and 63 are the last of their kind)
    if (stride_max > 31) most = most > 31 * small_replicas(31) ? most : 31 * small_replicas(31);
    if (stride_max > 63) most = most > 63 * small_replicas(63) ? most : 63 * small_replicas(63);
    return most;
}
template <int NT, bool SS, bool RES>
__global__ __launch_bounds__(NT) void k_profile_small(const BsigWorkItem *__restrict__ items, uint32_t n_tiles,
                                                      int32_t *__restrict__ out,
                                                      const uint2 *__restrict__ windows,
                                                      const BsigReadsDev R, const BsigKParams P)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    constexpr int S = SS ? 2 : 1;
    const int tid = threadIdx.x;
    const uint32_t tile = tile_of_block(blockIdx.x, n_tiles);
    const BsigWorkItem w = items[tile];
    uint2 win[BSIG_MAX_CLASSES], clip;
    PackedWin pk;
    load_windows<RES>(R, P, BSIG_MODE_PROFILE, w, items, windows, win, tile, pk, clip);
    // the replicas go by THIS tile's values (a plan's tile_cells is at least 64 cells); the launch reserves the image of
    // the worst tile a plan of this tile_cells can hold
    const int nv = w.nc * S;
    const int stride = nv | 1;                                 // odd: replicas shift by one bank
    const int n_rep = small_replicas(stride);
    for (int v = tid; v < n_rep * stride; v += NT) lds[v] = 0;
    uint8_t *ptab = reinterpret_cast<uint8_t *>(lds + ((small_image_dwords((P.tile_cells * S) | 1) + 3) & ~3));
    build_ptab<NT>(ptab, R, P, tid);
    block_sync<NT>();
    const bool neg_range = (w.units_strand & BSIG_ITEM_NEG) != 0u;
    int32_t *mine = lds + (tid & (n_rep - 1)) * stride;

    const SmallOne<SS> one{P, mine, w.loc, w.len, w.c0, w.nc, neg_range};
#ifndef BSIG_SMALL_PRE
#define BSIG_SMALL_PRE 4
#endif
    for_each_read<NT, BSIG_SMALL_PRE>(R, P, win, pk.base, ptab, tid, one);
    if (pk.n_chunks > 1) packed_later_chunks<NT>(R, P, BSIG_MODE_PROFILE, w, pk.n_chunks, clip, ptab, tid, one);
    block_sync<NT>();

    for (int v = tid; v < nv; v += NT) {
        int acc = 0;
        for (int r = 0; r < n_rep; ++r) acc += lds[r * stride + v];
        if (!P.accumulate) out[w.out_off + v] = acc;
        else if (acc) atomicAdd(out + w.out_off + v, acc);
    }
}

// ------------------------------------------------------------------------------------------
// bamCount: one (or two, strand-specific) counters per range
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void k_count(const BsigWorkItem *__restrict__ items, uint32_t n_tiles,
                                              int32_t *__restrict__ out,
                                              const uint2 *__restrict__ windows,
                                              const BsigReadsDev R, const BsigKParams P)
{
    __shared__ int32_t wsum[2 * (NT / kWave)];
    __shared__ __attribute__((aligned(16))) uint32_t ptab[BSIG_PACK_CODES];      // (the count family's table: build_ctab)
    const int tid = threadIdx.x;
    const uint32_t tile = tile_of_block(blockIdx.x, n_tiles);
    const BsigWorkItem w = items[tile];
    uint2 win[BSIG_MAX_CLASSES], clip;
    PackedWin pk;
    load_windows(R, P, BSIG_MODE_COUNT, w, items, windows, win, tile, pk, clip);
    build_ctab<NT>(ptab, P, tid);
    block_sync<NT>();
    const bool neg_range = (w.units_strand & BSIG_ITEM_NEG) != 0u;
    const int glo = w.loc + w.c0;           // sub-interval of the range, genomic coordinates
    const int gn = w.nc;
    // Counting is all this launch does per read, and the launch is bound by its vector instructions (PMC, config 3's
    // tiling: 95e6 VALU instructions x 4 cycles over 1,024 SIMDs = the whole 155 us): no branch per read -- a read
    // that is no read, rejected or outside adds 0 -- and ONE packed counter (all reads in its low half, reverse-strand
    // ones in its high half) that becomes sense and antisense once, behind the loop.
    uint32_t acc = 0;
    const CountOne one{P, glo, gn, acc};
    for_each_read<NT>(R, P, win, pk.base, ptab, tid, one);
    if (pk.n_chunks > 1) packed_later_chunks<NT>(R, P, BSIG_MODE_COUNT, w, pk.n_chunks, clip, ptab, tid, one);
    const int c_all = (int)(acc & 0xFFFFu), c_neg = (int)(acc >> 16);
    int c_anti = neg_range ? c_all - c_neg : c_neg, c_sense = c_all - c_anti;

    // wave reduction, then across the waves of the workgroup
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        c_sense += __shfl_down(c_sense, d);
        c_anti += __shfl_down(c_anti, d);
    }
    if (NT > kWave) {
        if ((tid & (kWave - 1)) == 0) { wsum[2 * (tid / kWave)] = c_sense; wsum[2 * (tid / kWave) + 1] = c_anti; }
        __syncthreads();
        if (tid == 0) {
            c_sense = 0; c_anti = 0;
            for (int k = 0; k < NT / kWave; ++k) { c_sense += wsum[2 * k]; c_anti += wsum[2 * k + 1]; }
        }
    }
    if (tid == 0) {
        const bool atomic = (w.units_strand & BSIG_ITEM_ATOMIC) != 0u;
        int32_t *o = out + w.out_off;
        if (P.ss) {
            if (atomic) { if (c_sense) atomicAdd(o, c_sense); if (c_anti) atomicAdd(o + 1, c_anti); }
            else        { o[0] = c_sense; o[1] = c_anti; }
        } else {
            if (atomic) { if (c_sense + c_anti) atomicAdd(o, c_sense + c_anti); }
            else        { o[0] = c_sense + c_anti; }
        }
    }
}

// bamCount, several consecutive tiles per wave.  A count tile moves one dword out and has no LDS image,
// so nothing but the dependent chain item -> index -> reads fills a wave's lifetime.  Here lane t of
// the wave fetches the work item of tile t of its group and looks up that tile's windows (T chains
// side by side instead of one behind the other); the tiles are then streamed one after the other
// with the windows broadcast from their lane, and lane t stores tile t's counters at the end.
template <int T, int PRE>
__global__ __launch_bounds__(kWave) void k_count_multi(const BsigWorkItem *__restrict__ items, uint32_t n_tiles,
                                                       int32_t *__restrict__ out,
                                                       const uint2 *__restrict__ windows,
                                                       const BsigReadsDev R, const BsigKParams P)
{
    __shared__ uint32_t stage[T][16];      // per tile: 5 windows, first base, bases, flags, packed base and chunks
    __shared__ __attribute__((aligned(16))) uint32_t ptab[BSIG_PACK_CODES];      // (the count family's table: build_ctab)
    const int lane = threadIdx.x;
    const uint32_t n_groups = (n_tiles + T - 1) / T;
    const uint32_t first = tile_of_block(blockIdx.x, n_groups) * T;
    const uint32_t n_here = n_tiles - first < (uint32_t)T ? n_tiles - first : (uint32_t)T;      // uniform
    const bool have = (uint32_t)lane < n_here;
    int64_t out_off = 0;
    bool atomic = false;
    if (have) {
        const BsigWorkItem w = items[first + lane];
        uint2 win[BSIG_MAX_CLASSES], clip;
        PackedWin pk;
        load_windows(R, P, BSIG_MODE_COUNT, w, items, windows, win, first + lane, pk, clip);
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) { stage[lane][2 * c] = win[c].x; stage[lane][2 * c + 1] = win[c].y; }
        stage[lane][10] = (uint32_t)(w.loc + w.c0);
        stage[lane][11] = (uint32_t)w.nc;
        stage[lane][12] = w.units_strand;
        stage[lane][13] = (uint32_t)pk.base;
        stage[lane][14] = (uint32_t)pk.n_chunks;
        out_off = w.out_off;
        atomic = (w.units_strand & BSIG_ITEM_ATOMIC) != 0u;
    }
    build_ctab<kWave>(ptab, P, lane);
    block_sync<kWave>();
    int my_sense = 0, my_anti = 0;
#pragma unroll 1
    for (int t = 0; t < (int)n_here; ++t) {
        uint2 wn[BSIG_MAX_CLASSES];
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c)
            wn[c] = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)stage[t][2 * c]),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)stage[t][2 * c + 1]));
        const int glo = __builtin_amdgcn_readfirstlane((int)stage[t][10]);
        const int gn = __builtin_amdgcn_readfirstlane((int)stage[t][11]);
        const bool neg_range = ((uint32_t)__builtin_amdgcn_readfirstlane((int)stage[t][12]) & BSIG_ITEM_NEG) != 0u;
        const int pbase = __builtin_amdgcn_readfirstlane((int)stage[t][13]);
        const int pchunks = __builtin_amdgcn_readfirstlane((int)stage[t][14]);
        uint32_t acc = 0;                          // (see k_count: nothing branches per read)
        const CountOne one{P, glo, gn, acc};
        for_each_read<kWave, PRE>(R, P, wn, pbase, ptab, lane, one);
        if (pchunks > 1) {
            const BsigWorkItem w2 = items[first + t];
            packed_later_chunks<kWave>(R, P, BSIG_MODE_COUNT, w2, pchunks, make_uint2(0u, 0xFFFFFFFFu), ptab, lane, one);
        }
        // (the wave's sum stays packed: a tile that is not sliced has at most 32,768 reads in its windows)
        // (six DPP adds and a v_readlane; the butterfly of __shfl_xor was six trips through LDS and their addresses)
        const uint32_t sum = (uint32_t)__builtin_amdgcn_readlane(wave_inclusive_scan((int)acc), kWave - 1);
        const int c_all = (int)(sum & 0xFFFFu), c_neg = (int)(sum >> 16);
        const int c_anti = neg_range ? c_all - c_neg : c_neg, c_sense = c_all - c_anti;
        if (lane == t) { my_sense = c_sense; my_anti = c_anti; }
    }
    if (have) {
        int32_t *o = out + out_off;
        if (P.ss) {
            if (atomic) { if (my_sense) atomicAdd(o, my_sense); if (my_anti) atomicAdd(o + 1, my_anti); }
            else        { o[0] = my_sense; o[1] = my_anti; }
        } else {
            if (atomic) { if (my_sense + my_anti) atomicAdd(o, my_sense + my_anti); }
            else        { o[0] = my_sense + my_anti; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// bamCoverage: +1/-1 difference array in LDS, workgroup prefix scan, coalesced store
// ------------------------------------------------------------------------------------------
// The difference array holds SIGNED 16-bit cells, two per LDS dword, as k_profile's image does with
// unsigned ones: the dword is kept equal to hi * 65536 + lo (mod 2^32) by plain integer adds of
// +-1 and +-65536 (a borrow out of the low half is part of that sum, not an error), and is taken
// apart at the end as lo = sign-extended low half, hi = sign-extended high half of (word + 0x8000).
// A tile that is not cut into slices has at most 32,767 reads in its windows (bsig_plan_create's
// ceiling for coverage), so every cell stays inside [-32767, 32767].  4 KiB instead of 8 KiB of
// LDS per 2,048-cell tile: 24 instead of 18 single-wave workgroups per CU.
template <int NT, int PRE, bool RES>
__global__ __launch_bounds__(NT) void k_coverage(const BsigWorkItem *__restrict__ items, uint32_t n_tiles,
                                                 int32_t *__restrict__ out,
                                                 const uint2 *__restrict__ windows,
                                                 const BsigReadsDev R, const BsigKParams P)
{
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    const int img_vec = (P.tile_cells + 8 + 7) / 8;          // 16-B vectors of the image (8 cells each)
    // per-wave scan totals live behind the tile image, inside the dynamic region, so that the
    // image itself starts at the 16-B aligned LDS base
    int32_t *wtot = lds + 4 * img_vec;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const uint32_t tile = tile_of_block(blockIdx.x, n_tiles);
    const BsigWorkItem w = items[tile];
    uint2 win[BSIG_MAX_CLASSES], clip;
    PackedWin pk;
    load_windows<RES>(R, P, BSIG_MODE_COVERAGE, w, items, windows, win, tile, pk, clip);
    int4 *lds4 = reinterpret_cast<int4 *>(lds);
    for (int v = tid; v < img_vec; v += NT) lds4[v] = make_int4(0, 0, 0, 0);
    // the packed class's filter table: behind the image and the scan totals (16-B aligned)
    uint8_t *ptab = reinterpret_cast<uint8_t *>(lds4 + img_vec + (NT / kWave + 3) / 4);
    build_ptab<NT>(ptab, R, P, tid);
    const int nv = w.nc;
    const int sh = (int)(w.out_off & 3);
    const int nvec = (sh + nv + 3) >> 2;
    block_sync<NT>();

    const bool neg_range = (w.units_strand & BSIG_ITEM_NEG) != 0u;
    const int rend1 = w.loc + w.len - 1;     // last base of the range

    const CoverOne one{P, lds, w.loc, w.c0, w.nc, sh, neg_range, rend1};
    for_each_read<NT, PRE>(R, P, win, pk.base, ptab, tid, one);
    if (pk.n_chunks > 1) packed_later_chunks<NT>(R, P, BSIG_MODE_COVERAGE, w, pk.n_chunks, clip, ptab, tid, one);
    block_sync<NT>();

    // cumsum (:464-470): each lane owns 4 consecutive cells (two packed dwords), wave scan of the
    // lane totals, carry across waves and across passes
    int32_t *gbase = out + (w.out_off - sh);
    const uint2 *lds2 = reinterpret_cast<const uint2 *>(lds);
    auto lo16 = [](uint32_t d) { return (int)(int16_t)(uint16_t)d; };
    auto hi16 = [](uint32_t d) { return (int)(int16_t)(uint16_t)((d + 0x8000u) >> 16); };
    int carry = 0;
    for (int base = 0; base < nvec; base += NT) {
        const int v = base + tid;
        const uint2 d = v < nvec ? lds2[v] : make_uint2(0u, 0u);
        int4 x = make_int4(lo16(d.x), hi16(d.x), lo16(d.y), hi16(d.y));
        x.y += x.x; x.z += x.y; x.w += x.z;
        const int tot = x.w;
        const int incl = wave_inclusive_scan(tot);
        int pre = carry;
        int all = __builtin_amdgcn_readlane(incl, kWave - 1);
        if (NT > kWave) {
            if (lane == kWave - 1) wtot[tid / kWave] = incl;
            __syncthreads();
            all = 0;
            for (int k = 0; k < NT / kWave; ++k) {
                const int t = wtot[k];
                if (k < tid / kWave) pre += t;
                all += t;
            }
            __syncthreads();
        }
        const int add = pre + incl - tot;
        x.x += add; x.y += add; x.z += add; x.w += add;
        if (v < nvec) {
            if (P.accumulate) add_vec(gbase, v, x, sh, nv);
            else store_vec(gbase, v, x, sh, nv);
        }
        carry += all;
    }
}

// ------------------------------------------------------------------------------------------
// one-time layout of the reads in HBM
// ------------------------------------------------------------------------------------------

// bam_endpos(b) - 1 (htslib): sum of M(0) D(2) N(3) =(7) X(8) lengths; 0x4 or empty -> 1 base.
__global__ void k_cigar_end(int64_t n, const int32_t *__restrict__ pos,
                            const uint16_t *__restrict__ flag,
                            const int64_t *__restrict__ cigar_off,
                            const uint32_t *__restrict__ cigar, int32_t *__restrict__ end_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t rlen = 0;
    if (!(flag[i] & 0x4)) {
        const int64_t k1 = cigar_off[i + 1];
        for (int64_t k = cigar_off[i]; k < k1; ++k) {
            const uint32_t c = cigar[k];
            const uint32_t op = c & 0xFu;
            if ((0x18Du >> op) & 1u) rlen += c >> 4;    // bits 0,2,3,7,8
        }
    }
    if (rlen == 0) rlen = 1;
    end_out[i] = (int32_t)(pos[i] + rlen - 1);
}

// Span classes 0 and 1 keep `span - 1` next to flag and mapq in ONE word (no end column: 8 B per visit):
//   class 0 (span <= 256):   flag (16 bits) | mapq << 16 | (span - 1) << 24
//   class 1 (span <= 4096):  flag (12 bits) | mapq << 12 | (span - 1) << 20
// The SAM specification defines 12 flag bits; a read that sets a higher one (a uint16 can) and spans more
// than 256 bp goes to class 2, whose words hold all 16.
// A short read (span <= 256) whose (flag, mapq) pair has a code in the file's pair table, with 12-bit flags
// and pos inside its reference (the bucket index clamps positions to the reference; a packed word cannot
// carry a position the index does not vouch for), goes to the packed class instead: one word per read.
// codemap: 2^20 entries indexed by flag | mapq << 12, 0xFFFF = no code; NULL = no packed class.
__device__ __forceinline__ int read_class(int span, uint32_t flag, uint32_t mapq, int p, int64_t ref_bp,
                                          const uint16_t *__restrict__ codemap, uint32_t &code)
{
    if (span <= 256) {
        if (codemap && span >= 1 && flag < 4096u && p >= 0 && (int64_t)p < ref_bp) {
            code = codemap[flag | (mapq << 12)];
            if (code != 0xFFFFu) return BSIG_CLASS_PACKED;
        }
        return 0;
    }
    return (span <= 4096 && flag < 4096u) ? 1 : span <= 65536 ? 2 : 3;
}

// reference of read i: last r with ref_off[r] <= i
__device__ __forceinline__ int ref_of_read(const int64_t *__restrict__ ref_off, int n_ref, int64_t i)
{
    int lo = 0, hi = n_ref;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ref_off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

constexpr int kPrepThreads = 256;
constexpr int kPrepChunk = 2048;     // reads per workgroup in k_span_hist / k_scatter

// ---- the file's pair table: the (flag, mapq) pairs of short reads, counted on a sample --------------
// Any table is correct (a read whose pair has no code stays in class 0); a sample of the file finds the
// frequent pairs.  hist: 2^20 counters indexed by flag | mapq << 12.
__global__ __launch_bounds__(kPrepThreads) void k_pair_sample(int64_t n, int64_t chunk_stride,
                                                              const int32_t *__restrict__ pos, const int32_t *__restrict__ end,
                                                              const uint16_t *__restrict__ flag, const uint8_t *__restrict__ mapq,
                                                              uint32_t *__restrict__ hist)
{
    const int64_t base = (int64_t)blockIdx.x * chunk_stride * kPrepChunk;
    for (int r = 0; r < kPrepChunk / kPrepThreads; ++r) {
        const int64_t i = base + r * kPrepThreads + threadIdx.x;
        if (i >= n) break;
        const int span = end[i] - pos[i] + 1;
        const uint32_t f = flag[i];
        if (span >= 1 && span <= 256 && f < 4096u && pos[i] >= 0) atomicAdd(&hist[f | ((uint32_t)mapq[i] << 12)], 1u);
    }
}
// the non-empty counters as (key, count) pairs; *n_out may exceed cap (the caller then reads the counters themselves)
__global__ void k_pair_compact(const uint32_t *__restrict__ hist, uint32_t n_keys, uint2 *__restrict__ out, uint32_t cap,
                               uint32_t *__restrict__ n_out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_keys) return;
    const uint32_t c = hist[k];
    if (!c) return;
    const uint32_t slot = atomicAdd(n_out, 1u);
    if (slot < cap) out[slot] = make_uint2(k, c);
}
// codemap[key of code c] = c (codemap was filled with 0xFFFF); fmtab[c] = flag | mapq << 16
__global__ void k_codemap_fill(const uint32_t *__restrict__ fmtab, int n_codes, uint16_t *__restrict__ codemap)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_codes) return;
    const uint32_t fm = fmtab[c];
    codemap[(fm & 0xFFFu) | ((fm >> 16) & 0xFFu) << 12] = (uint16_t)c;
}

// per-chunk class counts + per-class max span; also checks that the reads are sorted by position
// inside every reference (maxspan[BSIG_MAX_CLASSES] is set to 1 otherwise)
__global__ __launch_bounds__(kPrepThreads) void k_span_hist(int64_t n, int32_t n_ref,
                                                            const int64_t *__restrict__ ref_off,
                                                            const uint32_t *__restrict__ ref_units,
                                                            const int32_t *__restrict__ pos,
                                                            const int32_t *__restrict__ end,
                                                            const uint16_t *__restrict__ flag,
                                                            const uint8_t *__restrict__ mapq,
                                                            const uint16_t *__restrict__ codemap,
                                                            uint32_t *__restrict__ chunk_counts,
                                                            int32_t *__restrict__ maxspan)
{
    __shared__ uint32_t cnt[BSIG_MAX_CLASSES];
    __shared__ int32_t mx[BSIG_MAX_CLASSES];
    const int tid = threadIdx.x;
    if (tid < BSIG_MAX_CLASSES) { cnt[tid] = 0; mx[tid] = 0; }
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kPrepChunk;
    // per lane: how many of its reads fell into each class and their longest span; ONE LDS atomic per wave and class
    // at the end (an atomic per read had all 64 lanes of a wave queue up on one address: 1.6 of the kernel's 3.9 ms
    // on 5e8 reads)
    uint32_t my_cnt[BSIG_MAX_CLASSES];
    int32_t my_max[BSIG_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) { my_cnt[c] = 0; my_max[c] = 0; }
    // (a chunk of 2,048 reads nearly always lies inside one reference: the search over the references' first reads,
    // four dependent loads per read, is made for the chunk's two ends -- uniform, scalar loads -- and per read only
    // where they differ)
    const int64_t last = (base + kPrepChunk < n ? base + kPrepChunk : n) - 1;
    const int rf_lo = ref_of_read(ref_off, n_ref, base < n ? base : n - 1), rf_hi = ref_of_read(ref_off, n_ref, last);
    for (int r = 0; r < kPrepChunk / kPrepThreads; ++r) {
        const int64_t i = base + r * kPrepThreads + tid;
        if (i < n) {
            const int p = pos[i];
            const int span = end[i] - p + 1;
            const int rf = rf_lo == rf_hi ? rf_lo : ref_of_read(ref_off, n_ref, i);
            uint32_t code;
            const int c = read_class(span, flag[i], mapq[i], p, (int64_t)ref_units[rf] << BSIG_REF_UNIT_SHIFT, codemap, code);
#pragma unroll
            for (int k = 0; k < BSIG_MAX_CLASSES; ++k) {
                my_cnt[k] += c == k ? 1u : 0u;
                my_max[k] = c == k && span > my_max[k] ? span : my_max[k];
            }
            // a position below its predecessor's is allowed only where a new reference starts
            if (i > 0 && p < pos[i - 1] && ref_off[rf] != i) maxspan[BSIG_MAX_CLASSES] = 1;
        }
    }
#pragma unroll
    for (int k = 0; k < BSIG_MAX_CLASSES; ++k) {
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) {
            my_cnt[k] += __shfl_xor(my_cnt[k], d);
            const int32_t o = __shfl_xor(my_max[k], d);
            my_max[k] = o > my_max[k] ? o : my_max[k];
        }
        if ((tid & (kWave - 1)) == 0) {
            if (my_cnt[k]) atomicAdd(&cnt[k], my_cnt[k]);
            if (my_max[k] > 0) atomicMax(&mx[k], my_max[k]);
        }
    }
    __syncthreads();
    if (tid < BSIG_MAX_CLASSES) {
        chunk_counts[(int64_t)blockIdx.x * BSIG_MAX_CLASSES + tid] = cnt[tid];
        if (mx[tid] > 0) atomicMax(&maxspan[tid], mx[tid]);
    }
}

// The exclusive scan of the chunks' class counts (chunk_base[k][c] = reads of class c in the chunks before k) and the
// class totals, on the device: the counts used to travel to the host (4.9 MB for 5e8 reads), be summed there and come
// back as 9.8 MB of bases, 3 ms of every layout.  One workgroup: every thread sums a slab of consecutive chunks, the
// slab sums are scanned in LDS, every thread writes its slab's bases.
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void k_chunk_scan(int64_t n_chunks, const uint32_t *__restrict__ counts,
                                                             uint64_t *__restrict__ chunk_base, uint64_t *__restrict__ totals)
{
    __shared__ uint64_t slab[kScanThreads][BSIG_MAX_CLASSES];
    const int tid = threadIdx.x;
    const int64_t per = (n_chunks + kScanThreads - 1) / kScanThreads;
    const int64_t k0 = (int64_t)tid * per, k1 = k0 + per < n_chunks ? k0 + per : n_chunks;
    uint64_t sum[BSIG_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) sum[c] = 0;
    for (int64_t k = k0; k < k1; ++k)
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) sum[c] += counts[k * BSIG_MAX_CLASSES + c];
#pragma unroll
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) slab[tid][c] = sum[c];
    __syncthreads();
    // inclusive scan over the slabs, in place (Hillis-Steele: ten steps)
    for (int d = 1; d < kScanThreads; d <<= 1) {
        uint64_t add[BSIG_MAX_CLASSES];
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) add[c] = tid >= d ? slab[tid - d][c] : 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) slab[tid][c] += add[c];
        __syncthreads();
    }
    uint64_t run[BSIG_MAX_CLASSES];
#pragma unroll
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) run[c] = slab[tid][c] - sum[c];
    for (int64_t k = k0; k < k1; ++k)
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
            chunk_base[k * BSIG_MAX_CLASSES + c] = run[c];
            run[c] += counts[k * BSIG_MAX_CLASSES + c];
        }
    if (tid == kScanThreads - 1)
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) totals[c] = slab[tid][c];
}

struct ScatterOut {
    int32_t *pos[BSIG_MAX_CLASSES];
    int32_t *end[BSIG_MAX_CLASSES];
    uint32_t *fm[BSIG_MAX_CLASSES];
    int32_t *tlen[BSIG_MAX_CLASSES];
    uint32_t *gb[BSIG_MAX_CLASSES];     // bucket number of every read (temporary)
    int32_t kshift[BSIG_MAX_CLASSES];
};

// stable partition of the reads into their classes; chunk_base = exclusive scan of chunk_counts
__global__ __launch_bounds__(kPrepThreads) void k_scatter(int64_t n, int32_t n_ref,
                                                          const int64_t *__restrict__ ref_off,
                                                          const uint32_t *__restrict__ ref_unit0,
                                                          const uint32_t *__restrict__ ref_units,
                                                          const int32_t *__restrict__ pos,
                                                          const int32_t *__restrict__ end,
                                                          const uint16_t *__restrict__ flag,
                                                          const uint8_t *__restrict__ mapq,
                                                          const int32_t *__restrict__ tlen,
                                                          const uint16_t *__restrict__ codemap,
                                                          const uint64_t *__restrict__ chunk_base,
                                                          const ScatterOut O)
{
    constexpr int NW = kPrepThreads / kWave;
    __shared__ uint32_t wcnt[NW][BSIG_MAX_CLASSES];
    __shared__ uint64_t run[BSIG_MAX_CLASSES];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    if (tid < BSIG_MAX_CLASSES) run[tid] = chunk_base[(int64_t)blockIdx.x * BSIG_MAX_CLASSES + tid];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kPrepChunk;
    // (the chunk's reference from its two ends, per read only where they differ: see k_span_hist)
    const int64_t last = (base + kPrepChunk < n ? base + kPrepChunk : n) - 1;
    const int rf_lo = ref_of_read(ref_off, n_ref, base < n ? base : n - 1), rf_hi = ref_of_read(ref_off, n_ref, last);
    for (int r = 0; r < kPrepChunk / kPrepThreads; ++r) {
        const int64_t i = base + r * kPrepThreads + tid;
        const bool valid = i < n;
        int p = 0, e = 0, cls = -1, rf = 0;
        uint32_t code = 0;
        if (valid) {
            p = pos[i]; e = end[i];
            rf = rf_lo == rf_hi ? rf_lo : ref_of_read(ref_off, n_ref, i);
            cls = read_class(e - p + 1, flag[i], mapq[i], p, (int64_t)ref_units[rf] << BSIG_REF_UNIT_SHIFT, codemap, code);
        }
        uint32_t rank = 0;
#pragma unroll
        for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
            const unsigned long long m = __ballot(cls == c);
            if (cls == c) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[wv][c] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (valid) {
            uint64_t dst = run[cls] + rank;
            for (int k = 0; k < wv; ++k) dst += wcnt[k][cls];
            int64_t pp = p < 0 ? 0 : p;
            const int64_t ref_bp = (int64_t)ref_units[rf] << BSIG_REF_UNIT_SHIFT;
            if (pp >= ref_bp) pp = ref_bp - 1;
            const uint64_t g = ((uint64_t)ref_unit0[rf] << BSIG_REF_UNIT_SHIFT) + (uint64_t)pp;
            uint32_t fmw = (uint32_t)flag[i] | ((uint32_t)mapq[i] << 16);
            if (cls == BSIG_CLASS_PACKED) {
                fmw = ((uint32_t)p & kPackPosMask) | ((uint32_t)(e - p) << BSIG_PACK_POS_BITS) | (code << 23);   // span - 1 <= 255
            } else {
                O.pos[cls][dst] = p;
                if (cls == 0) fmw |= (uint32_t)(e - p) << 24;      // span - 1 <= 255
                else if (cls == 1) fmw = (uint32_t)flag[i] | ((uint32_t)mapq[i] << 12) | ((uint32_t)(e - p) << 20);   // flag < 4096, span - 1 <= 4095
                else O.end[cls][dst] = e;
            }
            O.fm[cls][dst] = fmw;
            O.tlen[cls][dst] = tlen[i];
            O.gb[cls][dst] = (uint32_t)(g >> O.kshift[cls]);
        }
        __syncthreads();
        if (tid < BSIG_MAX_CLASSES) {
            uint64_t t = 0;
            for (int k = 0; k < NW; ++k) t += wcnt[k][tid];
            run[tid] += t;
        }
        __syncthreads();
    }
}

// idx[b] = first read of the class with bucket >= b (gb is sorted)
__global__ void k_build_idx(int64_t n, const uint32_t *__restrict__ gb, uint64_t n_buckets,
                            uint32_t *__restrict__ idx)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)gb[mid] < b) lo = mid + 1; else hi = mid;
    }
    idx[b] = (uint32_t)lo;
}

// ------------------------------------------------------------------------------------------
// integrity of a resident layout that came from a reads file (bsig_reads_load)
// ------------------------------------------------------------------------------------------
// Order-independent 64-bit checksum of a run of 32-bit words: sum of mix(word, position).  Computed on the
// device where the data lies anyway (save: before the download; load: after the upload), so a reads file
// that rotted on disk is caught at HBM speed instead of a host CRC pass over 12 B per read.
__global__ __launch_bounds__(256) void k_checksum(const uint32_t *__restrict__ w, uint64_t n, uint64_t salt,
                                                  unsigned long long *__restrict__ acc)
{
    unsigned long long h = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        unsigned long long x = ((unsigned long long)w[i] << 32 | (uint32_t)(i * 0x9E3779B1ull)) ^ (i + salt) * 0xD6E8FEB86659FD93ull;
        x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 29;
        h += x;
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) h += __shfl_xor(h, d);
    if ((threadIdx.x & (kWave - 1)) == 0 && h) atomicAdd(acc, h);
}

// a bucket index the pileup kernels may follow blindly: non-decreasing, ending at the class's read count
__global__ void k_check_idx(const uint32_t *__restrict__ idx, uint64_t n_buckets, uint32_t n_reads, int *__restrict__ bad)
{
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets) return;
    const uint32_t v = idx[b];
    if (v > n_reads || (b < n_buckets && v > idx[b + 1]) || (b == n_buckets && v != n_reads)) *bad = 1;
}

// ------------------------------------------------------------------------------------------
// read visits of a plan, for the roofline's algorithmic bytes:
//   acc[c], c = 0..3 = reads of span class c whose pos lies in the exact candidate window of
//                      their tile (SURVEY 8d's V, per class because class 0 reads are 4 B shorter)
//   acc[4]           = reads actually streamed (windows rounded to index buckets and to 4 reads)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lower_bound_pos(const int32_t *pos, uint32_t lo, uint32_t hi, int64_t key)
{
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((int64_t)pos[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// the same over the packed words of one chunk (positions are base + the low bits' distance from base)
__device__ __forceinline__ uint32_t lower_bound_packed(const uint32_t *words, uint32_t lo, uint32_t hi, int64_t base, int64_t key)
{
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (base + (int64_t)((words[mid] - (uint32_t)base) & kPackPosMask) < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void k_visits(const BsigReadsDev R, const BsigKParams P, int mode,
                         const BsigWorkItem *__restrict__ items, int64_t n_items,
                         unsigned long long *__restrict__ acc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_items) return;
    const BsigWorkItem w = items[t];
    int64_t tlo, thi;
    item_interval(w, P, mode, tlo, thi);
    unsigned long long streamed = 0;
    for (int c = 0; c < BSIG_SPAN_CLASSES; ++c) {
        const BsigClassCols &C = R.cls[c];
        if (C.n == 0) continue;
        uint32_t j_lo, j_hi;
        if (!class_window(C, w, tlo, thi, P.ext, j_lo, j_hi)) continue;
        streamed += ((j_hi + 3u) & ~3u) - (j_lo & ~3u);
        int64_t wlo = tlo - P.ext - C.maxspan + 1, whi = thi + P.ext;
        if (wlo < 0) wlo = 0;
        const uint32_t a = lower_bound_pos(C.pos, j_lo, j_hi, wlo);
        const uint32_t b = lower_bound_pos(C.pos, a, j_hi, whi);
        if (b > a) atomicAdd(&acc[c], (unsigned long long)(b - a));
    }
    {
        const BsigClassCols &C = R.cls[BSIG_CLASS_PACKED];
        int64_t rlo, rhi;
        if (packed_window(C, w, tlo, thi, P.ext, rlo, rhi)) {
            int64_t wlo = tlo - P.ext - C.maxspan + 1;
            const int64_t whi = thi + P.ext;
            if (wlo < 0) wlo = 0;
            const uint64_t g0 = (uint64_t)w.ref_unit0 << BSIG_REF_UNIT_SHIFT;
            unsigned long long v = 0;
            for (int64_t a = rlo; a < rhi; a += kPackChunk) {
                const int64_t b = a + kPackChunk < rhi ? a + kPackChunk : rhi;
                const uint32_t j_lo = C.idx[(g0 + (uint64_t)a) >> C.kshift], j_hi = C.idx[(g0 + (uint64_t)b) >> C.kshift];
                if (j_lo >= j_hi) continue;
                streamed += ((j_hi + 3u) & ~3u) - (j_lo & ~3u);
                const uint32_t x = lower_bound_packed(C.fm, j_lo, j_hi, a, wlo);
                const uint32_t y = lower_bound_packed(C.fm, x, j_hi, a, whi);
                v += y - x;
            }
            if (v) atomicAdd(&acc[BSIG_CLASS_PACKED], v);
        }
    }
    atomicAdd(&acc[BSIG_MAX_CLASSES], streamed);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers (called by runtime.hip)
// ------------------------------------------------------------------------------------------
namespace bsig {

// Tuning knobs (defaults from the environment once, changeable at run time through bsig_debug_set_knob for
// the sweep scripts): 0 = k_profile class-0 passes in flight (BAMSIGNALS_PROFILE_PRE), 1 = count tiles per
// wave (BAMSIGNALS_COUNT_TILES), 2 = count passes in flight (BAMSIGNALS_COUNT_PRE).
static int g_knobs[6] = {-1, -1, -1, -1, -1, -1};
static int knob(int k)
{
    static const char *const names[6] = {"BAMSIGNALS_PROFILE_PRE", "BAMSIGNALS_COUNT_TILES", "BAMSIGNALS_COUNT_PRE", "BAMSIGNALS_KNOB3",
                                         "BAMSIGNALS_KNOB4", "BAMSIGNALS_PROFILE_TILES"};
    static const int dflt[6] = {2, 0, 0, 0, 0, 0};
    if (g_knobs[k] < 0) {
        const char *e = getenv(names[k]);
        g_knobs[k] = e ? atoi(e) : dflt[k];
    }
    return g_knobs[k];
}

template <int NT>
static hipError_t launch_mode(int mode, int ss, const BsigReadsDev &R, const BsigKParams &P,
                              const BsigWorkItem *items, int64_t n_items, int tile_cells,
                              uint2 *windows, bool resolve_first, int32_t *out, hipStream_t st)
{
    if (n_items <= 0) return hipSuccess;
    // count family: consecutive tiles per wave and packed passes in flight (knobs 1 and 2, 0 = by the launch's size:
    // 8 x 4 from 65,536 tiles on -- eight tiles per wave still fill every SIMD eight times over --, 4 x 2 below;
    // scripts/count_sweep.py on config 3's tiling, final build: 1 x 4 0.139 ms, 2 x 2 0.122, 4 x 2 0.112, 8 x 4 0.107)
    const bool count_large = n_items >= 65536;
    const int count_tiles = knob(1) > 0 ? knob(1) : count_large ? 8 : 4, count_pre = knob(2) > 0 ? knob(2) : count_large ? 4 : 2;
    if (windows && resolve_first) {
        // P.resolved is set: the windows of every tile first (one lane per tile), with the lookup form of the parameters
        BsigKParams Q = P;
        Q.resolved = 0;
        hipLaunchKernelGGL(k_resolve_tiles, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st,
                           R, Q, mode, items, (uint32_t)n_items, reinterpret_cast<BsigResolved *>(windows));
    }
    const dim3 grid((unsigned)n_items), block(NT);
    if (mode == BSIG_MODE_PROFILE && tile_cells * (ss ? 2 : 1) <= kSmallCells && P.binsize > 1) {
        const int stride = (tile_cells * (ss ? 2 : 1)) | 1;
        const size_t lds = (size_t)((small_image_dwords(stride) + 3) & ~3) * sizeof(int32_t) + BSIG_PACK_CODES;   // + the packed class's table
        // (the form for resolved windows has no lookup code in it, as k_profile's)
#define BSIG_KS(SS_, RES_) hipLaunchKernelGGL((k_profile_small<NT, SS_, RES_>), grid, block, lds, st, items, (uint32_t)n_items, out, windows, R, P)
        if (ss) { if (P.resolved) BSIG_KS(true, true); else BSIG_KS(true, false); }
        else    { if (P.resolved) BSIG_KS(false, true); else BSIG_KS(false, false); }
#undef BSIG_KS
    } else if (mode == BSIG_MODE_PROFILE) {
        const size_t lds = (size_t)((tile_cells * (ss ? 2 : 1) + 8 + 7) / 8) * 16 + BSIG_PACK_CODES;     // 16-bit counters + the packed class's table
        // class-0 passes requested before anything is consumed (knob 0: 2, 3 or 4; fewer = fewer VGPRs = more
        // resident waves, more = one round trip for denser windows)
        const int pre = knob(0);
#define BSIG_KP(SS_, PRE_, W_)                                                                                                          \
    do {                                                                                                                                \
        if (P.resolved) hipLaunchKernelGGL((k_profile<NT, SS_, PRE_, W_, true>), grid, block, lds, st, items, (uint32_t)n_items, out, windows, R, P); \
        else hipLaunchKernelGGL((k_profile<NT, SS_, PRE_, W_, false>), grid, block, lds, st, items, (uint32_t)n_items, out, windows, R, P); \
    } while (0)
        // the build for 8 waves per SIMD (96 SGPRs, the rest kept in VGPR lanes; 57 VGPRs in the resolved form, which
        // alone has them to spare): narrow tiles -- many workgroups per byte moved -- gain from the eighth wave,
        // 2-kb tiles lose (same bases in 500 / 1,000 / 1,500 / 2,000-cell tiles, two launches, 7 -> 8 waves:
        // 354 -> 316, 227 -> 215, 212 -> 212, 187 -> 192 us).  knob 3: 8 = always, 1 = never, 0 = by the tile image
        const bool w8 = NT == kWave && P.resolved && (knob(3) == 8 || (knob(3) == 0 && lds <= 3072));
        // consecutive tiles per wave of a large launch (k_profile_multi).  One workgroup per tile, refilled by the
        // hardware as workgroups retire, is the better schedule down to 1-kb tiles (same bases, 7-8 waves, 1 / 2 / 4 tiles
        // per wave: 2-kb tiles 188 / 199 / 213 us, config 5's 1-kb tiles 145 / 155 / 166, config 4 398 / 420 / 431); for
        // 500-bp tiles four per wave win (317 / 299 / 282).  knob 5: 0 = four per wave for images of up to 2 KB
        // (about 760 cells), 1 = never, 2 / 4 = always
        const int pt = knob(5) == 0 ? (lds <= 2048 ? 4 : 1) : knob(5);
        if (NT == kWave && P.resolved && !P.accumulate && pt > 1) {
            const int T = pt >= 4 ? 4 : 2;
            const dim3 g2((unsigned)((n_items + T - 1) / T));
            const size_t lds_m = lds + (size_t)T * 20 * sizeof(uint32_t);
#define BSIG_KM(SS_, W_, T_) hipLaunchKernelGGL((k_profile_multi<SS_, 2, W_, T_>), g2, dim3(kWave), lds_m, st, items, (uint32_t)n_items, out, windows, R, P)
            if (ss) { if (w8) { if (T == 4) BSIG_KM(true, 8, 4); else BSIG_KM(true, 8, 2); } else { if (T == 4) BSIG_KM(true, 1, 4); else BSIG_KM(true, 1, 2); } }
            else    { if (w8) { if (T == 4) BSIG_KM(false, 8, 4); else BSIG_KM(false, 8, 2); } else { if (T == 4) BSIG_KM(false, 1, 4); else BSIG_KM(false, 1, 2); } }
#undef BSIG_KM
            return hipGetLastError();
        }
        if (ss) { if (pre <= 2) { if (w8) BSIG_KP(true, 2, 8); else BSIG_KP(true, 2, 1); } else if (pre == 3) BSIG_KP(true, 3, 1); else BSIG_KP(true, 4, 1); }
        else    { if (pre <= 2) { if (w8) BSIG_KP(false, 2, 8); else BSIG_KP(false, 2, 1); } else if (pre == 3) BSIG_KP(false, 3, 1); else BSIG_KP(false, 4, 1); }
#undef BSIG_KP
    } else if (mode == BSIG_MODE_COVERAGE) {
        const size_t lds = (size_t)((tile_cells + 8 + 7) / 8) * 16 + (size_t)((NT / 64 + 3) / 4) * 16 + BSIG_PACK_CODES;   // signed 16-bit cells, scan totals, the packed class's table
        // (the packed class's passes hold 256 reads each: two in flight cover a 2-kb tile at 100-fold coverage; the form
        // for resolved windows has no lookup code in it)
        if (P.resolved) hipLaunchKernelGGL((k_coverage<NT, 2, true>), grid, block, lds, st, items, (uint32_t)n_items, out, windows, R, P);
        else hipLaunchKernelGGL((k_coverage<NT, 2, false>), grid, block, lds, st, items, (uint32_t)n_items, out, windows, R, P);
    } else if (NT == kWave && (!windows || P.resolved) && count_tiles > 1) {
        // several consecutive tiles per wave (the slices of heavy tiles, which come with fixed windows, and
        // the wider workgroups keep the one-tile kernel)
        const int T = count_tiles >= 8 ? 8 : count_tiles >= 4 ? 4 : 2;
        const dim3 g2((unsigned)((n_items + T - 1) / T));
#define BSIG_CM(T_, PRE_) hipLaunchKernelGGL((k_count_multi<T_, PRE_>), g2, dim3(kWave), 0, st, items, (uint32_t)n_items, out, windows, R, P)
        if (count_pre <= 2)      { if (T == 8) BSIG_CM(8, 2); else if (T == 4) BSIG_CM(4, 2); else BSIG_CM(2, 2); }
        else if (count_pre == 3) { if (T == 8) BSIG_CM(8, 3); else if (T == 4) BSIG_CM(4, 3); else BSIG_CM(2, 3); }
        else                     { if (T == 8) BSIG_CM(8, 4); else if (T == 4) BSIG_CM(4, 4); else BSIG_CM(2, 4); }
#undef BSIG_CM
    } else {
        hipLaunchKernelGGL((k_count<NT>), grid, block, 0, st, items, (uint32_t)n_items, out, windows, R, P);
    }
    return hipGetLastError();
}

hipError_t launch_pileup(int mode, int ss, int threads, const BsigReadsDev &R, const BsigKParams &P,
                         const BsigWorkItem *items, int64_t n_items, int tile_cells,
                         void *windows, bool resolve_first, int32_t *out, hipStream_t st)
{
    switch (threads) {
    case 64:  return launch_mode<64>(mode, ss, R, P, items, n_items, tile_cells, (uint2 *)windows, resolve_first, out, st);
    case 128: return launch_mode<128>(mode, ss, R, P, items, n_items, tile_cells, (uint2 *)windows, resolve_first, out, st);
    case 256: return launch_mode<256>(mode, ss, R, P, items, n_items, tile_cells, (uint2 *)windows, resolve_first, out, st);
    default:  return hipErrorInvalidValue;
    }
}

hipError_t launch_make_ptab(const BsigReadsDev &R, const BsigKParams &P, uint8_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_make_ptab, dim3(1), dim3(128), 0, st, R, P, out);
    return hipGetLastError();
}

hipError_t launch_resolve(const BsigReadsDev &R, const BsigKParams &P, int mode, const BsigWorkItem *items,
                          int64_t n_items, void *windows, hipStream_t st)
{
    if (n_items <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n_items * BSIG_MAX_CLASSES + 255) / 256)), dim3(256), 0, st,
                       R, P, mode, items, n_items, (uint2 *)windows);
    return hipGetLastError();
}

hipError_t launch_count_heavy(const void *windows, int64_t n_items, int64_t heavy_reads,
                              unsigned long long *count, hipStream_t st)
{
    if (n_items <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_count_heavy, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st,
                       (const uint2 *)windows, n_items, heavy_reads, count);
    return hipGetLastError();
}

hipError_t launch_cigar_end(int64_t n, const int32_t *pos, const uint16_t *flag, const int64_t *cigar_off,
                            const uint32_t *cigar, int32_t *end_out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cigar_end, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       n, pos, flag, cigar_off, cigar, end_out);
    return hipGetLastError();
}

int64_t prep_chunks(int64_t n) { return (n + kPrepChunk - 1) / kPrepChunk; }

hipError_t launch_pair_sample(int64_t n, const int32_t *pos, const int32_t *end, const uint16_t *flag, const uint8_t *mapq,
                              uint32_t *hist, uint2 *pairs, uint32_t cap, uint32_t *n_pairs, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    // up to 1,024 chunks of 2,048 reads, evenly spread over the file (all of it up to 2 M reads)
    const int64_t chunks = prep_chunks(n);
    const int64_t stride = std::max<int64_t>(1, chunks / 1024);
    const int64_t blocks = (chunks + stride - 1) / stride;
    hipLaunchKernelGGL(k_pair_sample, dim3((unsigned)blocks), dim3(kPrepThreads), 0, st, n, stride, pos, end, flag, mapq, hist);
    hipLaunchKernelGGL(k_pair_compact, dim3((1u << 20) / 256), dim3(256), 0, st, hist, 1u << 20, pairs, cap, n_pairs);
    return hipGetLastError();
}

hipError_t launch_codemap_fill(const uint32_t *fmtab, int n_codes, uint16_t *codemap, hipStream_t st)
{
    if (n_codes <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_codemap_fill, dim3((unsigned)((n_codes + 255) / 256)), dim3(256), 0, st, fmtab, n_codes, codemap);
    return hipGetLastError();
}

hipError_t launch_span_hist(int64_t n, int32_t n_ref, const int64_t *ref_off, const uint32_t *ref_units, const int32_t *pos,
                            const int32_t *end, const uint16_t *flag, const uint8_t *mapq, const uint16_t *codemap,
                            uint32_t *chunk_counts, int32_t *maxspan, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_span_hist, dim3((unsigned)prep_chunks(n)), dim3(kPrepThreads), 0, st,
                       n, n_ref, ref_off, ref_units, pos, end, flag, mapq, codemap, chunk_counts, maxspan);
    return hipGetLastError();
}

hipError_t launch_chunk_scan(int64_t n_chunks, const uint32_t *counts, uint64_t *chunk_base, uint64_t *totals, hipStream_t st)
{
    hipLaunchKernelGGL(k_chunk_scan, dim3(1), dim3(kScanThreads), 0, st, n_chunks, counts, chunk_base, totals);
    return hipGetLastError();
}

hipError_t launch_scatter(int64_t n, int32_t n_ref, const int64_t *ref_off, const uint32_t *ref_unit0,
                          const uint32_t *ref_units, const int32_t *pos, const int32_t *end,
                          const uint16_t *flag, const uint8_t *mapq, const int32_t *tlen, const uint16_t *codemap,
                          const uint64_t *chunk_base, const ScatterPtrs &S, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    ScatterOut O;
    for (int c = 0; c < BSIG_MAX_CLASSES; ++c) {
        O.pos[c] = S.pos[c]; O.end[c] = S.end[c]; O.fm[c] = S.fm[c]; O.tlen[c] = S.tlen[c];
        O.gb[c] = S.gb[c]; O.kshift[c] = S.kshift[c];
    }
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)prep_chunks(n)), dim3(kPrepThreads), 0, st,
                       n, n_ref, ref_off, ref_unit0, ref_units, pos, end, flag, mapq, tlen, codemap, chunk_base, O);
    return hipGetLastError();
}

hipError_t launch_build_idx(int64_t n, const uint32_t *gb, uint64_t n_buckets, uint32_t *idx, hipStream_t st)
{
    const uint64_t total = n_buckets + 1;
    hipLaunchKernelGGL(k_build_idx, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       n, gb, n_buckets, idx);
    return hipGetLastError();
}

hipError_t launch_checksum(const void *words, uint64_t n_words, uint64_t salt, unsigned long long *acc, hipStream_t st)
{
    if (n_words == 0) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<uint64_t>((n_words + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_checksum, dim3(blocks), dim3(256), 0, st, (const uint32_t *)words, n_words, salt, acc);
    return hipGetLastError();
}

hipError_t launch_check_idx(const uint32_t *idx, uint64_t n_buckets, uint32_t n_reads, int *bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_check_idx, dim3((unsigned)((n_buckets + 1 + 255) / 256)), dim3(256), 0, st, idx, n_buckets, n_reads, bad);
    return hipGetLastError();
}

hipError_t launch_visits(const BsigReadsDev &R, const BsigKParams &P, int mode, const BsigWorkItem *items,
                         int64_t n_items, unsigned long long *acc, hipStream_t st)
{
    if (n_items <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_visits, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st,
                       R, P, mode, items, n_items, acc);
    return hipGetLastError();
}

}  // namespace bsig

namespace { __global__ void k_warm_pileup() {} }
// loads this file's code object onto the current device (first launch of a process: ~10 ms per code object)
hipError_t bsig::warm_pileup_module(hipStream_t st)
{
    hipLaunchKernelGGL(k_warm_pileup, dim3(1), dim3(64), 0, st);
    return hipGetLastError();
}

extern "C" int bsig_debug_set_resolve_min(long long n_tiles);     // runtime.hip
extern "C" int bsig_debug_set_knob(int which, int value)
{
    if (which == 4) return bsig_debug_set_resolve_min(value);
    if (which < 0 || which >= 6 || value < 0) return -1;
    bsig::g_knobs[which] = value;
    return 0;
}

// (debug: what the compiler made of the pileup kernels the BASELINE configurations run -- registers per lane and
// scratch bytes per lane (spills: there must be none); tests/test_gpu_parity.py.  which: 0 k_profile resolved,
// 1 its 8-wave build, 2 k_profile fused, 3 k_profile_multi (8 waves, four tiles), 4 k_coverage resolved,
// 5 k_count_multi<4, 2>, 6 k_profile resolved with strands)
extern "C" int bsig_debug_pileup_attrs(int which, int *vgprs, int *scratch_bytes)
{
    const void *f = nullptr;
    switch (which) {
    case 0: f = reinterpret_cast<const void *>(&k_profile<64, false, 2, 1, true>); break;
    case 1: f = reinterpret_cast<const void *>(&k_profile<64, false, 2, 8, true>); break;
    case 2: f = reinterpret_cast<const void *>(&k_profile<64, false, 2, 1, false>); break;
    case 3: f = reinterpret_cast<const void *>(&k_profile_multi<false, 2, 8, 4>); break;
    case 4: f = reinterpret_cast<const void *>(&k_coverage<64, 2, true>); break;
    case 5: f = reinterpret_cast<const void *>(&k_count_multi<4, 2>); break;
    case 6: f = reinterpret_cast<const void *>(&k_profile<64, true, 2, 1, true>); break;
    default: return 1;
    }
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, f) != hipSuccess) return 2;
    if (vgprs) *vgprs = a.numRegs;
    if (scratch_bytes) *scratch_bytes = (int)a.localSizeBytes;
    return 0;
}
#ifdef BSIG_STAMPS
extern "C" int bsig_debug_set_stamp_buffer(void *buf)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &buf, sizeof(buf));
}
extern "C" int bsig_debug_set_ablate(int bits)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &bits, sizeof(bits));
}
#endif
