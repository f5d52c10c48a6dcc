// Structures shared by the host runtime and the gfx950 kernels.
//
// HBM layout of the reads (built once per BAM by the prep kernels):
//   * reads are split by reference span into "span classes" (span <= 256 | <= 4096 | <= 65536 |
//     longer = classes 0..3), and the short ones (span <= 256) once more: those whose (flag, mapq)
//     pair is one of the file's BSIG_PACK_CODES most frequent pairs form the PACKED class
//     (BSIG_CLASS_PACKED), ONE 32-bit word per read
//        word : pos & 0x7FFF | (span - 1) << 15 | code << 23
//     -- a read window is narrower than 32,768 bases (wider ones are walked in chunks of that
//     width), so the low 15 bits of pos and the window's start give pos back; `code` indexes the
//     file's pair table fmtab[code] = flag | mapq << 16.  The filter on flag and mapq
//     (src/bamsignals.cpp:328-331) is evaluated once per code and launch, not once per read.
//     Short reads with a rarer pair, with flag bits above the 12 SAM defines or with pos outside
//     their reference stay in class 0.  Inside a class reads keep BAM order
//     (sorted by reference id, then pos).  Classes 0..3 have 32-bit columns
//        pos  : 0-based leftmost position                         (core.pos)
//        end  : bam_endpos - 1, inclusive                         (src/bamsignals.cpp:16-18)
//               -- classes 2..3 only; classes 0 and 1 keep span - 1 inside fm
//        fm   : class 0 (span <= 256):  flag | mapq << 16 | (span - 1) << 24
//               class 1 (span <= 4096, flag < 4096): flag | mapq << 12 | (span - 1) << 20
//               classes 2..3: flag | mapq << 16
//        tlen : template length                                   (core.isize)
//     and a bucket index  idx[b] = first read of the class whose global coordinate
//     g = (ref_unit0[rid] << 16) + pos  falls in bucket >= b, bucket = g >> kshift.
//     References are laid out back to back in units of 64 kbp, so one flat index
//     serves every reference.
//   * splitting by span bounds the left extension of a range's candidate window by the
//     class's own maximum span: one 2 kb intron-spanning read does not widen the window
//     of the 100-bp reads.
#ifndef BSIG_TYPES_H
#define BSIG_TYPES_H
#include <stdint.h>

#define BSIG_MAX_CLASSES 5
#define BSIG_SPAN_CLASSES 4      // classes 0..3: by span
#define BSIG_CLASS_PACKED 4      // span <= 256 and a frequent (flag, mapq) pair: one word per read (in .fm)
#define BSIG_PACK_POS_BITS 15    // low bits of pos kept in a packed word = log2 of a window chunk
#define BSIG_PACK_CODES 512      // entries of the pair table (9-bit codes)
#define BSIG_REF_UNIT_SHIFT 16   // references are laid out in units of 65536 bp

struct BsigClassCols {
    const int32_t *pos;
    const int32_t *end;
    const uint32_t *fm;
    const int32_t *tlen;
    const uint32_t *idx;     // n_buckets + 1 entries
    int64_t n;               // reads in the class
    int32_t maxspan;         // max(end - pos + 1) over the class
    int32_t kshift;          // log2(bucket width in bp)
};

struct BsigReadsDev {
    BsigClassCols cls[BSIG_MAX_CLASSES];
    const uint32_t *fmtab;   // BSIG_PACK_CODES entries: flag | mapq << 16 of a packed word's code (zero-padded)
    int32_t n_codes;         // codes in use
};

// One unit of GPU work: a tile of at most `tile_cells` output cells of one range
// (the device-side analogue of GArray, src/bamsignals.cpp:32-50).
struct BsigWorkItem {
    int32_t loc;        // range start, 0-based
    int32_t len;        // range width
    int32_t c0;         // profile/coverage: first cell of the tile (range orientation)
                        // count: first base of the sub-interval, relative to loc
    int32_t nc;         // profile/coverage: cells in the tile; count: bases in the sub-interval
    int64_t out_off;    // flat int32 offset of the tile's first output cell
    uint32_t ref_unit0; // first 64-kbp unit of the range's reference in the global coordinate
    uint32_t units_strand;  // BSIG_ITEM_* bits below
};
#define BSIG_ITEM_UNITS_MASK 0x0FFFFFFFu   // 64-kbp units of the range's reference
#define BSIG_ITEM_HEAVY (1u << 29)         // its reads are piled up by slice items of a second launch
#define BSIG_ITEM_NEG (1u << 30)           // range is on the '-' strand
#define BSIG_ITEM_ATOMIC (1u << 31)        // count mode: add with a global atomic (range was split)

// What the index says about one tile (k_resolve_tiles -> the pileup kernels of a large launch): the read
// windows of the five classes, and the packed class's first chunk
struct BsigResolved {
    uint32_t win[2 * BSIG_MAX_CLASSES];   // [j_lo, j_hi) per class
    int32_t pbase, pchunks;
};

struct BsigKParams {
    int32_t mapqual;
    uint32_t requiredF;
    uint32_t filteredF;
    int32_t has_tlen_filter;
    int32_t tf0, tf1;
    int32_t use_tlen;       // tlen column needed (filter, midpoint or tspan)
    int32_t shift;
    int32_t midpoint;
    int32_t tspan;
    int32_t ss;
    int32_t binsize;        // >= 1 (profile)
    uint32_t div_magic;     // exact n / binsize for 0 <= n < 2^31:
    int32_t div_shift;      //   __umulhi(n, div_magic) >> div_shift        (binsize >= 2)
    uint32_t div_m15;       // ... and for 0 <= n < 2^15 (a range shorter than 32,768 bases) at the full rate of the
    int32_t div_s15;        //   vector unit: (n * div_m15) >> div_s15, a 24-bit multiply (binsize 2..8192; else 0)
    int32_t ext;            // window extension on both sides (src/bamsignals.cpp:457,487)
    int32_t tile_cells;     // output cells per tile (sizes the dynamic LDS image)
    int32_t accumulate;     // 1: add the tile image into the result with integer atomics (slices of
                            // a heavy tile) instead of storing it
    int32_t resolved;       // 1: `windows` holds one BsigResolved per tile, written by k_resolve_tiles in
                            // front of this launch (large launches: the tile's item and its windows then
                            // arrive in ONE memory round trip instead of two dependent ones)
    int32_t pad_;
    const uint8_t *ptab;    // the packed class's filter table for THESE parameters (BSIG_PACK_CODES bytes on the
                            // device: bit 0 rejected, bit 1 reverse strand), made once per plan by k_make_ptab
};

#endif
