// Structures shared by the host runtime and the gfx950 kernels.
//
// HBM layout of the reads (built once per BAM by the prep kernels):
//   * reads are split by reference span into up to BSIG_MAX_CLASSES "span classes"
//     (span <= 256 | <= 4096 | <= 65536 | longer).  Inside a class they keep BAM order
//     (sorted by reference id, then pos).  A class has four 32-bit columns
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

#define BSIG_MAX_CLASSES 4
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
    int32_t ext;            // window extension on both sides (src/bamsignals.cpp:457,487)
    int32_t tile_cells;     // output cells per tile (sizes the dynamic LDS image)
    int32_t accumulate;     // 1: add the tile image into the result with integer atomics (slices of
                            // a heavy tile) instead of storing it
};

#endif
