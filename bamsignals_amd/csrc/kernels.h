// Launchers of the gfx950 kernels (kernels.hip), used by the host runtime (runtime.hip).
#ifndef BSIG_KERNELS_H
#define BSIG_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bamsignals_abi.h"
#include "bsig_types.h"

namespace bsig {

struct ScatterPtrs {
    int32_t *pos[BSIG_MAX_CLASSES];
    int32_t *end[BSIG_MAX_CLASSES];
    uint32_t *fm[BSIG_MAX_CLASSES];
    int32_t *tlen[BSIG_MAX_CLASSES];
    uint32_t *gb[BSIG_MAX_CLASSES];
    int32_t kshift[BSIG_MAX_CLASSES];
};

hipError_t launch_pileup(int mode, int ss, int threads, const BsigReadsDev &R, const BsigKParams &P,
                         const BsigWorkItem *items, int64_t n_items, int tile_cells,
                         void *windows /* n_items * BSIG_MAX_CLASSES * 8 bytes (fixed read ranges: slices of heavy tiles), or NULL */,
                         bool resolve_first /* fill `windows` with k_resolve before the pileup launch */,
                         int32_t *out, hipStream_t st);
// the packed class's filter table for P (BSIG_PACK_CODES bytes at `out`): once per plan
hipError_t launch_make_ptab(const BsigReadsDev &R, const BsigKParams &P, uint8_t *out, hipStream_t st);
hipError_t launch_resolve(const BsigReadsDev &R, const BsigKParams &P, int mode, const BsigWorkItem *items,
                          int64_t n_items, void *windows, hipStream_t st);
hipError_t launch_count_heavy(const void *windows, int64_t n_items, int64_t heavy_reads,
                              unsigned long long *count, hipStream_t st);
hipError_t launch_cigar_end(int64_t n, const int32_t *pos, const uint16_t *flag, const int64_t *cigar_off,
                            const uint32_t *cigar, int32_t *end_out, hipStream_t st);
int64_t prep_chunks(int64_t n);
// the (flag, mapq) pairs of a sample of the short reads: hist = 2^20 zeroed counters (flag | mapq << 12), the non-empty
// ones as (key, count) in pairs[0 .. min(*n_pairs, cap))
hipError_t launch_pair_sample(int64_t n, const int32_t *pos, const int32_t *end, const uint16_t *flag, const uint8_t *mapq,
                              uint32_t *hist, uint2 *pairs, uint32_t cap, uint32_t *n_pairs, hipStream_t st);
// codemap (2^20 x uint16, filled with 0xFFFF) receives the code of every pair of fmtab (flag | mapq << 16)
hipError_t launch_codemap_fill(const uint32_t *fmtab, int n_codes, uint16_t *codemap, hipStream_t st);
hipError_t launch_span_hist(int64_t n, int32_t n_ref, const int64_t *ref_off, const uint32_t *ref_units, const int32_t *pos,
                            const int32_t *end, const uint16_t *flag, const uint8_t *mapq, const uint16_t *codemap /* or NULL */,
                            uint32_t *chunk_counts,
                            int32_t *maxspan /* BSIG_MAX_CLASSES + 1: the last = "not sorted" flag */, hipStream_t st);
// exclusive scan of the chunks' class counts (BSIG_MAX_CLASSES per chunk) and the class totals, on the device
hipError_t launch_chunk_scan(int64_t n_chunks, const uint32_t *counts, uint64_t *chunk_base, uint64_t *totals, hipStream_t st);
hipError_t launch_scatter(int64_t n, int32_t n_ref, const int64_t *ref_off, const uint32_t *ref_unit0,
                          const uint32_t *ref_units, const int32_t *pos, const int32_t *end,
                          const uint16_t *flag, const uint8_t *mapq, const int32_t *tlen, const uint16_t *codemap,
                          const uint64_t *chunk_base, const ScatterPtrs &S, hipStream_t st);
hipError_t launch_build_idx(int64_t n, const uint32_t *gb, uint64_t n_buckets, uint32_t *idx, hipStream_t st);
// 64-bit order-independent checksum of n_words 32-bit words, ADDED into *acc (device)
hipError_t launch_checksum(const void *words, uint64_t n_words, uint64_t salt, unsigned long long *acc, hipStream_t st);
// *bad (device) = 1 unless idx[0..n_buckets] is non-decreasing, <= n_reads, and ends at n_reads
hipError_t launch_check_idx(const uint32_t *idx, uint64_t n_buckets, uint32_t n_reads, int *bad, hipStream_t st);
hipError_t warm_pileup_module(hipStream_t st);
hipError_t launch_visits(const BsigReadsDev &R, const BsigKParams &P, int mode, const BsigWorkItem *items,
                         int64_t n_items, unsigned long long *acc, hipStream_t st);

}  // namespace bsig
#endif
