/* TEST INFRASTRUCTURE ONLY — CPU restatement of the reference hot path.
 *
 * Plain-C, single-thread restatement of lamortenera/bamsignals'
 * src/bamsignals.cpp:222-494 over *columnar* reads (the arrays the decode
 * stage produces) instead of htslib records.  It is the checker for the HIP
 * path and the timed "port" CPU baseline of bench.py.  Nothing under
 * bamsignals_amd/ may include, link or call it.
 *
 * PARITY UNPINNED (by the evidence rule): the reference holds no literal
 * expected outputs (its tests draw unseeded regions, tests/testthat/
 * test_methods.R:11-20) and cannot be built or run here (Rcpp.h, htslib/sam.h,
 * R and Rhtslib are absent from the image), so there is no oracle/_ref and no
 * number under tests/golden/ was computed by the reference or by R.  What IS
 * tied to reference-held data: the decode (BAM columns == the data.frame of
 * tests/testthat/randomReads.RData, read for read) and the BAI layout; the
 * arithmetic rests on three restatements by this project agreeing on the
 * reference's own test grid (this file from src/bamsignals.cpp,
 * oracle/oracle_np.py likewise in all-pairs form, oracle/r_oracle.py from the
 * R text of tests/testthat/utils.R:178-311).  The acceptance run that would pin
 * it needs R: bamsignals_amd/r_package/graft_into_reference.sh + the
 * reference's own testthat suite.
 */
#ifndef BAMSIGNALS_ORACLE_H
#define BAMSIGNALS_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int64_t n_reads;
    int32_t n_ref;
    const int64_t *ref_off;  /* n_ref+1: reads of reference r are [ref_off[r], ref_off[r+1]) */
    const int32_t *pos;      /* 0-based leftmost position, sorted within a reference      */
    const int32_t *end;      /* bam_endpos - 1 (inclusive)                                 */
    const uint16_t *flag;
    const uint8_t *mapq;
    const int32_t *tlen;
    int32_t max_span;        /* max(end - pos + 1); stands in for the BAI when emulating the
                                htslib region iterator.  Fill with bsor_max_span().       */
} bsor_reads;

/* bam_endpos(b) - 1 from packed CIGAR (htslib; call site src/bamsignals.cpp:16-18) */
void bsor_cigar_end(int64_t n, const int32_t *pos, const uint16_t *flag,
                    const int64_t *cigar_off, const uint32_t *cigar, int32_t *end_out);

int32_t bsor_max_span(int64_t n, const int32_t *pos, const int32_t *end);

/* allocateList (src/bamsignals.cpp:139-192): off[i] = first flat cell of range i,
 * off[n] = total.  binsize <= 0 -> one cell (x mult) per range.                 */
int64_t bsor_layout(int64_t n, const int32_t *len, int binsize, int ss, int64_t *off);

/* pileup_core (src/bamsignals.cpp:444-461).  out must hold off[n] ints; it is zeroed here.
 * returns 0, or -1 for "negative 'ext' values don't make sense" / bad arguments. */
int bsor_pileup_core(const bsor_reads *reads, int64_t n,
                     const int32_t *rid, const int32_t *loc, const int32_t *len,
                     const int32_t *strand,
                     const int32_t *tlen_filter, int n_tlen_filter,
                     int mapqual, int binsize, int shift, int ss,
                     int requiredF, int filteredF, int pe_mid, int maxgap,
                     int32_t *out, const int64_t *off);

/* coverage_core (src/bamsignals.cpp:474-494) */
int bsor_coverage_core(const bsor_reads *reads, int64_t n,
                       const int32_t *rid, const int32_t *loc, const int32_t *len,
                       const int32_t *strand,
                       const int32_t *tlen_filter, int n_tlen_filter,
                       int mapqual, int requiredF, int filteredF, int tspan, int maxgap,
                       int32_t *out, const int64_t *off);

#ifdef __cplusplus
}
#endif
#endif
