/* TEST INFRASTRUCTURE ONLY — see bamsignals_oracle.h.
 *
 * Faithful (chunked-driver) restatement of the reference algorithm; every
 * function cites the reference lines it follows.  htslib is replaced by sorted
 * columns + an emulation of its region iterator.
 */
#include "bamsignals_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- third-party arithmetic: htslib bam_endpos / bam_cigar2rlen -------------------
 * rlen = sum of op lengths for M(0) D(2) N(3) =(7) X(8); unmapped (0x4) -> 0; 0 -> 1.  */
void bsor_cigar_end(int64_t n, const int32_t *pos, const uint16_t *flag,
                    const int64_t *cigar_off, const uint32_t *cigar, int32_t *end_out)
{
    for (int64_t i = 0; i < n; ++i) {
        int64_t rlen = 0;
        if (!(flag[i] & 0x4)) {
            for (int64_t k = cigar_off[i]; k < cigar_off[i + 1]; ++k) {
                uint32_t op = cigar[k] & 0xF;
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8)
                    rlen += cigar[k] >> 4;
            }
        }
        if (rlen == 0) rlen = 1;
        end_out[i] = (int32_t)(pos[i] + rlen - 1);   /* readEnd, src/bamsignals.cpp:16-18 */
    }
}

int32_t bsor_max_span(int64_t n, const int32_t *pos, const int32_t *end)
{
    int32_t m = 1;
    for (int64_t i = 0; i < n; ++i) {
        int32_t s = end[i] - pos[i] + 1;
        if (s > m) m = s;
    }
    return m;
}

/* allocateList, src/bamsignals.cpp:139-192 (shapes only; R allocation is the shim's job) */
int64_t bsor_layout(int64_t n, const int32_t *len, int binsize, int ss, int64_t *off)
{
    int64_t mult = ss ? 2 : 1, acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        off[i] = acc;
        if (binsize <= 0) {
            acc += mult;                                  /* :148-169 */
        } else {
            int64_t w = ((int64_t)len[i] + binsize - 1) / binsize;   /* ceil(len/dbinsize) :175 */
            if (len[i] <= 0) w = 0;
            acc += mult * w;
        }
    }
    off[n] = acc;
    return acc;
}

/* GArray, src/bamsignals.cpp:32-50 */
typedef struct {
    int rid, loc, len, strand;
    int32_t *array;
} garray;

static inline int ga_end(const garray *g) { return g->loc + g->len; }

/* sortByStart, src/bamsignals.cpp:222-226 */
static int cmp_garray(const void *a, const void *b)
{
    const garray *x = (const garray *)a, *y = (const garray *)b;
    if (x->rid != y->rid) return x->rid < y->rid ? -1 : 1;
    if (x->loc != y->loc) return x->loc < y->loc ? -1 : 1;
    return 0;
}

typedef struct {
    /* shared filter parameters */
    int mapqual;
    uint32_t requiredF, filteredF;
    const int32_t *tlen_filter;   /* NULL when empty (:456, :486) */
    /* Pileupper */
    int binsize, shift, ss, midpoint;
    /* Coverager */
    int tspan;
    /* per-read state set by set_read */
    int pos5, negstrand;          /* Pileupper (:313-316) */
    int start, end;               /* Coverager (:377-380) */
} pile_state;

/* the filter shared by Pileupper::setRead :328-333 and Coverager::setRead :394-399 */
static inline int read_rejected(const pile_state *p, uint16_t flag, uint8_t mapq, int32_t isize)
{
    uint32_t nf = ~(uint32_t)(int)flag;                    /* ~flag after int promotion */
    if ((int)mapq < p->mapqual) return 1;
    if (p->requiredF & nf) return 1;                       /* invalidFlag(read, requiredF) */
    if (!(p->filteredF & nf)) return 1;                    /* !invalidFlag(read, filteredF) */
    if (p->tlen_filter) {
        int a = isize < 0 ? -isize : isize;
        if (a < p->tlen_filter[0] || a > p->tlen_filter[1]) return 1;
    }
    return 0;
}

/* Pileupper::setRead, src/bamsignals.cpp:326-346 (filter handled by the caller) */
static inline void pileupper_set(pile_state *p, int32_t pos, int32_t read_end, uint16_t flag, int32_t isize)
{
    p->negstrand = (flag & 0x10) != 0;
    int a = isize < 0 ? -isize : isize;
    int offset = p->midpoint ? (a / 2 + p->shift) : p->shift;
    p->pos5 = p->negstrand ? read_end - offset : pos + offset;
}

/* Pileupper::pileup, src/bamsignals.cpp:349-363 */
static inline void pileupper_pile(const pile_state *p, garray *r)
{
    int rel = p->pos5 - r->loc;
    if (rel < 0 || rel >= r->len) return;
    int anti = p->negstrand ? 1 : 0;
    if (r->strand < 0) { rel = r->len - rel - 1; anti = 1 - anti; }
    if (p->ss) ++r->array[2 * (rel / p->binsize) + anti];
    else       ++r->array[rel / p->binsize];
}

/* Coverager::setRead, src/bamsignals.cpp:392-415 */
static inline void coverager_set(pile_state *p, int32_t pos, int32_t read_end, uint16_t flag, int32_t isize)
{
    p->start = pos; p->end = read_end;
    if (p->tspan) {
        int neg = (flag & 0x10) != 0;
        if (neg && isize < 0)       p->start = p->end + isize + 1;
        else if (!neg && isize > 0) p->end = p->start + isize - 1;
    }
}

/* Coverager::pileup, src/bamsignals.cpp:418-438.  Zero-width ranges are skipped:
 * the reference writes array[0] of an empty vector there (UB).                       */
static inline void coverager_pile(const pile_state *p, garray *r)
{
    if (r->len <= 0) return;
    if (p->start >= ga_end(r) || p->end < r->loc) return;
    int a, b;
    if (r->strand >= 0) { a = p->start - r->loc;      b = p->end + 1 - r->loc; }
    else                { a = ga_end(r) - 1 - p->end; b = ga_end(r) - p->start; }
    ++r->array[a > 0 ? a : 0];
    if (b < r->len) --r->array[b];
}

/* first read index j in [lo,hi) with pos[j] >= key */
static int64_t lower_bound_pos(const int32_t *pos, int64_t lo, int64_t hi, int64_t key)
{
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if ((int64_t)pos[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* overlapAndPileup<T>, src/bamsignals.cpp:240-291.  coverage != 0 selects Coverager. */
static int overlap_and_pileup(const bsor_reads *R, garray *ranges, int64_t n, int ext,
                              pile_state *p, int maxgap, int coverage)
{
    if (ext < 0) return -1;                                /* :243 */
    qsort(ranges, (size_t)n, sizeof(garray), cmp_garray);  /* :246 */

    int64_t processed = 0;
    while (processed < n) {
        int64_t chunk_start = processed;
        int rid = ranges[chunk_start].rid;
        int64_t start = (int64_t)ranges[chunk_start].loc - ext;
        int64_t end = (int64_t)ga_end(&ranges[chunk_start]) + ext;
        int64_t chunk_end = chunk_start + 1;
        for (; chunk_end < n; ++chunk_end) {               /* :255-264 */
            int64_t next_start = (int64_t)ranges[chunk_end].loc - ext;
            if (ranges[chunk_end].rid != rid || next_start - end > maxgap) break;
            int64_t e = (int64_t)ga_end(&ranges[chunk_end]) + ext;
            if (e > end) end = e;
        }
        /* bam_itr_queryi(idx, rid, start, end) :267 — htslib clamps beg to 0, returns no
         * iterator if end < beg, and yields records with pos < end && endpos > beg.      */
        int64_t beg = start < 0 ? 0 : start;
        if (rid >= 0 && rid < R->n_ref && end >= beg) {
            int64_t lo = R->ref_off[rid], hi = R->ref_off[rid + 1];
            int64_t j = lower_bound_pos(R->pos, lo, hi, beg - R->max_span + 1);
            int64_t curr = chunk_start;
            for (; j < hi; ++j) {                          /* bam_itr_next loop :271 */
                int32_t rpos = R->pos[j];
                if (rpos >= end) break;
                int32_t rend = R->end[j];
                if ((int64_t)rend + 1 <= beg) continue;    /* endpos > beg */
                if (read_rejected(p, R->flag[j], R->mapq[j], R->tlen[j])) continue;   /* :272-273 */
                if (coverage) coverager_set(p, rpos, rend, R->flag[j], R->tlen[j]);
                else          pileupper_set(p, rpos, rend, R->flag[j], R->tlen[j]);
                int64_t ov_start = (int64_t)rpos - ext;    /* :275-276 */
                int64_t ov_end = (int64_t)rend + ext;
                while (curr < chunk_end && ov_start >= ga_end(&ranges[curr])) ++curr;   /* :278 */
                if (curr == chunk_end) break;              /* :280 */
                for (int64_t r = curr; r < chunk_end && ranges[r].loc <= ov_end; ++r) {  /* :282-285 */
                    if (coverage) coverager_pile(p, &ranges[r]);
                    else          pileupper_pile(p, &ranges[r]);
                }
            }
        }
        processed = chunk_end;
    }
    return 0;
}

static garray *make_ranges(int64_t n, const int32_t *rid, const int32_t *loc, const int32_t *len,
                           const int32_t *strand, int32_t *out, const int64_t *off)
{
    garray *g = (garray *)malloc(sizeof(garray) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        g[i].rid = rid[i]; g[i].loc = loc[i]; g[i].len = len[i]; g[i].strand = strand[i];
        g[i].array = out + off[i];
    }
    return g;
}

/* pileup_core, src/bamsignals.cpp:444-461 */
int bsor_pileup_core(const bsor_reads *reads, int64_t n,
                     const int32_t *rid, const int32_t *loc, const int32_t *len,
                     const int32_t *strand,
                     const int32_t *tlen_filter, int n_tlen_filter,
                     int mapqual, int binsize, int shift, int ss,
                     int requiredF, int filteredF, int pe_mid, int maxgap,
                     int32_t *out, const int64_t *off)
{
    if (pe_mid && n_tlen_filter < 2) return -1;   /* reference reads tlen_filter[1] here (:457) */
    memset(out, 0, sizeof(int32_t) * (size_t)off[n]);
    garray *g = make_ranges(n, rid, loc, len, strand, out, off);
    if (binsize <= 0) {                           /* allocateList count mode :160-167 */
        int maxw = -1;
        for (int64_t i = 0; i < n; ++i) if (len[i] > maxw) maxw = len[i];
        binsize = maxw;
    }
    pile_state p;
    memset(&p, 0, sizeof p);
    p.mapqual = mapqual; p.requiredF = (uint32_t)requiredF; p.filteredF = (uint32_t)filteredF;
    p.tlen_filter = n_tlen_filter ? tlen_filter : NULL;
    p.binsize = binsize; p.shift = shift; p.ss = ss; p.midpoint = pe_mid;
    int ext = abs(shift) + (pe_mid ? tlen_filter[1] : 0);      /* :457 */
    int rc = overlap_and_pileup(reads, g, n, ext, &p, maxgap, 0);
    free(g);
    return rc;
}

/* cumsum, src/bamsignals.cpp:464-470 */
static void cumsum(int32_t *c, int len)
{
    if (len < 2) return;
    int32_t acc = c[0];
    for (int i = 1; i < len; ++i) c[i] = (acc += c[i]);
}

/* coverage_core, src/bamsignals.cpp:474-494 */
int bsor_coverage_core(const bsor_reads *reads, int64_t n,
                       const int32_t *rid, const int32_t *loc, const int32_t *len,
                       const int32_t *strand,
                       const int32_t *tlen_filter, int n_tlen_filter,
                       int mapqual, int requiredF, int filteredF, int tspan, int maxgap,
                       int32_t *out, const int64_t *off)
{
    if (tspan && n_tlen_filter < 2) return -1;    /* reference reads tlen_filter[1] here (:487) */
    memset(out, 0, sizeof(int32_t) * (size_t)off[n]);
    garray *g = make_ranges(n, rid, loc, len, strand, out, off);
    pile_state p;
    memset(&p, 0, sizeof p);
    p.mapqual = mapqual; p.requiredF = (uint32_t)requiredF; p.filteredF = (uint32_t)filteredF;
    p.tlen_filter = n_tlen_filter ? tlen_filter : NULL;
    p.tspan = tspan;
    int ext = tspan ? tlen_filter[1] : 0;                       /* :487 */
    int rc = overlap_and_pileup(reads, g, n, ext, &p, maxgap, 1);
    if (rc == 0)
        for (int64_t i = 0; i < n; ++i) cumsum(g[i].array, g[i].len);   /* :490-492 */
    free(g);
    return rc;
}
