/* bamsignals_abi.h — C ABI of the MI355X-native bamsignals hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no R, Rcpp, torch or HIP types.
 * The shared object is bamsignals_amd/libbamsignals_hip.so.  Callers: the plain-C R shim
 * (bamsignals_amd/r_package/src/shim.c, the replacement of the reference's Rcpp glue
 * src/RcppExports.cpp:10-82) and the Python ctypes host (bamsignals_amd/_lib.py), which
 * mirrors the reference's R interface because R is not available in the build image.
 *
 * Citations "ref:" are file:line in lamortenera/bamsignals v1.41.1.
 *
 * Conventions
 *   - every function returns BSIG_OK (0) or a negative BSIG_ERR_*; bsig_last_error() returns the
 *     message of the last failure on the calling thread (the R shim passes it to Rf_error);
 *   - ranges are flat arrays: rid (reference id in the BAM header), loc (0-based start =
 *     GRanges start - 1, ref: src/bamsignals.cpp:131), len (width), strand (+1, -1, 0 for '*',
 *     ref: src/bamsignals.cpp:123-129);
 *   - results are ONE flat int32 buffer owned by the caller: range i owns
 *     out[off[i] .. off[i+1]) with off from bsig_layout().  With ss the element
 *     2*bin + antisense (the column-major 2 x width matrix of ref: src/bamsignals.cpp:172-190,361).
 *     bamCount (binsize <= 0): mult cells per range, i.e. the single vector / 2 x n matrix of
 *     ref: src/bamsignals.cpp:148-169.
 */
#ifndef BAMSIGNALS_ABI_H
#define BAMSIGNALS_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSIG_ABI_VERSION 4

enum {
    BSIG_OK = 0,
    BSIG_ERR_ARG = -1,       /* invalid argument                                              */
    BSIG_ERR_IO = -2,        /* "Fail to open BAM file X"          ref: src/bamsignals.cpp:204 */
    BSIG_ERR_NOINDEX = -3,   /* "BAM indexing file is not available for file X"      ref: :209 */
    BSIG_ERR_CHROM = -4,     /* "chromosome X not present in the bam file"           ref: :119 */
    BSIG_ERR_EXT = -5,       /* "negative 'ext' values don't make sense"             ref: :243 */
    BSIG_ERR_DEVICE = -6,    /* HIP runtime failure / no GPU                                  */
    BSIG_ERR_NOMEM = -7,
    BSIG_ERR_FORMAT = -8     /* malformed BAM / BAI / SAM                                     */
};

enum { BSIG_MODE_PROFILE = 0, BSIG_MODE_COUNT = 1, BSIG_MODE_COVERAGE = 2 };

int bsig_abi_version(void);
/* CPUs the library sizes its host thread pools by: hardware threads, cut down by the affinity mask and
 * by a cgroup CPU quota (env BAMSIGNALS_THREADS overrides the pool size itself)                      */
int32_t bsig_effective_cpus(void);
const char *bsig_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Output layout — replaces allocateList (ref: src/bamsignals.cpp:139-192).
 * off must hold n+1 entries; returns the total number of int32 cells (off[n]).
 * binsize <= 0 selects bamCount's layout.  Also the native half of fastWidth
 * (ref: src/CountSignals.cpp:19-29): width[i] = (off[i+1]-off[i]) / (ss ? 2 : 1).
 * ------------------------------------------------------------------------------------------ */
int64_t bsig_layout(int64_t n, const int32_t *len, int32_t binsize, int32_t ss, int64_t *off);

/* checkList (ref: src/CountSignals.cpp:4-16): is a list of n signals a valid `signals` slot?  The
 * caller reports per element whether it is an integer vector (INTSXP), how long its `dim` attribute is
 * (0: none) and dim[0].  Valid: every element an integer vector and, if ss, a matrix (2 dims) with 2
 * rows.  Returns 1 (valid) or 0.                                                                */
int32_t bsig_check_list(int64_t n, const int32_t *is_int, const int32_t *n_dim, const int32_t *dim0,
                        int32_t ss);
/* fastWidth (ref: src/CountSignals.cpp:19-29): width[i] = length[i] / (ss ? 2 : 1)              */
void bsig_fast_width(int64_t n, const int64_t *length, int32_t ss, int32_t *width);

/* ------------------------------------------------------------------------------------------
 * Device context: one per GPU (and per host thread that drives it).
 * stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL to create one.
 * ------------------------------------------------------------------------------------------ */
typedef struct bsig_ctx bsig_ctx;
int bsig_device_count(int32_t *n);
/* env BAMSIGNALS_ARENA_GB=<n> (default 0): the first context of a device reserves n GB of HBM in ONE allocation;
 * scratch, resident reads and result buffers are carved out of it first, so that calls whose memory fits make no
 * allocation of their own (every allocation is a trip into the driver, and on a shared host that is where a call's
 * time goes astray).  An arena nobody holds a block of is released by bsig_cache_clear().               */
int bsig_ctx_create(int32_t device, void *stream, bsig_ctx **ctx);
void bsig_ctx_destroy(bsig_ctx *ctx);
int bsig_ctx_sync(bsig_ctx *ctx);
void *bsig_ctx_stream(bsig_ctx *ctx);
/* page-locked host memory: results copied into it travel over PCIe by DMA at full rate
 * (bsig_plan_run_host into pageable memory is staged by the runtime and several times slower)   */
int bsig_host_alloc(int64_t bytes, void **ptr);
void bsig_host_free(void *ptr);

/* ------------------------------------------------------------------------------------------
 * Reads resident in HBM.  Input = the columnar arrays the CPU decode stage produces
 * (what htslib's bam1_core_t holds for each record returned by bam_itr_next,
 * ref: src/bamsignals.cpp:271), sorted by (reference id, pos) as in a coordinate-sorted BAM.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int64_t n_reads;
    int32_t n_ref;
    const int32_t *ref_len;    /* n_ref    : reference lengths (BAM header l_ref)               */
    const int64_t *ref_off;    /* n_ref+1  : reads of reference r are [ref_off[r], ref_off[r+1]) */
    const int32_t *pos;        /* core.pos, 0-based                                              */
    const uint16_t *flag;      /* core.flag                                                      */
    const uint8_t *mapq;       /* core.qual                                                      */
    const int32_t *tlen;       /* core.isize                                                     */
    const int32_t *end;        /* bam_endpos-1 (ref: src/bamsignals.cpp:16-18), or NULL to let   */
    const int64_t *cigar_off;  /*   the GPU compute it from the packed CIGAR:                    */
    const uint32_t *cigar;     /*   read i owns cigar[cigar_off[i] .. cigar_off[i+1]), len<<4|op */
} bsig_columns;

typedef struct bsig_reads bsig_reads;

/* Classes of the resident layout: 0..3 by reference span (<= 256 | <= 4096 | <= 65536 | longer); 4 = the
 * packed class: short reads (span <= 256) whose (flag, mapq) pair is one of the file's 512 most frequent
 * pairs -- ONE 32-bit word per read (15 position bits, span - 1, a 9-bit code into the pair table), so a
 * visit of such a read moves 4 bytes (8 with the template length); the other short reads stay in class 0 */
#define BSIG_N_CLASSES 5
typedef struct {
    int64_t n_reads;
    int64_t hbm_bytes;              /* device bytes held: class columns + bucket indexes (blocks as allocated) */
    int32_t n_classes;              /* classes in use                                            */
    int32_t n_codes;                /* (flag, mapq) pairs in the packed class's table            */
    int64_t class_n[BSIG_N_CLASSES];
    int32_t class_maxspan[BSIG_N_CLASSES];
    int32_t class_bucket_shift[BSIG_N_CLASSES];
} bsig_reads_info;

int bsig_reads_upload(bsig_ctx *ctx, const bsig_columns *cols, bsig_reads **reads);
int bsig_reads_get_info(const bsig_reads *reads, bsig_reads_info *info);
void bsig_reads_free(bsig_reads *reads);
/* a copy of resident reads on another GPU (device-to-device over xGMI where peer access exists):
 * how the single-process multi-GPU path replicates a BAM that was decoded once                 */
int bsig_reads_clone(const bsig_reads *src, bsig_ctx *dst_ctx, bsig_reads **reads);
/* The resident layout as a file (the decoded-column "sidecar" of a BAM): a later process loads it and
 * skips BGZF inflate and record parsing (the reference pays both on every call, ref:
 * src/bamsignals.cpp:449,479 + :271).  `stamp` ties the file to what it was made from (the
 * file-level calls use size + mtime of the BAM and of every index file next to it); bsig_reads_load fails
 * with BSIG_ERR_FORMAT when the stamp differs or the file is damaged, BSIG_ERR_IO when it is absent.
 * Nothing in the file is trusted: the shapes must follow from the counts, the bucket indexes are
 * checked on the device (non-decreasing, within the read count) and a checksum of what reached HBM
 * must equal the one the writer took of its resident layout.                                      */
int bsig_reads_save(const bsig_reads *reads, const char *path, const char *stamp);
int bsig_reads_load(bsig_ctx *ctx, const char *path, const char *stamp, bsig_reads **reads);

/* ------------------------------------------------------------------------------------------
 * A plan = ranges + call parameters resident in HBM, ready to run any number of times.
 * Replaces parseRegions' GArray vector + sort + the Pileupper/Coverager construction
 * (ref: src/bamsignals.cpp:92-135, 246, 455-457, 485-487).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t mode;              /* BSIG_MODE_*                                                   */
    int32_t mapqual;
    int32_t binsize;           /* profile: >= 1; ignored otherwise                              */
    int32_t shift;
    int32_t ss;
    int32_t requiredF;
    int32_t filteredF;
    int32_t pe_mid;            /* profile/count: paired.end == "midpoint"                       */
    int32_t tspan;             /* coverage:      paired.end == "extend"                         */
    int32_t n_tlen_filter;     /* 0 or 2 (ref: R/wrappers.R:84-98)                              */
    int32_t tlen_filter[2];
    /* tuning knobs, 0 = default */
    int32_t tile_cells;        /* output cells per workgroup tile (default 2048)                */
    int32_t threads;           /* 64, 128 or 256 threads per workgroup (default 64)             */
} bsig_params;

typedef struct bsig_plan bsig_plan;

typedef struct {
    int64_t n_ranges;
    int64_t n_items;           /* workgroup tiles                                               */
    int64_t cells;             /* int32 output cells                                            */
    int64_t visits;            /* reads in the exact candidate windows of all tiles (V)         */
    int64_t visits_short;      /* ... of them in span classes 0-1 (span <= 4096: no end column) */
    int64_t streamed;          /* reads actually loaded (windows rounded to index buckets)      */
    int64_t algorithmic_bytes; /* of one run of the resident plan: sum(bytes_per_visit*V) + 32*items + 4*cells
                                * + 8*items*classes (the index entries; small launches look them up in the pileup kernel)
                                * or + 48*items (large launches: the windows kept with the plan from its first run;
                                * with BAMSIGNALS_CACHE_WINDOWS=0, looked up by a launch of their own in every run:
                                * 8*items*classes + (32 + 2*48)*items) */
    int32_t bytes_per_visit_short;   /* 8  (pos + flag|mapq|span - 1), 12 with the tlen column  */
    int32_t bytes_per_visit_long;    /* 12 (pos + end + flag|mapq), 16 with the tlen column     */
    int64_t visits_packed;     /* ... of V in the packed class (not part of visits_short)       */
    int32_t bytes_per_visit_packed;  /* 4  (one word), 8 with the tlen column                   */
    int32_t heavy_tiles;       /* tiles whose read windows hold more reads than a tile image's 16-bit counters may
                                * see (32,768; 32,767 for coverage; BAMSIGNALS_HEAVY_READS lowers it): their reads are
                                * cut into slices that a second launch adds with integer atomics; 0: one launch     */
} bsig_plan_stats;

int bsig_plan_create(bsig_ctx *ctx, const bsig_reads *reads, int64_t n_ranges,
                     const int32_t *rid, const int32_t *loc, const int32_t *len,
                     const int32_t *strand, const bsig_params *params, bsig_plan **plan);
const int64_t *bsig_plan_offsets(const bsig_plan *plan);     /* n_ranges+1, host memory          */
int64_t bsig_plan_cells(const bsig_plan *plan);
int bsig_plan_get_stats(bsig_plan *plan, bsig_plan_stats *stats);
/* run on the context's stream; out_dev: device buffer of bsig_plan_cells() int32, 16-B aligned.
 * Asynchronous: call bsig_ctx_sync() (or synchronise the stream) before reading out_dev.       */
int bsig_plan_run(bsig_plan *plan, int32_t *out_dev);
/* run + copy to host memory + synchronise                                                      */
int bsig_plan_run_host(bsig_plan *plan, int32_t *out_host);
/* the same without the final synchronisation (out_host should be page-locked, bsig_host_alloc):
 * lets one host thread keep several GPUs busy; finish with bsig_ctx_sync() on the plan's context */
int bsig_plan_run_host_async(bsig_plan *plan, int32_t *out_host);
void bsig_plan_free(bsig_plan *plan);

/* one-shot: columns already in HBM -> host result (upload ranges, run, download)               */
int bsig_pileup_columns(bsig_ctx *ctx, const bsig_reads *reads, int64_t n_ranges,
                        const int32_t *rid, const int32_t *loc, const int32_t *len,
                        const int32_t *strand, const bsig_params *params,
                        int32_t *out_host, const int64_t *off);

/* ------------------------------------------------------------------------------------------
 * CPU decode stage: BAM (+ BAI) -> columnar arrays.  Replaces what the reference gets from
 * htslib: sam_open / bam_index_load (ref: src/bamsignals.cpp:200-214), sam_hdr_read +
 * bam_name2id (ref: :26-28, :95), bam_itr_queryi / bam_itr_next (ref: :267-271).
 * ------------------------------------------------------------------------------------------ */
typedef struct bsig_bam bsig_bam;
/* opens <path> and loads <path>.csi, <stem>.csi, <path>.bai or <stem>.bai (the first that exists, in the order
 * of htslib's bam_index_load, ref: src/bamsignals.cpp:207; a CSI index may have any min_shift / depth --
 * references beyond 2^29 bp need one; the first file that exists IS the index: one of a .csi name that is not a CSI index fails the open as it does in htslib, an older .bai beside it is not consulted); errors BSIG_ERR_IO / BSIG_ERR_NOINDEX with the reference's messages
 * (ref: src/bamsignals.cpp:204,209).                                                             */
int bsig_bam_open(const char *path, bsig_bam **bam);
void bsig_bam_close(bsig_bam *bam);
const char *bsig_bam_path(const bsig_bam *bam);
int32_t bsig_bam_n_ref(const bsig_bam *bam);
const char *bsig_bam_ref_name(const bsig_bam *bam, int32_t rid);
int32_t bsig_bam_ref_len(const bsig_bam *bam, int32_t rid);
int32_t bsig_bam_name2id(const bsig_bam *bam, const char *name);          /* -1 if absent        */
/* Decode the records the index lists for the regions [beg, end) (0-based) into host columns
 * owned by the handle (valid until the next decode or close).  n_regions < 0: whole file.
 * threads <= 0: all hardware threads (env BAMSIGNALS_THREADS).  cols->end is NULL and
 * cols->cigar_off / cols->cigar are filled: the GPU derives bam_endpos.                        */
int bsig_bam_decode(bsig_bam *bam, int64_t n_regions, const int32_t *rid, const int64_t *beg,
                    const int64_t *end, int32_t threads, bsig_columns *cols);
/* stage timers (seconds) of the calling thread's last whole-file decode: BGZF block scan, waiting
 * for inflate, record-boundary scan, column extraction, total, inflate time on the producer side */
void bsig_bam_decode_timing(double *t6);

/* Whole BAM -> reads resident in HBM.  The BGZF blocks are inflated by the CPU thread pool straight
 * into page-locked buffers that travel to HBM while the next batch inflates; record boundaries
 * (the block_size links bam_itr_next follows, ref: src/bamsignals.cpp:271), the core fields and
 * bam_endpos are then taken from the uncompressed stream by GPU kernels (csrc/devdecode.hip).
 * Records that cross BGZF block borders (htsjdk) and CG-tag CIGARs are handled there; damaged or
 * unsorted files, and any parse the host check could not prove, take the CPU decode
 * (bsig_bam_decode + bsig_reads_upload) inside this call: same result, and the CPU path's error
 * messages.  env BAMSIGNALS_DEVICE_DECODE=0 forces the CPU decode, =require fails instead of
 * falling back (testing).                                                                      */
int bsig_reads_from_bam(bsig_ctx *ctx, bsig_bam *bam, int32_t threads, bsig_reads **reads);
/* Whole BAM -> resident reads on each of n contexts (one per GPU; the single-process multi-GPU route):
 * GPU g inflates and parses share g of the BGZF blocks (the stage the reference spends its wall time
 * in, ref: src/bamsignals.cpp:271), the column shares are all-gathered over xGMI (RCCL grouped
 * send/recv, or peer copies: env BAMSIGNALS_EXCHANGE=rccl|peer), every GPU builds its resident layout.
 * The shares are accepted only if their record chains tile the stream exactly; otherwise (and for
 * files the device path declines, or smaller than 16 blocks per GPU) the file is decoded on the first
 * GPU and cloned.  *sharded (may be NULL) receives 1 if the sharded route was taken.
 * env BAMSIGNALS_SHARDED_DECODE=0 / =require as BAMSIGNALS_DEVICE_DECODE.                           */
int bsig_reads_from_bam_multi(bsig_ctx *const *ctxs, int32_t n, bsig_bam *bam, int32_t threads,
                              bsig_reads **reads, int32_t *sharded);
/* The same for an index-driven query: the records the BAI lists for the regions [beg, end) (what
 * one bam_itr_queryi per chunk of ranges returns, ref: src/bamsignals.cpp:252-271): a superset of
 * the overlapping records, each at most once, in file order.  Falls back like bsig_reads_from_bam. */
int bsig_reads_from_bam_regions(bsig_ctx *ctx, bsig_bam *bam, int64_t n_regions, const int32_t *rid,
                                const int64_t *beg, const int64_t *end, int32_t threads,
                                bsig_reads **reads);
/* stage seconds of the calling thread's last device-side decode: block scan, CPU inflate, waiting
 * for the copies, record walk + extraction kernels, total, HBM layout (all 0 after a CPU decode) */
void bsig_device_decode_timing(double *t6);

/* ------------------------------------------------------------------------------------------
 * File-level drop-in entry points: what the R shim's .Call routines bind.
 * Ranges come as GRanges slots flattened by the shim (ref: parseRegions, src/bamsignals.cpp:
 * 92-135): seq_code[i] indexes seq_levels (the factor levels of seqnames, mapped to BAM ids BY
 * NAME), start is 1-based, strand is +1 / -1 / 0.  out/off as in bsig_layout().
 * device < 0: the GPUs listed in env BAMSIGNALS_DEVICES ("0,1,...,7"), else env BAMSIGNALS_DEVICE,
 * else GPU 0.  With several GPUs every GPU inflates and parses its share of the BGZF blocks and the
 * column shares are all-gathered over xGMI; the (rid, loc)-sorted ranges are dealt round-robin to the
 * GPUs (one host thread and stream each); the result shards are gathered on the first GPU over xGMI
 * (RCCL grouped send/recv; env BAMSIGNALS_EXCHANGE=peer: peer copies; env BAMSIGNALS_GATHER=direct: the
 * first GPU reads its peers' shard buffers in place, one pass), put into range order there and
 * downloaded once (env BAMSIGNALS_GATHER=pcie: every GPU's shard over its own PCIe link instead,
 * reassembled by host threads).  Index-driven decodes (queries that need less than a third of the genome)
 * are shared between the GPUs island by island and the last few are kept (env BAMSIGNALS_REGION_CACHE,
 * default 8, 0 = none): a repeated call whose ranges lie inside an earlier call's finds its reads
 * resident.  maxgap is accepted for signature parity with the
 * reference (ref: src/bamsignals.cpp:446,476) and does not influence the result.
 * ------------------------------------------------------------------------------------------ */
/* replaces bamsignals_pileup_core (ref: src/RcppExports.cpp:33-52 -> src/bamsignals.cpp:444-461) */
int bsig_pileup_core(const char *bampath, int64_t n_ranges, const int32_t *seq_code,
                     int32_t n_seq_levels, const char *const *seq_levels, const int32_t *start,
                     const int32_t *width, const int32_t *strand,
                     const int32_t *tlen_filter, int32_t n_tlen_filter,
                     int32_t mapqual, int32_t binsize, int32_t shift, int32_t ss,
                     int32_t requiredF, int32_t filteredF, int32_t pe_mid, int32_t maxgap,
                     int32_t device, int32_t *out, const int64_t *off);
/* replaces bamsignals_coverage_core (ref: src/RcppExports.cpp:54-70 -> src/bamsignals.cpp:474-494) */
int bsig_coverage_core(const char *bampath, int64_t n_ranges, const int32_t *seq_code,
                       int32_t n_seq_levels, const char *const *seq_levels, const int32_t *start,
                       const int32_t *width, const int32_t *strand,
                       const int32_t *tlen_filter, int32_t n_tlen_filter,
                       int32_t mapqual, int32_t requiredF, int32_t filteredF, int32_t tspan,
                       int32_t maxgap, int32_t device, int32_t *out, const int64_t *off);
/* The same two calls with the result delivered IN PLACE, as the reference delivers it: allocateList (ref:
 * src/bamsignals.cpp:139-192) makes the R vectors first and the pileup counts straight into them (:361-362,
 * :423-436) -- ONE copy of the result in host memory.  dst[i] = where range i's cells go (the payload of its
 * vector / 2 x width matrix: (off[i+1] - off[i]) int32 of bsig_layout(); never touched for an empty range);
 * bamCount's layout (binsize <= 0) is one vector: dst[0] receives the n (ss: 2 x n) counts.  Large results
 * cross PCIe by DMA into two page-locked halves and are moved on range by range by a few threads: no flat
 * staging copy of the result exists in host memory.                                                     */
int bsig_pileup_core_into(const char *bampath, int64_t n_ranges, const int32_t *seq_code,
                          int32_t n_seq_levels, const char *const *seq_levels, const int32_t *start,
                          const int32_t *width, const int32_t *strand,
                          const int32_t *tlen_filter, int32_t n_tlen_filter,
                          int32_t mapqual, int32_t binsize, int32_t shift, int32_t ss,
                          int32_t requiredF, int32_t filteredF, int32_t pe_mid, int32_t maxgap,
                          int32_t device, int32_t *const *dst);
int bsig_coverage_core_into(const char *bampath, int64_t n_ranges, const int32_t *seq_code,
                            int32_t n_seq_levels, const char *const *seq_levels, const int32_t *start,
                            const int32_t *width, const int32_t *strand,
                            const int32_t *tlen_filter, int32_t n_tlen_filter,
                            int32_t mapqual, int32_t requiredF, int32_t filteredF, int32_t tspan,
                            int32_t maxgap, int32_t device, int32_t *const *dst);
/* replaces bamsignals_writeSamAsBamAndIndex (ref: src/bamsignals.cpp:496-534): text SAM ->
 * BAM + <bampath>.bai                                                                          */
int bsig_write_sam_as_bam_and_index(const char *sampath, const char *bampath);
/* columnar writer used for synthetic BAMs: coordinate-sorted columns -> BAM + BAI             */
int bsig_write_columns_as_bam(const char *bampath, int32_t n_ref, const char *const *ref_names,
                              const bsig_columns *cols, int32_t level);
/* the same with real-shaped records: a read name, l_seq random bases and qualities (about 3 bits of
 * entropy per quality, like binned Illumina data), an NM tag -- 204 bytes per 100-bp read instead of
 * 52, literal-heavy DEFLATE blocks that compress about 2 : 1.  For benchmarks of the decode stage on
 * data shaped like real BAMs; the alignment columns are the caller's, the rest follows (seed, index). */
int bsig_write_columns_as_bam_with_seq(const char *bampath, int32_t n_ref, const char *const *ref_names,
                                       const bsig_columns *cols, int32_t level, int32_t l_seq, uint64_t seed);
/* The file-level entry points keep, per process: one context per listed GPU, the parsed header +
 * BAI of the last 16 BAMs, and whole BAMs decoded to HBM -- least recently used first out above
 * env BAMSIGNALS_CACHE_GB (per GPU, default 96).  A file is identified by path + size + mtime of the
 * BAM and of its index: a rewritten file is decoded again.  env BAMSIGNALS_SIDECAR=1 (next to the BAM,
 * <bam>.bsig) or BAMSIGNALS_SIDECAR_DIR=<dir> additionally keeps the resident layout on disk
 * (bsig_reads_save) so that another process skips the decode.  File-level calls may be made from
 * several host threads at once (the cache is locked for look-ups only; cold decodes take turns).
 * Every device list seen (device argument / BAMSIGNALS_DEVICES) keeps its own contexts and resident
 * BAMs: alternating between two GPUs does not evict anything.  Multi-GPU calls on one device list take
 * turns at the run stage (they share the slots' cached result buffers and the RCCL communicators).
 * bsig_cache_clear drops all of it (not the sidecar files); whatever a running call uses -- contexts,
 * resident reads, communicators -- stays alive until that call returns, so it may be called at any time. */
void bsig_cache_clear(void);
/* diagnostic: device / page-locked allocations made so far for the buffers the multi-GPU result path keeps
 * between calls (a call on a resident BAM with shapes seen before adds none)                        */
int64_t bsig_debug_scratch_allocs(void);
/* diagnostic: number and a checksum of the BGZF blocks of a file as the device-side decode tabulates them
 * (env BAMSIGNALS_SCAN=mmap: the walk through the mapped file; default: through pread())              */
int bsig_debug_block_table(const char *path, int64_t *n_blocks, uint64_t *checksum);
/* ... and in two steps, as the whole-file decode tabulates large files: the head first (*n_head blocks), the
 * rest in the background                                                                             */
int bsig_debug_block_table_progressive(const char *path, int64_t head_bytes, int64_t *n_head, int64_t *n_blocks,
                                       uint64_t *checksum);
/* how the calling thread's last file-level call was carried out, e.g.
 * "8 GPU slot(s); reads: sharded decode, columns over rccl; result: xgmi/rccl"                  */
const char *bsig_last_call_route(void);
/* stage seconds of the calling thread's last bsig_pileup_core / bsig_coverage_core: open (header +
 * BAI), decode, upload + HBM layout, plan + kernels + download, total; t6[5] = 1 if the BAM was
 * already resident in HBM                                                                      */
void bsig_last_call_timing(double *t6);
/* the same (slots 0..5) and where the stages' time went, n <= 16 slots:
 *   6  seconds inside the driver's allocator (hipMalloc / hipFree / hipHostMalloc ...) during decode + layout
 *   7  ... during plan + kernels + download          8  ... during the whole call
 *   9  plan creation (ranges -> tiles in HBM, heavy-tile probe)   10  kernels (launch .. stream idle; includes the
 *      result buffer's allocation)   11  download of the result to host memory   (9-11: one GPU; 0 with several)
 *   12 driver allocation calls of the whole call
 *   13 reserved
 *   14 bytes reserved for the resident columns while the file was still being inflated (0: no reservation)
 *   15 seconds the layout waited for that reservation
 * The allocator meter is per process: calls made by other threads at the same time are counted too.         */
void bsig_last_call_timing_ex(double *t, int32_t n);

/* ------------------------------------------------------------------------------------------
 * Reassembly of sharded results (multi-GPU): segment k of src (src_off[k] .. src_off[k+1]) is
 * copied to dst at dst_off[which[k]].  The host half of "each GArray owns its own output
 * buffer" (ref: src/bamsignals.cpp:164,181,186) once ranges were dealt round-robin to GPUs.
 * ------------------------------------------------------------------------------------------ */
int bsig_scatter_segments(int64_t n, const int32_t *src, const int64_t *src_off, int32_t *dst,
                          const int64_t *dst_off, const int64_t *which);

/* The same on the device, for hosts that gather the shards into ONE device buffer on the root GPU (one
 * process per GPU: torch.distributed / RCCL gather): the segment tables live in HBM with the map, so a
 * run is one kernel launch on the context's stream, asynchronous, and the result stays in HBM in the
 * caller's range order.  The tables are checked like bsig_scatter_segments checks them (every segment fits
 * its destination, inside n_src_cells / n_dst_cells).  src_off: n+1 entries, dst_off: n_dst+1, which: n. */
typedef struct bsig_segmap bsig_segmap;
int bsig_segmap_create(bsig_ctx *ctx, int64_t n, const int64_t *src_off, int64_t n_dst, const int64_t *dst_off,
                       const int64_t *which, bsig_segmap **map);
int bsig_segmap_run(bsig_segmap *map, const int32_t *src_dev, int32_t *dst_dev);
/* A NARROW WIRE for result shards that travel between GPUs (round 5; the gather of the shards to one GPU is bounded by that
 * GPU's ingress, and the cells of a per-base profile are almost all 0, 1 or 2): a shard of n_cells int32 cells as a message
 * of bsig_narrow_bytes(n_cells, cap) bytes -- two bits a cell (the value, or 3 = see the list) and a list of up to `cap`
 * (cell, value) exceptions; lossless for any int32.  bsig_narrow_pack writes the message on the context's stream (src_dev
 * and msg_dev 16-B aligned); bsig_narrow_count (synchronises) says how many exceptions a packed shard has -- a plan's result
 * is a function of plan and reads, so a first run tells the `cap` of every later one; bsig_segmap_run_narrow is
 * bsig_segmap_run from such a message; bsig_segmap_narrow_overflowed (synchronises) reports whether any message so far had
 * more exceptions than its list held (its result is then wrong).  The message, in 32-bit words: [0] the number of
 * exceptions the shard has (it may exceed cap: then the list is incomplete), [1..3] zero, [4 .. 4 + ceil(n_cells / 16)) the
 * codes (cell c in bits 2 (c % 16) .. + 1 of word c / 16), then cap pairs (cell, value) of which the first min([0], cap)
 * are in use, in no particular order; n_cells < 2^32.  Replaces nothing of the reference: each range owns its
 * output there (ref: src/bamsignals.cpp:164,181,186), which is what makes shards -- and their reassembly -- legal.       */
int64_t bsig_narrow_bytes(int64_t n_cells, int64_t cap);
int bsig_narrow_pack(bsig_ctx *ctx, const int32_t *src_dev, int64_t n_cells, void *msg_dev, int64_t cap);
int bsig_narrow_count(bsig_ctx *ctx, const void *msg_dev, int64_t *n_exceptions);
int bsig_segmap_run_narrow(bsig_segmap *map, const void *msg_dev, int64_t n_cells, int64_t cap, int32_t *dst_dev);
int bsig_segmap_narrow_overflowed(bsig_segmap *map, int *overflowed);
void bsig_segmap_free(bsig_segmap *map);

#ifdef __cplusplus
}
#endif
#endif /* BAMSIGNALS_ABI_H */
