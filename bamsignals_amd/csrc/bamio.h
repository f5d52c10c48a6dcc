// Self-contained BGZF / BAM / BAI reader and writer (CPU decode stage).
//
// The reference reaches BAM files only through htslib (via Rhtslib; call sites
// src/bamsignals.cpp:95,202,207,267,271,505-531).  htslib is not part of this build, so the
// binary layouts are implemented here from the SAM/BAM specification (sections 4.1, 4.2, 5.2).
// Inflate/deflate use libdeflate when the shared object can be dlopen()ed, zlib otherwise.
#ifndef BSIG_BAMIO_H
#define BSIG_BAMIO_H
#include <stdint.h>

#include <functional>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace bsig {

struct BamHeader {
    std::string text;                 // SAM header text
    std::vector<std::string> names;   // reference names, in BAM order
    std::vector<int32_t> lens;
    int name2id(const std::string &n) const
    {
        for (size_t i = 0; i < names.size(); ++i)
            if (names[i] == n) return (int)i;
        return -1;
    }
};

// allocator whose resize() leaves new elements uninitialised: the decode threads are the first
// to touch (and page in) the column memory, instead of a serial zero fill; big buffers ask for
// transparent huge pages so that paging in 1+ GB of columns is thousands, not hundreds of
// thousands, of faults
void *column_alloc(size_t bytes);      // 2-MiB aligned + MADV_HUGEPAGE for large buffers
void column_free(void *p);

template <typename T>
struct NoInitAlloc {
    using value_type = T;
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(column_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t) { column_free(p); }
    bool operator==(const NoInitAlloc &) const { return true; }
    bool operator!=(const NoInitAlloc &) const { return false; }
    template <typename U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
template <typename T> using ColVec = std::vector<T, NoInitAlloc<T>>;

// columns of the placed records (refID >= 0), in file order = sorted by (refID, pos)
struct HostColumns {
    std::vector<int64_t> ref_off;     // n_ref + 1
    ColVec<int32_t> pos, tlen;
    ColVec<uint16_t> flag;
    ColVec<uint8_t> mapq;
    ColVec<int64_t> cigar_off;        // n + 1
    ColVec<uint32_t> cigar;
    int64_t n_unplaced = 0;           // records with refID < 0 (skipped)
    int64_t size() const { return (int64_t)pos.size(); }
};

struct BaiChunk { uint64_t beg, end; };
struct BaiRef {
    std::vector<std::pair<uint32_t, std::vector<BaiChunk>>> bins;   // sorted by bin number
    std::vector<uint64_t> linear;                                    // BAI: 16 kbp windows
    std::vector<std::pair<uint32_t, uint64_t>> loff;                 // CSI: per-bin loffset, sorted by bin number
};
// A BAI (min_shift 14, depth 5: SAM spec 5.2) or a CSI index (any min_shift / depth: CSIv1 spec; htslib's
// bam_index_load, ref: src/bamsignals.cpp:207, accepts both -- references beyond 2^29 bp need a CSI).
struct BaiIndex {
    std::vector<BaiRef> refs;
    uint64_t n_no_coor = 0;
    int min_shift = 14, depth = 5;
};

struct Region { int32_t rid; int64_t beg, end; };   // 0-based half-open

// one BGZF block (SAM spec 4.1)
struct BgzfBlock {
    uint64_t coff;       // file offset of the block
    uint32_t csize;      // whole block
    uint32_t doff;       // offset of the deflate data inside the block
    uint32_t dlen;       // deflate bytes
    uint32_t isize;      // uncompressed bytes
    uint32_t crc;        // CRC32 of the uncompressed bytes (the block's trailer)
};

// The BGZF layer of a file on its own, for the device-side record decode (devdecode.hip): the
// memory-mapped file, its block table, and the thread-pool inflate of a run of blocks.
class BgzfFile {
public:
    BgzfFile();
    ~BgzfFile();
    int open(const std::string &path);                        // maps the file and scans every block header
    // the same in two steps: returns once the blocks of the first head_bytes of the file are tabulated (blocks()
    // then holds exactly those, from the start of the file) while the rest is walked in the background; finish()
    // waits for it and completes blocks().  Small files, and any doubt, are tabulated whole at once.
    int open_progressive(const std::string &path, uint64_t head_bytes);
    bool complete() const;                                    // is blocks() the whole table?
    int finish();
    int map(const std::string &path);                         // maps the file only (index-driven access)
    // ... whose table then grows in file order as somebody else reads the headers (devdecode.hip: RawStream)
    void append_blocks(const BgzfBlock *b, size_t n);
    const std::vector<BgzfBlock> &blocks() const;
    const uint8_t *data() const;                              // the mapped (compressed) file
    size_t size() const;
    // inflates blocks [b0, b1) back to back into dst (sum of their isize bytes); threads <= 0: default
    int inflate(size_t b0, size_t b1, uint8_t *dst, int threads) const;
    // maps the pages of these byte spans of the file in (several threads) before they are read
    void populate(const std::vector<std::pair<uint64_t, uint64_t>> &spans) const;
    // copies file bytes [off, off + len) to dst with pread(): page cache -> dst without touching the mapping
    bool read_span(uint64_t off, size_t len, uint8_t *dst) const;
    // the block whose header sits at file offset `off`; false if there is none / it is malformed
    bool block_at(uint64_t off, BgzfBlock &b) const;
    // inflates the listed blocks back to back into dst
    int inflate_list(const BgzfBlock *list, size_t n, uint8_t *dst, int threads) const;
private:
    struct Impl;
    Impl *p_;
};

// The file chunks htslib's iterator would visit for `regions` (bins of reg2bins + the linear
// index's lower bound, SAM spec 5.3), sorted by file offset and merged: every chunk starts and
// ends at a record boundary, no two overlap.
std::vector<BaiChunk> bai_region_chunks(const BaiIndex &idx, const std::vector<Region> &regions);
// false: the CRC32 of inflated BGZF blocks is not compared with their trailers (env BAMSIGNALS_NO_CRC)
bool crc_check_enabled();
// the 8 x 256 lookup tables of CRC32 (IEEE, reflected) for 8 bytes per step
const uint32_t *crc32_slice8_tables();
// CPUs this process may use (hardware threads, affinity mask, cgroup quota)
int effective_cpus();
// the number of decode threads a request of `t` (<= 0: default) resolves to
int decode_threads(int t);
// runs body(i) for i in [0, n) on the decode thread pool
void pool_for(int64_t n, int threads, const std::function<void(int64_t)> &body);

// bytes in front of the first record of an uncompressed BAM stream (magic, text, reference
// table), or -1 if `n` bytes do not hold the whole header yet, -2 if it is not a BAM stream
int64_t bam_header_bytes(const uint8_t *p, size_t n);

// Whole file -> columns.  threads <= 0: hardware concurrency.
int bam_decode_all(const std::string &path, int threads, BamHeader &hdr, HostColumns &cols);
// Only the BGZF blocks the index lists for `regions` (a superset of the overlapping records,
// each record at most once, file order kept).
int bam_decode_regions(const std::string &path, const BaiIndex &idx, const std::vector<Region> &regions,
                       int threads, BamHeader &hdr, HostColumns &cols);
int bam_read_header(const std::string &path, BamHeader &hdr);
// seconds of the last bam_decode_all on this thread: block scan, inflate wait, boundary scan,
// column extraction, total
extern thread_local double g_decode_timing[6];
int bai_load(const std::string &bai_path, BaiIndex &idx);
// BSIG_ERR_NOINDEX if absent, BSIG_ERR_FORMAT if the file does not inflate to a CSIv1 index
int csi_load(const std::string &csi_path, BaiIndex &idx);

// One alignment for the writer
struct BamRecord {
    int32_t rid = -1, pos = -1;
    uint8_t mapq = 0;
    uint16_t flag = 0;
    int32_t next_rid = -1, next_pos = -1, tlen = 0;
    std::string name;                 // without NUL
    std::vector<uint32_t> cigar;      // len << 4 | op
    std::string seq;                  // bases, or empty for '*'
    std::string qual;                 // phred+33 text, or empty for '*'
    std::vector<uint8_t> aux;         // binary tags
};

// Streaming BAM writer that also builds the BAI (bins + 16-kbp linear index + pseudo-bin).
class BamWriter {
public:
    BamWriter();
    ~BamWriter();
    int open(const std::string &path, const BamHeader &hdr, int level = 6);
    int write(const BamRecord &r);
    // columnar fast path used for synthetic data: name "*", no sequence
    int write_core(int32_t rid, int32_t pos, uint16_t flag, uint8_t mapq, int32_t tlen,
                   const uint32_t *cigar, int n_cigar);
    // a whole coordinate-sorted file from columns: write_core() for every read, with the BGZF blocks
    // built and deflated by the worker pool (same bytes as the record-by-record calls)
    // l_seq > 0: real-shaped records instead of bare ones -- a read name, l_seq random bases and
    // qualities, an NM tag (204 bytes for 100 bp) -- generated from (seed, read index)
    int write_columns(int32_t n_ref, const int64_t *ref_off, const int32_t *pos, const uint16_t *flag,
                      const uint8_t *mapq, const int32_t *tlen, const int64_t *cigar_off,
                      const uint32_t *cigar, int threads, int l_seq = 0, uint64_t seed = 0);
    int close();                              // flushes, writes EOF block and <path>.bai
private:
    struct Impl;
    Impl *p_;
};

// writeSamAsBamAndIndex (ref: src/bamsignals.cpp:496-534): text SAM -> BAM + BAI
int sam_to_bam_and_index(const std::string &sam_path, const std::string &bam_path);

}  // namespace bsig
#endif
