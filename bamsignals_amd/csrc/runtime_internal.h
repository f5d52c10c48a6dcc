// Internals shared by the host runtime (runtime.hip) and the device-side BAM decode (devdecode.hip).
#ifndef BSIG_RUNTIME_INTERNAL_H
#define BSIG_RUNTIME_INTERNAL_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/bamsignals_abi.h"
#include "bsig_types.h"
#include "host_util.h"

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? BSIG_ERR_NOMEM : BSIG_ERR_DEVICE,          \
                        "HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_),     \
                        __FILE__, __LINE__, #expr);                                            \
    } while (0)

namespace bsig {
// Large device blocks are kept and reused instead of going back to the driver.  Measured on MI355X / ROCm 7.2
// with plain hipMalloc / hipFree (scripts/hipmalloc_stalls.py): a hipMalloc normally returns in 0.3 ms, but
// once about 70 GB have been freed since the last time, ONE hipMalloc takes 2.2-3.4 s (the freed memory is
// handed back in bulk).  A decode of the north star's BAM allocates and frees 25 GB of scratch, so every third
// cold decode of a session paid that.  Blocks of kBlockCacheMin bytes or more therefore come from, and
// return to, a per-process cache of free blocks (best fit within `max_waste`), bounded by env
// BAMSIGNALS_SCRATCH_CACHE_GB per GPU (default 48; above it the largest free blocks are released first);
// bsig_cache_clear() releases all of it.
constexpr size_t kBlockCacheMin = (size_t)8 << 20;
// *got receives the block's real size (>= bytes), which block_free wants back
hipError_t block_alloc(int device, size_t bytes, double max_waste, void **p, size_t *got);
void block_free(int device, void *p, size_t bytes);
void block_cache_release();
// env BAMSIGNALS_ARENA_GB > 0: one allocation of that size made with the first context of a device; block_alloc
// carves out of it first (runtime.hip)
void arena_reserve(int device);
}  // namespace bsig

// owns a set of device allocations (all of them through block_alloc: the arena if there is one, the cache of
// free blocks for large ones, hipMalloc otherwise)
struct DevPool {
    struct Blk { void *p; size_t bytes; int device; };
    std::vector<Blk> blks;
    int64_t bytes = 0;
    template <typename T>
    hipError_t alloc(T **p, size_t count)
    {
        void *q = nullptr;
        const size_t nbytes = std::max<size_t>(count * sizeof(T), 16);
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        size_t got = 0;
        if (e == hipSuccess) e = bsig::block_alloc(dev, nbytes, 1.125, &q, &got);      // long-lived: little slack
        if (e != hipSuccess) { *p = nullptr; return e; }
        blks.push_back(Blk{q, got, dev});
        bytes += (int64_t)nbytes;
        *p = (T *)q;
        return hipSuccess;
    }
    void release()
    {
        for (const Blk &b : blks) bsig::block_free(b.device, b.p, b.bytes);
        blks.clear();
        bytes = 0;
    }
    ~DevPool() { release(); }
};

struct bsig_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    bool warm_pending = false;
};

struct bsig_reads {
    bsig_ctx *ctx = nullptr;
    BsigReadsDev dev{};
    DevPool pool;
    bsig_reads_info info{};
    int32_t n_ref = 0;
    uint64_t col_cap[BSIG_MAX_CLASSES] = {};      // elements allocated per class column
    uint64_t idx_entries[BSIG_MAX_CLASSES] = {};  // entries allocated per class index
    std::vector<uint32_t> fmtab;                  // host copy of dev.fmtab (BSIG_PACK_CODES entries, or empty)
    std::vector<uint32_t> ref_unit0, ref_units;
    std::vector<int32_t> ref_len;
};

namespace bsig {
struct BaiIndex;
// Builds the resident HBM layout of R (span classes + bucket indexes, bsig_types.h) from device
// columns of n reads in BAM order; ref_off is a host array of n_ref + 1 entries.  The input
// columns are only read.
int layout_from_device(bsig_ctx *ctx, bsig_reads *R, int64_t n, int32_t n_ref, const int32_t *ref_len,
                       const int64_t *ref_off, const int32_t *d_pos, const int32_t *d_end,
                       const uint16_t *d_flag, const uint8_t *d_mapq, const int32_t *d_tlen);
// first launches of a process load each source file's code object onto the device: done with the context
hipError_t warm_decode_module(hipStream_t st);      // devdecode.hip
hipError_t warm_collect_module(hipStream_t st);     // collect.hip
// hands the device-side decode's cached scratch back to the driver (devdecode.hip)
void release_decode_scratch();
// > 0: the file (or this build's limits) needs another decode path; nothing was allocated
constexpr int kNeedsCpuPath = 1;
// Whole BAM -> resident reads on every listed GPU: each GPU decodes one share of the BGZF blocks, the
// column shares are all-gathered over xGMI (devdecode.hip).  BSIG_OK, kNeedsCpuPath or an error.
int reads_from_bam_sharded(const std::vector<bsig_ctx *> &ctxs, const std::string &path, int threads,
                           std::vector<bsig_reads *> &out, const char **transport);
// Index-driven decode with several GPUs: the merged BAI islands of the regions [beg, end) are dealt to the
// GPUs in contiguous runs, the column shares all-gathered (devdecode.hip).  BSIG_OK, kNeedsCpuPath or an error.
int reads_from_regions_sharded(const std::vector<bsig_ctx *> &ctxs, const std::string &path, const BaiIndex &idx, int64_t n_regions,
                               const int32_t *rid, const int64_t *beg, const int64_t *end, int threads,
                               std::vector<bsig_reads *> &out, const char **transport);
// a device buffer to (pageable or page-locked) host memory, staged through page-locked halves where
// that is faster (runtime.hip)
int download_to_host(bsig_ctx *ctx, const void *src_dev, void *dst_host, size_t bytes);
}  // namespace bsig
#endif
