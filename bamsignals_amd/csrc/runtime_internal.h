// Internals shared by the host runtime (runtime.hip) and the device-side BAM decode (devdecode.hip).
#ifndef BSIG_RUNTIME_INTERNAL_H
#define BSIG_RUNTIME_INTERNAL_H
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
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
constexpr size_t kBlockCacheMin = (size_t)1 << 20;
constexpr size_t kSlabBytes = (size_t)1 << 20;       // what a pool carves its small allocations out of
// Every trip into the driver's allocator (hipMalloc, hipFree, hipHostMalloc, hipHostFree) goes through these and
// is metered, per process: where a call's time goes when it is not in a kernel or a copy is then a number
// (bsig_last_call_timing_ex) instead of a guess.
struct AllocMeter {
    std::atomic<int64_t> ns{0}, calls{0};
};
extern AllocMeter g_alloc_meter;
struct AllocTimer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~AllocTimer()
    {
        g_alloc_meter.ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count());
        g_alloc_meter.calls.fetch_add(1);
    }
};
inline hipError_t metered_malloc(void **p, size_t bytes) { AllocTimer t; return hipMalloc(p, bytes); }
inline hipError_t metered_free(void *p) { AllocTimer t; return hipFree(p); }
inline hipError_t metered_host_malloc(void **p, size_t bytes) { AllocTimer t; return hipHostMalloc(p, bytes, hipHostMallocDefault); }
inline hipError_t metered_host_free(void *p) { AllocTimer t; return hipHostFree(p); }
// Two page-locked halves of at least `bytes` each, one pair per GPU, for whoever moves bulk data through the host on
// that GPU: the streamed decode on the way in, the result download on the way out.  Pinning 2 x 32 MB costs 10 ms, and
// a session's first call used to pay it once for each.  The holder keeps `mu` for as long as it uses the halves.
struct PinnedPair {
    std::mutex mu;
    uint8_t *buf[2] = {nullptr, nullptr};
    size_t cap = 0;
    int ensure(size_t bytes);          // (holding mu; the device current)
};
PinnedPair &pinned_pair_for(int device);
// *got receives the block's real size (>= bytes), which block_free wants back
hipError_t block_alloc(int device, size_t bytes, double max_waste, void **p, size_t *got);
void block_free(int device, void *p, size_t bytes);
void block_cache_release();
// env BAMSIGNALS_ARENA_GB > 0: one allocation of that size made with the first context of a device; block_alloc
// carves out of it first (runtime.hip)
void arena_reserve(int device);
}  // namespace bsig

// owns a set of device allocations (all of them through block_alloc: the arena if there is one, the cache of
// free blocks for large ones, hipMalloc otherwise).  Requests below kSmall bytes are carved out of slabs of
// kSlabBytes (which the cache of free blocks keeps between uses): a layout or a plan makes a dozen
// allocations of a few hundred bytes each, and every one of them would be a trip into the driver.
// A pool may adopt a RESERVATION: one block made ahead of time (reads_from_bam_device reserves the resident
// columns on a side thread while the file is still being inflated); allocations are carved out of it while
// it lasts.
struct DevPool {
    struct Blk { void *p; size_t bytes; int device; };
    std::vector<Blk> blks;
    int64_t bytes = 0;
    double max_waste;                 // how much larger than asked a cached block may be (long-lived data: little)
    static constexpr size_t kSmall = (size_t)256 << 10;
    uint8_t *slab = nullptr;          // current slab for small requests (one of blks)
    size_t slab_left = 0;
    uint8_t *res = nullptr;           // the adopted reservation (one of blks)
    size_t res_left = 0;
    explicit DevPool(double waste = 1.125) : max_waste(waste) {}
    DevPool(const DevPool &) = delete;
    DevPool &operator=(const DevPool &) = delete;
    void adopt(int device, void *p, size_t got)
    {
        blks.push_back(Blk{p, got, device});
        res = (uint8_t *)p;
        res_left = got;
    }
    template <typename T>
    hipError_t alloc(T **p, size_t count)
    {
        const size_t nbytes = (std::max<size_t>(count * sizeof(T), 16) + 255) & ~(size_t)255;
        bytes += (int64_t)nbytes;
        if (res && nbytes <= res_left) {
            *p = (T *)res;
            res += nbytes;
            res_left -= nbytes;
            return hipSuccess;
        }
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) { *p = nullptr; return e; }
        if (nbytes <= kSmall) {
            if (nbytes > slab_left) {
                void *q = nullptr;
                size_t got = 0;
                e = bsig::block_alloc(dev, bsig::kSlabBytes, 2.0, &q, &got);
                if (e != hipSuccess) { *p = nullptr; return e; }
                blks.push_back(Blk{q, got, dev});
                slab = (uint8_t *)q;
                slab_left = got;
            }
            *p = (T *)slab;
            slab += nbytes;
            slab_left -= nbytes;
            return hipSuccess;
        }
        void *q = nullptr;
        size_t got = 0;
        e = bsig::block_alloc(dev, nbytes, max_waste, &q, &got);
        if (e != hipSuccess) { *p = nullptr; return e; }
        blks.push_back(Blk{q, got, dev});
        *p = (T *)q;
        return hipSuccess;
    }
    // device bytes this pool holds (slabs and an over-sized reservation included: what the cache budgets count)
    int64_t footprint() const
    {
        int64_t t = 0;
        for (const Blk &b : blks) t += (int64_t)b.bytes;
        return t;
    }
    void release()
    {
        for (const Blk &b : blks) bsig::block_free(b.device, b.p, b.bytes);
        blks.clear();
        bytes = 0;
        slab = res = nullptr;
        slab_left = res_left = 0;
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
    // which layout this is: a number no other layout of the process has (next_layout_gen(), taken whenever the resident
    // columns and indexes are (re)built or loaded).  A plan's cached tile windows are valid for exactly one layout.
    uint64_t layout_gen = 0;
};
namespace bsig { uint64_t next_layout_gen(); }

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
// the calling thread's last whole-file decode: bytes reserved ahead of time for the resident columns, and the
// seconds the layout waited for that reservation (devdecode.hip)
void decode_reservation_info(double *bytes, double *wait_s);
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
// Where a call's result goes in host memory: ONE flat buffer (range i at flat + off[i]), or one destination per
// range (ptrs[i]: the payload of the R vector allocateList made for range i, ref: src/bamsignals.cpp:172-190).
// `off` (n + 1 entries) is the flat layout either way.
struct HostDest {
    int32_t *flat = nullptr;
    int32_t *const *ptrs = nullptr;
    const int64_t *off = nullptr;
    int64_t n = 0;
    int32_t *range(int64_t i) const { return flat ? flat + off[i] : ptrs[i]; }
    // cells [c0, c0 + count) of the flat layout, wherever their ranges live
    void put(int64_t c0, const int32_t *src, int64_t count) const;
};
// a device buffer of `cells` int32 in the flat layout to the destination (runtime.hip)
int download_to_dest(bsig_ctx *ctx, const int32_t *src_dev, const HostDest &dst, int64_t cells);
// ... a slice of the result: `cells` int32 at src_dev belong at cell dst_cell0 of the flat layout.  copy_threads
// host threads move the page-locked halves on (0: the default, env BAMSIGNALS_COPY_THREADS)
int download_slice_to_dest(bsig_ctx *ctx, const int32_t *src_dev, const HostDest &dst, int64_t dst_cell0, int64_t cells, int copy_threads);
// ... several slices as ONE pipelined stream: slice k = cells[k] int32 at src_dev + src_c0[k], bound for cell dst_c0[k]
int download_slices_to_dest(bsig_ctx *ctx, const int32_t *src_dev, int64_t n_slices, const int64_t *src_c0, const int64_t *dst_c0,
                            const int64_t *cells, const HostDest &dst, int copy_threads);
// bsig_plan_run_host with the kernels' and the download's seconds told apart (runtime.hip)
int plan_run_host_timed(bsig_plan *p, const HostDest &dst, double *t_kernels, double *t_download);
}  // namespace bsig
#endif
